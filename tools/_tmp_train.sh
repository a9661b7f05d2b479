set -x
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_backward_gpu.py -x -q 2>&1 | tail -8
MADTP_TRAIN_PRECISION=f16x3 timeout 600 python tools/train_step_bench.py 64 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
MADTP_TRAIN_PRECISION=f16x3 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py 64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find /tmp/prof_train -name "*_results.db" | head -1) "train step profile" > gpurun_out/tmp_train_stats.txt 2>&1
