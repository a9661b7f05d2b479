set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/tl.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/tl -name "*_results.db" | head -1)
python tools/rocpd_timeline.py $DB patchify 1 1 345 > gpurun_out/s4_timeline_nlvr.txt
python tools/rocpd_step.py $DB
wc -l gpurun_out/s4_timeline_nlvr.txt
