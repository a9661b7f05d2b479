cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_backward_gpu.py -x -q 2>&1 | tail -8
MADTP_TRAIN_PRECISION=f16x3 timeout 600 python tools/train_step_bench.py 64 2>&1 | tail -1
MADTP_TRAIN_SAVE=0 MADTP_TRAIN_PRECISION=f16x3 timeout 600 python tools/train_step_bench.py 64 2>&1 | tail -1
