#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | cut -c1-220
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_g.log 2>&1
DB=$(find /tmp/prof_g -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB "nlvr" | grep -E "attn_|total kernel" | cut -c1-200
