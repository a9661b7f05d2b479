#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
python -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q -k "gather or select" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o r02 -- python $R/bench.py --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_g.log 2>&1
DB=$(find /tmp/prof_g -name "*_results.db" | head -1)
python $R/tools/rocpd_stats.py $DB "gather" | grep -E "token_gather|total kernel" | cut -c1-200
