#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -3
for st in 2 3 0; do MADTP_ATTN_LARGE_STAGES=$st python bench.py --config vqa --steps 10 --warmup 3 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('STAGES=$st', d['value'], d['ms_per_step'])"; done
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o r02 -- python $GRAFT_REPO_ROOT/bench.py --config vqa --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_v.log 2>&1
DB=$(find /tmp/prof_v -name "*_results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB "vqa" | head -8 | cut -c1-200
