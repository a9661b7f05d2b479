#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for st in 0 1; do echo DESC=$st; MADTP_GEMM_DESC=$st python tools/gemm_bench.py bf16 small 2>&1 | grep "^M=" | grep -E "lp_out"; done
for st in 0 1 0 1; do MADTP_GEMM_DESC=$st python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('DESC=$st', d['value'], d['ms_per_step'])"; done
for st in 0 1; do MADTP_GEMM_DESC=$st python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('B=1 DESC=$st', d['value'], d['ms_per_step'])"; done
