#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
MADTP_GEMM_SPB2=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -2
for st in 0 1; do echo SPB2=$st; MADTP_GEMM_SPB2=$st python tools/gemm_bench.py bf16 small 2>&1 | grep "^M=" | grep -E "lp_out" | head -7; done
for st in 0 1 0 1; do MADTP_GEMM_SPB2=$st python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('SPB2=$st', d['value'], d['ms_per_step'])"; done
for st in 0 1; do MADTP_GEMM_SPB2=$st python bench.py --config retrieval --steps 20 --warmup 5 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('retrieval SPB2=$st', d['value'], d['ms_per_step'])"; done
