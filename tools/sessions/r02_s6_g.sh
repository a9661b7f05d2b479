#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for st in 0 1; do echo ST5=$st; MADTP_GEMM_ST5=$st python tools/gemm_bench.py bf16 small 2>&1 | grep "^M=" | grep -E "lp_out|f32\+res"; done
for st in 0 1 0 1; do MADTP_GEMM_ST5=$st python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ST5=$st', d['value'], d['ms_per_step'])"; done
for st in 0 1; do MADTP_GEMM_ST5=$st python bench.py --batch 1 --steps 50 --warmup 10 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('B=1 ST5=$st', d['value'], d['ms_per_step'])"; done
