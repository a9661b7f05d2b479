#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
for sk in 0 1 0 1; do MADTP_GEMM_SK=$sk python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('SK=$sk', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"; done
