#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -5
timeout 600 python tools/gemm_bench.py bf16 ab > gpurun_out/r02_gemm_sk_ab.txt 2>&1
cat gpurun_out/r02_gemm_sk_ab.txt
for sk in 0 1; do MADTP_GEMM_SK=$sk python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('SK=$sk', d['value'], d['ms_per_step'], d['roofline'])"; done
