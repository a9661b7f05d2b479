#!/bin/bash
# library yardstick + exact attention 96-key instantiation check
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/gemm_bench.py bf16 lib > gpurun_out/r02_gemm_lib.txt 2>&1
cat gpurun_out/r02_gemm_lib.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/libprof -o lib -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py bf16 lib > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/libprof 2>/dev/null | head -30 | cut -c1-260) > gpurun_out/r02_gemm_lib_kernels.txt 2>&1
cat gpurun_out/r02_gemm_lib_kernels.txt
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -3
python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | cut -c1-300
