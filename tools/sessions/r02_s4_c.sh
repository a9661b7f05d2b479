# 256x256 GEMM kernel: parity tests, then the A/B table against the wave-specialised kernel on the forward's shapes
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" 2>&1 | tail -8
timeout 600 python tools/gemm_bench.py bf16 ab > gpurun_out/s4_gemm_ab.txt 2>&1
cat gpurun_out/s4_gemm_ab.txt
