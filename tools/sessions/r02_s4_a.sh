# session 4 baseline: GPU tests, per-shape GEMM table, per-shape breakdown inside the forward
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python tools/gemm_bench.py bf16 > gpurun_out/s4_gemm_bench.txt 2>&1
python bench.py --steps 10 --traffic off --no-cpu-baseline --no-parity --gemm-breakdown 2>gpurun_out/s4_bench.err | tail -1 > gpurun_out/s4_bench.json
cat gpurun_out/s4_gemm_bench.txt
grep -v "^$" gpurun_out/s4_bench.err | tail -40
cut -c1-400 gpurun_out/s4_bench.json
