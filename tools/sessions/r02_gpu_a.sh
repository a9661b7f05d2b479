set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py 2>gpurun_out/bench_bf16.err | tail -1 > gpurun_out/bench_bf16.json
python bench.py --precision f16x3 --steps 10 --no-cpu-baseline --no-parity 2>gpurun_out/bench_f16x3.err | tail -1 > gpurun_out/bench_f16x3.json
cat gpurun_out/bench_bf16.json; cat gpurun_out/bench_f16x3.json; tail -5 gpurun_out/bench_bf16.err gpurun_out/bench_f16x3.err
