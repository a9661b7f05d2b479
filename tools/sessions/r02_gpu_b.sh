set -x
mkdir -p gpurun_out
python -m pytest tests/test_workloads_gpu.py -x -q -s 2>&1 | tail -15
for c in retrieval clip vqa; do
  python bench.py --config $c --steps 10 --traffic off 2>gpurun_out/bench_$c.err | tail -1 > gpurun_out/bench_$c.json
  cut -c1-900 gpurun_out/bench_$c.json; tail -3 gpurun_out/bench_$c.err
done
python bench.py 2>gpurun_out/bench_bf16.err | tail -1 > gpurun_out/bench_bf16.json
cat gpurun_out/bench_bf16.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_detail'], d['parity_mode']['value'])"
tail -3 gpurun_out/bench_bf16.err
