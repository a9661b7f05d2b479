# full GPU test suite + headline bench (with and without the 256x256 kernel) 
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 20 --traffic off --no-cpu-baseline --no-parity --gemm-breakdown 2>gpurun_out/s4f_bench.err | tail -1 > gpurun_out/s4f_bench.json
MADTP_GEMM_SQ=0 python bench.py --steps 20 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 > gpurun_out/s4f_bench_nosq.json
python bench.py --config clip --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 > gpurun_out/s4f_bench_clip.json
python bench.py --config retrieval --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 > gpurun_out/s4f_bench_retr.json
python bench.py --config vqa --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 > gpurun_out/s4f_bench_vqa.json
grep "bf16  M" gpurun_out/s4f_bench.err | head -50
for f in gpurun_out/s4f_bench*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('kernel','')[:60])"; done
