set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "token_score" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_model_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -5
python bench.py --config vqa --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c1-200
MADTP_TS_SPLIT=0 python bench.py --config vqa --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c1-200
python bench.py --config retrieval --image-size 384 --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c1-200
MADTP_TS_SPLIT=0 python bench.py --config retrieval --image-size 384 --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c1-200
