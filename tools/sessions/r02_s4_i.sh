set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_parity_gpu.py -m gpu -x -q -k "encoder_level" 2>&1 | tail -15
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for e in 1 0; do
MADTP_ENCODER_CALL=$e python bench.py --steps 20 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c1-200
done
for b in 1 8 16; do for e in 1 0; do
MADTP_ENCODER_CALL=$e python bench.py --batch $b --steps 20 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c90-200
done; done
