set -x
timeout 1200 python -m pytest tests/test_model_parity_gpu.py tests/test_workloads_gpu.py tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do for e in 1 0; do
echo "ENC=$e $(MADTP_ENCODER_CALL=$e python bench.py --steps 30 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c90-170)"
done; done
for b in 1 16; do for e in 1 0; do
echo "B$b ENC=$e $(MADTP_ENCODER_CALL=$e python bench.py --batch $b --steps 30 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c90-170)"
done; done
