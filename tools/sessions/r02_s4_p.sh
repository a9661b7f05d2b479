set -x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for P in f16x3 bf16; do echo "$P $(python bench.py --precision $P --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c90-170)"; done
