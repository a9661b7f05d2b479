set -x
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention" 2>&1 | grep -E "AssertionError|passed|failed" | head
timeout 1200 python -m pytest tests/test_model_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pg -o r -- python $GRAFT_REPO_ROOT/bench.py --config vqa --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/pg.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pg -name "*_results.db" | head -1) "vqa" | grep -E "attn_|total kernel"
cd $GRAFT_REPO_ROOT
for C in vqa; do echo "$C $(python bench.py --config $C --steps 20 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c60-170)"; done
