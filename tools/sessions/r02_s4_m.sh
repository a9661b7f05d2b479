set -x
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gather or select" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_parity_gpu.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for C in nlvr vqa; do
rocprofv3 --kernel-trace --stats -d /tmp/pg_$C -o r -- python $GRAFT_REPO_ROOT/bench.py --config $C --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/pg_$C.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pg_$C -name "*_results.db" | head -1) "$C" | grep -E "token_gather|token_select|token_score|total kernel"
done
cd $GRAFT_REPO_ROOT
for C in nlvr vqa; do echo "$C $(python bench.py --config $C --steps 20 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>/dev/null | tail -1 | cut -c90-170)"; done
