set -x
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 2>&1 | tail -5 | tee gpurun_out/bench_bf16.json
python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --no-parity 2>&1 | tail -3 | tee gpurun_out/bench_fp32.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16.log 2>&1
ls -R $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 | head -20
