# one gpurun call: smoke, bench (bf16 + fp32 parity mode), rocprofv3 kernel-trace stats of the bench command -> gpurun_out/
set -x
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py 2>&1 | tail -1 > gpurun_out/bench_bf16.json
python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --no-parity 2>&1 | tail -1 > gpurun_out/bench_fp32.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bf16
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_bf16 -name "*_results.db" | head -1) "$PROFILE_HEADER" > gpurun_out/kernel_stats.txt
head -30 gpurun_out/kernel_stats.txt
cut -c1-400 gpurun_out/bench_bf16.json; cut -c1-300 gpurun_out/bench_fp32.json
