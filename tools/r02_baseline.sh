# round-2 baseline: bf16 bench, fp32 bench + kernel stats of the fp32 (parity) mode at the headline batch
set -x
mkdir -p gpurun_out
python bench.py --no-cpu-baseline 2>gpurun_out/bench_bf16.err | tail -1 > gpurun_out/bench_bf16.json
python bench.py --steps 5 --warmup 2 --precision fp32 --no-cpu-baseline --no-parity 2>gpurun_out/bench_fp32.err | tail -1 > gpurun_out/bench_fp32.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_fp32 -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --precision fp32 --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_fp32.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_fp32 -name "*_results.db" | head -1) "fp32 parity mode, B=64, round-2 start" > gpurun_out/kernel_stats_fp32.txt
rm -rf gpurun_out/prof_fp32
head -40 gpurun_out/kernel_stats_fp32.txt
cut -c1-600 gpurun_out/bench_bf16.json; cut -c1-400 gpurun_out/bench_fp32.json
