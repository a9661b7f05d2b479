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
# PMC passes (HBM-side bytes of the GEMM kernels): counters in their own runs, kernel-trace only
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-gemm-events"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o f -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o w -- $CMD > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_gemm_json.py $(find gpurun_out/pmc_fetch -name "*_results.db" | head -1) $(find gpurun_out/pmc_write -name "*_results.db" | head -1) "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-gemm-events" > gpurun_out/pmc_gemm_bf16.json
head -c 900 gpurun_out/pmc_gemm_bf16.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/prof_bf16
