# usage: bash tools/r02_prof.sh <precision> <tag>   -> gpurun_out/kernel_stats_<tag>.txt
set -x
PREC=${1:-f16x3}; TAG=${2:-$PREC}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --precision $PREC --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find gpurun_out/prof_$TAG -name "*_results.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --precision $PREC --no-cpu-baseline --no-parity --no-gemm-events (8 forwards)" > gpurun_out/kernel_stats_$TAG.txt
rm -rf gpurun_out/prof_$TAG
cut -c1-150 gpurun_out/kernel_stats_$TAG.txt | head -45
tail -2 gpurun_out/prof_$TAG.log | cut -c1-300
