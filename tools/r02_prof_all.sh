# round-2 profiles: kernel stats of the headline (bf16), the parity mode (f16x3) and the VQA config
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for spec in "nlvr bf16" "nlvr f16x3" "vqa bf16" "retrieval bf16" "clip bf16"; do
  set -- $spec; C=$1; P=$2; TAG=${C}_${P}
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o r02 -- python $GRAFT_REPO_ROOT/bench.py --config $C --precision $P --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -name "*_results.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --config $C --precision $P --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events (7 forwards incl. warm-up)" > $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
done
cd $GRAFT_REPO_ROOT
for f in gpurun_out/kernel_stats_*.txt; do echo == $f; cut -c1-150 $f | head -14; done
