# round-2 final measurement set: GPU tests, full bench lines of the four configurations, kernel stats per configuration,
# one headline timeline  ->  gpurun_out/r02_f_*  (copied to profiles/ afterwards)
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
for C in nlvr retrieval clip vqa; do
  python bench.py --config $C 2>gpurun_out/r02_f_bench_$C.err | tail -1 > gpurun_out/r02_f_bench_$C.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_f_bench_$C.json")); r=d.get("roofline") or {}; p=d.get("parity_mode") or {}
print("$C", d["value"], d["ms_per_step"], "frac", r.get("frac"), "traffic", r.get("traffic"), "parity", p.get("value"), (p.get("index_match") or {}))
PY
done
cd /tmp && export TMPDIR=/tmp
for spec in "nlvr bf16" "nlvr f16x3" "vqa bf16" "retrieval bf16" "clip bf16"; do
  set -- $spec; C=$1; P=$2; TAG=${C}_${P}
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o r02 -- python $GRAFT_REPO_ROOT/bench.py --config $C --precision $P --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
  DB=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -name "*_results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --config $C --precision $P --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events (7 forwards incl. warm-up)" > $GRAFT_REPO_ROOT/gpurun_out/r02_f_${TAG}_kernel_stats.txt
  if [ "$TAG" = "nlvr_bf16" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB patchify 1 1 320 > $GRAFT_REPO_ROOT/gpurun_out/r02_f_timeline_nlvr.txt; python $GRAFT_REPO_ROOT/tools/rocpd_step.py $DB; fi
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
done
cd $GRAFT_REPO_ROOT
head -24 gpurun_out/r02_f_nlvr_bf16_kernel_stats.txt | cut -c1-160
