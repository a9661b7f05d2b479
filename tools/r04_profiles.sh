# round-4 measurement set: GPU tests, bench lines of the four configurations, kernel stats per configuration (serial loop, so
# that per-kernel durations are undisturbed), one headline timeline, the two-forwards-in-flight overlap trace, HBM counters per
# kernel  ->  gpurun_out/r03_*  (copied to profiles/ afterwards).   usage: bash tools/r03_profiles.sh [tag]
TAG=${1:-r04}
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
for C in nlvr retrieval clip vqa; do
  python bench.py --config $C 2>gpurun_out/${TAG}_bench_$C.err | tail -1 > gpurun_out/${TAG}_bench_$C.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_$C.json")); r=d.get("roofline") or {}; p=d.get("parity_mode") or {}
print("$C", d["value"], d["ms_per_step"], (d.get("single_stream") or {}).get("value"), "frac", r.get("frac"), "traffic", r.get("traffic"), "parity", p.get("value"), (p.get("index_match") or {}))
PY
done
python tools/retrieval_bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_retrieval_evaluate.json; cut -c1-300 gpurun_out/${TAG}_retrieval_evaluate.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "nlvr f16" "nlvr bf16" "nlvr f16x3" "vqa bf16" "vqa f16x3" "retrieval bf16" "clip bf16"; do
  set -- $spec; C=$1; P=$2; T=${C}_${P}
  CMD="bench.py --config $C --precision $P --inflight 1 --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o p -- python $R/$CMD > $R/gpurun_out/prof_$T.log 2>&1
  DB=$(find $R/gpurun_out/prof_$T -name "*_results.db" | head -1)
  python $R/tools/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python $CMD (7 forwards incl. warm-up)" > $R/gpurun_out/${TAG}_${T}_kernel_stats.txt
  if [ "$T" = "nlvr_f16" ]; then python $R/tools/rocpd_timeline.py $DB patchify 1 1 330 > $R/gpurun_out/${TAG}_timeline_nlvr.txt; python $R/tools/rocpd_step.py $DB; fi
  rm -rf $R/gpurun_out/prof_$T
done
# two forwards in flight: who overlaps whom
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_inflight -o p -- python $R/bench.py --steps 24 --warmup 3 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > $R/gpurun_out/prof_inflight.log 2>&1
python $R/tools/rocpd_overlap.py $(find $R/gpurun_out/prof_inflight -name "*_results.db" | head -1) 400 0.8 > $R/gpurun_out/${TAG}_inflight_overlap.txt
rm -rf $R/gpurun_out/prof_inflight
# HBM-side counters per kernel (separate passes, MI355X_MICROARCH.md)
CMD="bench.py --inflight 1 --steps 2 --warmup 1 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d $R/gpurun_out/pmc_$CTR -o p -- python $R/$CMD > $R/gpurun_out/pmc_$CTR.log 2>&1
done
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes) -- python $CMD   (3 forwards)";
  echo "# read bytes = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of 16-B/lane streams at 64 B, MI355X_MICROARCH.md); durations are those of the counter pass";
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find $R/gpurun_out/pmc_WRITE_SIZE -name "*_results.db" | head -1); } > $R/gpurun_out/${TAG}_pmc_per_kernel.txt
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
cd $R
head -24 gpurun_out/${TAG}_nlvr_f16_kernel_stats.txt | cut -c1-160
head -30 gpurun_out/${TAG}_pmc_per_kernel.txt | cut -c1-170
