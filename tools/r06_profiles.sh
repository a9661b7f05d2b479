# round-6 measurement set (one gpurun call): GPU tests + smoke, the driver's own bench command, bench lines of the four configurations
# (index_match at the FULL BASELINE batch of each), kernel stats per configuration (serial loop: per-kernel durations undisturbed),
# the headline timeline, the in-flight overlap trace, HBM counters / L2 hit rates / SQ (MFMA-busy) counters per kernel, the CU-mask
# A/B, latency table, retrieval / caption / training-step timings  ->  gpurun_out/r06_*  (copied to profiles/ afterwards).
# usage: bash tools/r06_profiles.sh [tag]
TAG=${1:-r06}
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/${TAG}_bench_driver_cmd.err | tail -1 > gpurun_out/${TAG}_bench_driver_cmd.json
for C in nlvr retrieval clip vqa; do
  python bench.py --config $C 2>gpurun_out/${TAG}_bench_$C.err | tail -1 > gpurun_out/${TAG}_bench_$C.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_$C.json")); r=d.get("roofline") or {}; p=d.get("parity_mode") or {}
print("$C", d["value"], d["dtype"], d["ms_per_step"], (d.get("single_stream") or {}).get("value"), "f16", d.get("f16_value"), "frac", r.get("frac"), "mfma_busy", r.get("mfma_busy"), "traffic", r.get("traffic"), "parity", p.get("value"), (p.get("index_match") or {}))
PY
done
python tools/retrieval_bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_retrieval_evaluate.json; cut -c1-300 gpurun_out/${TAG}_retrieval_evaluate.json
python tools/caption_bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_caption_bench.json; cut -c1-300 gpurun_out/${TAG}_caption_bench.json
python tools/latency_table.py f16 > gpurun_out/${TAG}_latency_table_f16.txt 2>&1
{ for m in fp32 f16x3; do MADTP_TRAIN_PRECISION=$m python tools/train_step_bench.py 64 2>&1 | grep "^B="; done; } > gpurun_out/${TAG}_train_step.txt
bash tools/cumask_ab.sh gpurun_out/${TAG}_cumask_ab.txt f16 96 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "nlvr f16" "nlvr bf16" "nlvr f16x3" "vqa bf16" "retrieval bf16" "clip bf16"; do
  set -- $spec; C=$1; P=$2; T=${C}_${P}
  CMD="bench.py --config $C --precision $P --inflight 1 --steps 5 --warmup 2 --min-seconds 0 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o p -- python $R/$CMD > $R/gpurun_out/prof_$T.log 2>&1
  DB=$(find $R/gpurun_out/prof_$T -name "*_results.db" | head -1)
  MADTP_STATS_SKIP=1 python $R/tools/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python $CMD (12 forwards: warm-up 2, one untimed pass of 5, the timed 5)" > $R/gpurun_out/${TAG}_${T}_kernel_stats.txt
  if [ "$T" = "nlvr_f16" ]; then python $R/tools/rocpd_timeline.py $DB patchify 1 1 330 > $R/gpurun_out/${TAG}_timeline_nlvr.txt; python $R/tools/rocpd_step.py $DB; fi
  rm -rf $R/gpurun_out/prof_$T
done
# four forwards in flight: who overlaps whom
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_inflight -o p -- python $R/bench.py --precision f16 --steps 24 --warmup 3 --min-seconds 0 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > $R/gpurun_out/prof_inflight.log 2>&1
python $R/tools/rocpd_overlap.py $(find $R/gpurun_out/prof_inflight -name "*_results.db" | head -1) 400 0.8 > $R/gpurun_out/${TAG}_inflight_overlap.txt
rm -rf $R/gpurun_out/prof_inflight
# counters per kernel (separate passes, MI355X_MICROARCH.md): HBM-side bytes, L2 hit / miss, SQ (MFMA-busy, wave-cycle split)
CMD="bench.py --precision f16 --inflight 1 --steps 2 --warmup 1 --min-seconds 0 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d $R/gpurun_out/pmc_$CTR -o p -- python $R/$CMD > $R/gpurun_out/pmc_$CTR.log 2>&1
done
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/gpurun_out/pmc_L2 -o p -- python $R/$CMD > $R/gpurun_out/pmc_L2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_sq1 -o p -- python $R/$CMD > $R/gpurun_out/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_sq2 -o p -- python $R/$CMD > $R/gpurun_out/pmc_sq2.log 2>&1
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes) -- python $CMD   (5 forwards)";
  echo "# read bytes = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of 16-B/lane streams at 64 B, MI355X_MICROARCH.md); durations are those of the counter pass";
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find $R/gpurun_out/pmc_WRITE_SIZE -name "*_results.db" | head -1); } > $R/gpurun_out/${TAG}_pmc_per_kernel.txt
{ echo "# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -- python $CMD   (5 forwards)";
  python $R/tools/rocpd_l2.py $(find $R/gpurun_out/pmc_L2 -name "*_results.db" | head -1); } > $R/gpurun_out/${TAG}_l2_per_kernel.txt
{ echo "# rocprofv3 --pmc <SQ set 1 | SQ set 2> --kernel-trace (separate passes) -- python $CMD   (5 forwards; serial loop)";
  python $R/tools/rocpd_sq.py $(find $R/gpurun_out/pmc_sq1 -name "*_results.db" | head -1) $(find $R/gpurun_out/pmc_sq2 -name "*_results.db" | head -1); } > $R/gpurun_out/${TAG}_mfma_busy.txt
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE $R/gpurun_out/pmc_L2 $R/gpurun_out/pmc_sq1 $R/gpurun_out/pmc_sq2
cd $R
head -24 gpurun_out/${TAG}_nlvr_f16_kernel_stats.txt | cut -c1-160
head -12 gpurun_out/${TAG}_mfma_busy.txt | cut -c1-200
cat gpurun_out/${TAG}_cumask_ab.txt
