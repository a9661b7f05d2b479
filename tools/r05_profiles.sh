# round-5 measurement set (one gpurun call): GPU tests + smoke, bench lines of the four configurations, kernel stats per
# configuration (serial loop, so that per-kernel durations are undisturbed), the headline timeline, the in-flight overlap trace, HBM
# counters and L2 hit rates per kernel, GEMM A/B table, latency table, training-step timings  ->  gpurun_out/r05_*  (copied to
# profiles/ afterwards).   usage: bash tools/r05_profiles.sh [tag]
TAG=${1:-r05}
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -2
for C in nlvr retrieval clip vqa; do
  python bench.py --config $C 2>gpurun_out/${TAG}_bench_$C.err | tail -1 > gpurun_out/${TAG}_bench_$C.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_$C.json")); r=d.get("roofline") or {}; p=d.get("parity_mode") or {}
print("$C", d["value"], d["ms_per_step"], (d.get("single_stream") or {}).get("value"), "bf16", d.get("bf16_value"), "frac", r.get("frac"), "traffic", r.get("traffic"), "parity", p.get("value"), (p.get("index_match") or {}))
PY
done
python tools/retrieval_bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_retrieval_evaluate.json; cut -c1-300 gpurun_out/${TAG}_retrieval_evaluate.json
python tools/caption_bench.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_caption_bench.json; cut -c1-300 gpurun_out/${TAG}_caption_bench.json
python tools/gemm_bench.py bf16 ab > gpurun_out/${TAG}_gemm_pp_ab.txt 2>&1
python tools/latency_table.py f16 > gpurun_out/${TAG}_latency_table_f16.txt 2>&1
{ for m in fp32 f16x3; do MADTP_TRAIN_PRECISION=$m python tools/train_step_bench.py 4 16 64 2>&1 | grep "^B="; done;
  echo "# model.train(): dropout 0.1 + DropPath (counter-based masks)"; MADTP_TRAIN_DROPOUT=1 MADTP_TRAIN_PRECISION=f16x3 python tools/train_step_bench.py 64 2>&1 | grep "^B=";
  echo "# MADTP_TRAIN_SAVE=0: fused layer calls + recompute in the backward (the round-4 scheme)"; MADTP_TRAIN_SAVE=0 MADTP_TRAIN_PRECISION=f16x3 python tools/train_step_bench.py 64 2>&1 | grep "^B=";
  echo "# MADTP_TRAIN_FUSED_ADAM=1: torch.optim.AdamW(fused=True) instead of the default foreach implementation"; MADTP_TRAIN_FUSED_ADAM=1 MADTP_TRAIN_PRECISION=f16x3 python tools/train_step_bench.py 64 2>&1 | grep "^B="; } > gpurun_out/${TAG}_train_step.txt
{ for hv in 0 1; do echo "MADTP_ATTN_HV=$hv"; MADTP_ATTN_HV=$hv python tools/attn_large_bench.py 2>&1 | grep "N= 901\|N= 577"; done; } > gpurun_out/${TAG}_attn_large_hv_ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "nlvr f16" "nlvr bf16" "nlvr f16x3" "vqa bf16" "vqa f16x3" "retrieval bf16" "clip bf16"; do
  set -- $spec; C=$1; P=$2; T=${C}_${P}
  CMD="bench.py --config $C --precision $P --inflight 1 --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o p -- python $R/$CMD > $R/gpurun_out/prof_$T.log 2>&1
  DB=$(find $R/gpurun_out/prof_$T -name "*_results.db" | head -1)
  MADTP_STATS_SKIP=1 python $R/tools/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python $CMD (7 forwards)" > $R/gpurun_out/${TAG}_${T}_kernel_stats.txt
  if [ "$T" = "nlvr_f16" ]; then python $R/tools/rocpd_timeline.py $DB patchify 1 1 330 > $R/gpurun_out/${TAG}_timeline_nlvr.txt; python $R/tools/rocpd_step.py $DB; fi
  rm -rf $R/gpurun_out/prof_$T
done
# training step in the f16x3 mode: where the kernel time goes
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o p -- env MADTP_TRAIN_PRECISION=f16x3 python $R/tools/train_step_bench.py 64 > $R/gpurun_out/prof_train.log 2>&1
MADTP_STATS_SKIP=6 python $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_train -name "*_results.db" | head -1) "rocprofv3 --kernel-trace --stats -- MADTP_TRAIN_PRECISION=f16x3 python tools/train_step_bench.py 64 (the 3 timed training steps)" > $R/gpurun_out/${TAG}_train_step_kernel_stats.txt
rm -rf $R/gpurun_out/prof_train
# caption generation (incremental decoding + device-side beam search): the kernels of the timed generate calls
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cap -o p -- python $R/tools/caption_bench.py > $R/gpurun_out/prof_cap.log 2>&1
MADTP_STATS_SKIP=1 python $R/tools/rocpd_stats.py $(find $R/gpurun_out/prof_cap -name "*_results.db" | head -1) "rocprofv3 --kernel-trace --stats -- python tools/caption_bench.py (the generate calls after the first)" > $R/gpurun_out/${TAG}_caption_kernel_stats.txt
rm -rf $R/gpurun_out/prof_cap
# four forwards in flight: who overlaps whom
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_inflight -o p -- python $R/bench.py --steps 24 --warmup 3 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > $R/gpurun_out/prof_inflight.log 2>&1
python $R/tools/rocpd_overlap.py $(find $R/gpurun_out/prof_inflight -name "*_results.db" | head -1) 400 0.8 > $R/gpurun_out/${TAG}_inflight_overlap.txt
rm -rf $R/gpurun_out/prof_inflight
# HBM-side counters per kernel (separate passes, MI355X_MICROARCH.md) + one pass of L2 hit / miss / fabric read requests
CMD="bench.py --inflight 1 --steps 2 --warmup 1 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $CTR --kernel-trace -d $R/gpurun_out/pmc_$CTR -o p -- python $R/$CMD > $R/gpurun_out/pmc_$CTR.log 2>&1
done
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $R/gpurun_out/pmc_L2 -o p -- python $R/$CMD > $R/gpurun_out/pmc_L2.log 2>&1
{ echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes) -- python $CMD   (3 forwards)";
  echo "# read bytes = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of 16-B/lane streams at 64 B, MI355X_MICROARCH.md); durations are those of the counter pass";
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/pmc_FETCH_SIZE -name "*_results.db" | head -1) $(find $R/gpurun_out/pmc_WRITE_SIZE -name "*_results.db" | head -1); } > $R/gpurun_out/${TAG}_pmc_per_kernel.txt
{ echo "# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -- python $CMD   (3 forwards)";
  python $R/tools/rocpd_l2.py $(find $R/gpurun_out/pmc_L2 -name "*_results.db" | head -1); } > $R/gpurun_out/${TAG}_l2_per_kernel.txt
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE $R/gpurun_out/pmc_L2
cd $R
head -24 gpurun_out/${TAG}_nlvr_f16_kernel_stats.txt | cut -c1-160
head -12 gpurun_out/${TAG}_l2_per_kernel.txt | cut -c1-170
cat gpurun_out/${TAG}_train_step.txt
