#!/bin/bash
# Round 6, review item 2: MFMA-busy / issue-stall counters per kernel of the headline forward (serial loop: undisturbed launches).
# Two SQ passes (8 SQ slots per pass) + GRBM_GUI_ACTIVE in each; --kernel-trace only (gpurun refuses --pmc with sys / hip traces).
# usage: bash tools/r06_mfma_busy.sh [precision] [out]
P=${1:-f16}
OUT=${2:-r06_mfma_busy_$P.txt}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_INST_ANY\|SQ_WAIT_ANY\|SQ_ACTIVE_INST_ANY\|GRBM_GUI_ACTIVE\|SQ_WAVES\b" | sort -u > $R/gpurun_out/r06_counters_available.txt
CMD="bench.py --precision $P --inflight 1 --steps 2 --warmup 1 --traffic off --no-cpu-baseline --no-parity --no-gemm-events"
timeout 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_sq1 -o p -- python $R/$CMD > $R/gpurun_out/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_sq2 -o p -- python $R/$CMD > $R/gpurun_out/pmc_sq2.log 2>&1
{ echo "# rocprofv3 --pmc <SQ set 1 | SQ set 2> --kernel-trace (separate passes) -- python $CMD   (3 forwards; serial loop)";
  echo "# counters available on this box: $(tr '\n' ' ' < $R/gpurun_out/r06_counters_available.txt)";
  python $R/tools/rocpd_sq.py $(find $R/gpurun_out/pmc_sq1 -name "*_results.db" | head -1) $(find $R/gpurun_out/pmc_sq2 -name "*_results.db" | head -1); } > $R/gpurun_out/$OUT
for f in pmc_sq1 pmc_sq2; do tail -n 3 $R/gpurun_out/$f.log | cut -c1-300; done
rm -rf $R/gpurun_out/pmc_sq1 $R/gpurun_out/pmc_sq2
head -40 $R/gpurun_out/$OUT | cut -c1-230
