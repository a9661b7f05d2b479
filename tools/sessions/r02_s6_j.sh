#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for C in retrieval vqa nlvr; do
rocprofv3 --kernel-trace --stats -d /tmp/prof_$C -o r02 -- python $R/bench.py --config $C --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_$C.log 2>&1
DB=$(find /tmp/prof_$C -name "*_results.db" | head -1)
python $R/tools/rocpd_timeline.py $DB patchify 2 1 800 > $R/gpurun_out/r02_h_timeline_$C.txt
python $R/tools/rocpd_step.py $DB | tail -2
done
