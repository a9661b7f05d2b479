#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for b in 1 8; do
python $R/bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-parity --traffic off --no-gemm-events 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('B=$b', d['value'], d['ms_per_step'])"
done
rocprofv3 --kernel-trace --stats -d /tmp/prof_b1 -o r02 -- python $R/bench.py --batch 1 --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_b1.log 2>&1
DB=$(find /tmp/prof_b1 -name "*_results.db" | head -1)
python $R/tools/rocpd_stats.py $DB "b1" | head -14 | cut -c1-190
python $R/tools/rocpd_step.py $DB
python $R/tools/rocpd_timeline.py $DB patchify 2 1 330 > $R/gpurun_out/r02_b1_timeline.txt
