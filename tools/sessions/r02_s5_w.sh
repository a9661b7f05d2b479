#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o r02 -- python $R/bench.py --config vqa --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_v.log 2>&1
DB=$(find /tmp/prof_v -name "*_results.db" | head -1)
python $R/tools/rocpd_stats.py $DB "vqa" | head -16 | cut -c1-200
python $R/tools/rocpd_step.py $DB
