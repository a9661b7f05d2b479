#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for sk in 0 1; do
  MADTP_GEMM_SK=$sk rocprofv3 --kernel-trace --stats -d /tmp/prof_sk$sk -o r02 -- python $R/bench.py --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > /tmp/prof_sk$sk.log 2>&1
  DB=$(find /tmp/prof_sk$sk -name "*_results.db" | head -1)
  python $R/tools/rocpd_stats.py $DB "SK=$sk" | head -12 | cut -c1-200
  python $R/tools/rocpd_timeline.py $DB patchify 1 1 130 > $R/gpurun_out/r02_sk${sk}_timeline.txt
done
