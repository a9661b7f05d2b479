set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d /tmp/p1 -o p -- python $GRAFT_REPO_ROOT/tools/sq_pmc_probe.py > /tmp/p1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/p2 -o p -- python $GRAFT_REPO_ROOT/tools/sq_pmc_probe.py > /tmp/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_dump.py $(find /tmp/p1 -name "*_results.db" | head -1) > gpurun_out/s4_pmc_hit.txt 2>&1
python tools/pmc_dump.py $(find /tmp/p2 -name "*_results.db" | head -1) > gpurun_out/s4_pmc_fetch.txt 2>&1
tail -5 /tmp/p1.log
cat gpurun_out/s4_pmc_hit.txt | awk 'NR%4==0'
cat gpurun_out/s4_pmc_fetch.txt | awk 'NR%4==0'
