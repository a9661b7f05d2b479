# Where does a `rocprofv3 --pmc` child of bench.py hang (it does in some runs, until the parent's timeout)?  The profiled python is
# started under `timeout -s ABRT` with faulthandler on, so a hang leaves the Python stack of every thread in the log.
cd /tmp; export TMPDIR=/tmp PYTHONFAULTHANDLER=1 MADTP_BENCH_WATCHDOG=25
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-6}
for i in $(seq 1 $N); do
 for FILTER in none gemm; do
  rm -rf /tmp/hp; s=$(date +%s)
  if [ $FILTER = gemm ]; then RX="--kernel-include-regex gemm_(ws|pp|sq)_kernel"; else RX=""; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace $RX -d /tmp/hp -o p -- timeout -s ABRT 45 python3 $R/bench.py --config nlvr --precision bf16 --steps 1 --warmup 1 --traffic off --min-seconds 0 --no-cpu-baseline --no-parity --no-bf16-leg --no-gemm-events --inflight 1 ${EXTRA_ARGS} > /tmp/hp.out 2> /tmp/hp.err
  rc=$?; e=$(date +%s)
  echo "run $i filter=$FILTER rc=$rc $((e - s)) s json=$(grep -c '^{' /tmp/hp.out) pmc_rows=$(python3 -c "
import sqlite3,glob
f=glob.glob('/tmp/hp/**/*_results.db',recursive=True)
print(sqlite3.connect(f[0]).execute('select count(*), count(distinct name) from pmc_events').fetchall() if f else None)" 2>/dev/null)"
  if [ $rc -ne 0 ]; then grep -v "^W2026\|^E2026" /tmp/hp.err | grep -A6 "most recent" | head -8 | cut -c1-160; fi
 done
done
