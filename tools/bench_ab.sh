# A/B of bench.py's headline leg under environment variants, same box, interleaved rounds.
#   usage: bash tools/bench_ab.sh OUT.txt ROUNDS "NAME1:ENV1 ENV2" "NAME2:..." ...   (":" alone = no extra environment)
OUT=$1; ROUNDS=$2; shift 2
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for spec in "$@"; do
    NAME=${spec%%:*}; ENVS=${spec#*:}
    LINE=$(env $ENVS python bench.py --no-parity --no-cpu-baseline --traffic off ${BENCH_ARGS:-} 2>/dev/null | tail -1)
    python - "$NAME" "$r" "$LINE" >> $OUT <<'PY'
import json, sys
name, r, line = sys.argv[1:4]
try:
    d = json.loads(line); rf = d.get("roofline") or {}
    print(f"round {r} {name:28s} value {d['value']:8.1f}  serial {d.get('serial_value')}  frac {rf.get('frac')}  avg_launch_us {rf.get('avg_launch_us')}  gemm_ms/step {rf.get('kernel_ms_per_step')}  all_gemm_ms {(rf.get('all_gemm_launches') or {}).get('ms_per_step')}")
except Exception as e:
    print(f"round {r} {name}: failed ({e!r}): {line[-200:]}")
PY
  done
done
cat $OUT
