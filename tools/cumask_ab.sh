#!/bin/bash
# Round 6, review item 1: spatial partition of the in-flight forwards (CU-masked streams, one slice of every XCD per worker)
# against round 5's priority time-slicing, same box, interleaved.  Usage: tools/cumask_ab.sh [out-file] [precision] [steps]
OUT=${1:-gpurun_out/r06_cumask_ab.txt}
PREC=${2:-f16}
STEPS=${3:-96}
mkdir -p "$(dirname "$OUT")"
run() {  # label, inflight, partition, extra env
  local label=$1 nf=$2 part=$3; shift 3
  local line
  line=$(env "$@" MADTP_INFLIGHT_CUMASK="$part" python bench.py --precision "$PREC" --steps "$STEPS" --warmup 3 --inflight "$nf" \
         --no-parity --no-cpu-baseline --no-gemm-events --traffic off 2>/dev/null | tail -1)
  python - "$label" "$line" >> "$OUT" <<'PY'
import json, sys
label, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    c = d["config"]
    print(f"{label:44s} value {d['value']:9.1f} images/s  ms/step {d['ms_per_step']:7.3f}  serial {d.get('serial_value', 0):9.1f}  "
          f"part {c.get('inflight_cus_per_xcd')} high {c.get('inflight_high_priority_streams')} shared {c.get('inflight_workers_share_weights')} "
          f"hbm {c['hbm_bytes_allocated'] / 2**30:.2f} GiB")
except Exception as e:
    print(f"{label:44s} FAILED {e!r}: {line[-300:]}")
PY
}
echo "# $(date -u) precision $PREC steps $STEPS (bench.py --no-parity; value = in-flight throughput, serial = one forward at a time)" >> "$OUT"
for rep in 1 2; do
  run "4 prio (h,h,n,n), 4 replicas [round 5]" 4 off MADTP_INFLIGHT_SHARE=0
  run "4 prio (h,h,n,n), shared weights" 4 off
  run "4 x 8 CUs/XCD" 4 8,8,8,8
  run "2 x 16 CUs/XCD" 2 16,16
  run "3 x (11,11,10)" 3 11,11,10
  run "4 x (12,12,4,4)" 4 12,12,4,4
  run "4 x (10,10,6,6)" 4 10,10,6,6
  run "2 prio (h,n), shared" 2 off
done
cat "$OUT"
