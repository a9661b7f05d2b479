#!/bin/bash
# Rehearsal of the driver's multi-GPU launch on a ONE-GPU box: the exact torchrun command line with N ranks (default 2; the
# driver's node has 8) that share GPU 0 (MADTP_BENCH_ONE_GPU=1: gloo instead of RCCL).  Checks that every rank gets through
# build / core pinning / in-flight runner / barriers / reductions with N x (4 workers + main) host threads on one box, and that
# rank 0 prints one JSON line; per_rank_ms_per_step min / max show stragglers.  The throughput it prints is meaningless (N ranks on
# one GPU).   usage: bash tools/bench_rehearsal.sh [ranks] [configs...]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export MADTP_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-2}; shift
CONFIGS=${@:-nlvr retrieval}
mkdir -p gpurun_out
for C in $CONFIGS; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus $N \
      --config $C --steps 8 --warmup 2 --parity-steps 8 --bf16-steps 8 2>gpurun_out/rehearsal_${C}_n$N.err | tail -1 > gpurun_out/rehearsal_${C}_n$N.json
  python - <<PY
import json
d = json.load(open("gpurun_out/rehearsal_${C}_n$N.json"))
assert d["n_gpus"] == $N and d["config"]["parallelism"] == "dp$N", d
pr = d["per_rank_ms_per_step"]
print("$C", "ranks", d["n_gpus"], "value", d["value"], "scaling", d["scaling"], "per_rank_ms_per_step min/max", pr["min"], pr["max"],
      "parity", (d.get("parity_mode") or {}).get("value"), "bf16", d.get("bf16_value"))
PY
done
