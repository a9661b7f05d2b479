#!/bin/bash
# Rehearsal of the driver's multi-GPU launch on a ONE-GPU box: the exact torchrun command line with 2 ranks that share GPU 0
# (MADTP_BENCH_ONE_GPU=1: gloo instead of RCCL).  Checks that every rank gets through build / pinning / in-flight runner /
# barriers / reductions and that rank 0 prints one JSON line; the throughput it prints is meaningless (two ranks on one GPU).
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export MADTP_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for C in nlvr retrieval; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 \
      --config $C --steps 8 --warmup 2 --parity-steps 8 2>gpurun_out/rehearsal_$C.err | tail -1 > gpurun_out/rehearsal_$C.json
  python - <<PY
import json
d = json.load(open("gpurun_out/rehearsal_$C.json"))
assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2", d
print("$C", "n_gpus", d["n_gpus"], "value", d["value"], "scaling", d["scaling"], "parity", (d.get("parity_mode") or {}).get("value"), "keys", sorted(d)[:6])
PY
done
