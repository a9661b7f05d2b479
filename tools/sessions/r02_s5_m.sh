#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for kv in 0 1 0 1; do MADTP_KV_SIDE=$kv python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('KV_SIDE=$kv', d['value'], d['ms_per_step'])"; done
MADTP_KV_SIDE=1 timeout 600 python -m pytest tests/test_model_parity_gpu.py -m gpu -x -q -k "nlvr" 2>&1 | tail -2
