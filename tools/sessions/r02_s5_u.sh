#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -3
for hs in 0 1 0 1; do MADTP_ATTN_HEAD_SPLIT=$hs python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('f16x3 HEAD_SPLIT=$hs', d['value'], d['ms_per_step'])"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('parity_mode') or {}).get('value'), (d.get('parity_mode') or {}).get('index_match'))"
