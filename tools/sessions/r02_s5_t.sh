#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -3
for hs in 0 1; do echo HEAD_SPLIT=$hs; MADTP_ATTN_HEAD_SPLIT=$hs python tools/attn_large_bench.py 2>&1 | grep "^N=" | grep -E "B= +(8|16|32) "; done
for hs in 0 1 0 1; do MADTP_ATTN_HEAD_SPLIT=$hs python bench.py --config vqa --steps 10 --warmup 3 --no-cpu-baseline --no-parity --traffic off 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('HEAD_SPLIT=$hs', d['value'], d['ms_per_step'])"; done
timeout 900 python -m pytest tests/test_model_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -2
