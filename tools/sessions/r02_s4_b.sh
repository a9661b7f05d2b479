# bf16 long-sequence attention: parity tests, then the VQA config and the 384x384 retrieval pieces
set -x
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -6
python -m pytest tests/test_model_parity_gpu.py tests/test_workloads_gpu.py -m gpu -x -q 2>&1 | tail -6
python bench.py --config vqa --steps 10 --traffic off --no-cpu-baseline --no-parity --no-gemm-events 2>gpurun_out/s4_vqa.err | tail -1 > gpurun_out/s4_vqa.json
cut -c1-300 gpurun_out/s4_vqa.json; tail -3 gpurun_out/s4_vqa.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_vqa -o r02 -- python $GRAFT_REPO_ROOT/bench.py --config vqa --precision bf16 --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events > $GRAFT_REPO_ROOT/gpurun_out/prof_vqa.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $GRAFT_REPO_ROOT/gpurun_out/prof_vqa -name "*_results.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --config vqa --precision bf16 --steps 5 --warmup 2 --traffic off --no-cpu-baseline --no-parity --no-gemm-events (7 forwards incl. warm-up)" > $GRAFT_REPO_ROOT/gpurun_out/s4_kernel_stats_vqa.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_vqa
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/s4_kernel_stats_vqa.txt | head -24
