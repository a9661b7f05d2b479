"""GPU-side durations of small GEMMs need rocprofv3 (the host-side loop is launch-bound):
   rocprofv3 --kernel-trace -d out -o x -- python tools/gemm_small_probe.py ; python tools/rocpd_stats.py out/x_results.db"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load()
dt = torch.bfloat16
M, N, K = [int(v) for v in os.environ.get("SHAPE", "1280,768,768").split(",")]
ws = [(torch.randn(N, K, device="cuda") * 0.05).to(dt) for _ in range(40)]   # distinct weights: HBM-cold like the model
a = torch.randn(M, K, device="cuda").to(dt)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=dt)
for rep in range(3):
    for w in ws:
        hip.gemm(a, w, bias, n=N, out=out)
torch.cuda.synchronize()
