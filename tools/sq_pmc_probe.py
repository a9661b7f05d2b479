"""One launch set per configuration (5 = 256x128 wave-specialised, 6 / 7 = 256x256 kernels) on three ViT shapes, for rocprofv3 --pmc
passes (L2 hit rate, fabric bytes) - see tools/r02_s4_e.sh."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load()
dt = torch.bfloat16
for M, N, K in ((14208, 2304, 768), (25216, 3072, 768), (17152, 768, 3072)):
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for cfg in (5, 6):
        with hip.gemm_config(cfg):
            for _ in range(4): hip.gemm(a, w, bias, n=N, out=out)
    torch.cuda.synchronize()
