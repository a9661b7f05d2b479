"""Timing-only ablations of the 256x256 GEMM kernels (MADTP_SQ_ABLATE builds: bit 0 no LDS-DMA, bit 1 no MFMA, bit 2 no fragment
reads; results are wrong by construction).  One process per library: python tools/sq_ablate.py <lib.so>"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load(sys.argv[1] if len(sys.argv) > 1 else None)
dt = torch.bfloat16
for M, N, K in ((14208, 2304, 768), (25216, 3072, 768), (17152, 768, 3072)):
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dt)
    line = f"{os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'default':24s} M={M:6d} N={N:5d} K={K:5d}"
    for cfg in (6,):
        with hip.gemm_config(cfg):
            for _ in range(3): hip.gemm(a, w, bias, n=N, out=out)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): hip.gemm(a, w, bias, n=N, out=out)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        line += f"   cfg{cfg} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF"
    print(line, flush=True)
