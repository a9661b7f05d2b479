import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
dt = torch.bfloat16
for M, N, K in ((5120, 18432, 768), (5120, 1536, 768), (10496, 18432, 768)):
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=dt)
    for _ in range(3): hip.gemm(a, w, bias, n=N, out=out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.gemm(a, w, bias, n=N, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    ref = (a[:64].double() @ w.double().t() + bias.double())
    err = (out[:64].double() - ref).abs().max().item()
    print(f"NGRP={os.environ.get('MADTP_GEMM_NGRP')} M={M} N={N} K={K} {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF  err {err:.3f}", flush=True)
