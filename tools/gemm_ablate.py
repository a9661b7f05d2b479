import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load(os.environ.get('MADTP_ABLATE_LIB'))
shapes = [(25216, 2304, 768), (25216, 768, 3072), (10496, 768, 768)]
dt = torch.bfloat16
for M, N, K in shapes:
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
    print(f"lib={os.path.basename(os.environ.get('MADTP_ABLATE_LIB') or 'default')} dbg={os.environ.get('MADTP_GEMM_DEBUG')} cfg={os.environ.get('MADTP_GEMM_CFG')} M={M:6d} N={N:5d} K={K:5d} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
