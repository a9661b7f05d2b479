#!/usr/bin/env python
"""Timing ablations of gemm_pp_kernel (library built with -DMADTP_PP_ABLATIONS; MADTP_PP_ABLATE selects the variant per process)."""
import os, sys, subprocess
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from madtp_amd import hip
    hip.load()
    td = torch.bfloat16
    out_line = f"ABL={os.environ.get('MADTP_PP_ABLATE', '0'):>2s}"
    for M, N, K in ((4096, 4096, 4096), (17152, 3072, 768), (12288, 2304, 768), (25216, 2304, 768)):
        a = torch.randn(M, K, device="cuda").to(td); w = (torch.randn(N, K, device="cuda") * 0.05).to(td)
        out = torch.empty(M, N, device="cuda", dtype=td)
        with hip.gemm_config(9):
            for _ in range(3): hip.gemm(a, w, None, n=N, out=out)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): hip.gemm(a, w, None, n=N, out=out)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out_line += f"  {M}x{N}x{K} {us:7.1f} us"
    print(out_line, flush=True)
    sys.exit(0)
print("ABL bits: 1 no steady DMA, 2 no MFMA, 4 no steady fragment reads, 8 DMA issued at the head of the MFMA part, 16 no setprio")
for abl in (0, 1, 2, 4, 3, 5, 6, 8, 16, 0):
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MADTP_PP_ABLATE=str(abl)))
