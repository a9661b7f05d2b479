#!/usr/bin/env python
"""madtp_gemm on the text encoder's problem sizes against K (runs on the GPU box).  Below ~10 us per call the loop is bound by the
Python/ctypes launch path (a 1-slab problem takes the same 9.7 us as a 12-slab one), so only the long-K rows measure the kernel:
48 slabs in 20.6 us = 0.43 us per 16 KiB slab step of a lone 64x64 workgroup (not a DMA limit: tools/probes/probe_stage_rate.hip)."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from madtp_amd import hip
hip.load()
for M, N in ((1280, 768), (1280, 2304), (197, 768)):
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = (torch.randn(((N + 127) // 128) * 128, K, device="cuda") * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(5): hip.gemm(a, w, None, n=N, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): hip.gemm(a, w, None, n=N, out=out)
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} N={N} K={K:5d} slabs={K//64:3d} {e0.elapsed_time(e1)*20:7.2f} us", flush=True)
