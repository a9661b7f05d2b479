#!/usr/bin/env python
"""Microbenchmark of madtp_gemm on the shapes of the NLVR forward (runs on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import hip
hip.load()
shapes = [(25216, 2304, 768), (25216, 768, 768), (25216, 3072, 768), (25216, 768, 3072),
          (10496, 2304, 768), (10496, 768, 768), (10496, 3072, 768), (10496, 768, 3072),
          (5120, 1536, 768), (1280, 2304, 768), (1280, 768, 768), (1280, 3072, 768), (1280, 768, 3072), (1280, 768, 1536),
          (1280, 128, 768), (10496, 128, 768), (4096, 4096, 4096)]
if len(sys.argv) > 2 and sys.argv[2] == "quant":  # tile-count quantisation of the N=768 problems between 10k and 18k rows
    shapes = [(M, 768, K) for M in (10496, 11008, 11648, 12160, 14208, 17152) for K in (768, 3072)]
if len(sys.argv) > 2 and sys.argv[2] == "small":
    shapes = [sh for sh in shapes if sh[0] <= 5120 or sh[1] == 128]
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
print("dtype", dt, "MADTP_GEMM_DEBUG", os.environ.get("MADTP_GEMM_DEBUG"), "CFG", os.environ.get("MADTP_GEMM_CFG"))
for M, N, K in shapes:
    if dt == torch.float32 and M * N * K > 4096**3: continue
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    for name, kw in (("lp_out", dict(out_dtype=dt)), ("f32+res", dict(out_dtype=torch.float32, residual=res)), ("gelu", dict(out_dtype=dt, act=hip.ACT_GELU))):
        out = torch.empty(M, N, device="cuda", dtype=kw.get("out_dtype"))
        for _ in range(3): hip.gemm(a, w, bias, n=N, out=out, **{k: v for k, v in kw.items() if k != "out_dtype"})
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps): hip.gemm(a, w, bias, n=N, out=out, **{k: v for k, v in kw.items() if k != "out_dtype"})
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"M={M:6d} N={N:5d} K={K:5d} {name:8s} {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
