#!/usr/bin/env python
"""Microbenchmark of madtp_gemm on the shapes of the NLVR forward (runs on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import hip, runtime
hip.load()
if os.environ.get("MADTP_PRECISION"):  # "f16": the same 2-byte containers hold IEEE f16 (random bits either way for timing)
    runtime.set_precision(os.environ["MADTP_PRECISION"])
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
if len(sys.argv) > 2 and sys.argv[2] == "ab":  # 256x128 wave-specialised (cfg 5) vs 256x256 (cfg 6) vs automatic on the forward's ViT shapes
    rows = [25216, 17152, 14336, 12416, 12288, 11776, 11136, 10752, 10496]  # the headline forward's row counts (128 images x tokens per layer) + 12288
    if len(sys.argv) > 3 and sys.argv[3] == 'small_tiles': rows = [17152, 12288, 10752, 10496]
    for M in rows:
        for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
            a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
            bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
            kw = dict(residual=res, out_dtype=torch.float32) if N == 768 else (dict(act=hip.ACT_GELU) if N == 3072 else {})
            out = torch.empty(M, N, device="cuda", dtype=kw.get("out_dtype", dt))
            line = f"M={M:6d} N={N:5d} K={K:5d}"
            for cfg in ((7, 1, 2, 3, 0) if (len(sys.argv) > 3 and sys.argv[3] == "small_tiles") else (7, 9, 10, 0)):  # 9: 256x256 ping-pong; 7: wave-specialised, 16x16x32 MFMA; 8: wave-specialised, 32x32x16 MFMA; 6: 256x256; 0: automatic (5 = 7 + stream-K tail)
                with hip.gemm_config(cfg):
                    for _ in range(3): hip.gemm(a, w, bias, n=N, out=out, **kw)
                    torch.cuda.synchronize()
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20): hip.gemm(a, w, bias, n=N, out=out, **kw)
                    e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / 20
                line += f"   cfg{cfg} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF"
            print(line, flush=True)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "cube":  # long-K yardstick: 4096^3 / 8192^3, every big-tile kernel and the vendor library, random operands
    for M in (4096, 8192):
        a = torch.randn(M, M, device="cuda").to(dt); w = (torch.randn(M, M, device="cuda") * 0.05).to(dt)
        out = torch.empty(M, M, device="cuda", dtype=dt); lib_out = torch.empty(M, M, device="cuda", dtype=dt)
        def t(fn, reps=10):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        line = f"M=N=K={M}"
        for rnd in range(2):
            for cfg in (7, 6, 9):
                with hip.gemm_config(cfg):
                    us = t(lambda: hip.gemm(a, w, None, n=M, out=out))
                line += f"   cfg{cfg} {us:8.1f} us {2.0*M*M*M/us/1e6:7.1f} TF"
            us = t(lambda: torch.matmul(a, w.t(), out=lib_out))
            line += f"   library {us:8.1f} us {2.0*M*M*M/us/1e6:7.1f} TF |"
        print(line, flush=True)
        print("   max|pp - lib| =", (out.float() - lib_out.float()).abs().max().item(), flush=True)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "lib":  # yardstick: the vendor library (torch -> hipBLASLt / rocBLAS) on the same shapes, plain bf16 out
    import torch.nn.functional as F
    rows = [25216, 12288, 10496]
    for M in rows + [8192]:
        for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)) if M != 8192 else ((8192, 8192),):
            a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
            bias = torch.randn(N, device="cuda")
            out = torch.empty(M, N, device="cuda", dtype=dt)
            lib_out = torch.empty(M, N, device="cuda", dtype=dt)
            line = f"M={M:6d} N={N:5d} K={K:5d}"
            def t(fn):
                for _ in range(3): fn()
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / 20
            us = t(lambda: hip.gemm(a, w, None, n=N, out=out))
            line += f"   madtp {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF"
            us = t(lambda: torch.matmul(a, w.t(), out=lib_out))
            line += f"   library {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF"
            wt = w.t().contiguous()
            us = t(lambda: torch.matmul(a, wt, out=lib_out))
            line += f"   library(NN) {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF"
            print(line, flush=True)
    sys.exit(0)
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
