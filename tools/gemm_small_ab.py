#!/usr/bin/env python
"""The 1280-row GEMMs of the text encoder (64 samples x 20 tokens) under every small-tile configuration (round 5: the deep-ring
variants - 64x64 tiles x 6 stages, 64x128 x 5, one workgroup per CU - were measured with this tool and dropped):
40 distinct weights per shape (HBM / MALL-cold like the model's), event-timed back-to-back launches (GPU-bound: >= 6 us each).
usage: python tools/gemm_small_ab.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import hip, runtime
hip.load()
runtime.set_precision(os.environ.get("MADTP_PRECISION", "f16"))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
dt = torch.bfloat16


def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


for N, K, kind, S in ((2304, 768, "lp", 1), (768, 768, "f32res", 1), (1536, 768, "lp", 1), (3072, 768, "gelu", 1),
                      (768, 768, "splitk_ln", 2), (768, 1536, "splitk_ln", 2), (768, 3072, "splitk_ln", 3), (768, 3072, "splitk_ln", 4)):
    ws = [hip.cast_lp_weight(torch.randn((N + 127) // 128 * 128, K, device="cuda") * 0.05) for _ in range(40)]
    a = hip.cast_bf16(torch.randn(M, K, device="cuda"))
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda")
    g, b = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    line = f"M={M} N={N:5d} K={K:5d} {kind:9s} S={S}"
    for cfg in (0, 4, 2, 3, 1):
        def run():
            for w in ws:
                if kind == "lp": hip.gemm(a, w, bias, n=N)
                elif kind == "gelu": hip.gemm(a, w, bias, n=N, act=hip.ACT_GELU)
                elif kind == "f32res": hip.gemm(a, w, bias, res, out_dtype=torch.float32, n=N)
                else: hip.gemm_splitk_ln(a, w, bias, res, g, b, 1e-12, S, N, want_bf16=True)
        with hip.gemm_config(cfg):
            try:
                us = timeit(run, len(ws))
            except Exception as e:
                us = float("nan")
        line += f"  cfg{cfg} {us:6.1f}"
    print(line + "   (us per call" + ("; incl. the splitk_ln pass" if kind == "splitk_ln" else "") + ")", flush=True)
