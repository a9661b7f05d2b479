#!/usr/bin/env python
"""Long-sequence self-attention with score side outputs (attn_bf16_large_kernel): time per launch against the batch, i.e. against
the number of waves per SIMD (B x ceil(N/64) workgroups of 4 waves on 256 CUs).  Runs on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import hip
hip.load()
H = 12
for N in (320, 420, 577, 901):
    for B in (8, 16, 32, 64, 128):
        if B * N > 70000: continue
        qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
        q, k, v = qkv[:, :H * 64], qkv[:, H * 64:2 * H * 64], qkv[:, 2 * H * 64:]
        for _ in range(3): hip.attention(q, k, v, B, H, N, N, 0.125, scores=True)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): hip.attention(q, k, v, B, H, N, N, 0.125, scores=True)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        wgs = B * ((N + 63) // 64)
        steps = H * 2 * ((N + 127) // 128)
        print(f"N={N:4d} B={B:4d} workgroups={wgs:5d} {us:8.1f} us  {us / steps:6.2f} us per chunk step  {6.0 * B * H * N * N * 64 / us / 1e6:7.1f} TF (QK twice + PV)", flush=True)
