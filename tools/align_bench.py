"""Alignment-logits kernel (madtp_align_logits: x @ sd^T as three MFMA products of split operands) timed per launch over the row
counts of the headline forward, with a bit-level fingerprint of every output.  MADTP_ALIGN_ROWS=64|128 selects the tile height
(align_ws_kernel / align_ws2_kernel): run once per setting and compare the fingerprints (they must be identical: same product order
per accumulator).   usage: [MADTP_ALIGN_ROWS=128] python tools/align_bench.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madtp_amd import hip
hip.load(os.environ.get('MADTP_ABLATE_LIB'))
D = 768
g = torch.Generator().manual_seed(0)
sd = torch.randn(128, D, generator=g).cuda()
sd[100:] = 0
hi = hip.cast_bf16(sd); lo = hip.cast_bf16((sd - hi.float()).contiguous())
q = hip.split_f16_weight(sd.contiguous())          # f16x3 flavour: planes [128, 2D] of sd * 2^s
q0, q1 = q[:, :D].contiguous(), q[:, D:].contiguous()
tag = os.environ.get("MADTP_ALIGN_ROWS", "auto")
for M in (1280, 1000, 10001, 10496, 11136, 12288, 14336, 17152, 25216):
    x = torch.randn(M, D, generator=g).cuda() * 3
    for name, call in (("bf16x3", lambda: hip.align_logits(x, hi, lo)), ("f16x3", lambda: hip.align_logits(x, q0, q1, hip.w_scale_of(q)))):
        out = call()
        torch.cuda.synchronize()
        fp = int(out.view(torch.int32).to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        for _ in range(5): call()
        e0.record()
        for _ in range(50): call()
        e1.record(); torch.cuda.synchronize()
        print(f"rows={tag:>4s} {name:7s} M={M:6d} {e0.elapsed_time(e1) * 1e3 / 50:8.1f} us  fingerprint {fp:012x}", flush=True)
