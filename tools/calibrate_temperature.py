#!/usr/bin/env python
"""p -> temperature calibration with the CPU oracle (mirrors the reference controller
compress_nlvr_dtp.py:162-201 which steps `temperature` until fvcore GFLOPs ~= Ori_Gflops*(1-p)).

For the synthetic weights there is no trained checkpoint/temperature, so T is found by bisection such that the
analytic FLOPs of BLIP_NLVR.forward computed from the OBSERVED per-layer token counts equal (1-p) x the unpruned
FLOPs (+-2 %).  The batch matters (k = max over the local batch, vit.py:145), so calibrate at the bench batch.

usage: python tools/calibrate_temperature.py --batch 64 --p 0.5 [--size 224 --len 20 --seed 0]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madtp_amd import harness, specs, synth  # noqa: E402
from oracle import madtp_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--p", type=float, default=0.5)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--len", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lo", type=float, default=1.0)
    ap.add_argument("--hi", type=float, default=400.0)
    a = ap.parse_args()
    W = specs.synth_weights(specs.blip_nlvr_shapes(a.size), 0)
    images = synth.synth_images(2 * a.batch, a.size, a.seed)
    ids = synth.synth_token_ids(a.batch, a.len, a.seed)
    n0 = (a.size // 16) ** 2 + 1
    full = harness.nlvr_forward_flops([n0] * 12, [a.len] * 12, n0, a.len)
    target = (1 - a.p) * full

    def flops_at(T):
        tr = {}
        t0 = time.time()
        with torch.no_grad():
            O.blip_nlvr_forward(W, images, ids, torch.ones_like(ids), T, trace=tr)
        vl = harness.token_lengths(tr["vit"], n0)
        tl = harness.token_lengths(tr["text"], a.len)
        f = harness.nlvr_forward_flops(vl, tl, n0, a.len)
        print(f"T={T:.4f} ratio={f / full:.4f} vit={vl} txt={tl} ({time.time() - t0:.1f}s)", flush=True)
        return f, vl, tl

    lo, hi = a.lo, a.hi
    best = None
    for _ in range(14):
        mid = (lo * hi) ** 0.5
        f, vl, tl = flops_at(mid)
        best = (mid, f, vl, tl)
        if abs(f - target) / target < 0.01:
            break
        if f > target:
            lo = mid
        else:
            hi = mid
    T, f, vl, tl = best
    out = {"batch": a.batch, "p": a.p, "size": a.size, "len": a.len, "seed": a.seed, "temperature": T,
           "flops_ratio": f / full, "vit_lens": vl, "txt_lens": tl, "full_gflops_per_sample": full / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
