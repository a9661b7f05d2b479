#!/usr/bin/env python
"""p -> temperature calibration with the CPU oracle for every bench workload (mirrors the reference
controller compress_nlvr_dtp.py:162-201, which steps `temperature` until the fvcore GFLOPs reach Ori_Gflops*(1-p)).

For the synthetic weights there is no trained checkpoint/temperature, so T is found by bisection such that the analytic
FLOPs of the workload's forward (madtp_amd/workloads.py) computed from the OBSERVED per-layer token counts equal (1-p) x the
unpruned FLOPs (+-1 %).  The batch matters (k = max over the local batch, vit.py:145), so calibrate at the bench batch.

usage: python tools/calibrate_temperature.py --task {nlvr,retrieval,clip,vqa} [--batch B --p P --size S --seed 0]
The last stdout line is the JSON record committed verbatim in madtp_amd/configs.py."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madtp_amd import workloads  # noqa: E402
from oracle import workloads as OW  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="nlvr", choices=workloads.NAMES)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--p", type=float, default=-1.0)
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lo", type=float, default=1.0)
    ap.add_argument("--hi", type=float, default=400.0)
    a = ap.parse_args()
    w = workloads.get(a.task, **({"size": a.size} if (a.size and a.task == "retrieval") else {}))
    B = a.batch or w.default_batch
    p = a.p if a.p >= 0 else w.p
    size = a.size or getattr(w, "size", 224)
    W = OW.weights(a.task, size)
    full = w.flops(None)
    target = (1 - p) * full

    def flops_at(T):
        t0 = time.time()
        _, lens = OW.forward(a.task, W, B, T, a.seed, size)
        f = w.flops(lens)
        print(f"T={T:.4f} ratio={f / full:.4f} lens={lens} ({time.time() - t0:.1f}s)", flush=True)
        return f, lens

    lo, hi = a.lo, a.hi
    best = None
    for _ in range(14):
        mid = (lo * hi) ** 0.5
        f, lens = flops_at(mid)
        best = (mid, f, lens)
        if abs(f - target) / target < 0.01:
            break
        if f > target:
            lo = mid
        else:
            hi = mid
    T, f, lens = best
    print(json.dumps({"task": a.task, "batch": B, "p": p, "size": size, "seed": a.seed, "temperature": T, "flops_ratio": f / full,
                      "lens": lens, "full_gflops_per_sample": full / 1e9}))


if __name__ == "__main__":
    main()
