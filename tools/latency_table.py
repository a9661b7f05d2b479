#!/usr/bin/env python
"""Latency of the vision encoder and of the whole NLVR forward at small batches, for the three host/device hand-overs of k:
per-layer path (Python between layers), encoder-level call with the per-layer host read of k, sync-free encoder call
(madtp_vit_encoder_async: device-side lengths).  usage: latency_table.py [precision]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import configs, harness, hip, runtime, vit, bert
hip.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "f16"
T = configs.temperature_for("nlvr", 64, 0.5)[0]
model = harness.build_nlvr(224, 0, "cuda")
venc = model.visual_encoder


def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


paths = (("per-layer", False, False), ("encoder call, host k", True, False), ("encoder call, sync-free", True, True))
print(f"precision {mode}, temperature {T:.3f}; ms per call (mean of 30)")
print("ViT encoder alone (VisionTransformer.forward), images per call:")
print(f"{'path':28s}" + "".join(f"{n:>9d}" for n in (1, 2, 4, 8, 16)))
with runtime.precision(mode), torch.no_grad():
    for name, ec, sf in paths:
        vit._ENCODER_CALL, vit._SYNC_FREE = ec, sf
        row = []
        for n in (1, 2, 4, 8, 16):
            images, _, _ = harness.nlvr_inputs(max(1, n), 224, 20, seed=3)
            img = images[:n].contiguous()
            row.append(timed(lambda: venc(img, space_dict=model.space_dict, temperature=T)))
        print(f"{name:28s}" + "".join(f"{v:9.3f}" for v in row), flush=True)
    # Round 6: rounds 4-5 timed harness.run_nlvr here, i.e. the forward PLUS the test harness's read-back of every layer's pruning
    # record to the host (24 layers x several .cpu() copies, ~2.2 ms at one sample, and more fields in round 5 than in round 4 -
    # the "20 % small-batch regression" of the round-5 review was the harness, not the forward).  Both are printed now.
    for what, call in (("BLIP_NLVR.forward(train=False) alone", lambda im, tx, tg: model(im, tx, tg, temperature=T, train=False)),
                       ("harness.run_nlvr = forward + per-layer records copied to the host (what rounds 4-5 printed)",
                        lambda im, tx, tg: harness.run_nlvr(model, im, tx, tg, T))):
        print(f"whole NLVR forward - {what}; samples per call (2 images each):")
        print(f"{'path':28s}" + "".join(f"{n:>9d}" for n in (1, 2, 4, 8)))
        for name, ec, sf in paths:
            vit._ENCODER_CALL, vit._SYNC_FREE, bert._ENCODER_CALL = ec, sf, ec
            row = []
            for n in (1, 2, 4, 8):
                images, text, targets = harness.nlvr_inputs(n, 224, 20, seed=3)
                row.append(timed(lambda: call(images, text, targets)))
            print(f"{name:28s}" + "".join(f"{v:9.3f}" for v in row), flush=True)
