#!/usr/bin/env python
"""Secondary benchmark (inference half): captioning with beam search on one MI355X.
One JSON line: captions per second of BLIP_Decoder.generate(sample=False, num_beams=3, max_length=20, min_length=5) - the
evaluation call of compress_caption_dtp.py:86 with configs/caption_coco.yaml's generation settings - on synthetic 384x384
images with random-init weights (the captions are noise; the work per caption is that of the reference: pruned ViT + up to 16
decoder steps over 3 beams), next to the CPU oracle on a bounded sample.  bench.py's headline metric is unaffected.

usage: python tools/caption_bench.py [--batch 32] [--size 384] [--temperature 6.0] [--precision bf16] [--steps 5]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madtp_amd import runtime, specs, synth  # noqa: E402
from madtp_amd.blip import BLIP_Decoder  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=384)
ap.add_argument("--temperature", type=float, default=6.0)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--max-length", type=int, default=20)
ap.add_argument("--min-length", type=int, default=5)
ap.add_argument("--no-cpu-baseline", action="store_true")
args = ap.parse_args()

model = BLIP_Decoder(image_size=args.size, evaluate=True)
model.load_state_dict(specs.synth_weights(specs.blip_decoder_shapes(args.size), 0, device="cuda"), strict=False)
model = model.eval().cuda()
images = synth.synth_images(args.batch, args.size, 0).cuda()
kw = dict(sample=False, num_beams=3, max_length=args.max_length, min_length=args.min_length, temperature=args.temperature)
with runtime.precision(args.precision), torch.no_grad():
    out = model.generate(images, **kw)  # warm-up (weight preparation, allocator)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        out = model.generate(images, **kw)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
lens = [blk.last_prune["k"] + 2 if blk.last_prune is not None and blk.last_prune["pruned"] else None for blk in model.visual_encoder.blocks]
res = {"metric": "captions/s, BLIP_Decoder.generate beam search (3 beams) incl. the pruned ViT", "value": round(args.batch / dt, 1),
       "unit": "captions/s", "ms_per_batch": round(dt * 1e3, 2), "dtype": args.precision, "data": "synthetic",
       "config": {"workload": f"{args.batch} images {args.size}x{args.size}, num_beams 3, max_length {args.max_length}, min_length "
                              f"{args.min_length}, temperature {args.temperature}, random-init weights", "generated_length": int(out.shape[1]),
                  "vit_tokens_kept_per_layer": lens}}
if not args.no_cpu_baseline:
    from oracle import madtp_oracle as O
    W = specs.synth_weights(specs.blip_decoder_shapes(args.size), 0)
    n = 2
    t0 = time.time()
    with torch.no_grad():
        O.blip_decoder_generate_forward(W, images[:n].cpu(), args.temperature, max_length=args.max_length, min_length=args.min_length)
    cdt = time.time() - t0
    res["cpu_baseline"] = {"value": round(n / cdt, 2), "unit": "captions/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"{n} images (oracle/madtp_oracle.py blip_decoder_generate_forward), {cdt:.1f}s"}
print(json.dumps(res))
