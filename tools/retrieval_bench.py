#!/usr/bin/env python
"""Secondary benchmark: retrieval evaluation with ITM re-ranking on one MI355X.
One JSON line: re-ranked (image, text) pairs per second of blip_retrieval.evaluate() on a synthetic evaluation set at the
reference's geometry (384x384 images, 35-token captions, k_test candidates per query), features included; next to it the
CPU oracle on a bounded sample.  bench.py's headline metric is unaffected.

usage: python tools/retrieval_bench.py [--images 128] [--texts 256] [--k-test 128] [--size 384] [--precision bf16]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madtp_amd import blip_retrieval as br, harness, runtime, specs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=128)
ap.add_argument("--texts", type=int, default=256)
ap.add_argument("--img-bs", type=int, default=32)
ap.add_argument("--k-test", type=int, default=128)
ap.add_argument("--size", type=int, default=384)
ap.add_argument("--temperature", type=float, default=6.0)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--no-cpu-baseline", action="store_true")
args = ap.parse_args()

dev = torch.device("cuda")
model = harness.build_retrieval(args.size, 0)
batches, ids, att = harness.retrieval_inputs(args.images, args.img_bs, args.texts, args.size, 35, 0, device="cuda")
loader = harness.RetrievalLoader(batches, ids, att)
cfg = {"k_test": args.k_test}
with runtime.precision(args.precision):
    br.evaluate(model, loader, dev, cfg, args.temperature)  # warm-up (weight preparation, allocator)
    torch.cuda.synchronize()
    t0 = time.time()
    i2t, t2i, _ = br.evaluate(model, loader, dev, cfg, args.temperature)
    torch.cuda.synchronize()
    dt = time.time() - t0
pairs = (args.images + args.texts) * args.k_test
out = {"metric": "ITM re-ranked (image,text) pairs/s, BLIP retrieval evaluate() incl. feature extraction", "value": round(pairs / dt, 1),
       "unit": "pairs/s", "seconds": round(dt, 3), "dtype": args.precision, "data": "synthetic",
       "config": {"workload": f"{args.images} images {args.size}x{args.size}, {args.texts} captions x 35 tokens, k_test {args.k_test}, "
                              f"temperature {args.temperature}", "queries": args.images + args.texts}}
if not args.no_cpu_baseline:
    from oracle import madtp_oracle as O
    W = specs.synth_weights(specs.blip_retrieval_shapes(args.size), 0)
    n_i, n_t, k = 4, 8, 4
    b2, i2, a2 = harness.retrieval_inputs(n_i, n_i, n_t, args.size, 35, 0)
    t0 = time.time()
    with torch.no_grad():
        O.retrieval_evaluate(W, b2, i2, a2, args.temperature, k)
    cdt = time.time() - t0
    out["cpu_baseline"] = {"value": round((n_i + n_t) * k / cdt, 2), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"{n_i} images, {n_t} captions, k_test {k} (oracle/madtp_oracle.py retrieval_evaluate), {cdt:.1f}s"}
print(json.dumps(out))
