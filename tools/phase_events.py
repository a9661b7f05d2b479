"""Unprofiled phase times of the serial NLVR forward from HIP events recorded around the vision / text encoder calls.
usage: python tools/phase_events.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madtp_amd import configs, runtime, workloads

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
w = workloads.get("nlvr")
B = w.default_batch
T, _ = configs.temperature_for("nlvr", B, w.p)
runtime.set_precision("bf16")
model = w.build("cuda")
inp = w.inputs(B, seed=0)
marks = []


def wrap(mod, tag):
    orig = mod.forward

    def f(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        h0 = time.perf_counter()
        out = orig(*a, **k)
        h1 = time.perf_counter()
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        marks.append((tag, e0, e1, h0, h1))
        return out
    mod.forward = f


wrap(model.visual_encoder, "vision")
wrap(model.text_encoder, "text")
with torch.no_grad():
    for _ in range(5):
        w.step(model, inp, T)
    torch.cuda.synchronize()
    marks.clear()
    t0 = time.perf_counter()
    fw = []
    for _ in range(steps):
        h0 = time.perf_counter()
        w.step(model, inp, T)
        fw.append((h0, time.perf_counter()))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
vis = [m for m in marks if m[0] == "vision"]
txt = [m for m in marks if m[0] == "text"]
gv = sum(m[1].elapsed_time(m[2]) for m in vis) / len(vis)
gt = sum(m[1].elapsed_time(m[2]) for m in txt) / len(txt)
gap_vt = sum(v[2].elapsed_time(t[1]) for v, t in zip(vis, txt)) / len(vis)          # vision end -> text start (GPU)
gap_tv = sum(t[2].elapsed_time(v[1]) for t, v in zip(txt[:-1], vis[1:])) / (len(vis) - 1)  # text end -> next vision start
hv = sum(m[4] - m[3] for m in vis) / len(vis) * 1e3
ht = sum(m[4] - m[3] for m in txt) / len(txt) * 1e3
hf = sum(b - a for a, b in fw) / len(fw) * 1e3
print(f"wall {wall:.3f} ms/forward | GPU: vision {gv:.3f}  v->t {gap_vt:.3f}  text {gt:.3f}  t->next v {gap_tv:.3f} | host: vision call {hv:.3f} "
      f"text call {ht:.3f} forward call {hf:.3f}")
