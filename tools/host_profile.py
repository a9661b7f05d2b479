"""Host-side cost of one forward: cProfile over N serial forwards of a bench.py workload (MI355X box).
usage: python tools/host_profile.py [config] [steps] [top]   (env MADTP_ENCODER_CALL / MADTP_TEXT_ENCODER_CALL_MAX as in bench.py)"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madtp_amd import configs, runtime, workloads

cfg = sys.argv[1] if len(sys.argv) > 1 else "nlvr"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
w = workloads.get(cfg)
B = w.default_batch
T, _ = configs.temperature_for(cfg, B, w.p)
runtime.set_precision("bf16")
model = w.build("cuda")
inp = w.inputs(B, seed=0)
with torch.no_grad():
    for _ in range(5):
        w.step(model, inp, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        w.step(model, inp, T)
    torch.cuda.synchronize()
    print(f"{cfg} B={B}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per forward (unprofiled)")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        w.step(model, inp, T)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(top)
if os.environ.get("HOST_PROFILE_CALLEES"):
    for pat in os.environ["HOST_PROFILE_CALLEES"].split(","):
        st.print_callees(pat)
