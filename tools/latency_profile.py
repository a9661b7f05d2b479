"""Where a 1-sample NLVR forward spends its time (round 6, review item 6): wall time per path, cProfile of the host side, and -
run under `rocprofv3 --kernel-trace --stats` - the kernel side.   usage: latency_profile.py [samples] [precision] [top]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from madtp_amd import bert, configs, harness, hip, runtime, vit

hip.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "f16"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
T = configs.temperature_for("nlvr", 64, 0.5)[0]
model = harness.build_nlvr(224, 0, "cuda")
images, text, targets = harness.nlvr_inputs(n, 224, 20, seed=3)


def timed(fn, reps=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with runtime.precision(mode), torch.no_grad():
    for name, ec, sf in (("per-layer", False, False), ("encoder call, host k", True, False), ("encoder call, sync-free", True, True)):
        vit._ENCODER_CALL, vit._SYNC_FREE, bert._ENCODER_CALL = ec, sf, ec
        if hasattr(bert, "_SYNC_FREE"):
            bert._SYNC_FREE = sf
        full = timed(lambda: harness.run_nlvr(model, images, text, targets, T))
        bare = timed(lambda: model(images, text, targets, temperature=T, train=False))
        print(f"{name:26s} harness.run_nlvr {full:7.3f} ms   model(...) alone {bare:7.3f} ms", flush=True)
    if os.environ.get("LAT_CPROFILE", "1") == "1":
        vit._ENCODER_CALL, vit._SYNC_FREE, bert._ENCODER_CALL = True, False, True
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(40):
            model(images, text, targets, temperature=T, train=False)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(top)
