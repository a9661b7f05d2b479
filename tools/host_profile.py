"""cProfile of the host side of one small-batch forward (where the launch-bound floor of ~4 ms per forward goes)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import harness, runtime
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runtime.set_precision("bf16")
T = 8.6
model = harness.build_nlvr(224, 0, "cuda")
images, text, targets = harness.nlvr_inputs(B, 224, 20, seed=0)
with torch.no_grad():
    for _ in range(5):
        model(images, text, targets, temperature=T, train=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        model(images, text, targets, temperature=T, train=False)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
