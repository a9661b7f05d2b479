"""Host-side cost of one training step (cProfile, cumulative): python tools/train_host_profile.py [B] [precision]"""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import harness, hip, runtime
hip.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MODE = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
T = 8.612223847001898
model = harness.build_nlvr(224, 0, "cuda")
opt = torch.optim.AdamW(model.parameters(), lr=1e-6, weight_decay=0.05)
images, text, _ = harness.nlvr_inputs(B, 224, 20, 0, "cuda")
targets = (torch.arange(B) % 2).cuda()


def step():
    opt.zero_grad(set_to_none=True)
    lo, lf = model(images, text, targets, temperature=T, train=True)
    (lo + 0.1 * lf).backward()
    opt.step()


with runtime.precision(MODE), runtime.training_f16x3(MODE == "f16x3"):
    step(); step(); torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
