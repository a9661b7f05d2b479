import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from madtp_amd import build, harness, hip, runtime
build.build(verbose=False); hip.load()
T = 8.612223847001898
model = harness.build_nlvr(224, 0, "cuda")
opt = torch.optim.AdamW(model.parameters(), lr=1e-6, weight_decay=0.05)
B = 64
images, text, _ = harness.nlvr_inputs(B, 224, 20, 0, "cuda")
targets = (torch.arange(B) % 2).cuda()
with runtime.precision("f16x3"), runtime.training_f16x3(True):
    def step():
        opt.zero_grad(set_to_none=True)
        lo, lf = model(images, text, targets, temperature=T, train=True)
        (lo + 0.1 * lf).backward()
        opt.step()
    step(); step()
    torch.cuda.synchronize()
    hip.profile_begin()
    step()
    torch.cuda.synchronize()
    rows = hip.profile_end()
rows.sort(key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in rows)
print("total gemm ms", tot, "launches", sum(r["launches"] for r in rows))
for r in rows[:60]:
    print(f'{r["dtype"]:5s} M={r["M"]:6d} N={r["N"]:5d} K={r["K"]:6d} x{r["launches"]:3d} {r["ms"]:8.3f} ms  {r["flops"]*(3 if r["dtype"]=="f16s" else 1)/r["ms"]/1e9:8.1f} TF(mfma)')
