import os, sys, time, torch
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
    def step(log=None):
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        lo, lf = model(images, text, targets, temperature=T, train=True)
        t1 = time.time()
        (lo + 0.1 * lf).backward()
        t2 = time.time()
        opt.step()
        t3 = time.time()
        torch.cuda.synchronize()
        t4 = time.time()
        if log is not None:
            log.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    step(); step()
    log = []
    for _ in range(4):
        step(log)
    for l in log:
        print("host fwd %.1f ms, bwd %.1f ms, opt %.1f ms, tail sync %.1f ms" % tuple(1e3 * x for x in l))
    # with syncs between the phases: GPU time of each
    def step2():
        torch.cuda.synchronize(); t0 = time.time()
        opt.zero_grad(set_to_none=True)
        lo, lf = model(images, text, targets, temperature=T, train=True)
        torch.cuda.synchronize(); t1 = time.time()
        (lo + 0.1 * lf).backward()
        torch.cuda.synchronize(); t2 = time.time()
        opt.step()
        torch.cuda.synchronize(); t3 = time.time()
        print("synced fwd %.1f ms, bwd %.1f ms, opt %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    step2(); step2()
