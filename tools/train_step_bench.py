"""Training-step timing of the headline model on the HIP path (madtp_amd/backward.py; MADTP_TRAIN_PRECISION=fp32 (default) or
f16x3 - round 5: every GEMM of the step as three f16 MFMA products): forward(train=True),
loss_ori + 0.1 loss_fdt, backward, AdamW step - next to the inference forward of the same mode.  Not a BASELINE metric (the
reference publishes none for training); recorded for DESIGN.md section 5.   python tools/train_step_bench.py [B ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from madtp_amd import build, harness, hip, runtime  # noqa: E402

build.build(verbose=False)
hip.load()
T = 8.612223847001898
MODE = os.environ.get("MADTP_TRAIN_PRECISION", "fp32")
model = harness.build_nlvr(224, 0, "cuda")
DROPOUT = os.environ.get("MADTP_TRAIN_DROPOUT", "0") == "1"  # model.train(): dropout 0.1 / DropPath as the reference's loops run
# (the reference's create_optimizer builds torch.optim.AdamW with its defaults = the foreach implementation on a GPU;
#  MADTP_TRAIN_FUSED_ADAM=1: torch's fused=True variant, for the record)
opt = torch.optim.AdamW(model.parameters(), lr=1e-6, weight_decay=0.05, **({"fused": True} if os.environ.get("MADTP_TRAIN_FUSED_ADAM") == "1" else {}))
for B in [int(a) for a in sys.argv[1:]] or [4, 16, 64]:
    images, text, _ = harness.nlvr_inputs(B, 224, 20, 0, "cuda")
    targets = (torch.arange(B) % 2).cuda()
    with runtime.precision(MODE), runtime.training_f16x3(MODE == "f16x3"), runtime.training_amp(MODE in ("bf16", "f16")):  # (bf16 / f16: the --amp route, round 6)
        with torch.no_grad():
            for _ in range(2):
                model(images, text, targets, temperature=T, train=False)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(images, text, targets, temperature=T, train=False)
            torch.cuda.synchronize()
            t_inf = (time.time() - t0) / 3

        model.train(DROPOUT)

        def step():
            opt.zero_grad(set_to_none=True)
            lo, lf = model(images, text, targets, temperature=T, train=True)
            (lo + 0.1 * lf).backward()
            opt.step()
            return float(lo.detach())
        step()
        torch.cuda.synchronize()
        t0 = time.time()
        losses = [step() for _ in range(3)]
        torch.cuda.synchronize()
        t_tr = (time.time() - t0) / 3
    model.eval()
    print(f"B={B:3d} samples ({2 * B} images): inference forward ({MODE} mode) {t_inf * 1e3:8.1f} ms, training step{' (train mode, dropout)' if DROPOUT else ''} {t_tr * 1e3:8.1f} ms "
          f"({2 * B / t_tr:7.1f} images/s), peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB, loss_ori {losses[0]:.4f} -> {losses[-1]:.4f}")
