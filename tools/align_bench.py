import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load(os.environ.get('MADTP_ABLATE_LIB'))
D = 768
sd = torch.randn(128, D, device="cuda")
hi = hip.cast_bf16(sd); lo = hip.cast_bf16((sd - hi.float()).contiguous())
for M in (1280, 10496, 25216):
    x = torch.randn(M, D, device="cuda")
    for _ in range(3): hip.align_logits(x, hi, lo)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): hip.align_logits(x, hi, lo)
    e1.record(); torch.cuda.synchronize()
    print(f"lib={os.path.basename(os.environ.get('MADTP_ABLATE_LIB') or 'default')} M={M:6d} {e0.elapsed_time(e1) * 1e3 / 50:8.1f} us", flush=True)
