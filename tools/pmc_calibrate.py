"""FETCH_SIZE calibration for the LDS-DMA access pattern of the GEMM kernels: N = 128 (one column tile), so every A row
panel is fetched by exactly one workgroup and W (196 KB) at most once per XCD - the true HBM read volume is known."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load()
for M in (10496, 25216):
    K, N = 768, 128
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    junk = torch.empty(1 << 28, device="cuda", dtype=torch.float32)  # 1 GiB: flush L2 + Infinity Cache between launches
    for _ in range(4):
        junk.fill_(1.0)
        hip.gemm(a, w, None, n=N, out=out)
    torch.cuda.synchronize()
    print(f"M={M}: A+W = {(M * K + N * K) * 2 / 2**20:.2f} MiB read, C = {M * N * 2 / 2**20:.2f} MiB written")
