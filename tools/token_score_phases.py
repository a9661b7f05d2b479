"""Phase timestamps (wall_clock64, 100 MHz) of token_score workgroup 0 - needs the -DMADTP_TS_TIMING build
(ABLATE=tstime python tools/build_ablate.py 1)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
lib = hip.load(os.environ["MADTP_ABLATE_LIB"])
for B, N in ((128, 97), (128, 197), (64, 20)):
    H, K = 12, 100
    nrt = (N + 15) // 16
    cs = torch.rand(B, nrt, N, device="cuda"); p0 = torch.rand(B, H, N, device="cuda"); on = torch.rand(B, H, N, device="cuda")
    ta = torch.randn(B, N, 128, device="cuda")[:, 1:, :K]
    for _ in range(3):
        hip.token_score((cs, p0, on), ta, 5.0, B, H, N)
    out = (ctypes.c_longlong * 16)()
    lib.madtp_debug_read_ts.argtypes = [ctypes.c_void_p]
    lib.madtp_debug_read_ts(out)
    t = list(out)[:7]
    print(f"B={B} N={N}: " + " ".join(f"{(t[i + 1] - t[i]) * 10}ns" for i in range(6)), " total", (t[6] - t[0]) * 10, "ns")

# long sequences: token_score_split_kernel (workgroup (0, 0)): staging issue | row max | per-token terms | sums + I | phase B | publish
for B, N in ((32, 901), (32, 605), (32, 430)):
    H, K = 12, 100
    nrt = (N + 15) // 16
    cs = torch.rand(B, nrt, N, device="cuda"); p0 = torch.rand(B, H, N, device="cuda"); on = torch.rand(B, H, N, device="cuda")
    ta = torch.randn(B, N, 128, device="cuda")[:, 1:, :K]
    for _ in range(3):
        hip.token_score_sync((cs, p0, on), ta, 5.0, B, H, N)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.madtp_debug_read_ts(out)
    t = list(out)[:7]
    print(f"split B={B} N={N}: " + " ".join(f"{(t[i + 1] - t[i]) * 10}ns" for i in range(6)), " total", (t[6] - t[0]) * 10, "ns")
