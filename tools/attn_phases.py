"""Phase timestamps (wall_clock64, 100 MHz) of one head iteration of attn_bf16_kernel, wave 0 of workgroup 0 - needs the
-DMADTP_TS_TIMING build of attention.hip (ABLATE=attime python tools/build_ablate.py 1)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
lib = hip.load(os.environ["MADTP_ABLATE_LIB"])
lib.madtp_debug_read_attn_ts.argtypes = [ctypes.c_void_p]
for B, N in ((128, 81), (128, 95), (128, 134), (128, 197)):
    H = 12
    qkv = torch.randn(B * N, 3 * H * 64, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        hip.attention(qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:], B, H, N, N, 0.125, scores=True)
    out = (ctypes.c_longlong * 8)()
    lib.madtp_debug_read_attn_ts(out)
    t = list(out)[:7]
    names = ["barrier", "stage-issue", "QK", "softmax+scores", "PV", "store+onorm"]
    print(f"B={B} N={N}: " + "  ".join(f"{n} {(t[i + 1] - t[i]) * 10}ns" for i, n in enumerate(names)), " | head total", (t[6] - t[0]) * 10, "ns")
