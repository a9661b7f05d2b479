"""Main-loop / epilogue split of gemm_ws_kernel as seen by consumer wave 0 of workgroup 0 (wall_clock64, 100 MHz) - needs
the -DMADTP_WS_TIMING build of gemm.hip (ABLATE=wstime python tools/build_ablate.py 1)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
lib = hip.load(os.environ["MADTP_ABLATE_LIB"])
lib.madtp_debug_read_ws_ts.argtypes = [ctypes.c_void_p]
dt = torch.bfloat16
for M, N, K, f32res in ((25216, 2304, 768, 0), (25216, 768, 3072, 0), (25216, 768, 3072, 1), (10496, 768, 768, 0), (10496, 768, 768, 1),
                        (10496, 3072, 768, 0)):
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda") if f32res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32res else dt)
    for _ in range(3):
        hip.gemm(a, w, bias, n=N, out=out, residual=res, out_dtype=out.dtype)
    torch.cuda.synchronize()
    ts = (ctypes.c_longlong * 8)()
    lib.madtp_debug_read_ws_ts(ts)
    main, epi, tiles, total = [int(x) for x in list(ts)[:4]]
    print(f"M={M} N={N} K={K} {'f32+res' if f32res else 'bf16   '}: tiles {tiles}  main loop {main * 10 / tiles:.0f} ns/tile  "
          f"epilogue {epi * 10 / tiles:.0f} ns/tile  wave total {total / 100:.1f} us")
