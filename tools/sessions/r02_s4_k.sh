set -x
for g in 0 2 3 5; do echo "== NGRP=$g (256-wide column tiles for cfg6, 128-wide x2 for cfg5)"; MADTP_GEMM_NGRP=$g timeout 300 python tools/sq_ablate.py 2>&1 | grep -v amdgpu; done
for g in 0 4 6 9 12; do echo "== ws NGRP=$g"; MADTP_GEMM_NGRP=$g timeout 300 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from madtp_amd import hip
hip.load()
dt = torch.bfloat16
for M, N, K in ((14208, 2304, 768), (25216, 3072, 768), (17152, 768, 3072), (10496, 2304, 768)):
    a = torch.randn(M, K, device="cuda").to(dt); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt)
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=dt)
    with hip.gemm_config(5):
        for _ in range(3): hip.gemm(a, w, bias, n=N, out=out)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): hip.gemm(a, w, bias, n=N, out=out)
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"ws M={M} N={N} K={K} {us:7.1f} us {2.0*M*N*K/us/1e6:7.1f} TF", flush=True)
PY
done
