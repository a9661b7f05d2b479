"""Builds timing-only variants of the library with parts of the wave-specialised GEMM main loop removed
(MADTP_WS_ABLATE bit mask, see gemm.hip): madtp_amd/lib/libmadtp_hip_abl<N>.so.  Used with tools/gemm_ablate.py."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from madtp_amd import build as b

b.build()
src, macro = {"align": ("prune.hip", "MADTP_AL_ABLATE"), "tstime": ("prune.hip", "MADTP_TS_TIMING"), "attime": ("attention.hip", "MADTP_TS_TIMING"),
             "wstime": ("gemm.hip", "MADTP_WS_TIMING"), "sq": ("gemm.hip", "MADTP_SQ_ABLATE")}.get(
    os.environ.get("ABLATE"), ("gemm.hip", "MADTP_WS_ABLATE"))
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    o = os.path.join(b.LIBDIR, f"{src[:-4]}_abl{n}.o")
    subprocess.check_call([b._hipcc()] + b.FLAGS + [f"-D{macro}={n}", "-c", os.path.join(b.CSRC, src), "-o", o])
    objs = [o] + [os.path.join(b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES if s != src]
    lib = os.path.join(b.LIBDIR, f"libmadtp_hip_abl{n}.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)
