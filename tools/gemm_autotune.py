#!/usr/bin/env python
"""Per-shape kernel choice of the big GEMMs, measured (runs on the MI355X box).

For every (operand class, N, K, output) the forward issues with M >= 4096 and every 64-row bucket of M in [4096, 32768], time the
three big-tile kernels - 0 wave-specialised 256x128 (madtp_gemm_set_config 7), 1 ping-pong 256x256 (9), 2 ping-pong 192x256 (10) -
with the forward's epilogue (f32 residual stream for the f32-output classes, GELU for N = 3072), interleaved rounds, median, and
write the winners to madtp_amd/csrc/gemm_table.h (+ a human-readable log).  The previous bucket's kernel is kept while it is
within 1.5 % of the best, so that noise does not fragment the table.

usage: python tools/gemm_autotune.py [--step 64] [--out madtp_amd/csrc/gemm_table.h] [--log gpurun_out/gemm_autotune.txt] [--quick]
"""
import argparse
import os
import sys
import statistics

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madtp_amd import hip, runtime  # noqa: E402

CLASSES = [  # (N, K, f32out)
    (2304, 768, 0), (768, 768, 1), (3072, 768, 0), (768, 3072, 1),
    (1536, 768, 0),   # cross-attention K/V of one text layer on the image tokens (single launches; the pair launch stays on ws)
]
CFG = {0: 7, 1: 9, 2: 10}


def bench(fn, reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=int, default=64)
    ap.add_argument("--mmax", type=int, default=32768)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "madtp_amd", "csrc", "gemm_table.h"))
    ap.add_argument("--log", default="gpurun_out/gemm_autotune.txt")
    ap.add_argument("--quick", action="store_true", help="step 512 (smoke run of the tool)")
    ap.add_argument("--modes", default="lp,x3")
    args = ap.parse_args()
    if args.quick:
        args.step = 512
    hip.load()
    os.environ["MADTP_GEMM_TABLE"] = "0"
    nb = args.mmax // 64
    log = []
    tables = []
    for mode in args.modes.split(","):
        x3 = mode == "x3"
        runtime.set_precision("f16x3" if x3 else "f16")
        for N, K, f32out in CLASSES:
            sel = [255] * nb
            w32 = torch.randn(N, K, device="cuda") * 0.05
            npad = (N + 127) // 128 * 128
            wp = torch.zeros(npad, K, device="cuda"); wp[:N] = w32
            wd = hip.split_f16_weight(wp) if x3 else hip.cast_lp_weight(wp)
            bias = torch.randn(N, device="cuda")
            a_full = torch.randn(args.mmax, K, device="cuda")
            ad_full = hip.split_f16(a_full) if x3 else hip.cast_bf16(a_full)
            res_full = torch.randn(args.mmax, N, device="cuda") if f32out else None
            out_full = torch.empty(args.mmax, (2 * N if (x3 and not f32out) else N), device="cuda",
                                   dtype=torch.float32 if f32out else (torch.float16 if x3 else torch.bfloat16))
            act = hip.ACT_GELU if N == 3072 else hip.ACT_NONE
            prev = None
            for M in range(4096, args.mmax + 1, args.step):
                ad, out = ad_full[:M], out_full[:M]
                kw = dict(residual=res_full[:M]) if f32out else {}

                def run():
                    hip.gemm(ad, wd, bias, n=N, out=out, act=act, **kw)
                ts = {c: [] for c in CFG}
                ok = {}
                for c, cfg in CFG.items():
                    with hip.gemm_config(cfg):
                        try:
                            run(); run(); ok[c] = True
                        except Exception:
                            ok[c] = False
                torch.cuda.synchronize()
                for rnd in range(3):
                    for c, cfg in CFG.items():
                        if not ok[c]:
                            continue
                        with hip.gemm_config(cfg):
                            ts[c].append(bench(run, 8))
                med = {c: statistics.median(v) for c, v in ts.items() if v}
                with hip.gemm_config(0):
                    run(); torch.cuda.synchronize()
                    t_auto = min(bench(run, 8) for _ in range(2))   # the cost model's choice (the table is off in this process)
                best = min(med, key=med.get)
                # hysteresis along M: keep the previous bucket's kernel while it is within 1.5 % of the best (noise must not fragment the table)
                pick = prev if (prev in med and med[prev] < 1.015 * med[best]) else best
                prev = pick
                for i in range((M - args.step) // 64, M // 64):
                    sel[i] = pick
                line = (f"{mode} N={N:5d} K={K:5d} f32out={f32out} M={M:6d}  ws {med.get(0, 0):7.1f}  pp256 {med.get(1, 0):7.1f}  pp192 {med.get(2, 0):7.1f}"
                        f"  auto(model) {t_auto:7.1f} -> {pick}  ({2.0 * M * N * K / med[pick] / 1e6:7.1f} TF)")
                print(line, flush=True)
                log.append(line)
            for i in range(0, 4096 // 64 - 1):
                sel[i] = 255
            tables.append((1 if x3 else 0, N, K, f32out, sel))
    os.makedirs(os.path.dirname(args.log) or ".", exist_ok=True)
    with open(args.log, "w") as f:
        f.write("\n".join(log) + "\n")
    with open(args.out, "w") as f:
        f.write("// Per-shape kernel choice of the big GEMMs (M >= 4096, 2-byte operand planes), written by tools/gemm_autotune.py from\n"
                "// measurements on an idle MI355X (log: profiles/r05_gemm_autotune.txt).  sel[i] is the choice for M in (64 i, 64 (i + 1)]:\n"
                "// 0 = wave-specialised 256x128, 1 = ping-pong 256x256, 2 = ping-pong 192x256, 255 = not measured (cost model).\n"
                "#pragma once\nnamespace {\nstruct GemmTabClass { int x3, N, K, f32out; const unsigned char* sel; int n; };\n")
        for x3, N, K, fo, sel in tables:
            f.write(f"static const unsigned char kTab_{x3}_{N}_{K}_{fo}[{len(sel)}] = {{" + ",".join(str(v) for v in sel) + "};\n")
        f.write("static const GemmTabClass kGemmTab[] = {\n")
        for x3, N, K, fo, sel in tables:
            f.write(f"    {{{x3}, {N}, {K}, {fo}, kTab_{x3}_{N}_{K}_{fo}, {len(sel)}}},\n")
        f.write("};\nstatic int gemm_table_lookup(bool x3, int M, int N, int K, bool f32out) {\n"
                "    for (const GemmTabClass& c : kGemmTab) {\n"
                "        if (!c.sel || c.x3 != (int)x3 || c.N != N || c.K != K || c.f32out != (int)f32out) continue;\n"
                "        const int i = (M - 1) / 64;\n"
                "        if (i < 0 || i >= c.n || c.sel[i] == 255) return -1;\n"
                "        return c.sel[i];\n    }\n    return -1;\n}\n}  // namespace\n")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
