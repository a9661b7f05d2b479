#!/usr/bin/env python
"""Headline benchmark: images/sec of the pruned BLIP-base NLVR2 forward (BLIP_NLVR.forward(train=False) dataflow,
reference models/blip_nlvr.py:63-100) at p=0.5, 64 samples (= 128 images) per GPU, bf16 GEMM operands, on MI355X.

    python bench.py [--gpus N --steps K --warmup W --precision bf16|fp32 --batch 64]

N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`:
one process per GPU, batch sharded data-parallel (each rank owns 64 whole samples - both images of a pair stay on
one rank), weights replicated, NO collective inside the forward (the path has no exchange step);
RCCL is used only for the barrier and the MAX-reduction of the elapsed time.  Scaling is therefore "weak".

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel (the bf16 MFMA GEMM): algorithmic FLOPs of every GEMM launch in the timed
                 region / the sum of their durations, measured live with HIP events on the launch stream.
  cpu_baseline - oracle/ (the CPU restatement of the reference forward, kind "port") timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md.  f16x3: three f16 MFMA products per logical product, so the
# roofline of the fp32-accurate GEMM in ALGORITHMIC flops (2MNK) is a third of the f16 dense peak.
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "f16x3": 2500.0 / 3.0}
GEMM_DT = {"bf16": "bf16", "fp32": "f32", "f16x3": "f16s"}


def load_calibration(batch, p=0.5):
    from madtp_amd import configs
    return configs.temperature_for("nlvr", batch, p)


def gemm_summary(rows, dtype, min_m=0):
    sel = [r for r in rows if r["dtype"] == dtype and r["M"] >= min_m]
    return sum(r["ms"] for r in sel), sum(r["flops"] for r in sel), sum(r["launches"] for r in sel)


WS_MIN_M = 4096  # madtp_gemm runs bf16 problems with M >= 4096 (and no split-K) on gemm_ws_kernel (csrc/gemm.hip)


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (collected and corrected
    as MI355X_MICROARCH.md prescribes: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled on gfx950); produced
    by tools/rocpd_pmc.py, see profiles/.  None when no such file is present."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_gemm_bf16.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        d = json.load(f)
    # traffic of the dominant kernel alone: launch-weighted over gemm_ws_kernel<true> / <false>
    fk = {k: v for k, v in d.get("per_kernel_fetch", {}).items() if "gemm_ws_kernel" in k}
    wk = {k: v for k, v in d.get("per_kernel_write", {}).items() if "gemm_ws_kernel" in k}
    n = sum(v["launches"] for v in fk.values())
    if n:
        fetch = sum(v["launches"] * v["kib_per_launch"] for v in fk.values()) / n
        write = sum(v["launches"] * v["kib_per_launch"] for v in wk.values()) / max(1, sum(v["launches"] for v in wk.values()))
        d["ws_hbm_bytes_per_launch"] = int((2 * fetch + write) * 1024)
    return d


def gemm_breakdown(rows, steps):
    out = []
    for r in sorted(rows, key=lambda r: -r["ms"]):
        out.append(f"{r['dtype']:5s} M={r['M']:6d} N={r['N']:5d} K={r['K']:5d} calls/step={r['launches'] / steps:5.1f} "
                   f"ms/step={r['ms'] / steps:7.3f} TFLOP/s={r['flops'] / r['ms'] / 1e9:7.1f}")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "f16x3"])
    ap.add_argument("--parity-steps", type=int, default=10, help="timed steps of the parity_mode leg (f16x3 precision)")
    ap.add_argument("--parity-batch", type=int, default=0, help="samples of the index_match check (0 = --batch)")
    ap.add_argument("--batch", type=int, default=64, help="NLVR samples per GPU (2 images each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--gemm-breakdown", action="store_true", help="per-shape GEMM time table on stderr")
    args = ap.parse_args()

    from madtp_amd import dist as mdist
    world, rank, local_rank = mdist.env_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    mdist.init("nccl")  # "nccl" is RCCL on ROCm; no-op for a single process
    dist = torch.distributed if world > 1 else None

    from madtp_amd import build, harness, hip, runtime
    if rank == 0:  # one builder per node (the prebuilt .so normally travels with the snapshot: no-op); the others wait
        build.build(verbose=False)
    if dist is not None:
        dist.barrier()
    hip.load()
    runtime.set_precision(args.precision)
    prof_rows = []

    T, calib = load_calibration(args.batch, 0.5)
    model = harness.build_nlvr(224, 0, "cuda")
    images, text, targets = harness.nlvr_inputs(args.batch, 224, 20, seed=rank)  # resident in HBM before timing

    def step():
        return model(images, text, targets, temperature=T, train=False)

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            logits = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # Roofline leg: the SAME K steps once more with a HIP event pair around every madtp_gemm launch (recorded by
        # the library on the launch stream).  Kept out of the `value` region because ~180 event pairs per step
        # perturb it by ~10 % (measured); its own wall time is reported as instrumented_ms_per_step.
        instr_elapsed = None
        if not args.no_gemm_events:
            hip.profile_begin()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            instr_elapsed = time.perf_counter() - t1
            prof_rows = hip.profile_end()
    elapsed = mdist.max_over_ranks(elapsed, device="cuda")

    images_per_step = 2 * args.batch * world
    value = images_per_step * args.steps / elapsed
    _, trace = harness.run_nlvr(model, images, text, targets, T)
    vit_lens = harness.token_lengths(trace["vit"], 197)
    txt_lens = harness.token_lengths(trace["text"], 20)
    flops_sample = harness.nlvr_forward_flops(vit_lens, txt_lens)
    flops_full = harness.nlvr_forward_flops([197] * 12, [20] * 12)

    roof = None
    if not args.no_gemm_events:
        dt_name = GEMM_DT[args.precision]
        ms_all, fl_all, cnt_all = gemm_summary(prof_rows, dt_name)
        # dominant kernel: gemm_ws_kernel (2-byte operand planes, M >= 4096) in the bf16 / f16x3 modes; gemm_kernel<float> in fp32
        min_m = WS_MIN_M if args.precision != "fp32" else 0
        ms, fl, cnt = gemm_summary(prof_rows, dt_name, min_m)
        if ms > 0:
            ach = fl / (ms * 1e-3) / 1e12
            peak = MFMA_PEAK_TFLOPS[args.precision]
            pmc = pmc_traffic() if args.precision == "bf16" else None
            alg_bytes = sum(r["bytes"] for r in prof_rows if r["dtype"] == dt_name and r["M"] >= min_m)
            kname = {"bf16": "gemm_ws_kernel (all bf16 madtp_gemm launches with M >= 4096: ViT qkv/proj/fc1/fc2, cross-attention K/V)",
                     "f16x3": "gemm_ws_kernel<f16-split> (all madtp_gemm launches with M >= 4096; 3 f16 MFMA products per "
                              "logical product: achieved/peak are in algorithmic 2MNK flops, peak = f16 dense / 3)",
                     "fp32": "gemm_kernel<float> (madtp_gemm)"}[args.precision]
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 1),
                    "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": pmc.get("ws_hbm_bytes_per_launch") if pmc else None,
                    "algorithmic_bytes_per_launch": round(alg_bytes / cnt),
                    "launches_per_step": cnt // args.steps, "kernel_ms_per_step": round(ms / args.steps, 3),
                    "avg_launch_us": round(1e3 * ms / cnt, 2), "algorithmic_gflop_per_launch": round(fl / cnt / 1e9, 3),
                    "instrumented_ms_per_step": round(1e3 * instr_elapsed / args.steps, 3),
                    "all_gemm_launches": {"achieved": round(fl_all / (ms_all * 1e-3) / 1e12, 1),
                                          "frac": round(fl_all / (ms_all * 1e-3) / 1e12 / peak, 4),
                                          "launches_per_step": cnt_all // args.steps,
                                          "ms_per_step": round(ms_all / args.steps, 3)}}

    out = {
        "metric": "images/sec forward, BLIP-base NLVR2 p=0.5 b64; pruned-token index match",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"bf16": "bf16", "fp32": "f32", "f16x3": "f16x3 (fp32-accurate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": "BLIP-base NLVR2 forward (BLIP_NLVR.forward(train=False)), p=0.5, 64 samples = 128 "
                               "images 224x224 + 20 text tokens per GPU, random-init weights",
                   "samples_per_gpu": args.batch, "images_per_gpu": 2 * args.batch, "temperature": T,
                   "flops_ratio_vs_unpruned": round(flops_sample / flops_full, 4), "vit_tokens_per_layer": vit_lens,
                   "text_tokens_per_layer": txt_lens, "parallelism": f"dp{world}"},
        "samples_per_s": round(value / 2, 1),
        "model_tflops": round(flops_sample * args.batch * world * args.steps / elapsed / 1e12, 1),
        "roofline": roof,
    }

    if not args.no_parity:
        # parity_mode leg: the SAME workload timed in the precision mode that carries the parity claim (f16x3: fp32-accurate
        # GEMMs on the f16 MFMA, everything else the fp32 mode's kernels), same barrier / max-over-ranks protocol
        pm = "f16x3"
        with runtime.precision(pm), torch.no_grad():
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t2 = time.perf_counter()
            for _ in range(args.parity_steps):
                step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            pel = time.perf_counter() - t2
        pel = mdist.max_over_ranks(pel, device="cuda")
        out["parity_mode"] = {"precision": pm, "value": round(images_per_step * args.parity_steps / pel, 1), "unit": "images/s",
                              "ms_per_step": round(1e3 * pel / args.parity_steps, 3), "steps": args.parity_steps,
                              "what": "same workload, every Linear as 3 f16 MFMA products of f16-split operands (fp32-accurate), "
                                      "attention / LayerNorm / scores on the exact-f32 kernels; kept sets vs the oracle below"}
    if rank == 0 and world == 1:
        if not args.no_parity:
            im = parity_report(model, harness, runtime, T, args.precision, B=args.parity_batch or args.batch)
            out["index_match"] = im
            out["parity_mode"]["index_match"] = im.get("f16x3")
            out["parity_mode"]["index_match_batch"] = im["batch"]
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T)
    if rank == 0 and args.gemm_breakdown and not args.no_gemm_events:
        print(gemm_breakdown(prof_rows, args.steps), file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def parity_report(model, harness, runtime, T, precision, B=64, seed=11):
    """kept-token index match of the timed precision mode, the f16x3 mode and the fp32 mode vs the CPU oracle at the
    HEADLINE batch (k = max_b count couples the samples of a batch, so a smaller batch is a different computation;
    the oracle is the checker here, never the thing measured)."""
    from tests.parity_util import nlvr_index_match
    return nlvr_index_match(model, T, sorted({"fp32", "f16x3", precision}), B=B, seed=seed)


def cpu_baseline(T, B=8, budget_s=12.0):
    """The CPU oracle (a port of the reference forward; the reference itself cannot travel to this box) timed on the
    host cores: B=8 samples (16 images) of the same synthetic workload, repeated for ~budget_s seconds."""
    from madtp_amd import specs, synth
    from oracle import madtp_oracle as O
    W = specs.synth_weights(specs.blip_nlvr_shapes(224), 0)
    images = synth.synth_images(2 * B, 224, 0)
    ids = synth.synth_token_ids(B, 20, 0)
    att = torch.ones_like(ids)
    with torch.no_grad():
        O.blip_nlvr_forward(W, images, ids, att, T)  # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            O.blip_nlvr_forward(W, images, ids, att, T)
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or n >= 50:
                break
    return {"value": round(2 * B * n / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} forwards of {B} samples ({2 * B} images 224x224 + 20 tokens), same weights/temperature, "
                      f"PyTorch CPU eager fp32 via oracle/madtp_oracle.py, {dt:.1f}s",
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
