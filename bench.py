#!/usr/bin/env python
"""Headline benchmark: images/sec of the pruned BLIP-base NLVR2 forward (BLIP_NLVR.forward(train=False) dataflow,
reference models/blip_nlvr.py:63-100) at p=0.5, 64 samples (= 128 images) per GPU, bf16 GEMM operands, on MI355X.

    python bench.py [--gpus N --steps K --warmup W --precision bf16|f16x3|fp32 --config nlvr|retrieval|clip|vqa --batch B]

--config selects one of BASELINE.json's GPU configurations (madtp_amd/workloads.py); the default is the one the metric is
quoted on (config 2, NLVR2 b64).  N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...`: one process per GPU, batch sharded data-parallel (whole samples per rank - both images of an NLVR
pair stay on one rank), weights replicated, NO collective inside the forward (the path has no exchange step, SURVEY.md
8(e)); RCCL is used only for the barrier and the MAX-reduction of the elapsed time.  Scaling is "weak" (retrieval, whose
BASELINE batch of 128 is a global one, is split over the ranks: "strong").

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     - the dominant kernel (the MFMA GEMM): algorithmic FLOPs of its launches in the timed region / the sum of their
                 durations, measured live with HIP events on the launch stream; `traffic` = HBM bytes per launch from two
                 rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that THIS run starts on a 1-step copy of itself; `mfma_busy` /
                 `sq_counters` from a third pass (SQ_VALU_MFMA_BUSY_CYCLES, wave-cycle split; counters for the GEMM kernels only);
                 `by_class` = every Linear class against its own bound (MFMA or HBM) with its tile-quantisation ceiling.
  value        - the bf16 leg (BASELINE.json config 2's dtype); f16_value = the same runner on IEEE f16 operands;
                 parity_qualified_value = the f16x3 leg (every kept set identical to the oracle).  The timed region repeats the K
                 steps in whole passes until it lasts >= 1 s (timed_passes_of_k_steps); ms_per_step is per step.
  parity_mode  - the same workload timed in the precision mode that carries the parity claim ("f16x3": fp32-accurate GEMMs
                 on the f16 MFMA) + index_match of every mode vs the CPU oracle at the headline batch.
  cpu_baseline - oracle/ (the CPU restatement of the reference forward, kind "port") timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md.  f16x3: three f16 MFMA products per logical product, so the
# roofline of the fp32-accurate GEMM in ALGORITHMIC flops (2MNK) is a third of the f16 dense peak.
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "fp32": 157.3, "f16x3": 2500.0 / 3.0}
GEMM_DT = {"bf16": "bf16", "f16": "f16", "fp32": "f32", "f16x3": "f16s"}
WS_MIN_M = 4096  # madtp_gemm runs 2-byte-operand problems with M >= 4096 (and no split-K) on gemm_ws_kernel / gemm_sq_kernel (csrc/gemm.hip)
METRIC = "images/sec forward, BLIP-base NLVR2 p=0.5 b64; pruned-token index match"


def gemm_summary(rows, dtype, min_m=0):
    sel = [r for r in rows if r["dtype"] == dtype and r["M"] >= min_m]
    return sum(r["ms"] for r in sel), sum(r["flops"] for r in sel), sum(r["launches"] for r in sel)


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured with a float4 copy)


def gemm_class(r, rows_patch):
    """the forward's big-GEMM classes by (N, K): the Linear each launch is (vit.py:77,92,31-35; nlvr_encoder.py:177-178)"""
    N, K = r["N"], r["K"]
    if (N, K) == (768, 768):
        return "patch_embed" if r["M"] == rows_patch else "proj"
    return {(2304, 768): "qkv", (3072, 768): "fc1", (768, 3072): "fc2", (1536, 768): "text_kv_pair"}.get((N, K), f"N{N}_K{K}")


def quant_ceiling(M, N, pair=False):
    """best share of full rounds over the three big tiles on 256 CUs: what tile quantisation alone allows a LONE launch"""
    best = 0.0
    for bm, bn in ((256, 256), (192, 256), (256, 128)):
        t = -(-M // bm) * -(-N // bn) * (2 if pair else 1)
        useful = (M * N * (2 if pair else 1)) / float(bm * bn)  # tiles' worth of real output
        best = max(best, useful / (-(-t // 256) * 256))
    return round(best, 3)


def roofline_by_class(rows, dtype, min_m, peak_tf, rows_patch):
    """every class against ITS bound: arithmetic intensity (algorithmic flop / algorithmic byte) below the ridge peak_tf / 8 TB/s
    = HBM-bound (proj: f16 A in, f32 residual in, f32 out), else MFMA-bound."""
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    agg = {}
    for r in rows:
        if r["dtype"] != dtype or r["M"] < min_m:
            continue
        a = agg.setdefault(gemm_class(r, rows_patch), {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0, "q": 0.0})
        a["ms"] += r["ms"]; a["flops"] += r["flops"]; a["bytes"] += r["bytes"]; a["launches"] += r["launches"]
        a["q"] += r["flops"] * quant_ceiling(r["M"], r["N"], pair=(r["N"], r["K"]) == (1536, 768))
    out = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        tf, gbs, ai = a["flops"] / (a["ms"] * 1e-3) / 1e12, a["bytes"] / (a["ms"] * 1e-3) / 1e9, a["flops"] / a["bytes"]
        hbm = ai < ridge
        out[k] = {"bound": "hbm" if hbm else "mfma", "launches": a["launches"], "avg_launch_us": round(1e3 * a["ms"] / a["launches"], 2),
                  "achieved_tflops": round(tf, 1), "achieved_gbs": round(gbs, 1), "flop_per_byte": round(ai, 1),
                  "frac": round(gbs / HBM_PEAK_GBS if hbm else tf / peak_tf, 4),
                  "frac_mfma": round(tf / peak_tf, 4), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
                  "tile_quantisation_ceiling": round(a["q"] / a["flops"], 3), "share_of_big_gemm_time": None}
    tot = sum(a["ms"] for a in agg.values()) or 1.0
    for k in out:
        out[k]["share_of_big_gemm_time"] = round(agg[k]["ms"] / tot, 3)
    return out


def gemm_breakdown(rows, steps):
    out = []
    for r in sorted(rows, key=lambda r: -r["ms"]):
        out.append(f"{r['dtype']:5s} M={r['M']:6d} N={r['N']:5d} K={r['K']:5d} calls/step={r['launches'] / steps:5.1f} "
                   f"ms/step={r['ms'] / steps:7.3f} TFLOP/s={r['flops'] / r['ms'] / 1e9:7.1f}")
    return "\n".join(out)


# the SQ counter pass of measure_traffic (8 SQ slots + the GRBM block; MI355X_MICROARCH.md "rocprofv3 PMC slots")
SQ_PASS = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
           "SQ_INSTS_VALU_MFMA_MOPS_F16", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "GRBM_GUI_ACTIVE")


def sq_summary(c, big):
    """MFMA-busy and the wave-cycle split of the big-GEMM launches of one --pmc pass (tools/rocpd_sq.py prints the same per kernel):
    mfma_busy = SUM SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) = share of the launches' time a SIMD's matrix pipe is busy;
    resident = waves per SIMD while the launch runs / 2 (the big kernels hold two waves per SIMD where they are resident: what tile
    quantisation, ramp-up and drain leave); wait / issue-stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over
    SQ_WAVE_CYCLES (disjoint)."""
    cols = [q[1] for q in c.execute("pragma table_info(pmc_events)").fetchall()]
    key = "dispatch_id" if "dispatch_id" in cols else "start"
    rows = c.execute(f"select name, counter_name, count(distinct {key}), count(*), sum(counter_value) from pmc_events group by name, counter_name").fetchall()
    tot, gui = {}, 0.0
    for name, ctr, nd, nrows, v in rows:
        if not big(name):
            continue
        if ctr == "GRBM_GUI_ACTIVE":
            gui += float(v) / (nrows / nd)  # a free-running cycle count per instance: the instances' average, summed over launches
        else:
            tot[ctr] = tot.get(ctr, 0.0) + float(v)
    if not gui or "SQ_VALU_MFMA_BUSY_CYCLES" not in tot:
        return None
    wc = tot.get("SQ_WAVE_CYCLES", 0.0)
    mops = tot.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0) + tot.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)
    rep = {"mfma_busy": round(tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024), 4),
           "mfma_flop_per_clk_per_cu": round(mops * 512.0 / (gui * 256), 1), "mfma_flop_per_clk_per_cu_peak": 4096,
           "waves_per_simd": round(wc * 4.0 / (gui * 1024), 3) if wc else None,
           "how": "this run: one rocprofv3 --pmc pass (" + " ".join(SQ_PASS) + ") over a 1-step serial copy of the command, big-GEMM launches only"}
    if wc:
        rep.update({"resident": round(min(1.0, wc * 4.0 / (gui * 1024) / 2.0), 4), "wave_cycles_wait": round(tot.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                    "wave_cycles_issue_stall": round(tot.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                    "wave_cycles_active": round(tot.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4)})
        rep["mfma_busy_while_resident"] = round(rep["mfma_busy"] / rep["resident"], 4) if rep["resident"] else None
    return rep


def run_bounded(cmd, cwd, env, timeout):
    """subprocess.run(capture_output) in its own session; on timeout the WHOLE process group is killed (rocprofv3 starts the profiled
    program as a child: killing only the launcher would leave a hung bench.py on the GPU under the timed legs).  -> CompletedProcess
    or None."""
    import signal
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return subprocess.CompletedProcess(cmd, p.returncode, out, err)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)  # (the group this call created - nobody else's processes)
        except ProcessLookupError:
            pass
        try:
            p.communicate(timeout=10)
        except Exception:
            pass
        return None


def measure_traffic(args):
    """HBM-side bytes per launch of the dominant kernels (gemm_ws_kernel / gemm_sq_kernel), collected as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE and WRITE_SIZE in SEPARATE `rocprofv3 --pmc X --kernel-trace` passes over a 1-step run of this very command,
    FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of 16-B/lane streams at 64 B; tools/pmc_calibrate.py confirmed the
    factor for this kernel's access pattern in round 1).  -> dict or None (no rocprofv3 / a pass failed)."""
    import sqlite3
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="madtp_pmc_", dir="/tmp")
    deadline = time.time() + float(os.environ.get("MADTP_TRAFFIC_BUDGET", "200"))
    try:
        big = lambda name: "gemm_ws_kernel" in name or "gemm_sq_kernel" in name or "gemm_pp_kernel" in name  # noqa: E731  (the big-GEMM kernels)
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ"):
            d = os.path.join(tmp, counter)
            pmc = [counter] if counter != "SQ" else list(SQ_PASS)
            # Counters are collected for the big-GEMM kernels ONLY (--kernel-include-regex): with every dispatch of the child under
            # counter collection - the on-device weight generator's torch kernels included - the profiled child hung before its first
            # forward in 6 of 10 runs on some boxes of the pool (stack: the first blocking copy behind those kernels); filtered, 0 of 10
            # on the same box, interleaved (tools/pmc_hang_probe.sh, profiles/r06_pmc_hang_probe.txt), and a pass takes 4 s instead of 6.
            cmd = [exe, "--pmc"] + pmc + ["--kernel-trace", "--kernel-include-regex", "gemm_(ws|pp|sq)_kernel", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__),
                   "--config", args.config, "--precision", args.precision, "--steps", "1", "--warmup", "1", "--traffic", "off", "--min-seconds", "0",
                   "--no-cpu-baseline", "--no-parity", "--no-bf16-leg", "--no-gemm-events", "--inflight", "1"] + (["--batch", str(args.batch)] if args.batch else [])
            env = dict(os.environ, TMPDIR="/tmp")
            # A profiled child occasionally hangs (seen about once in ten calls on the pool's boxes, also as the first GPU user of a
            # call): every pass gets a bounded time and ONE retry, the three passes together at most MADTP_TRAFFIC_BUDGET seconds
            # (a pass normally takes 12-15 s), so that a hang costs the line its counters at worst, never the run its minutes.
            r, dbs = None, []
            for attempt in range(2):
                left = deadline - time.time()
                if left < 20:
                    break
                shutil.rmtree(d, ignore_errors=True)
                r = run_bounded(cmd, "/tmp", env, min(left, float(os.environ.get("MADTP_TRAFFIC_TIMEOUT", "75"))))
                if r is None:
                    print(f"[bench] traffic pass {counter}: attempt {attempt + 1} timed out", file=sys.stderr)
                    continue
                dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
                if r.returncode == 0 and dbs:
                    break
            if r is None:
                if counter == "SQ":
                    break
                return None
            if r.returncode != 0 or not dbs:
                print(f"[bench] traffic pass {counter}: rc {r.returncode}, {len(dbs)} result files; stderr tail: {r.stderr[-600:]}", file=sys.stderr)
                if counter == "SQ":
                    break  # (the byte counters stand on their own)
                return None
            c = sqlite3.connect(dbs[0])
            if counter == "SQ":
                out["SQ"] = sq_summary(c, big)
                continue
            # one row per (dispatch, counter, hardware instance): launches = distinct dispatches, bytes = the sum over the instances
            cols = [q[1] for q in c.execute("pragma table_info(pmc_events)").fetchall()]
            key = "dispatch_id" if "dispatch_id" in cols else "start"
            rows = c.execute(f"select name, count(distinct {key}), sum(counter_value) from pmc_events where counter_name=? group by name",
                             (counter,)).fetchall()
            n = sum(cnt for name, cnt, _ in rows if big(name))
            v = sum(val for name, _, val in rows if big(name))
            if not n:
                return None
            out[counter] = (n, v * 1024.0 / n)  # the counters are in KiB
    except Exception as e:
        print(f"[bench] traffic passes failed: {e!r}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    f_n, f_b = out["FETCH_SIZE"]
    w_n, w_b = out["WRITE_SIZE"]
    return {"bytes_per_launch": int(2 * f_b + w_b), "fetch_size_raw_bytes_per_launch": int(f_b),
            "write_size_bytes_per_launch": int(w_b), "launches_profiled": f_n, "sq": out.get("SQ"),
            "how": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE --kernel-trace (separate passes) over a 1-step copy of the "
                   "command, gemm_ws_kernel + gemm_pp_kernel (+ gemm_sq_kernel) launches only; 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md)"}


def main():
    from madtp_amd import workloads
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)  # a multiple of the 2 / 3 / 4 forwards in flight (equal shares per worker; 24 forwards each: the ramp-up and the tail of the pipeline cost 24 steps ~5 %, 48 ~3 %, 96 ~1.5 % of the 192-step figure)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "fp32", "f16x3"],
                    help="fast modes: bf16 (default: the dtype BASELINE.json config 2 names, so `value` is the metric as written) or "
                         "f16 (IEEE f16 operands on the f16 MFMA - the same kernels and speed within 2 %%, 0.93 instead of 0.17 of the kept "
                         "sets identical to the fp32 oracle at the headline batch; reported as `f16_value` of a bf16 run); parity modes: "
                         "f16x3 (reported as `parity_qualified_value` of every run), fp32")
    ap.add_argument("--config", default="nlvr", choices=list(workloads.NAMES),
                    help="BASELINE.json configuration (default: the headline)")
    ap.add_argument("--batch", type=int, default=0, help="samples per GPU (0 = the configuration's BASELINE batch)")
    ap.add_argument("--image-size", type=int, default=0, help="retrieval only: 224 (default) or 384")
    ap.add_argument("--parity-steps", type=int, default=32, help="timed steps of the parity_mode leg (f16x3 precision)")
    ap.add_argument("--parity-inflight", type=int, default=4, help="forwards in flight in the parity_mode leg (when the headline "
                    "leg runs more than one in flight)")
    ap.add_argument("--parity-batch", type=int, default=0, help="samples of the index_match check (0 = the run's batch: the FULL BASELINE "
                    "batch of every configuration since round 6 - the oracle's CPU forward of 128 / 128 / 32 samples takes 15-60 s)")
    ap.add_argument("--traffic", default="auto", choices=["auto", "off"], help="roofline.traffic from live rocprofv3 --pmc passes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="headline leg only: no parity_mode / bf16 legs, no index_match")
    ap.add_argument("--no-gemm-events", action="store_true")
    ap.add_argument("--no-bf16-leg", action="store_true", help="skip the timed leg in the OTHER fast mode (f16_value of a bf16 run, bf16_value of an f16 run)")
    ap.add_argument("--bf16-steps", type=int, default=32, help="timed steps of that leg")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the timed region repeats the K steps (whole passes of K) until it lasts at least this long; ms_per_step is per step "
                         "(the driver's --steps 20 is 0.1 s of work: ramp-up and tail of the in-flight pipeline were ~5 %% of it)")
    ap.add_argument("--partition", default=None,
                    help="CUs per XCD of the in-flight workers, e.g. 8,8,8,8 (CU-masked streams, madtp_amd/pipeline.py); 'off' = priority "
                         "streams on the whole chip; default: MADTP_INFLIGHT_CUMASK, else the workload's measured default")
    ap.add_argument("--gemm-breakdown", action="store_true", help="per-shape GEMM time table on stderr")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("MADTP_INFLIGHT", "0")),
                    help="forwards in flight per GPU (madtp_amd.pipeline: one host thread + HIP stream + model replica each; "
                         "1 = the serial loop, 0 = the configuration's default).  The K timed steps are K whole forwards either way.")
    args = ap.parse_args()
    if os.environ.get("MADTP_BENCH_WATCHDOG"):  # debugging aid: dump every thread's Python stack after N seconds (tools/pmc_hang_probe.sh)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MADTP_BENCH_WATCHDOG"]), repeat=False, file=sys.stderr)

    from madtp_amd import dist as mdist
    world, rank, local_rank = mdist.env_world()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N` (how the driver starts the 1-GPU run): launch the N ranks ourselves, exactly
        # as the contract's command line does - one process per GPU, rendezvous on 127.0.0.1 (reference launch:
        # scripts/compress_nlvr_nlvr2_p0.5.sh:6 `python -m torch.distributed.run --nproc_per_node=8 ...`, utils.py:254-276)
        raise SystemExit(self_launch(args.gpus))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start {args.gpus} ranks (python -m torch.distributed.run "
                         f"--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or none (bench.py launches them itself)")
    if os.environ.get("MADTP_BENCH_DRY") == "1":
        return dry_run(args, world, rank)
    # roofline.traffic comes from two rocprofv3 --pmc passes over a one-step copy of this command (measure_traffic).  They run
    # FIRST, before this process touches the GPU: launched after the in-flight legs (four streams, two of them high-priority,
    # still alive in this process) the profiled child hung in about every second run until its timeout (profiles/README.md).
    pre_traffic = None
    if (args.traffic == "auto" and rank == 0 and world == 1 and args.precision != "fp32" and not args.no_gemm_events):
        from madtp_amd import build as _build
        _build.build(verbose=False)
        pre_traffic = measure_traffic(args)
    # MADTP_BENCH_ONE_GPU=1 (rehearsal on a one-GPU box: tools/bench_rehearsal.sh): all ranks share GPU 0 and reduce over gloo -
    # RCCL refuses two ranks on one device - so that the multi-rank code path of this file runs on hardware before the driver's
    # 8-GPU run; the figures of such a run mean nothing.
    one_gpu = world > 1 and os.environ.get("MADTP_BENCH_ONE_GPU") == "1"
    gpu_index = 0 if one_gpu else local_rank
    red_dev = "cpu" if one_gpu else "cuda"
    torch.cuda.set_device(gpu_index)
    mdist.init("gloo" if one_gpu else "nccl")  # "nccl" is RCCL on ROCm; no-op for a single process
    dist = torch.distributed if world > 1 else None

    if world > 1:  # each rank's host thread (k hand-over spin, launches) on its own cores, near its GPU
        mdist.pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), device_index=gpu_index)
    from madtp_amd import build, configs, hip, runtime
    if local_rank == 0:  # one builder per NODE (the prebuilt .so normally travels with the snapshot: no-op); the others wait
        build.build(verbose=False)
    if dist is not None:
        dist.barrier()
    hip.load()
    runtime.set_precision(args.precision)
    prof_rows = []

    w = workloads.get(args.config, **({"size": args.image_size} if (args.image_size and args.config == "retrieval") else {}))
    if args.inflight <= 0:
        args.inflight = w.default_inflight  # (a --precision f16x3 / fp32 run: 9.2 k serial, 10.4 / 10.9 k with two / three in flight)
    strong = args.config == "retrieval" and not args.batch  # BASELINE config 3: a GLOBAL batch of 128 over the ranks
    B = args.batch or (max(1, w.default_batch // world) if strong else w.default_batch)
    T, calib = configs.temperature_for(args.config, args.batch or w.default_batch, w.p)
    model = w.build("cuda")
    inp = w.inputs(B, seed=rank)  # resident in HBM before timing

    def step():
        return w.step(model, inp, T)

    runner = None
    if args.inflight > 1:
        from madtp_amd.pipeline import InflightRunner
        # one runner for both legs: the headline uses the first args.inflight workers, the parity_mode leg args.parity_inflight
        from madtp_amd.pipeline import partition_from_env
        n_slots = args.inflight if args.no_parity else max(args.inflight, args.parity_inflight)
        if args.partition:
            os.environ["MADTP_INFLIGHT_CUMASK"] = args.partition
        part = partition_from_env(n_slots, getattr(w, "default_partition", None))
        # ONE set of weights: the workers run shared replicas of `model` (round 6; four full replicas before: 20 GB allocated)
        share = os.environ.get("MADTP_INFLIGHT_SHARE", "1") != "0"
        runner = InflightRunner(w, n_slots, T, B, "cuda", seed0=rank * n_slots, partition=part,
                                models=model if share else [model] + [w.build("cuda") for _ in range(n_slots - 1)])
        runner.inputs[0] = inp

    def run_steps(n):  # n whole forwards: serial on the current stream, or spread over the in-flight workers
        if runner is not None:
            runner.run(n, workers=args.inflight)
        else:
            for _ in range(n):
                step()

    single = None
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        if runner is not None:
            runner.run(max(args.warmup, 3) * args.inflight, workers=args.inflight)  # every replica prepares its weights, grows its workspace and its
            #                                                  stream's allocator pool (>= 3 forwards per worker)
            # the serial loop of the same K steps, for the record (latency of one forward; rounds 1-2 reported this as `value`)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            single = time.perf_counter() - ts
        torch.cuda.synchronize()
        # how many passes of K steps make the timed region >= --min-seconds: from one untimed pass, the same count on every rank
        tp = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        est = mdist.max_over_ranks(time.perf_counter() - tp, device=red_dev)
        repeats = max(1, min(200, int(-(-args.min_seconds // max(est, 1e-4)))))
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(repeats):
            run_steps(args.steps)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0  # this rank's K steps, before it waits for the others
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        hl_high = runner.n_high if runner is not None else 0  # high-priority streams among the workers the headline leg used
        hl_part = runner.last_partition if runner is not None else None  # CUs per XCD of its workers (CU-masked streams), or None
        weight_bytes = sum(p.numel() * p.element_size() for p in model.parameters())
        hbm_alloc = torch.cuda.max_memory_allocated()  # every replica's weights (f32 + prepared compute-dtype copies), inputs, workspaces
        lens = w.lens(model)
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
    elapsed = mdist.max_over_ranks(elapsed, device=red_dev) / repeats  # per pass of K steps
    own_max, own_min = mdist.max_over_ranks(own, device=red_dev) / repeats, mdist.min_over_ranks(own, device=red_dev) / repeats

    images_per_step = w.images_per_sample * B * world
    value = images_per_step * args.steps / elapsed
    flops_sample, flops_full = w.flops(lens), w.flops(None)

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
            alg_bytes = sum(r["bytes"] for r in prof_rows if r["dtype"] == dt_name and r["M"] >= min_m)
            kname = {"f16": "gemm_ws_kernel + gemm_pp_kernel on IEEE f16 operands (v_mfma_f32_16x16x32_f16; all madtp_gemm launches with M >= 4096)",
                     "bf16": "gemm_ws_kernel + gemm_pp_kernel (all bf16 madtp_gemm launches with M >= 4096: ViT qkv/proj/fc1/fc2, cross-attention K/V; 256x128 wave-specialised or 256x256 ping-pong tiles by round count)",
                     "f16x3": "gemm_ws_kernel<f16-split> (all madtp_gemm launches with M >= 4096; 3 f16 MFMA products per "
                              "logical product: achieved/peak are in algorithmic 2MNK flops, peak = f16 dense / 3)",
                     "fp32": "gemm_kernel<float> (madtp_gemm)"}[args.precision]
            traffic = pre_traffic
            sq = (traffic or {}).get("sq")
            classes = roofline_by_class(prof_rows, dt_name, min_m, peak, w.images_per_sample * B * 196)
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 1),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "mfma_busy": sq["mfma_busy"] if sq else None, "sq_counters": sq,
                    "by_class": classes,
                    "note": "`frac` is over ALL big launches against the MFMA peak (the figure of rounds 1-5); by_class prices each Linear "
                            "against its own bound - proj moves an f32 residual stream in and out and sits below the ridge (HBM-bound)",
                    "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_detail": traffic,
                    "algorithmic_bytes_per_launch": round(alg_bytes / cnt),
                    "launches_per_step": cnt // args.steps, "kernel_ms_per_step": round(ms / args.steps, 3),
                    "avg_launch_us": round(1e3 * ms / cnt, 2), "algorithmic_gflop_per_launch": round(fl / cnt / 1e9, 3),
                    "instrumented_ms_per_step": round(1e3 * instr_elapsed / args.steps, 3),
                    "all_gemm_launches": {"achieved": round(fl_all / (ms_all * 1e-3) / 1e12, 1),
                                          "frac": round(fl_all / (ms_all * 1e-3) / 1e12 / peak, 4),
                                          "launches_per_step": cnt_all // args.steps,
                                          "ms_per_step": round(ms_all / args.steps, 3)}}

    headline = args.config == "nlvr"
    shared_weights = (runner is not None and runner.n > 1 and next(runner.models[1].parameters()) is next(model.parameters()))
    out = {
        "metric": METRIC if headline else f"images/sec forward, {args.config} configuration of BASELINE.json (p={w.p})",
        "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "timed_passes_of_k_steps": repeats,
        "timed_seconds": round(elapsed * repeats, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": {"bf16": "bf16", "f16": "f16", "fp32": "f32", "f16x3": "f16x3 (fp32-accurate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": w.describe(B), "samples_per_gpu": B, "images_per_gpu": w.images_per_sample * B, "temperature": T,
                   "p": w.p, "flops_ratio_vs_unpruned": round(flops_sample / flops_full, 4),
                   "tokens_per_layer": lens, "calibrated_at_batch": calib.get("batch"), "parallelism": f"dp{world}",
                   "inflight_per_gpu": args.inflight,
                   "inflight_high_priority_streams": (hl_high if runner is not None else 0),
                   "inflight_cus_per_xcd": (hl_part if runner is not None else None),
                   "model_replicas_resident": (runner.n if (runner is not None and not shared_weights) else 1),
                   "inflight_workers_share_weights": shared_weights,
                   "weight_bytes_per_replica": weight_bytes, "hbm_bytes_allocated": hbm_alloc,
                   "gemm_dispatch_hints_while_in_flight": ({"sq_cost": runner.sq_cost, "small_tile": runner.small_tile}
                                                           if runner is not None else None)},
        "per_rank_ms_per_step": {"min": round(1e3 * own_min / args.steps, 3), "max": round(1e3 * own_max / args.steps, 3),
                                 "what": "each rank's own K steps, clocked before the closing barrier (stragglers show as max >> min)"},
        "samples_per_s": round(value / w.images_per_sample, 1),
        "model_tflops": round(flops_sample * B * world * args.steps / elapsed / 1e12, 1),
        "roofline": roof,
    }
    # First-class companions of `value`: `value` is THROUGHPUT with args.inflight independent forwards in flight on one
    # GPU (args.inflight model replicas); the reference's eval loop runs one batch at a time (compress_nlvr_dtp.py:73-99), which is
    # `serial_value` / `ms_per_forward` here.  ms_per_step = elapsed / steps is the pipeline's issue interval, not a latency.
    if single is not None:
        out["serial_value"] = round(w.images_per_sample * B * world * args.steps / single, 1)
        out["ms_per_forward"] = round(1e3 * single / args.steps, 3)
        out["ms_per_forward_in_flight"] = round(1e3 * elapsed * args.inflight / args.steps, 3)
    else:
        out["serial_value"] = out["value"]
        out["ms_per_forward"] = out["ms_per_step"]
        out["ms_per_forward_in_flight"] = out["ms_per_step"]
    if single is not None:
        out["single_stream"] = {"value": round(w.images_per_sample * B * args.steps / single, 1), "unit": "images/s",
                                "ms_per_forward": round(1e3 * single / args.steps, 3),
                                "what": "the same K forwards one after the other on one stream (this rank; latency of a forward); "
                                        f"`value` runs them {args.inflight} at a time on separate HIP streams (madtp_amd/pipeline.py)"}
    if headline:  # keys of the round-1 line, kept for the driver's records
        out["config"]["vit_tokens_per_layer"], out["config"]["text_tokens_per_layer"] = lens["vit"], lens["text"]

    def timed_leg(mode, steps, workers):
        """`steps` whole forwards of the SAME workload in precision `mode`, same barrier / max-over-ranks protocol as the headline
        leg (warm-up first: every replica prepares that mode's weights)."""
        with runtime.precision(mode), torch.no_grad():
            for _ in range(2):
                step()
            if runner is not None:
                runner.run(3 * workers, workers=workers)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            t2 = time.perf_counter()
            if runner is not None:
                runner.run(steps, workers=workers)
            else:
                for _ in range(steps):
                    step()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t2
        return mdist.max_over_ranks(el, device=red_dev)

    if not args.no_parity:
        # parity_mode leg: the SAME workload timed in the precision mode that carries the parity claim (f16x3: fp32-accurate
        # GEMMs on the f16 MFMA, everything else the fp32 mode's kernels), same barrier / max-over-ranks protocol
        pm = "f16x3"
        # (measured, NLVR f16x3: 9.2 k serial, 10.4 k with two, 10.9 k with three, 11.3 k with four forwards in flight)
        pn = runner.n if runner is not None else 1
        pel = timed_leg(pm, args.parity_steps, pn)
        out["parity_mode"] = {"precision": pm, "value": round(images_per_step * args.parity_steps / pel, 1), "unit": "images/s",
                              "ms_per_step": round(1e3 * pel / args.parity_steps, 3), "steps": args.parity_steps,
                              "inflight_per_gpu": pn,
                              "what": "same workload, every Linear as 3 f16 MFMA products of f16-split operands (fp32-accurate), "
                                      "attention / LayerNorm / scores on the exact-f32 kernels; kept sets vs the oracle below"}
        # The figure that satisfies north_star's parity bar (bit-exact kept indices, logits within 1e-3), as a top-level key next
        # to `value` (the fast mode, whose match RATE is reported in index_match).
        out["parity_qualified_value"] = out["parity_mode"]["value"]
        out["parity_qualified_precision"] = pm
    if args.precision in ("bf16", "f16") and not args.no_bf16_leg and not args.no_parity:  # (--no-parity = the headline leg only: profiling runs)
        # BASELINE.json config 2 names bf16 (`value` of a default run); the same runner timed in the OTHER fast mode - the same
        # kernels on v_mfma_f32_16x16x32_f16 / _bf16 - so that one line carries the bf16 figure as BASELINE writes it, the f16
        # figure (0.93 of the kept sets identical to the oracle instead of 0.17) and the parity-qualified figure.
        other = "f16" if args.precision == "bf16" else "bf16"
        bn = args.inflight if runner is not None else 1
        bel = timed_leg(other, args.bf16_steps, bn)
        out[f"{other}_value"] = round(images_per_step * args.bf16_steps / bel, 1)
        out[f"{other}_ms_per_step"] = round(1e3 * bel / args.bf16_steps, 3)
        out[f"{other}_leg"] = {"steps": args.bf16_steps, "inflight_per_gpu": bn, "unit": "images/s",
                               "what": f"same workload and runner as `value`, {other} GEMM / attention operands; "
                                       f"its kept-set match vs the oracle is index_match.{other}"}
    if rank == 0 and world == 1:
        if not args.no_parity:
            modes = sorted({"fp32", "f16x3", "bf16", "f16", args.precision})
            if headline:
                from oracle.index_match import nlvr_index_match
                im = nlvr_index_match(model, T, modes, B=args.parity_batch or B, seed=11)
            else:
                im = generic_index_match(w, model, args.config, T, modes, args.parity_batch or B)
            out["index_match"] = im
            out["parity_mode"]["index_match"] = im.get("f16x3")
            out["parity_mode"]["index_match_batch"] = im["batch"]
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, w, T)
    if rank == 0 and args.gemm_breakdown and not args.no_gemm_events:
        print(gemm_breakdown(prof_rows, args.steps), file=sys.stderr, flush=True)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dry_run(args, world, rank):
    """MADTP_BENCH_DRY=1: the launch / rendezvous / barrier / max-over-ranks / one-JSON-line protocol of this file with a sleep in
    place of the GPU forward (gloo, no GPU touched) - what tests/test_distributed_cpu.py runs as `python bench.py --gpus 2` so that
    the first multi-GPU lease cannot fail in argument or launcher handling.  The figures mean nothing."""
    from madtp_amd import dist as mdist
    mdist.init("gloo")
    dist = torch.distributed if world > 1 else None
    for _ in range(args.warmup):
        time.sleep(0.0005)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.0005)
    if dist is not None:
        dist.barrier()
    elapsed = mdist.max_over_ranks(time.perf_counter() - t0, device="cpu")
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": round(128 * world * args.steps / elapsed, 1), "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
                          "data": "dry-run: sleep instead of the forward, no GPU work (MADTP_BENCH_DRY=1)",
                          "config": {"workload": "dry-run", "parallelism": f"dp{world}"}, "dry_run": True}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-executes this command line under torch.distributed.run with N ranks on this
    node and returns its exit code (rank 0 of the child prints the JSON line)."""
    import socket
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.call(cmd, env=env)


def _flat(o):
    if torch.is_tensor(o):
        return [o.detach().float().cpu()]
    return [t for x in o for t in _flat(x)]


def generic_index_match(w, model, name, T, modes, B, seed=11):
    """per-layer token counts and outputs of every precision mode vs the CPU oracle on B samples (the oracle is the checker
    here; kept-SET equality per layer is asserted by the -m gpu tests on the reference fixtures)."""
    from madtp_amd import runtime
    from oracle import workloads as OW
    size = getattr(w, "size", 224)
    W = OW.weights(name, size)
    t0 = time.perf_counter()
    ref_out, ref_lens = OW.forward(name, W, B, T, seed, size)
    rep = {"batch": B, "temperature": T, "oracle": "oracle/workloads.py -> oracle/madtp_oracle.py (CPU fp32 restatement)",
           "oracle_forward_s": round(time.perf_counter() - t0, 2), "oracle_tokens_per_layer": ref_lens}
    inp = w.inputs(B, seed)
    for mode in modes:
        with runtime.precision(mode), torch.no_grad():
            out = w.step(model, inp, T)
            lens = w.lens(model)
        same = all(lens[k] == ref_lens[k] for k in lens)
        outs, refs = _flat(out), _flat(ref_out)
        err = max((a - b).abs().max().item() for a, b in zip(outs, refs)) if all(a.shape == b.shape for a, b in zip(outs, refs)) else None
        rep[mode] = {"tokens_per_layer_equal": same, "max_abs_dout": None if err is None else round(err, 6)}
    return rep


def cpu_baseline(name, w, T, budget_s=12.0):
    """The CPU oracle (a port of the reference forward; the reference itself cannot travel to this box) timed on the
    host cores: a small batch of the same synthetic workload, repeated for ~budget_s seconds."""
    from oracle import workloads as OW
    B = {"nlvr": 8, "retrieval": 8, "clip": 8, "vqa": 2}[name]
    size = getattr(w, "size", 224)
    W = OW.weights(name, size)
    OW.forward(name, W, B, T, 0, size)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        OW.forward(name, W, B, T, 0, size)
        n += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or n >= 50:
            break
    rep = {"value": round(w.images_per_sample * B * n / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n} forwards of {B} samples ({w.images_per_sample * B} images) of the same workload, same weights/temperature, "
                     f"PyTorch CPU eager fp32 via oracle/madtp_oracle.py, {dt:.1f}s",
           "host_cpus": os.cpu_count()}
    ref = reference_proper(name)
    if ref:
        rep["reference_proper"] = ref
    return rep


def reference_proper(name):
    """The reference ITSELF (imported behind shims by tools/make_golden.py in the build container - it cannot travel to the GPU
    box) timed while the golden fixtures were recorded: one forward per fixture, wall seconds and thread count stored in the
    committed .npz.  A reported figure beside the port's, on different (container) cores - BASELINE.md section 3."""
    import glob
    import numpy as np
    pat = {"nlvr": "nlvr_b*.npz", "retrieval": "med_mm_*.npz", "clip": "clip_full_*.npz", "vqa": "vqa_*.npz"}[name]
    rows = []
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", pat))):
        try:
            g = np.load(path)
            if "ref_seconds" not in g.files:
                continue
            B = int(g["B"]) if "B" in g.files else 0
            imgs = B * (2 if name == "nlvr" else 1)
            rows.append({"fixture": os.path.basename(path), "images": imgs, "seconds": round(float(g["ref_seconds"]), 3),
                         "warm_median_of": (int(len(g["ref_seconds_runs"])) if "ref_seconds_runs" in g.files else 1),
                         "threads": int(g["threads"]) if "threads" in g.files else None,
                         "images_per_s": round(imgs / float(g["ref_seconds"]), 2) if imgs else None})
        except Exception:  # a fixture without the fields is simply not reported
            continue
    if not rows:
        return None
    return {"what": "the reference's own modules (CPU eager fp32, imported in the build container by tools/make_golden.py), one "
                    "forward per committed fixture: the median of 5 warm runs where the fixture records it (ref_seconds_runs), else one un-warmed "
                    "call", "fixtures": rows}


if __name__ == "__main__":
    main()
