"""One -m gpu test per BASELINE configuration at its CALIBRATED temperature (madtp_amd/configs.py): the workload of bench.py
--config X (madtp_amd/workloads.py) on a small batch vs the CPU oracle's forward of the same workload (oracle/workloads.py):
identical per-layer token counts in the fp32 and f16x3 modes, outputs within 1e-3; f16 stays close, bf16 in the neighbourhood."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _flat(o):
    if torch.is_tensor(o):
        return [o.detach().float().cpu()]
    return [t for x in o for t in _flat(x)]


@pytest.mark.parametrize("name,B", [("nlvr", 4), ("retrieval", 6), ("clip", 6), ("vqa", 2)])
def test_workload_matches_oracle_at_calibrated_temperature(name, B):
    from madtp_amd import build, configs, hip, runtime, workloads
    from oracle import workloads as OW
    build.build(verbose=False)
    hip.load()
    w = workloads.get(name)
    T, rec = configs.temperature_for(name, w.default_batch, w.p)
    assert abs(rec["flops_ratio"] - (1 - w.p)) < 0.02 * (1 - w.p) + 0.005, rec["flops_ratio"]   # +-2 %
    size = getattr(w, "size", 224)
    ref_out, ref_lens = OW.forward(name, OW.weights(name, size), B, T, 3, size)
    model = w.build("cuda")
    inp = w.inputs(B, 3)
    for mode in ("fp32", "f16x3"):
        with runtime.precision(mode), torch.no_grad():
            out = w.step(model, inp, T)
            lens = w.lens(model)
        for k in lens:
            assert lens[k] == ref_lens[k], (mode, k, lens[k], ref_lens[k])
        for a, b in zip(_flat(out), _flat(ref_out)):
            assert a.shape == b.shape and (a - b).abs().max().item() < 1e-3, (mode, (a - b).abs().max().item())
    # fast modes: no bit-exact claim (a rounding difference can flip a token decision, and the sequence lengths - hence the
    # outputs - then differ from the oracle's from that layer on); f16 (11 significand bits) stays close, bf16 (8 bits) in the
    # neighbourhood.  At the benchmark batch the decision-level agreement is measured by bench.py's index_match block.
    for mode, tol in (("f16", 0.05), ("bf16", 0.35)):
        with runtime.precision(mode), torch.no_grad():
            outb = w.step(model, inp, T)
        for a, b in zip(_flat(outb), _flat(ref_out)):
            if a.shape == b.shape:
                err = (a - b).abs().max().item()
                print(f"{name} {mode}: max |dout| {err:.4f} (|out| max {b.abs().max().item():.3f})")
                assert torch.isfinite(a).all() and err < tol * max(1.0, b.abs().max().item()), (mode, err)
    # the analytic FLOP counter is consistent: pruned < unpruned, ratio in (0, 1)
    assert 0 < w.flops(lens) < w.flops(None)


@pytest.mark.parametrize("name,task,B", [("nlvr", "retrieval", 8), ("clip", "retrieval_clip", 8)])
def test_controller_closes_around_the_hip_forward(name, task, B):
    """SURVEY 8(f) rank 3 on the GPU: the drivers' calculate_temperature() search (madtp_amd/controller.py, the ladder of the
    named driver) with Cur_Gflops measured on the HIP forward itself - the analytic counter on the token counts the kernels
    actually kept - converges to (1 - p) x the workload's unpruned GFLOPs within the driver's tolerance, and the per-epoch
    controller started there stays inside one of its small steps of the target."""
    from madtp_amd import build, controller as C, hip, runtime, workloads
    build.build(verbose=False)
    hip.load()
    w = workloads.get(name)
    model = w.build("cuda")
    inp = w.inputs(B, 5)
    full = C.workload_gflops(w, None)
    target = full * (1 - w.p)
    seen = []

    def measure(T):
        if T <= 0:
            return full  # temperature 0 = no pruning (vit.py:192)
        with runtime.precision("f16x3"), torch.no_grad():
            w.step(model, inp, T)
        g = C.workload_gflops(w, w.lens(model))
        seen.append((T, g))
        return g

    tol = C.SEARCH[task][1]
    # the drivers' figures are O(100) GFLOPs; rescale this workload's so that their absolute thresholds mean the same thing
    scale = C.ORI_GFLOPS[task] / full
    cur, T = C.calculate_temperature(lambda t: measure(t) * scale, full * scale, target * scale, task, max_iters=400)
    assert abs(cur - target * scale) <= tol and T > 0, (cur, target * scale, T, seen[-3:])
    assert len(seen) >= 3 and seen[-1][1] < 0.9 * full
    log = C.run_controller(lambda t: measure(t) * scale, T, w.p, C.ORI_GFLOPS[task], 6, task)
    assert all(abs(c - target * scale) <= 2 * tol for _, _, c in log), log


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,B,mode", [("nlvr", 4, "fp32"), ("nlvr", 6, "bf16"), ("retrieval", 6, "f16x3")])
def test_inflight_runner_equals_serial_forwards(name, B, mode):
    """madtp_amd/pipeline.py: forwards in flight on separate host threads / HIP streams / model replicas give EXACTLY the
    outputs of the same forwards run one after the other (same kernels, per-stream scratch, thread-local precision mode), also
    when the steps of the workers interleave many times; the workers ran the encoder-level C entry points."""
    from madtp_amd import build, configs, hip, runtime, workloads
    from madtp_amd.pipeline import InflightRunner
    build.build(verbose=False)
    hip.load()
    w = workloads.get(name)
    T, _ = configs.temperature_for(name, w.default_batch, w.p)
    with runtime.precision(mode), torch.no_grad():
        runner = InflightRunner(w, 2, T, B, "cuda", seed0=7)
        serial = [[t.clone() for t in _flat(w.step(runner.models[i], runner.inputs[i], T))] for i in range(2)]
        lens = [w.lens(runner.models[i]) for i in range(2)]
        for steps in (2, 7, 12):
            runner.run(steps)
            for i in range(2):
                for a, b in zip(_flat(runner.last[i]), serial[i]):
                    assert torch.equal(a, b), (steps, i, (a - b).abs().max().item())
                assert w.lens(runner.models[i]) == lens[i]
        # a subset of the slots (bench.py: headline on two, parity mode on three workers of ONE runner): the idle slot keeps its output
        keep = [t.clone() for t in _flat(runner.last[1])]
        runner.last[0] = None
        runner.run(3, workers=1)
        assert all(torch.equal(a, b) for a, b in zip(_flat(runner.last[0]), serial[0]))
        assert all(torch.equal(a, b) for a, b in zip(_flat(runner.last[1]), keep))
    assert runtime.get_precision() == "bf16"  # the caller's (default) mode is untouched outside the context
