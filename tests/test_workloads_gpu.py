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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["retrieval", "clip", "vqa"])
def test_workload_at_its_full_baseline_batch_matches_oracle(name):
    """BASELINE configurations 3-5 at their FULL batch (128 / 128 / 32 samples - the sizes bench.py --config X runs; the test above
    uses 6 / 6 / 2): in the parity-carrying mode (f16x3) every layer's token count equals the CPU oracle's and the outputs agree
    within 1e-3 (measured 2e-6 / 0.0 / 6e-6); the oracle's CPU forward of the full batch takes 11-20 s on the GPU box's host cores."""
    from madtp_amd import build, configs, hip, runtime, workloads
    from oracle import workloads as OW
    build.build(verbose=False)
    hip.load()
    w = workloads.get(name)
    B = w.default_batch
    T, _ = configs.temperature_for(name, B, w.p)
    size = getattr(w, "size", 224)
    ref_out, ref_lens = OW.forward(name, OW.weights(name, size), B, T, 11, size)
    model = w.build("cuda")
    inp = w.inputs(B, 11)
    with runtime.precision("f16x3"), torch.no_grad():
        out = w.step(model, inp, T)
        lens = w.lens(model)
    for k in lens:
        assert lens[k] == ref_lens[k], (name, k, lens[k], ref_lens[k])
    for a, b in zip(_flat(out), _flat(ref_out)):
        assert a.shape == b.shape and (a - b).abs().max().item() < 1e-3, (name, (a - b).abs().max().item())


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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,B,mode,part", [("nlvr", 4, "f16x3", [8, 8, 8, 8]), ("nlvr", 6, "f16", [16, 16]), ("nlvr", 4, "fp32", [12, 12, 4, 4]),
                                              ("retrieval", 6, "bf16", [11, 11, 10])])
def test_partitioned_runner_with_shared_weights_equals_serial_forwards(name, B, mode, part):
    """Round 6: the workers run on CU-masked streams (hip.MaskedStream: CUs [c0, c0+n) of every XCD, persistent GEMM grids sized
    for the slice) and SHARE one set of parameters and prepared weights (pipeline.shared_replica).  Outputs, per-layer token
    counts and every worker's records are bit-identical to the same forwards run one after the other on the whole chip."""
    from madtp_amd import build, configs, hip, runtime, workloads
    from madtp_amd.pipeline import InflightRunner, shared_replica
    build.build(verbose=False)
    hip.load()
    w = workloads.get(name)
    T, _ = configs.temperature_for(name, w.default_batch, w.p)
    n = len(part)
    with runtime.precision(mode), torch.no_grad():
        model = w.build("cuda")
        runner = InflightRunner(w, n, T, B, "cuda", seed0=3, models=model, partition=part)
        assert runner.models[0] is model
        p0 = {k: v.data_ptr() for k, v in model.named_parameters()}
        for r in runner.models[1:]:
            assert r is not model and {k: v.data_ptr() for k, v in r.named_parameters()} == p0  # one set of weights
            assert all(a is not b for a, b in zip(r.modules(), model.modules()))                   # distinct record holders
        serial = [[t.clone() for t in _flat(w.step(model, runner.inputs[i], T))] for i in range(n)]
        lens = []
        for i in range(n):
            w.step(runner.models[i], runner.inputs[i], T)
            lens.append(w.lens(runner.models[i]))
        mem0 = torch.cuda.memory_allocated()
        for steps in (n, 3 * n + 1, 5 * n):
            runner.run(steps)
            assert runner.last_partition == part and runner.n_high == 0
            for i in range(n):
                for a, b in zip(_flat(runner.last[i]), serial[i]):
                    assert torch.equal(a, b), (steps, i, (a - b).abs().max().item())
                assert w.lens(runner.models[i]) == lens[i]
        for i in range(n):
            assert hip.stream_get_sched(runner.streams[i])[0] == part[i]
        # the replicas added no parameter memory: what the runs left allocated is scratch, far below one more replica (~1 GB)
        assert torch.cuda.memory_allocated() - mem0 < 0.6 * sum(p.numel() * 4 for p in model.parameters())
        # fewer workers than slots: the same CUs split evenly
        runner.run(4, workers=2)
        assert len(runner.last_partition) == 2 and sum(runner.last_partition) == sum(part)
        for i in range(2):
            assert all(torch.equal(a, b) for a, b in zip(_flat(runner.last[i]), serial[i]))


def test_stream_sched_attributes_and_cu_mask_words():
    from madtp_amd import build, hip
    build.build(verbose=False)
    hip.load()
    words = hip.cu_mask_words(8, 8)
    assert len(words) == 8 and sum(bin(x).count("1") for x in words) == 64
    assert all(((words[i // 32] >> (i % 32)) & 1) == (1 if 8 <= i // 8 < 16 else 0) for i in range(256))
    s = torch.cuda.Stream()
    assert hip.stream_get_sched(s) == (32, -1.0, -2)
    hip.stream_set_sched(s, 8, 0.9, 0)
    assert hip.stream_get_sched(s) == (8, pytest.approx(0.9), 0)
    hip.stream_set_sched(s, 0, 0.0, -2)  # CUs unchanged, hints back to the process-wide ones
    assert hip.stream_get_sched(s) == (8, -1.0, -2)
    # a GEMM on a masked stream gives the bits of the same GEMM on the whole chip (persistent grid of 8 x 8 workgroups)
    a = torch.randn(12288, 768, device="cuda").bfloat16()
    wt = torch.randn(2304, 768, device="cuda").bfloat16()
    ref = hip.gemm(a, wt)
    m = hip.MaskedStream(4, 8)
    m.stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(m.stream):
        out = hip.gemm(a, wt)
    m.stream.synchronize()
    assert torch.equal(out, ref)
    ptr = m.ptr
    m.close()  # back to the pool (HIP streams under torch's allocator are never destroyed): the next handle of that range reuses it
    assert hip.masked_stream_info(ptr) is None
    m2 = hip.MaskedStream(4, 8)
    assert m2.ptr == ptr and hip.stream_get_sched(m2.stream)[0] == 8
