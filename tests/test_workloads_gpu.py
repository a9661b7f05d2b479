"""One -m gpu test per BASELINE configuration at its CALIBRATED temperature (madtp_amd/configs.py): the workload of bench.py
--config X (madtp_amd/workloads.py) on a small batch vs the CPU oracle's forward of the same workload (oracle/workloads.py):
identical per-layer token counts in the fp32 and f16x3 modes, outputs within 1e-3; bf16 stays close."""
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
    with runtime.precision("bf16"), torch.no_grad():
        outb = w.step(model, inp, T)
    for a, b in zip(_flat(outb), _flat(ref_out)):
        if a.shape == b.shape:
            err = (a - b).abs().max().item()
            print(f"{name} bf16: max |dout| {err:.4f} (|out| max {b.abs().max().item():.3f})")
            assert torch.isfinite(a).all() and err < 0.15 * max(1.0, b.abs().max().item())
    # the analytic FLOP counter is consistent: pruned < unpruned, ratio in (0, 1)
    assert 0 < w.flops(lens) < w.flops(None)
