"""Host-side helpers of the multi-GPU start-up: the on-device weight generator is the numpy generator bit for bit, and the
rank -> host-core pinning rule."""
import os

import numpy as np
import torch

from madtp_amd import dist as mdist, specs, synth


def test_torch_generator_equals_numpy_generator():
    for name, shape, seed in (("visual_encoder.blocks.3.attn.qkv.weight", (2304, 768), 0), ("space_dict", (100, 768), 2),
                              ("text_encoder.encoder.layer.0.attention.output.LayerNorm.weight", (768,), 1),
                              ("visual_encoder.pos_embed", (1, 197, 768), 5)):
        n = int(np.prod(shape))
        assert np.array_equal(synth.uniform_pm1(name, n, seed), synth.uniform_pm1_torch(name, n, seed, "cpu").numpy())
        a = synth.synth_tensor(name, shape, seed)
        # force the torch path through a non-cpu device type when one exists; on CPU-only boxes compare the pieces
        mean, std = synth._std_for(name, shape)
        v = synth.uniform_pm1_torch(name, n, seed, "cpu") * float(np.float32(std * synth._SQRT3))
        if mean != 0.0:
            v = v + float(np.float32(mean))
        assert torch.equal(a, v.reshape(shape))
    sd = specs.synth_weights({"a.weight": (4, 8), "a.position_ids": ("int64", 1, 6)}, 0, device="cpu")
    assert sd["a.position_ids"].tolist() == [[0, 1, 2, 3, 4, 5]] and sd["a.weight"].shape == (4, 8)


def test_pin_rank_to_cores_splits_the_mask():
    if not hasattr(os, "sched_getaffinity"):
        return
    before = os.sched_getaffinity(0)
    min_cores = mdist.MIN_CORES_PER_RANK
    try:
        allowed = sorted(before)
        if len(allowed) < 2:
            return
        # a slice smaller than the rank's own thread count is not applied (main thread + two in-flight workers)
        mdist.MIN_CORES_PER_RANK = len(allowed)
        assert mdist.pin_rank_to_cores(0, 2) == set() and os.sched_getaffinity(0) == before
        mdist.MIN_CORES_PER_RANK = 1
        sets = []
        for r in range(2):
            os.sched_setaffinity(0, before)
            sets.append(mdist.pin_rank_to_cores(r, 2))
            assert os.sched_getaffinity(0) == sets[-1]
        assert sets[0] and sets[1] and not (sets[0] & sets[1]) and (sets[0] | sets[1]) <= set(allowed)
        # GPU-local node given: the rank stays inside it, ranks sharing the node split it
        os.sched_setaffinity(0, before)
        node = set(allowed[: len(allowed) // 2])
        got = mdist.pin_rank_to_cores(1, 4, cpus_of_gpu=node)
        assert got and got <= node
        os.sched_setaffinity(0, before)
        assert mdist.pin_rank_to_cores(0, 1) == set()          # single process: untouched
        assert os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)
        mdist.MIN_CORES_PER_RANK = min_cores
    assert mdist._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}


def test_query_model_rejects_dictionaries_the_kernels_cannot_hold():
    """the encoder-level calls carve their logits slabs at 128 floats per row; a space dictionary of
    more than 128 entries must fail loudly before any kernel runs (reference configs: sd_num = 100)."""
    import pytest
    from madtp_amd.utils import Query_model
    qm = Query_model(ft_dim=768, sd_dim=768)
    with pytest.raises(NotImplementedError, match="128"):
        qm._dictionary(torch.randn(200, 768))
    with pytest.raises(NotImplementedError, match="128"):
        qm.encoder_args(torch.randn(200, 768), 2, 768, "cpu")


def test_synth_images_torch_path_gives_the_numpy_paths_bits():
    """synth_images(device=<GPU>) generates the pixels on the device with torch integer ops (no large host-to-device copy: round 6,
    the rocprofv3 --pmc hang); forced onto the CPU here, the torch path must give the numpy path's bits."""
    import torch
    from madtp_amd import synth
    for n, size, seed in ((2, 224, 0), (3, 32, 7), (1, 480, 11)):
        a = synth.synth_images(n, size, seed)
        b = synth.synth_images(n, size, seed, device="cpu", on_device=True)
        assert a.dtype == b.dtype == torch.float32 and a.shape == b.shape and torch.equal(a, b)
