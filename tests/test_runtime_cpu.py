"""Host logic of madtp_amd.runtime that needs no GPU."""
import torch


def test_registration_hooks_count_only_owned_modules():
    """The process-wide torch registration hooks (parameter re-assignment / replaced sub-module -> the layers re-collect their
    parameter lists) ignore modules that are not part of a mirror model which has collected its parameters: building unrelated
    modules elsewhere in the process must not invalidate anything (round-3 advice)."""
    from madtp_amd import runtime, vit
    e0 = runtime.param_epoch()
    torch.nn.Linear(4, 4)
    torch.nn.TransformerEncoderLayer(16, 2, 32)
    assert runtime.param_epoch() == e0, "a foreign module moved the epoch"
    blk = vit.Block(768, 12, qkv_bias=True)
    assert runtime.param_epoch() == e0, "constructing a mirror block is not a re-assignment"
    runtime.own_modules(blk)  # what Block._weights() does when it collects its parameter list
    blk.mlp.fc1.weight = torch.nn.Parameter(torch.zeros(3072, 768))
    assert runtime.param_epoch() == e0 + 1
    blk.mlp.fc2 = torch.nn.Linear(3072, 768)
    assert runtime.param_epoch() == e0 + 2
    other = vit.Block(768, 12, qkv_bias=True)  # never collected: nothing cached, nothing to invalidate
    other.mlp.fc1.weight = torch.nn.Parameter(torch.zeros(3072, 768))
    assert runtime.param_epoch() == e0 + 2


def test_score_fast_follows_the_precision_mode():
    from madtp_amd import hip, runtime
    prev = runtime.get_precision()
    try:
        for mode, want in (("fp32", 0), ("f16", 1), ("f16x3", 0), ("bf16", 1)):
            runtime.set_precision(mode)
            assert hip._score_fast == (want if runtime._SCORE_FAST else 0), mode
    finally:
        runtime.set_precision(prev)
