"""Deterministic synthetic weights and inputs.

There is no network, so no pretrained BLIP/CLIP checkpoint exists on either box; 259 M parameters are
also far too many to commit.  Both the golden-vector generator (this container, reference imported)
and the tests/bench on the GPU box therefore REGENERATE identical weights from a counter-based integer
hash: value = f(seed, parameter name, flat element index).  Pure uint32 arithmetic in numpy followed by
two exactly-rounded float32 ops, so the bits are identical on every machine and independent of the
torch RNG.  (SURVEY.md section 7 step 1 / section 8(c) "what never travels".)
"""
import math

import numpy as np
import torch

_U32 = np.uint32


def _fnv1a(name: str, seed: int) -> int:
    h = 0x811C9DC5 ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)
    for ch in name.encode():
        h ^= ch
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


def hash_u32(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n well-mixed uint32 values for stream `name` (lowbias32 finaliser over a Weyl sequence)."""
    x = np.arange(n, dtype=np.uint64)
    x = ((x * np.uint64(0x9E3779B9) + np.uint64(_fnv1a(name, seed))) & np.uint64(0xFFFFFFFF)).astype(_U32)
    x ^= x >> _U32(16)
    x *= _U32(0x7FEB352D)
    x ^= x >> _U32(15)
    x *= _U32(0x846CA68B)
    x ^= x >> _U32(16)
    return x


def uniform_pm1(name: str, n: int, seed: int = 0) -> np.ndarray:
    """float32 uniform in [-1, 1) with 24-bit resolution."""
    u = (hash_u32(name, n, seed) >> _U32(8)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return (u - np.float32(0.5)) * np.float32(2.0)


_SQRT3 = math.sqrt(3.0)


def _std_for(name: str, shape) -> tuple:
    """(mean, std) of the synthetic value for a parameter, by reference naming convention."""
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    is_norm = parent.startswith(("norm", "LayerNorm", "ln_")) or parent in ("norm",) or name.startswith("norm.")
    if len(shape) == 1:
        if is_norm and leaf == "weight":
            return 1.0, 0.05
        if is_norm and leaf == "bias":
            return 0.0, 0.02
        return 0.0, 0.02  # Linear / conv biases: non-zero so a dropped bias is caught by parity
    if leaf == "space_dict" or name.endswith("space_dict"):
        return 0.0, 1.0  # nn.Parameter(torch.randn(sd_num, sd_dim)) models/blip_nlvr.py:46
    if name.endswith(("positional_embedding", "class_embedding")):
        return 0.0, 0.02
    return 0.0, 0.02  # trunc_normal_(std=.02) vit.py:266-270; BERT initializer_range 0.02


def uniform_pm1_torch(name: str, n: int, seed: int, device) -> torch.Tensor:
    """uniform_pm1() evaluated with torch integer ops on `device` - the same bits: the 32-bit wrap-around products are the low
    32 bits of int64 products (two's-complement overflow does not touch them), and every float step is one exactly-rounded
    IEEE operation.  Lets each rank of a multi-GPU job generate its 259 M weights on its own GPU instead of on shared host
    cores (tests/test_synth_cpu.py pins it to the numpy path)."""
    M = 0xFFFFFFFF
    x = torch.arange(n, dtype=torch.int64, device=device)
    x = (x * 0x9E3779B9 + _fnv1a(name, seed)) & M
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & M
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & M
    x = x ^ (x >> 16)
    u = (x >> 8).to(torch.float32) * (1.0 / (1 << 24))
    return (u - 0.5) * 2.0


def synth_tensor(name: str, shape, seed: int = 0, dtype=torch.float32, device=None) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    mean, std = _std_for(name, shape)
    if device is not None and torch.device(device).type != "cpu":
        v = uniform_pm1_torch(name, n, seed, device) * float(np.float32(std * _SQRT3))
        if mean != 0.0:
            v = v + float(np.float32(mean))
        return v.reshape(shape).to(dtype)
    v = uniform_pm1(name, n, seed) * np.float32(std * _SQRT3)
    if mean != 0.0:
        v = v + np.float32(mean)
    return torch.from_numpy(v.reshape(shape)).to(dtype)


def fill_state_dict(module_or_sd, seed: int = 0, prefix: str = ""):
    """Returns {key: tensor} with synthetic values for every floating-point entry of a state_dict
    (integer buffers such as position_ids are left as they are)."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            out[k] = synth_tensor(prefix + k, v.shape, seed)
        else:
            out[k] = v.clone()
    return out


def synth_images(n: int, size: int = 224, seed: int = 0, device=None, on_device: bool = None) -> torch.Tensor:
    """[n,3,size,size] fp32, zero-mean unit-variance i.i.d. (uniform) pixels.  device = a GPU: the SAME bits generated there with torch
    integer ops (uniform_pm1_torch; tests/test_synth_cpu.py pins the two paths to each other) instead of a 77 MB pageable host-to-device
    copy - round 6: under `rocprofv3 --pmc` that copy hung in up to half of the runs on some boxes of the pool (the stack of the
    hung bench.py child, tools/pmc_hang_probe.sh), which cost bench.py's counter passes their time-outs.  on_device forces the path."""
    use_torch = on_device if on_device is not None else (device is not None and torch.device(device).type != "cpu")
    if use_torch:
        v = uniform_pm1_torch("images", n * 3 * size * size, seed, device if device is not None else "cpu") * float(np.float32(_SQRT3))
        return v.reshape(n, 3, size, size)
    v = uniform_pm1("images", n * 3 * size * size, seed) * np.float32(_SQRT3)
    t = torch.from_numpy(v.reshape(n, 3, size, size))
    return t if device is None else t.to(device)


def synth_clip_tokens(batch: int, ctx: int = 77, seed: int = 0, min_len: int = 6, max_len: int = 40, sot: int = 49406,
                      eot: int = 49407) -> torch.Tensor:
    """[batch, ctx] int64 CLIP-style token rows: SOT, `len_b` word ids in [1000, 40000), EOT (the highest id of the row -
    clip/model.py:501 finds it with argmax), zero padding."""
    h = hash_u32("clip_tokens", batch * ctx, seed).astype(np.int64)
    ln = hash_u32("clip_lens", batch, seed).astype(np.int64)
    ids = np.zeros((batch, ctx), dtype=np.int64)
    for b in range(batch):
        n = int(min_len + ln[b] % (max_len - min_len + 1))
        ids[b, 0] = sot
        ids[b, 1:1 + n] = 1000 + h[b * ctx:b * ctx + n] % 39000
        ids[b, 1 + n] = eot
    return torch.from_numpy(ids)


def synth_token_ids(batch: int, length: int, seed: int = 0, lo: int = 1000, hi: int = 30000,
                    first_id=None) -> torch.Tensor:
    """[batch,length] int64 ids uniform in [lo,hi); position 0 optionally overwritten (the reference
    writes tokenizer.enc_token_id there, models/blip_nlvr.py:69)."""
    h = hash_u32("token_ids", batch * length, seed).astype(np.int64)
    ids = lo + (h % (hi - lo))
    ids = torch.from_numpy(ids.reshape(batch, length))
    if first_id is not None:
        ids[:, 0] = first_id
    return ids


def synth_answer_ids(n: int, length: int, seed: int = 0, bos: int = 30522, sep: int = 102, min_len: int = 1, max_len=None):
    """(ids, attention_mask) [n, length] int64 of candidate answers as the VQA driver tokenises them
    (compress_vqa_dtp.py: tokenizer(answer_list, padding='longest'); input_ids[:,0] = bos_token_id): [DEC], 1..max_len word
    ids in [1000, 30000), [SEP], zero padding."""
    max_len = length - 2 if max_len is None else max_len
    h = hash_u32("answer_ids", n * length, seed).astype(np.int64)
    ln = hash_u32("answer_lens", n, seed).astype(np.int64)
    ids = np.zeros((n, length), dtype=np.int64)
    att = np.zeros((n, length), dtype=np.int64)
    for a in range(n):
        m = int(min_len + ln[a] % (max_len - min_len + 1))
        ids[a, 0] = bos
        ids[a, 1:1 + m] = 1000 + h[a * length:a * length + m] % 29000
        ids[a, 1 + m] = sep
        att[a, :m + 2] = 1
    return torch.from_numpy(ids), torch.from_numpy(att)
