"""ctypes binding of libmadtp_hip.so (include/madtp_hip.h) + thin tensor-level wrappers.

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic op on the product path is one
of the hand-written gfx950 kernels behind the C-ABI.  There is NO fallback: if the library is missing or a kernel
rejects its arguments a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_size_t, c_void_p

import torch

F32, BF16, F16S, F16 = 0, 1, 2, 3  # F16S: f16-split operand planes of the fp32-accurate GEMM (include/madtp_hip.h), torch.float16
#                                    F16: plain IEEE f16 operands (the "f16" fast mode), see set_lp_format below
ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_RELU = 0, 1, 2, 3
ABI_VERSION = 29

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmadtp_hip.so")

_SIGS = {
    "madtp_abi_version": (c_int, []),
    "madtp_strerror": (ctypes.c_char_p, [c_int]),
    "madtp_profile_begin": (c_int, []),
    "madtp_profile_end": (c_int, [ctypes.c_char_p, c_int]),
    "madtp_gemm": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "madtp_gemm_set_config": (c_int, [c_int]),
    "madtp_gemm_set_sq_cost": (c_float, [c_float]),
    "madtp_gemm_set_small_tile": (c_int, [c_int]),
    "madtp_stream_create_cumask": (c_int, [c_void_p, c_void_p, c_int]),
    "madtp_stream_set_sched": (c_int, [c_void_p, c_int, c_float, c_int]),
    "madtp_stream_get_sched": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "madtp_stream_destroy": (c_int, [c_void_p]),
    "madtp_set_score_fast": (c_int, [c_int]),
    "madtp_gemm_splitk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "madtp_splitk_ln": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_float, c_float, c_float, c_void_p]),
    "madtp_layernorm": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_float, c_void_p]),
    "madtp_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "madtp_assemble_tokens": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p]),
    "madtp_bert_embed": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "madtp_attention": (c_int, [c_void_p] * 8 + [c_int] * 8 + [c_float, c_int, c_void_p]),
    "madtp_attention_qk_mask": (c_int, [c_void_p] * 6 + [c_int] + [c_void_p] * 3 + [c_int] * 8 + [c_float, c_int, c_void_p]),
    "madtp_token_score": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float]
                          + [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p]),
    "madtp_token_select": (c_int, [c_void_p, c_int] + [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "madtp_token_gather": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "madtp_token_gather_ln": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "madtp_mask_gather": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "madtp_query_att_ft": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_float, c_int,
                                   c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "madtp_align_logits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "madtp_vector_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "madtp_vit_block_workspace": (c_size_t, [c_int] * 6),
    "madtp_vit_block_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_int, c_int,
                                     c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "madtp_vit_block_mlp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p]),
    "madtp_query_model": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                  c_int, c_float, c_int, c_int, c_int, c_void_p]),
    "madtp_bert_layer_workspace": (c_size_t, [c_int] * 7),
    "madtp_bert_layer_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int,
                                      c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "madtp_bert_layer_rest": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                                      c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                      c_void_p, c_void_p]),
    "madtp_token_score_sync": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float]
                               + [c_void_p] * 4 + [c_int, c_int, c_int, c_void_p]),
    "madtp_token_score_publish": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float]
                                  + [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p, c_void_p]),
    "madtp_token_score_wait": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "madtp_attention_indexed": (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_float, c_int, c_void_p]),
    "madtp_vit_block": (c_int, [c_void_p] * 5 + [c_size_t, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float]
                        + [c_void_p] * 7 + [c_void_p]),
    "madtp_vit_block_keep": (c_int, [c_void_p] * 5 + [c_size_t, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float]
                             + [c_void_p] * 5 + [c_int] + [c_void_p] * 2 + [c_void_p]),
    "madtp_bert_layer": (c_int, [c_void_p] * 7 + [c_size_t, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float]
                         + [c_void_p] * 5 + [c_int] + [c_void_p] * 9 + [c_int] + [c_void_p] * 3),
    "madtp_vit_encoder": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_float, c_void_p]),
    "madtp_vit_encoder_async": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_float, c_void_p,
                                         c_void_p, c_void_p]),
    "madtp_bert_encoder": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                                   c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p]),
    "madtp_bert_encoder_async": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int,
                                         c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p]),
    "madtp_bert_decode_step": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "madtp_kv_cache_reorder": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "madtp_sample_top_p": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_float, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                   c_int, c_void_p]),
    "madtp_add_scale": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_size_t, c_void_p]),
    "madtp_gemm_pair": (c_int, [c_void_p] * 8 + [c_int] * 8 + [c_float, c_float, c_void_p]),
    "madtp_attention_pair": (c_int, [c_void_p] * 11 + [c_int] * 8 + [c_float, c_int, c_void_p]),
    "madtp_query_att_ft_multi": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "madtp_query_att_ft_multi_split": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p]),
    "madtp_cast_bf16": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "madtp_cast_lp": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, c_void_p]),
    "madtp_range_status": (c_int, [c_int, c_void_p]),
    "madtp_split_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "madtp_split_f16_weight": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_void_p]),
    "madtp_lm_loss": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_void_p, c_int, c_void_p]),
    "madtp_token_prob": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "madtp_beam_topk": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "madtp_beam_topk_penalty": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_void_p,
                                         c_void_p, c_int, c_void_p]),
    # backward of the pruned ViT block (csrc/backward.hip)
    "madtp_transpose_pad": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "madtp_gemm_splitk_pp": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "madtp_splitk_sum": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_void_p]),
    "madtp_beam_update": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "madtp_weight_planes": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    "madtp_transpose_split": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "madtp_colsum": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "madtp_act_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "madtp_layernorm_bwd": (c_int, [c_void_p] * 8 + [c_int, c_int, c_float, c_void_p]),
    "madtp_token_gather_bwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "madtp_token_score_bwd": (c_int, [c_void_p] * 5 + [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int] + [c_void_p] * 4
                              + [c_int, c_int, c_int, c_void_p]),
    "madtp_attention_probs": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "madtp_attention_probs_x": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                        c_float, c_void_p]),
    "madtp_att_ft_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "madtp_attention_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "madtp_attention_bwd_cross_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "madtp_attention_bwd_cross": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint64,
                                          ctypes.c_uint64, c_void_p]),
    "madtp_attention_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 6
                            + [c_int, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint64, ctypes.c_uint64, c_void_p]),
    "madtp_dropout": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_float, ctypes.c_uint64, ctypes.c_uint64, c_void_p]),
    "madtp_attention_train": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                      c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, ctypes.c_uint64, ctypes.c_uint64, c_void_p]),
}



class LinStruct(ctypes.Structure):
    _fields_ = [("w", c_void_p), ("b", c_void_p), ("n", c_int), ("k", c_int), ("w_scale", c_float)]


class VitBlockW(ctypes.Structure):
    _fields_ = [("ln1_g", c_void_p), ("ln1_b", c_void_p), ("ln2_g", c_void_p), ("ln2_b", c_void_p),
                ("eps", c_float), ("scale", c_float),
                ("qkv", LinStruct), ("proj", LinStruct), ("fc1", LinStruct), ("fc2", LinStruct),
                ("heads", c_int), ("dim", c_int), ("dtype", c_int), ("act", c_int), ("attn_mask", c_void_p), ("ld_attn_mask", c_int)]


class BertLayerW(ctypes.Structure):
    _fields_ = [("qkv", LinStruct), ("attn_out", LinStruct), ("ln_att_g", c_void_p), ("ln_att_b", c_void_p),
                ("cross", c_int), ("variant_nlvr", c_int), ("has_merge", c_int),
                ("cq", LinStruct * 2), ("ckv", LinStruct * 2), ("cdense", LinStruct * 2), ("merge", LinStruct),
                ("fused_twin", c_int), ("cq_fused", LinStruct), ("cdense_fused", LinStruct),
                ("ln_cross_g", c_void_p), ("ln_cross_b", c_void_p),
                ("inter", LinStruct), ("out", LinStruct), ("ln_out_g", c_void_p), ("ln_out_b", c_void_p),
                ("eps", c_float), ("scale", c_float), ("heads", c_int), ("dim", c_int), ("dtype", c_int),
                ("self_mask_qk", c_void_p), ("ld_self_mask_qk", c_int)]


class QueryW(ctypes.Structure):
    _fields_ = [("sd_w", c_void_p), ("sd_hi", c_void_p), ("sd_lo", c_void_p), ("split_dtype", c_int), ("sd_scale", c_float),
                ("K", c_int), ("inv_sqrt_sd", c_float), ("att_ft", c_void_p), ("stats_ws", c_void_p)]


class LayerIO(ctypes.Structure):
    _fields_ = [("logits", c_void_p), ("x_attn", c_void_p), ("y", c_void_p), ("y_lp", c_void_p), ("mask_out", c_void_p),
                ("score", c_void_p), ("threshold", c_void_p), ("count", c_void_p), ("indices", c_void_p),
                ("indices_sort", c_void_p), ("k_out", c_int), ("k_used", c_int), ("n_out", c_int)]


def lin_struct(lin):
    """runtime.Lin -> LinStruct (the Lin object must stay alive while the struct is in use)."""
    k = lin.w.shape[1] // 2 if lin.w.dtype == torch.float16 else lin.w.shape[1]
    return LinStruct(lin.w.data_ptr(), 0 if lin.b is None else lin.b.data_ptr(), lin.n, k, w_scale_of(lin.w))


_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load(path=None):
    """Loads the library (no GPU needed for loading / symbol resolution)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found - run `python -m madtp_amd.build` (hipcc, gfx950). "
                           "There is no CPU/eager fallback for the product path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError => header/library mismatch
        fn.restype = res
        fn.argtypes = args
    v = lib.madtp_abi_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"libmadtp_hip ABI {v} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    lib.madtp_set_score_fast(_score_fast)
    return lib


E_RANGE = -6


def _check(code, what):
    if code != 0:
        msg = load().madtp_strerror(code).decode()
        if code == E_RANGE:
            load().madtp_range_status(1, None)  # the flag is sticky: clear it so that the process can go on in another mode
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


def _p(t):
    return 0 if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Element format of the fast modes' 2-byte operands.  PyTorch is plumbing here (it never does arithmetic on these buffers), and
# the two fast modes share every shape, stride and buffer size - so both keep their operands in torch.bfloat16-TYPED containers
# and the element FORMAT (bf16, or IEEE f16 in the "f16" mode) travels as the dtype code of the C-ABI (MADTP_BF16 / MADTP_F16),
# chosen by the thread's precision mode (runtime.set_precision -> set_lp_format).  Do not .float() such a tensor in the "f16"
# mode: use lp_to_f32().
import threading as _threading

_lp_state = _threading.local()


def set_lp_format(code):
    _lp_state.fmt = BF16 if code != F16 else F16


def lp_format():
    return getattr(_lp_state, "fmt", BF16)


def lp_to_f32(t):
    """a 2-byte operand container (see above) -> f32 values, in the current thread's element format"""
    if t.dtype == torch.bfloat16 and lp_format() == F16:
        return t.view(torch.float16).float()
    return t.float()


def _dt(t):
    return dt_code(t.dtype)


def dt_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return lp_format()
    if dtype == torch.float16:
        return F16S
    raise TypeError(f"unsupported dtype {dtype}")


def w_scale_of(w):
    """accumulator scale of a prepared weight: 2^-s for f16-split planes (runtime.prepare_linear tags the tensor), else 1."""
    return float(getattr(w, "_madtp_w_scale", 1.0))


def _req(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the MADTP hot path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


# --------------------------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, residual=None, out_dtype=None, act=ACT_NONE, n=None, out=None, out_scale=1.0):
    """act(a[M,K] @ w[Npad,K]^T + bias) * out_scale (+ residual) -> [M, n].  w must be padded to 128 rows."""
    _req(w, name="w")
    if not a.is_cuda or a.dim() != 2 or a.stride(1) != 1:
        raise RuntimeError("gemm: a must be a GPU row-major 2-D view (no CPU fallback)")
    M, K = a.shape
    split = a.dtype == torch.float16  # f16-split planes: a is [M, 2K], w [Npad, 2K], a split output [M, 2n]
    if split:
        K //= 2
    n = n if n is not None else w.shape[0]
    if w.shape[0] % 128 or w.shape[1] != (2 * K if split else K) or a.dtype != w.dtype:
        raise RuntimeError(f"gemm: bad weight {tuple(w.shape)} {w.dtype} for a {tuple(a.shape)} {a.dtype}")
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((M, 2 * n if out_dtype == torch.float16 else n), device=a.device, dtype=out_dtype)
    ldr = residual.stride(0) if residual is not None else 0
    if residual is not None:
        _req(residual, torch.float32, "residual")
    if bias is not None:
        _req(bias, torch.float32, "bias")
    _check(load().madtp_gemm(_p(a), _p(w), _p(bias), _p(residual), _p(out), M, n, K, a.stride(0), w.stride(0),
                             out.stride(0), ldr, _dt(a), _dt(out), act, w_scale_of(w), float(out_scale), _stream()), "madtp_gemm")
    return out


class gemm_config:
    """context manager (tests / A-B benchmarks): force a madtp_gemm tile configuration, see madtp_gemm_set_config."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        self.prev = load().madtp_gemm_set_config(self.cfg)

    def __exit__(self, *a):
        load().madtp_gemm_set_config(self.prev)


def gemm_set_sq_cost(cost):
    """madtp_gemm_set_sq_cost (include/madtp_hip.h): dispatch hint for callers that keep several forwards in flight; cost <= 0
    restores the default.  -> previous value."""
    return float(load().madtp_gemm_set_sq_cost(float(cost)))


_score_fast = 0  # (madtp_amd.runtime sets it with the precision mode, its default included)


def set_score_fast(on):
    """madtp_set_score_fast (include/madtp_hip.h): fast-mode arithmetic of token_score's softmax over tokens.  Remembered
    until the library is loaded (setting a precision mode must not need the library).  -> previous."""
    global _score_fast
    prev, _score_fast = _score_fast, 1 if on else 0
    if _lib is not None:
        _lib.madtp_set_score_fast(_score_fast)
    return prev


def gemm_set_small_tile(cfg):
    """madtp_gemm_set_small_tile (include/madtp_hip.h): tile configuration of the small problems, -1 = automatic.  -> previous."""
    return int(load().madtp_gemm_set_small_tile(int(cfg)))


# ---- per-stream scheduling attributes (include/madtp_hip.h, ABI 29) ------------------------------------------------------------------
XCDS, CUS_PER_XCD = 8, 32  # MI355X


def cu_mask_words(cu0, ncus):
    """The CU mask of "CUs [cu0, cu0 + ncus) of EVERY XCD" as 32-bit words: on MI355X mask bit i enables CU i // 8 of XCD i % 8
    (profiles/r06_cumask_probe.txt); an XCD cannot be excluded (an all-zero per-XCD mask means every CU)."""
    if not (0 <= cu0 and ncus >= 1 and cu0 + ncus <= CUS_PER_XCD):
        raise ValueError(f"CU range [{cu0}, {cu0 + ncus}) outside 0..{CUS_PER_XCD}")
    words = [0] * (XCDS * CUS_PER_XCD // 32)
    for cu in range(cu0, cu0 + ncus):
        for x in range(XCDS):
            i = cu * XCDS + x
            words[i // 32] |= 1 << (i % 32)
    return words


class MaskedStream:
    """A HIP stream that owns CUs [cu0, cu0 + ncus) of every XCD (madtp_stream_create_cumask) wrapped as a torch stream
    (`.stream`, a torch.cuda.ExternalStream); the library sizes its persistent launches on it for 8 * ncus CUs.
    HIP streams behind a handle are POOLED, never destroyed while the process lives: torch's caching allocator keeps blocks (and
    pending events) tagged with a stream it has seen, and destroying that stream under it aborts the process - close() / collection
    return the stream to the pool, the next MaskedStream of the same device and CU range takes it over."""

    def __init__(self, cu0, ncus, device=None, sq_cost=0.0, small_tile=-2):
        self.cu0, self.ncus = int(cu0), int(ncus)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        key = (self.device.index, self.cu0, self.ncus)
        with _MASKED_LOCK:
            free = _MASKED_POOL.get(key)
            self.ptr = free.pop() if free else 0
        if not self.ptr:
            words = cu_mask_words(self.cu0, self.ncus)
            arr = (ctypes.c_uint32 * len(words))(*words)
            out = c_void_p(0)
            with torch.cuda.device(self.device):
                _check(load().madtp_stream_create_cumask(ctypes.byref(out), arr, len(words)), "madtp_stream_create_cumask")
            self.ptr = int(out.value)
        _check(load().madtp_stream_set_sched(self.ptr, self.ncus, float(sq_cost), int(small_tile)), "madtp_stream_set_sched")
        self.stream = torch.cuda.ExternalStream(self.ptr, device=self.device)
        _MASKED[self.ptr] = (self.cu0, self.ncus, float(sq_cost), int(small_tile))

    def close(self):
        ptr, self.ptr = getattr(self, "ptr", 0), 0
        if ptr:
            _MASKED.pop(ptr, None)
            with _MASKED_LOCK:
                _MASKED_POOL.setdefault((self.device.index, self.cu0, self.ncus), []).append(ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_MASKED_POOL = {}  # (device, cu0, ncus) -> idle HIP stream pointers
_MASKED_LOCK = _threading.Lock()
_MASKED = {}  # stream pointer -> (cu0, ncus, sq_cost, small_tile) of the live MaskedStreams


def masked_stream_info(stream_ptr):
    """(cu0, ncus, sq_cost, small_tile) if the stream was made by MaskedStream, else None"""
    return _MASKED.get(int(stream_ptr))


def stream_set_sched(stream, cus_per_xcd=0, sq_cost=0.0, small_tile=-2):
    """madtp_stream_set_sched for any torch stream: cus_per_xcd 0 = unchanged, sq_cost <= 0 / small_tile -2 = the process-wide hints."""
    ptr = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    if ptr == 0:
        raise ValueError("the null stream cannot carry scheduling attributes")
    _check(load().madtp_stream_set_sched(ptr, int(cus_per_xcd), float(sq_cost), int(small_tile)), "madtp_stream_set_sched")


def stream_get_sched(stream):
    ptr = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    a, b, c = c_int(0), c_float(0), c_int(0)
    load().madtp_stream_get_sched(ptr, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return a.value, b.value, c.value


def gemm_pair(a0, a1, w0, w1, bias0, bias1, n, out_dtype=None):
    """(a0 @ w0^T + bias0, a1 @ w1^T + bias1): two GEMMs of identical shape in one launch where the kernel allows."""
    for t, name in ((a0, "a0"), (a1, "a1"), (w0, "w0"), (w1, "w1")):
        _req(t, name=name)
    M, K = a0.shape
    if a0.dtype == torch.float16:
        K //= 2
    if a1.shape != a0.shape or w1.shape != w0.shape or a0.stride(0) != a1.stride(0) or a0.dtype != w0.dtype:
        raise RuntimeError("gemm_pair: the two problems must have identical shapes and dtypes")
    out_dtype = out_dtype or a0.dtype
    nc = 2 * n if out_dtype == torch.float16 else n
    c0 = torch.empty((M, nc), device=a0.device, dtype=out_dtype)
    c1 = torch.empty((M, nc), device=a0.device, dtype=out_dtype)
    _check(load().madtp_gemm_pair(_p(a0), _p(a1), _p(w0), _p(w1), _p(bias0), _p(bias1), _p(c0), _p(c1), M, n, K, a0.stride(0),
                                  w0.stride(0), nc, _dt(a0), _dt(c0), w_scale_of(w0), w_scale_of(w1), _stream()), "madtp_gemm_pair")
    return c0, c1


def _lp_empty(shape, device, lp):
    """buffer of the low-precision copy of an f32 [..., dim] tensor: bf16 [..., dim] or f16-split planes [..., 2*dim]."""
    if lp == torch.float16:
        return torch.empty(tuple(shape[:-1]) + (2 * shape[-1],), device=device, dtype=torch.float16)
    return torch.empty(tuple(shape), device=device, dtype=torch.bfloat16)


def layernorm(x, gamma, beta, eps, want_f32=True, want_bf16=False, lp=None):
    """lp: torch.bfloat16 / torch.float16 (f16-split planes) selects the low-precision copy (want_bf16=True == lp=bf16)."""
    _req(x, torch.float32, "x")
    _req(gamma, torch.float32, "LayerNorm weight"); _req(beta, torch.float32, "LayerNorm bias")
    dim = x.shape[-1]
    rows = x.numel() // dim
    lp = lp or (torch.bfloat16 if want_bf16 else None)
    y32 = torch.empty_like(x) if want_f32 else None
    ylp = _lp_empty(x.shape, x.device, lp) if lp is not None else None
    _check(load().madtp_layernorm(_p(x), _p(gamma), _p(beta), _p(y32), _p(ylp), dt_code(lp) if lp is not None else BF16, rows,
                                  dim, float(eps), _stream()), "madtp_layernorm")
    return y32, ylp


def patchify(img, patch, out_dtype):
    _req(img, torch.float32, "img")
    B, C, S, _ = img.shape
    g = S // patch
    kc = C * patch * patch
    cols = torch.empty((B * g * g, 2 * kc if out_dtype == torch.float16 else kc), device=img.device, dtype=out_dtype)
    _check(load().madtp_patchify(_p(img), _p(cols), B, S, patch, _dt(cols), _stream()), "madtp_patchify")
    return cols


def assemble_tokens(patches, cls, pos, B, np_):
    dim = patches.shape[-1]
    x = torch.empty((B, np_ + 1, dim), device=patches.device, dtype=torch.float32)
    _check(load().madtp_assemble_tokens(_p(patches), _p(cls), _p(pos), _p(x), B, np_, dim, _stream()),
           "madtp_assemble_tokens")
    return x


def bert_embed(ids, word_emb, pos_emb, gamma, beta, eps, want_bf16=False, lp=None):
    _req(ids, torch.int64, "input_ids")
    B, L = ids.shape
    dim = word_emb.shape[1]
    lp = lp or (torch.bfloat16 if want_bf16 else None)
    y32 = torch.empty((B, L, dim), device=ids.device, dtype=torch.float32)
    ylp = _lp_empty((B, L, dim), ids.device, lp) if lp is not None else None
    _check(load().madtp_bert_embed(_p(ids), _p(word_emb), _p(pos_emb), _p(gamma), _p(beta), _p(y32), _p(ylp),
                                   dt_code(lp) if lp is not None else BF16, B, L, dim, float(eps), _stream()), "madtp_bert_embed")
    return y32, ylp


def attention(q, k, v, B, H, Nq, Nk, scale, add_mask=None, scores=False, mask_qk=None, split=False):
    """q,k,v: 2-D row views [B*N, >=H*64] (may be column slices of one fused projection).  Returns
    (out[B*Nq, H*64], (colsum_part, p0, onorm) or None).  mask_qk: optional additive f32 [>=Nq, >=Nk] mask (causal).
    split (f32 operands only): the products as three f16 MFMA products of f16-split operands (precision mode "f16x3")."""
    for t in (q, k, v):
        if not t.is_cuda or t.stride(1) != 1:
            raise RuntimeError("attention operands must be GPU row-major views")
    out = torch.empty((B * Nq, H * 64), device=q.device, dtype=q.dtype)
    side = None
    cs = p0 = on = None
    if scores:
        nrt = (Nq + 15) // 16
        cs = torch.empty((B, nrt, Nk), device=q.device, dtype=torch.float32)
        p0 = torch.empty((B, H, Nk), device=q.device, dtype=torch.float32)
        on = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
        side = (cs, p0, on)
    if add_mask is not None:
        _req(add_mask, torch.float32, "add_mask")
    io = F16S if (split and q.dtype == torch.float32) else _dt(q)
    if mask_qk is not None:
        _req(mask_qk, torch.float32, "mask_qk")
        _check(load().madtp_attention_qk_mask(_p(q), _p(k), _p(v), _p(out), _p(add_mask), _p(mask_qk), mask_qk.stride(0), _p(cs),
                                              _p(p0), _p(on), B, H, Nq, Nk, q.stride(0), k.stride(0), v.stride(0), out.stride(0),
                                              float(scale), io, _stream()), "madtp_attention_qk_mask")
        return out, side
    _check(load().madtp_attention(_p(q), _p(k), _p(v), _p(out), _p(add_mask), _p(cs), _p(p0), _p(on), B, H, Nq, Nk,
                                  q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale), io, _stream()),
           "madtp_attention")
    return out, side


def attention_probs(q, k, B, H, N, scale, key_mask=None):
    """P = softmax(scale q k^T [+ key_mask[b, j]]) f32 [B, H, N, N] from f32 row views q, k [B*N, >= H*64] (madtp_attention_probs)."""
    for t in (q, k):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(1) != 1:
            raise RuntimeError("attention_probs operands must be GPU f32 row-major views")
    if q.stride(0) != k.stride(0):
        raise RuntimeError("attention_probs: q and k must share their leading dimension")
    P = torch.empty((B, H, N, N), device=q.device, dtype=torch.float32)
    _check(load().madtp_attention_probs(_p(q), _p(k), q.stride(0), _p(key_mask), _p(P), B, H, N, float(scale), _stream()), "madtp_attention_probs")
    return P


def attention_probs_x(q, k, B, H, Nq, Nk, scale, key_mask=None, mask_qk=None):
    """P = softmax(scale q k^T [+ key_mask[b, j]] [+ mask_qk[i, j]]) f32 [B, H, Nq, Nk] from f32 row views q [B*Nq, >= H*64],
    k [B*Nk, >= H*64] (madtp_attention_probs_x): cross-attention or cached-key probabilities for output_attentions=True."""
    for t in (q, k):
        if not t.is_cuda or t.dtype != torch.float32 or t.stride(1) != 1:
            raise RuntimeError("attention_probs_x operands must be GPU f32 row-major views")
    P = torch.empty((B, H, Nq, Nk), device=q.device, dtype=torch.float32)
    _check(load().madtp_attention_probs_x(_p(q), _p(k), q.stride(0), k.stride(0), _p(key_mask), _p(mask_qk),
                                          mask_qk.stride(0) if mask_qk is not None else 0, _p(P), B, H, Nq, Nk, float(scale), _stream()),
           "madtp_attention_probs_x")
    return P


def attention_pair(q0, q1, k0, k1, v0, v1, B, H, Nq, Nk, scale, add_mask0=None, add_mask1=None):
    """Two attention problems of identical shape (no score outputs) in one launch where the kernel allows -> (out0, out1)."""
    for t in (q0, q1, k0, k1, v0, v1):
        if not t.is_cuda or t.stride(1) != 1:
            raise RuntimeError("attention operands must be GPU row-major views")
    if q0.stride(0) != q1.stride(0) or k0.stride(0) != k1.stride(0) or v0.stride(0) != v1.stride(0):
        raise RuntimeError("attention_pair: the two problems must share their leading dimensions")
    out0 = torch.empty((B * Nq, H * 64), device=q0.device, dtype=q0.dtype)
    out1 = torch.empty_like(out0)
    _check(load().madtp_attention_pair(_p(q0), _p(q1), _p(k0), _p(k1), _p(v0), _p(v1), None, _p(out0), _p(out1), _p(add_mask0),
                                       _p(add_mask1), B, H, Nq, Nk, q0.stride(0), k0.stride(0), v0.stride(0), out0.stride(0),
                                       float(scale), _dt(q0), _stream()), "madtp_attention_pair")
    return out0, out1


def _ta_view(token_attn):
    """token_attn: any [B,n,K] f32 GPU view with unit column stride -> (ptr, row stride, batch stride, K)."""
    if not token_attn.is_cuda or token_attn.dtype != torch.float32 or token_attn.dim() != 3 or token_attn.stride(2) != 1:
        raise RuntimeError("token_attn must be a GPU f32 [B,n,K] view with unit column stride")
    return token_attn.data_ptr(), token_attn.stride(1), token_attn.stride(0), token_attn.shape[2]


def token_score(side, token_attn, temperature, B, H, N):
    cs, p0, on = side
    n = N - 1
    tp, ldr, ldb, K = _ta_view(token_attn)
    dev = token_attn.device
    score = torch.empty((B, n), device=dev, dtype=torch.float32)
    thr = torch.empty((B,), device=dev, dtype=torch.float32)
    count = torch.empty((B,), device=dev, dtype=torch.int32)
    kmax = torch.zeros((1,), device=dev, dtype=torch.int32)
    _check(load().madtp_token_score(_p(cs), cs.shape[1], _p(p0), _p(on), tp, ldr, ldb, K,
                                    float(temperature), _p(score), _p(thr), _p(count), _p(kmax), B, H, N, _stream()),
           "madtp_token_score")
    return score, thr, count, kmax


def token_select(score, k):
    B, n = score.shape
    dev = score.device
    indices = torch.empty((B, k), device=dev, dtype=torch.int64)
    indices_sort = torch.empty((B, n), device=dev, dtype=torch.int64)
    dst_pos = torch.empty((B, n), device=dev, dtype=torch.int32)
    merge_w = torch.empty((B, n), device=dev, dtype=torch.float32)
    _check(load().madtp_token_select(_p(score), k, _p(indices), _p(indices_sort), _p(dst_pos), _p(merge_w), B, n,
                                     _stream()), "madtp_token_select")
    return indices, indices_sort, dst_pos, merge_w


def token_gather(x, dst_pos, merge_w, k):
    _req(x, torch.float32, "x")
    B, N, dim = x.shape
    y = torch.empty((B, k + 2, dim), device=x.device, dtype=torch.float32)
    _check(load().madtp_token_gather(_p(x), _p(dst_pos), _p(merge_w), _p(y), B, N, k, dim, _stream()),
           "madtp_token_gather")
    return y


def token_gather_ln(x, dst_pos, merge_w, k, gamma, beta, eps, want_f32=True, want_bf16=True, lp=None):
    """token_gather with the following LayerNorm fused in -> (y, LN(y) f32 or None, LN(y) low-precision copy or None)."""
    _req(x, torch.float32, "x")
    B, N, dim = x.shape
    lp = lp or (torch.bfloat16 if want_bf16 else None)
    y = torch.empty((B, k + 2, dim), device=x.device, dtype=torch.float32)
    h32 = torch.empty_like(y) if want_f32 else None
    hlp = _lp_empty(y.shape, x.device, lp) if lp is not None else None
    _check(load().madtp_token_gather_ln(_p(x), _p(dst_pos), _p(merge_w), _p(y), B, N, k, dim, _p(gamma), _p(beta), float(eps),
                                        _p(h32), _p(hlp), dt_code(lp) if lp is not None else BF16, _stream()),
           "madtp_token_gather_ln")
    return y, h32, hlp


def mask_gather(mask2d, order, k, order2=None):
    """mask2d f32 [B,N] additive; order int64 [B,>=k+1] (NLVR: indices_sort) or (MED) order=indices, order2=indices_sort."""
    _req(mask2d, torch.float32, "mask")
    B, N = mask2d.shape
    out = torch.empty((B, k + 2), device=mask2d.device, dtype=torch.float32)
    _check(load().madtp_mask_gather(_p(mask2d), _p(order), order.stride(0), _p(order2),
                                    order2.stride(0) if order2 is not None else 0, _p(out), B, N, k, _stream()),
           "madtp_mask_gather")
    return out


def query_att_ft(token_attn, ft, out=None, sd_dim=768, fast=False):
    """ft: [B,n,dim] f32 GPU view with unit column stride (e.g. x[:,1:,:]); token_attn [B,n,K] view.
    fast: bf16-MFMA variant (fast mode)."""
    fp, ldf, ldfb, dim = _ta_view(ft)
    B, n = ft.shape[0], ft.shape[1]
    tp, ldr, ldb, K = _ta_view(token_attn)
    acc = 1
    if out is None:
        out = torch.empty((B, K, dim), device=ft.device, dtype=torch.float32)
        acc = 0
    ws = torch.empty(B * 256, device=ft.device, dtype=torch.float32) if fast else None
    _check(load().madtp_query_att_ft(tp, ldr, ldb, K, fp, ldf, ldfb, _p(out), 1.0 / (sd_dim ** 0.5), acc,
                                     B, n, dim, 1 if fast else 0, _p(ws), _stream()), "madtp_query_att_ft")
    return out


class AttFtSeg(ctypes.Structure):
    _fields_ = [("token_attn", c_void_p), ("ft", c_void_p), ("n", c_int), ("ldt_row", c_int), ("ldt_batch", c_int),
                ("ldf_row", c_int), ("ldf_batch", c_int)]


def query_att_ft_multi(pairs, out=None, sd_dim=768, exact=False):
    """pairs: list of (token_attn [B,n,K] view, ft [B,n,dim] f32 view) of the layers of an encoder -> their summed att_ft
    [B,K,dim] in one launch (fast mode: bf16 MFMA; exact=True: the exact-f32 kernel, bit-identical to summing layer by layer;
    exact="split": f16-split operands on the f16 MFMA - the f16x3 mode)."""
    segs = (AttFtSeg * len(pairs))()
    for i, (ta, ft) in enumerate(pairs):
        fp, ldf, ldfb, dim = _ta_view(ft)
        tp, ldr, ldb, K = _ta_view(ta)
        segs[i] = AttFtSeg(tp, fp, ft.shape[1], ldr, ldb, ldf, ldfb)
    B = pairs[0][1].shape[0]
    acc = 1
    if out is None:
        out = torch.empty((B, K, dim), device=pairs[0][1].device, dtype=torch.float32)
        acc = 0
    ws = None if exact is True else torch.empty(len(pairs) * B * 256, device=out.device, dtype=torch.float32)
    fn = load().madtp_query_att_ft_multi_split if exact == "split" else load().madtp_query_att_ft_multi
    _check(fn(segs, len(pairs), K, _p(out), _p(ws), 1.0 / (sd_dim ** 0.5), acc, B, dim, _stream()), "madtp_query_att_ft_multi")
    return out


def query_att_ft_multi_ptrs(segs_ptrs, B, K, dim, device, sd_dim=768, exact=False):
    """query_att_ft_multi from raw (token_attn ptr, ft ptr, n, ldt_row, ldt_batch, ldf_row, ldf_batch) tuples (the layers of an
    encoder-level call: no tensor views are built)."""
    segs = (AttFtSeg * len(segs_ptrs))()
    for i, t in enumerate(segs_ptrs):
        segs[i] = AttFtSeg(*t)
    out = torch.empty((B, K, dim), device=device, dtype=torch.float32)
    ws = None if exact is True else torch.empty(len(segs_ptrs) * B * 256, device=device, dtype=torch.float32)
    fn = load().madtp_query_att_ft_multi_split if exact == "split" else load().madtp_query_att_ft_multi
    _check(fn(segs, len(segs_ptrs), K, _p(out), _p(ws), 1.0 / (sd_dim ** 0.5), 0, B, dim, _stream()), "madtp_query_att_ft_multi")
    return out


def add_scale(a, b, scale):
    out = torch.empty_like(a)
    _check(load().madtp_add_scale(_p(a), _p(b), _p(out), float(scale), a.numel(), _stream()), "madtp_add_scale")
    return out


def cast_bf16(src, scale=1.0):
    """f32 -> the fast modes' 2-byte operand (bf16, or f16 in the "f16" mode: lp_format()), of src * scale"""
    _req(src, torch.float32, "src")
    dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    _check(load().madtp_cast_lp(_p(src), _p(dst), src.numel(), lp_format(), float(scale), _stream()), "madtp_cast_lp")
    return dst


def cast_bf16_plain(src):
    """f32 -> bf16, whatever the thread's 2-byte element format: the alignment dictionary's hi / lo planes (madtp_align_logits
    splits x.sd^T into bf16 products in BOTH fast modes)."""
    _req(src, torch.float32, "src")
    dst = torch.empty(src.shape, device=src.device, dtype=torch.bfloat16)
    _check(load().madtp_cast_bf16(_p(src), _p(dst), src.numel(), _stream()), "madtp_cast_bf16")
    return dst


def split_code(t):
    """split_dtype of alignment-dictionary planes: float16 = the f16 planes of the f16x3 mode, bfloat16 = bf16 hi / lo planes"""
    return F16S if t.dtype == torch.float16 else BF16


def cast_lp_weight(w):
    """Prepared weight of a fast mode: bf16 cast, or - "f16" mode - f16 of w * 2^s with max|w| * 2^s in (2^13, 2^14] (small
    weights stay clear of the f16 subnormals; the tensor is tagged with the accumulator scale 2^-s like the split planes)."""
    if lp_format() != F16:
        return cast_bf16(w)
    amax = float(w.abs().max())
    s = 0
    if amax > 0 and amax == amax and amax != float("inf"):
        import math
        s = max(-100, min(100, 14 - math.ceil(math.log2(amax))))
    dst = cast_bf16(w, scale=2.0 ** s)
    dst._madtp_w_scale = float(2.0 ** -s)
    return dst


def range_status(reset=True):
    """madtp_range_status: 1 when a producer kernel met a value outside the f16 range since the last reset (f16 modes)."""
    return int(load().madtp_range_status(1 if reset else 0, _stream()))


def split_f16(src):
    """f32 [..., K] (rows contiguous) -> f16-split activation planes [..., 2K] (torch.float16) for an F16S GEMM."""
    _req(src, torch.float32, "src")
    K = src.shape[-1]
    rows = src.numel() // K
    dst = torch.empty(tuple(src.shape[:-1]) + (2 * K,), device=src.device, dtype=torch.float16)
    _check(load().madtp_split_f16(_p(src), K, _p(dst), 2 * K, rows, K, _stream()), "madtp_split_f16")
    return dst


def split_f16_weight(w, log2_scale=None):
    """f32 [n, K] -> [n, 2K] f16 planes [Q0 | Q1] of w * 2^s (max|w| * 2^s in (2^13, 2^14]); the tensor is tagged with 2^-s.
    log2_scale: a caller-chosen s instead of the one derived from max|w| (which costs a host read): the backward's wgrad passes 0
    for O(1) activations - their planes are then accurate to ~3e-8 ABSOLUTE, far inside its tolerance."""
    _req(w, torch.float32, "w")
    n, K = w.shape
    s = 0
    amax = float(w.abs().max()) if log2_scale is None else 0.0
    if log2_scale is not None:
        s = int(log2_scale)
    elif amax > 0 and amax == amax and amax != float("inf"):
        import math
        s = 14 - math.ceil(math.log2(amax))
        s = max(-100, min(100, s))
    dst = torch.empty((n, 2 * K), device=w.device, dtype=torch.float16)
    _check(load().madtp_split_f16_weight(_p(w), K, _p(dst), n, K, float(2.0 ** s), _stream()), "madtp_split_f16_weight")
    dst._madtp_w_scale = float(2.0 ** -s)
    dst._madtp_log2_scale = s
    return dst


def vector_gather(vectors, indices):
    _req(vectors, torch.float32, "vectors"); _req(indices, torch.int64, "indices")
    B, L, D = vectors.shape
    K = indices.shape[1]
    out = torch.empty((B, K, D), device=vectors.device, dtype=torch.float32)
    _check(load().madtp_vector_gather(_p(vectors), _p(indices), _p(out), B, L, K, D, _stream()), "madtp_vector_gather")
    return out


# ---- layer-level calls --------------------------------------------------------------------------------------------
_WS = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, current stream): the kernels of one stream run in order, so its layers can share
    it; forwards in flight on different streams (madtp_amd.pipeline) each get their own."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def prune_outputs(B, n, device):
    """(score, threshold, count, kmax): kmax is None - callers read `count` back and take the max on the host, which
    saves the memset + atomic of the device-side batch max (the read-back is the layer's one host sync either way)."""
    return (torch.empty((B, n), device=device, dtype=torch.float32), torch.empty((B,), device=device, dtype=torch.float32),
            torch.empty((B,), device=device, dtype=torch.int32), None)


def token_score_sync(side, token_attn, temperature, B, H, N):
    """token_score whose k = max_b count arrives on the HOST through pinned memory (madtp_token_score_sync).
    -> (score, threshold, count, k:int)"""
    cs, p0, on = side
    tp, ldr, ldb, K = _ta_view(token_attn)
    score, thr, count, _ = prune_outputs(B, N - 1, token_attn.device)
    k = ctypes.c_int32(-1)
    _check(load().madtp_token_score_sync(_p(cs), cs.shape[1], _p(p0), _p(on), tp, ldr, ldb, K, float(temperature), _p(score),
                                         _p(thr), _p(count), ctypes.byref(k), B, H, N, _stream()), "madtp_token_score_sync")
    return score, thr, count, k.value


def batch_max_count(count):
    """k = max_b count (vit.py:145 `.item()`): one D2H copy of B int32 values, max on the host."""
    return int(count.cpu().max())


def _carve(buf, *shape):
    """leading contiguous [*shape] view of a (larger) flat buffer."""
    n = 1
    for d in shape:
        n *= d
    return buf.view(-1)[:n].view(*shape)


def vit_block(wstruct, x, token_attn, temperature, max_keep=0):
    """Block.forward in ONE library call (attention half, host read of k, pruning rule, MLP half).
    -> (y [B,N',D], info or None); info = dict(k, score, threshold, count, pruned, indices, indices_sort).
    max_keep: CLIP's rule (clip/model.py:220-221: no pruning when k <= max_keep); 0 is the BLIP rule k < 1."""
    B, N, D = x.shape
    lib = load()
    nbytes = lib.madtp_vit_block_workspace(B, N, wstruct.dim, wstruct.fc1.n, wstruct.heads, wstruct.dtype)
    ws = workspace(nbytes, x.device)
    x_attn = torch.empty_like(x)
    ybuf = torch.empty_like(x)
    k_out, k_used = ctypes.c_int(0), ctypes.c_int(0)
    if temperature > 0:
        tp, ldr, ldb, K = _ta_view(token_attn)
        score, thr, count, _ = prune_outputs(B, N - 1, x.device)
        idx = torch.empty((B, N - 1), device=x.device, dtype=torch.int64)
        idx_sort = torch.empty((B, N - 1), device=x.device, dtype=torch.int64)
        _check(lib.madtp_vit_block_keep(ctypes.byref(wstruct), _p(x), _p(x_attn), _p(ybuf), _p(ws), ws.numel(), B, N, tp, ldr, ldb,
                                        K, float(temperature), _p(score), _p(thr), _p(count), _p(idx), _p(idx_sort), int(max_keep),
                                        ctypes.byref(k_out), ctypes.byref(k_used), _stream()), "madtp_vit_block_keep")
        info = {"k": k_out.value, "score": score, "threshold": thr, "count": count, "pruned": False, "indices": None,
                "indices_sort": None}
        if k_used.value > 0:
            k = k_used.value
            info.update(pruned=True, indices=_carve(idx, B, k), indices_sort=idx_sort)
            return _carve(ybuf, B, k + 2, D), info
        return ybuf, info
    _check(lib.madtp_vit_block(ctypes.byref(wstruct), _p(x), _p(x_attn), _p(ybuf), _p(ws), ws.numel(), B, N, 0, 0, 0, 0, 0.0,
                               0, 0, 0, 0, 0, ctypes.byref(k_out), ctypes.byref(k_used), _stream()), "madtp_vit_block")
    return ybuf, None


def bert_layer(wstruct, hidden, mask2d, token_attn, temperature, cross_mode, enc0, enc1, Nk, enc_mask0, enc_mask1,
               hidden_lp=None, kv_pre=(None, None), kv_index=None):
    """BertLayer.forward in ONE library call.  -> (y [B,L',D], mask_out [B,L'] or None, info or None, y_lp).
    hidden_lp / y_lp: bf16 copies of the layer input / output (fast mode; the LayerNorms emit them, saving the casts)."""
    B, L, D = hidden.shape
    lib = load()
    kv_ld = 0
    for t in kv_pre:  # cached [k|v] rows may be column slices of a wider projection (all layers side by side)
        if t is not None:
            if t.stride(-1) != 1 or (kv_ld and kv_ld != t.stride(0)):
                raise RuntimeError("bert_layer: kv_pre tensors must be row-major with one common row stride")
            kv_ld = t.stride(0)
    nbytes = lib.madtp_bert_layer_workspace(B, L, Nk, wstruct.dim, wstruct.inter.n, wstruct.heads, wstruct.dtype)
    ws = workspace(nbytes, hidden.device)
    att = torch.empty_like(hidden)
    ybuf = torch.empty_like(hidden)
    k_out, k_used = ctypes.c_int(0), ctypes.c_int(0)
    dev = hidden.device
    ylp = None
    if wstruct.dtype != F32:  # compute-dtype copy of the layer output for the next layer (bf16, or f16-split planes)
        ylp = _lp_empty(hidden.shape, dev, torch.float16 if wstruct.dtype == F16S else torch.bfloat16)
    if temperature > 0:
        tp, ldr, ldb, K = _ta_view(token_attn)
        score, thr, count, _ = prune_outputs(B, L - 1, dev)
        idx = torch.empty((B, L - 1), device=dev, dtype=torch.int64)
        idx_sort = torch.empty((B, L - 1), device=dev, dtype=torch.int64)
        mbuf = torch.empty((B, L), device=dev, dtype=torch.float32) if mask2d is not None else None
        _check(lib.madtp_bert_layer(ctypes.byref(wstruct), _p(hidden), _p(mask2d), _p(att), _p(ybuf), _p(mbuf), _p(ws), ws.numel(),
                                    B, L, Nk, tp, ldr, ldb, K, float(temperature), _p(score), _p(thr), _p(count), _p(idx),
                                    _p(idx_sort), int(cross_mode), _p(enc0), _p(enc1), _p(enc_mask0), _p(enc_mask1),
                                    _p(hidden_lp), _p(ylp), _p(kv_pre[0]), _p(kv_pre[1]), _p(kv_index), kv_ld, ctypes.byref(k_out),
                                    ctypes.byref(k_used), _stream()), "madtp_bert_layer")
        info = {"k": k_out.value, "score": score, "threshold": thr, "count": count, "pruned": False, "indices": None,
                "indices_sort": None}
        if k_used.value > 0:
            k = k_used.value
            info.update(pruned=True, indices=_carve(idx, B, k), indices_sort=idx_sort)
            return (_carve(ybuf, B, k + 2, D), (_carve(mbuf, B, k + 2) if mbuf is not None else None), info,
                    _carve(ylp, B, k + 2, ylp.shape[-1]) if ylp is not None else None)
        return ybuf, None, info, ylp
    _check(lib.madtp_bert_layer(ctypes.byref(wstruct), _p(hidden), _p(mask2d), _p(att), _p(ybuf), 0, _p(ws), ws.numel(), B, L, Nk,
                                0, 0, 0, 0, 0.0, 0, 0, 0, 0, 0, int(cross_mode), _p(enc0), _p(enc1), _p(enc_mask0), _p(enc_mask1),
                                _p(hidden_lp), _p(ylp), _p(kv_pre[0]), _p(kv_pre[1]), _p(kv_index), kv_ld, ctypes.byref(k_out),
                                ctypes.byref(k_used), _stream()), "madtp_bert_layer")
    return ybuf, None, None, ylp


def vit_block_attn(wstruct, x, token_attn, temperature):
    """x f32 [B,N,D] contiguous -> (x_attn, (score, thr, count, kmax) or None)."""
    B, N, D = x.shape
    lib = load()
    nbytes = lib.madtp_vit_block_workspace(B, N, wstruct.dim, wstruct.fc1.n, wstruct.heads, wstruct.dtype)
    ws = workspace(nbytes, x.device)
    out = torch.empty_like(x)
    if temperature > 0:
        tp, ldr, ldb, K = _ta_view(token_attn)
        po = prune_outputs(B, N - 1, x.device)
        _check(lib.madtp_vit_block_attn(ctypes.byref(wstruct), _p(x), _p(out), _p(ws), ws.numel(), B, N, tp, ldr, ldb, K,
                                        float(temperature), _p(po[0]), _p(po[1]), _p(po[2]), _p(po[3]), _stream()),
               "madtp_vit_block_attn")
        return out, po
    _check(lib.madtp_vit_block_attn(ctypes.byref(wstruct), _p(x), _p(out), _p(ws), ws.numel(), B, N, 0, 0, 0, 0, 0.0, 0, 0,
                                    0, 0, _stream()), "madtp_vit_block_attn")
    return out, None


def vit_block_mlp(wstruct, x, k, score):
    B, N, D = x.shape
    lib = load()
    nbytes = lib.madtp_vit_block_workspace(B, N, wstruct.dim, wstruct.fc1.n, wstruct.heads, wstruct.dtype)
    ws = workspace(nbytes, x.device)
    if k > 0:
        y = torch.empty((B, k + 2, D), device=x.device, dtype=torch.float32)
        indices = torch.empty((B, k), device=x.device, dtype=torch.int64)
        indices_sort = torch.empty((B, N - 1), device=x.device, dtype=torch.int64)
    else:
        y = torch.empty_like(x)
        indices = indices_sort = None
    _check(lib.madtp_vit_block_mlp(ctypes.byref(wstruct), _p(x), _p(y), _p(ws), ws.numel(), B, N, k, _p(score), _p(indices),
                                   _p(indices_sort), _stream()), "madtp_vit_block_mlp")
    return y, indices, indices_sort


def align_logits(x2d, sd_hi, sd_lo, sd_scale=1.0):
    """sd_hi / sd_lo: bf16 hi/lo planes (fast mode) or the f16 planes Q0 / Q1 of sd * 2^s with sd_scale = 2^-s (f16x3 mode)."""
    M, D = x2d.shape
    out = torch.empty((M, 128), device=x2d.device, dtype=torch.float32)
    _check(load().madtp_align_logits(_p(x2d), _p(sd_hi), _p(sd_lo), _p(out), M, D, split_code(sd_hi), float(sd_scale), _stream()),
           "madtp_align_logits")
    return out


def query_model(x, sd_w, K, att_ft=None, want_att_ft=True, sd_dim=768, sd_split=None):
    """x f32 [B,N,D] contiguous -> (token_attn view [B,N-1,K], att_ft).  sd_split=(hi, lo[, scale]): bf16 planes select the
    fast mode (bf16x3 logits + bf16 att_ft), f16 planes of sd * 2^s (scale = 2^-s) the fp32-accurate f16x3 logits."""
    B, N, D = x.shape
    kp = sd_w.shape[0]
    full = torch.empty((B * N, kp), device=x.device, dtype=torch.float32)
    acc = 1 if att_ft is not None else 0
    if want_att_ft and att_ft is None:
        att_ft = torch.empty((B, K, D), device=x.device, dtype=torch.float32)
    hi, lo = (sd_split[0], sd_split[1]) if sd_split is not None else (None, None)
    sd_scale = sd_split[2] if sd_split is not None and len(sd_split) > 2 else 1.0
    fast = hi is not None and hi.dtype == torch.bfloat16
    ws = torch.empty(B * 256, device=x.device, dtype=torch.float32) if (want_att_ft and fast) else None
    _check(load().madtp_query_model(_p(x), _p(sd_w), _p(hi), _p(lo), split_code(hi) if hi is not None else BF16, float(sd_scale), K,
                                    _p(full), _p(att_ft) if want_att_ft else 0, _p(ws), acc, 1.0 / (sd_dim ** 0.5), B, N, D,
                                    _stream()), "madtp_query_model")
    return full.view(B, N, kp)[:, 1:, :K], att_ft


def bert_layer_attn(wstruct, hidden, mask2d, token_attn, temperature, Nk):
    """First half of a BERT layer (madtp_bert_layer_attn: self-attention + output LayerNorm + importance score) for callers with
    their own pruning rule -> (attention_output, (score, threshold, count, kmax) or None)."""
    B, L, D = hidden.shape
    lib = load()
    nbytes = lib.madtp_bert_layer_workspace(B, L, Nk, wstruct.dim, wstruct.inter.n, wstruct.heads, wstruct.dtype)
    ws = workspace(nbytes, hidden.device)
    att = torch.empty_like(hidden)
    if temperature > 0:
        tp, ldr, ldb, K = _ta_view(token_attn)
        po = prune_outputs(B, L - 1, hidden.device)
        _check(lib.madtp_bert_layer_attn(ctypes.byref(wstruct), _p(hidden), _p(mask2d), _p(att), _p(ws), ws.numel(), B, L, Nk,
                                         tp, ldr, ldb, K, float(temperature), _p(po[0]), _p(po[1]), _p(po[2]), _p(po[3]),
                                         _stream()), "madtp_bert_layer_attn")
        return att, po
    _check(lib.madtp_bert_layer_attn(ctypes.byref(wstruct), _p(hidden), _p(mask2d), _p(att), _p(ws), ws.numel(), B, L, Nk, 0,
                                     0, 0, 0, 0.0, 0, 0, 0, 0, _stream()), "madtp_bert_layer_attn")
    return att, None


def bert_layer_rest(wstruct, att, mask2d, k, score, cross_mode, enc0, enc1, Nk, enc_mask0, enc_mask1):
    B, L, D = att.shape
    lib = load()
    nbytes = lib.madtp_bert_layer_workspace(B, L, Nk, wstruct.dim, wstruct.inter.n, wstruct.heads, wstruct.dtype)
    ws = workspace(nbytes, att.device)
    Lp = k + 2 if k > 0 else L
    y = torch.empty((B, Lp, D), device=att.device, dtype=torch.float32)
    indices = indices_sort = mask_out = None
    if k > 0:
        indices = torch.empty((B, k), device=att.device, dtype=torch.int64)
        indices_sort = torch.empty((B, L - 1), device=att.device, dtype=torch.int64)
        if mask2d is not None:
            mask_out = torch.empty((B, Lp), device=att.device, dtype=torch.float32)
    _check(lib.madtp_bert_layer_rest(ctypes.byref(wstruct), _p(att), _p(mask2d), _p(y), _p(mask_out), _p(ws), ws.numel(), B, L,
                                     k, _p(score), _p(indices), _p(indices_sort), int(cross_mode), _p(enc0), _p(enc1), Nk,
                                     _p(enc_mask0), _p(enc_mask1), _stream()), "madtp_bert_layer_rest")
    return y, mask_out, indices, indices_sort


def lm_loss(logits, labels, n_vocab, label_smoothing=0.1):
    """med.py:1036-1042: logits f32 [B, L, >= n_vocab] (prediction scores of every position), labels int64 [B, L] (-100 =
    ignored) -> loss f32 [B] = per-sequence sum of the label-smoothed next-token cross-entropy."""
    _req(logits, torch.float32, "logits")
    _req(labels, torch.int64, "labels")
    B, L, ld = logits.shape
    if not logits.is_contiguous() or not labels.is_contiguous() or tuple(labels.shape) != (B, L):
        raise RuntimeError("lm_loss: logits [B,L,ld] and labels [B,L] must be contiguous")
    out = torch.empty((B,), device=logits.device, dtype=torch.float32)
    _check(load().madtp_lm_loss(_p(logits), ld, L, L - 1, int(n_vocab), _p(labels), L, float(label_smoothing), _p(out), B,
                                _stream()), "madtp_lm_loss")
    return out


def token_prob(logits, tok, n_vocab):
    """blip_vqa.py:170-171: softmax(logits[:, :n_vocab], dim=1).index_select(1, tok); logits f32 [Q, >= n_vocab] (rows may be
    strided), tok int64 [A] -> f32 [Q, A]."""
    _req(tok, torch.int64, "tok")
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("token_prob: logits must be a GPU f32 [Q, V] tensor with unit column stride (rows may be strided)")
    Q, A = logits.shape[0], tok.shape[0]
    out = torch.empty((Q, A), device=logits.device, dtype=torch.float32)
    _check(load().madtp_token_prob(_p(logits), logits.stride(0), int(n_vocab), _p(tok), A, _p(out), Q, _stream()),
           "madtp_token_prob")
    return out


def beam_topk(logits, beam_scores, num_beams, n_vocab, n_top=None, suppress_token=-1, prev_ids=None, repetition_penalty=1.0):
    """One beam-search step's candidate selection (include/madtp_hip.h madtp_beam_topk): logits f32 [B * num_beams, >= n_vocab]
    (rows may be strided), beam_scores f32 [B * num_beams] -> (scores f32 [B, n_top], flat index int32 [B, n_top] = beam * V + token),
    descending, n_top = 2 * num_beams by default."""
    _req(beam_scores, torch.float32, "beam_scores")
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("beam_topk: logits must be a GPU f32 [rows, V] tensor with unit column stride (rows may be strided)")
    rows = logits.shape[0]
    if rows % num_beams or beam_scores.numel() != rows:
        raise ValueError("beam_topk: logits rows / beam_scores must be batch * num_beams long")
    B = rows // num_beams
    n_top = 2 * num_beams if n_top is None else int(n_top)
    sc = torch.empty((B, n_top), device=logits.device, dtype=torch.float32)
    ix = torch.empty((B, n_top), device=logits.device, dtype=torch.int32)
    if prev_ids is not None and repetition_penalty != 1.0:  # RepetitionPenaltyLogitsProcessor on the beams' sequences so far
        if not prev_ids.is_cuda or prev_ids.dtype != torch.int64 or prev_ids.dim() != 2 or prev_ids.stride(1) != 1:
            raise RuntimeError("beam_topk: prev_ids must be a GPU int64 [rows, cur_len] tensor with unit column stride")
        _check(load().madtp_beam_topk_penalty(_p(logits), logits.stride(0), int(n_vocab), _p(beam_scores), int(num_beams), n_top,
                                              int(suppress_token), _p(prev_ids), prev_ids.stride(0), prev_ids.shape[1],
                                              float(repetition_penalty), _p(sc), _p(ix), B, _stream()), "madtp_beam_topk_penalty")
        return sc, ix
    _check(load().madtp_beam_topk(_p(logits), logits.stride(0), int(n_vocab), _p(beam_scores), int(num_beams), n_top,
                                  int(suppress_token), _p(sc), _p(ix), B, _stream()), "madtp_beam_topk")
    return sc, ix


class BeamState:
    """Device-side hypothesis state of a beam search (madtp_beam_update): B items, num_beams beams, sequences of up to max_length."""

    def __init__(self, B, num_beams, max_length, pad_token_id, device):
        S, n = num_beams + 1, B * num_beams
        self.B, self.num_beams, self.max_length = B, num_beams, max_length
        self.ids = [torch.full((n, max_length), pad_token_id, dtype=torch.int64, device=device) for _ in range(2)]
        self.beam_scores = torch.zeros((B, num_beams), dtype=torch.float32, device=device)
        self.beam_scores[:, 1:] = -1e9
        self.beam_scores = self.beam_scores.view(-1)
        self.beam_src = torch.zeros((n,), dtype=torch.int64, device=device)
        self.hyp_n = torch.zeros((B,), dtype=torch.int32, device=device)
        self.hyp_order = torch.zeros((B, S), dtype=torch.int32, device=device)
        self.hyp_score = torch.zeros((B, S), dtype=torch.float64, device=device)
        self.hyp_len = torch.zeros((B, S), dtype=torch.int32, device=device)
        self.hyp_tok = torch.zeros((B, S, max_length), dtype=torch.int64, device=device)
        self.worst = torch.full((B,), 1e9, dtype=torch.float64, device=device)
        self.done = torch.zeros((B,), dtype=torch.int32, device=device)
        self.err = torch.zeros((1,), dtype=torch.int32, device=device)


def beam_update(st, cur, sc, ix, n_vocab, cur_len, denom, eos_token_id, pad_token_id, early_stopping):
    """One step of hypothesis book-keeping on the device: st.ids[cur] (+ candidates sc / ix of madtp_beam_topk) -> st.ids[1 - cur],
    st.beam_scores, st.beam_src and the hypothesis state."""
    _check(load().madtp_beam_update(_p(sc), _p(ix), sc.shape[1], int(n_vocab), _p(st.ids[cur]), _p(st.ids[1 - cur]), st.max_length,
                                    int(cur_len), _p(st.beam_scores), _p(st.beam_src), _p(st.hyp_n), _p(st.hyp_order), _p(st.hyp_score),
                                    _p(st.hyp_len), _p(st.hyp_tok), _p(st.worst), _p(st.done), _p(st.err), float(denom), st.num_beams,
                                    int(eos_token_id), int(pad_token_id), 1 if early_stopping else 0, st.B, _stream()),
           "madtp_beam_update")


def sample_top_p(logits, u, n_vocab, top_p, top_k=50, suppress_token=-1, prev_ids=None, repetition_penalty=1.0, want_prob=False):
    """One nucleus-sampling step (include/madtp_hip.h madtp_sample_top_p): logits f32 [rows, >= n_vocab], u f32 [rows] uniform numbers
    in [0, 1) -> next token int64 [rows] (and its probability among the survivors)."""
    _req(u, torch.float32, "u")
    if not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("sample_top_p: logits must be a GPU f32 [rows, V] tensor with unit column stride")
    rows = logits.shape[0]
    tok = torch.empty((rows,), device=logits.device, dtype=torch.int64)
    prob = torch.empty((rows,), device=logits.device, dtype=torch.float32) if want_prob else None
    if prev_ids is not None:
        _req(prev_ids, torch.int64, "prev_ids")
    _check(load().madtp_sample_top_p(_p(logits), logits.stride(0), int(n_vocab), _p(prev_ids), prev_ids.stride(0) if prev_ids is not None else 0,
                                     prev_ids.shape[1] if prev_ids is not None else 0, float(repetition_penalty), int(suppress_token),
                                     int(top_k), float(top_p), _p(u), _p(tok), _p(prob), rows, _stream()), "madtp_sample_top_p")
    return (tok, prob) if want_prob else tok


def profile_begin():
    load().madtp_profile_begin()


def profile_end():
    """-> list of dicts(dtype, M, N, K, launches, ms, flops) for every GEMM shape launched since profile_begin()."""
    buf = ctypes.create_string_buffer(1 << 20)
    n = load().madtp_profile_end(buf, len(buf))
    rows = []
    for line in buf.raw[:n].decode().splitlines():
        dt, M, N, K, c, ms, fl, by = line.split()
        rows.append({"dtype": {F32: "f32", BF16: "bf16", F16S: "f16s", F16: "f16"}[int(dt)], "M": int(M), "N": int(N), "K": int(K),
                     "launches": int(c), "ms": float(ms), "flops": float(fl), "bytes": float(by)})
    return rows


def gemm_splitk_ln(a, w, bias, residual, gamma, beta, eps, splits, n, scale=1.0, want_bf16=False, lp=None):
    """LayerNorm(scale*(a @ w^T + bias) + residual) via split-K partials; returns (y32, low-precision copy or None)."""
    M, K = a.shape
    if a.dtype == torch.float16:
        K //= 2
    lp = lp or (torch.bfloat16 if want_bf16 else None)
    part = torch.empty((splits, M, n), device=a.device, dtype=torch.float32)
    _check(load().madtp_gemm_splitk(_p(a), _p(w), _p(part), M, n, K, a.stride(0), w.stride(0), splits, _dt(a), _stream()),
           "madtp_gemm_splitk")
    y32 = torch.empty((M, n), device=a.device, dtype=torch.float32)
    ylp = _lp_empty((M, n), a.device, lp) if lp is not None else None
    _check(load().madtp_splitk_ln(_p(part), splits, _p(bias), _p(residual), _p(gamma), _p(beta), _p(y32), _p(ylp),
                                  dt_code(lp) if lp is not None else BF16, M, n, float(eps), w_scale_of(w), float(scale), _stream()),
           "madtp_splitk_ln")
    return y32, ylp


# ---- encoder-level calls (madtp_vit_encoder / madtp_bert_encoder) ---------------------------------------------------------
class LazyPruneInfo(dict):
    """last_prune record of a layer run by an encoder-level call: the tensors (views of the call's one buffer) are only
    created when the record is read - building ~70 views per forward eagerly would cost more host time than the call saved."""

    def __init__(self, fill):
        super().__init__()
        self._fill_fn = fill

    def _fill(self):
        if self._fill_fn is not None:
            fn, self._fill_fn = self._fill_fn, None
            dict.update(self, fn())

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, d=None):
        self._fill()
        return dict.get(self, k, d)

    def items(self):
        self._fill()
        return dict.items(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __len__(self):
        self._fill()
        return dict.__len__(self)


_ESZ = {torch.float32: 4, torch.int32: 4, torch.int64: 8, torch.bfloat16: 2, torch.float16: 2}


_IO_NP = None
_IO_LAYOUT = {}


def _io_dtype():
    """numpy mirror of LayerIO (same field order / offsets) so that a call's pointers are filled with one vector add."""
    global _IO_NP
    if _IO_NP is None:
        import numpy as np
        names = [f[0] for f in LayerIO._fields_]
        fmts = ["u8"] * 10 + ["i4"] * 3
        offs = [getattr(LayerIO, n).offset for n in names]
        _IO_NP = np.dtype({"names": names, "formats": fmts, "offsets": offs, "itemsize": ctypes.sizeof(LayerIO)})
    return _IO_NP


class EncoderRun:
    """Buffers and results of one encoder-level call: ONE device allocation carved per layer (sized for the unpruned sequence).
    The per-layer offsets of a shape are computed once and cached; a call adds the buffer's base address to them."""
    _FIELDS = ("logits", "x_attn", "y", "y_lp", "mask_out", "score", "threshold", "count", "indices", "indices_sort")

    def __init__(self, n_layers, B, N0, D, device, want_logits, want_prune, want_mask, lp_dtype):
        import numpy as np
        self.n_layers, self.B, self.N0, self.D, self.lp_dtype = n_layers, B, N0, D, lp_dtype
        key = (n_layers, B, N0, D, want_logits, want_prune, want_mask, lp_dtype)
        lay = _IO_LAYOUT.get(key)
        if lay is None:
            lp_w = 0 if lp_dtype is None else (2 * D * 2 if lp_dtype == torch.float16 else D * 2)  # bytes per row of the lp copy
            sizes = {"logits": B * N0 * 128 * 4 if want_logits else 0, "x_attn": B * N0 * D * 4, "y": B * N0 * D * 4,
                     "y_lp": B * N0 * lp_w, "mask_out": B * N0 * 4 if want_mask else 0,
                     "score": B * (N0 - 1) * 4 if want_prune else 0, "threshold": B * 4 if want_prune else 0,
                     "count": B * 4 if want_prune else 0, "indices": B * (N0 - 1) * 8 if want_prune else 0,
                     "indices_sort": B * (N0 - 1) * 8 if want_prune else 0}
            # x_attn is scratch inside a layer: ONE buffer shared by all layers, placed behind the per-layer blocks
            off, per = {}, 0
            for f in self._FIELDS:
                if f == "x_attn":
                    continue
                off[f] = per if sizes[f] else None
                per += (sizes[f] + 255) // 256 * 256
            xa = (sizes["x_attn"] + 255) // 256 * 256
            rel = np.zeros((n_layers, 10), dtype=np.uint64)     # offsets relative to the aligned base; 0 rows masked below
            present = np.zeros((10,), dtype=bool)
            for j, f in enumerate(self._FIELDS):
                if f == "x_attn":
                    rel[:, j] = per * n_layers
                    present[j] = True
                elif off[f] is not None:
                    rel[:, j] = np.arange(n_layers, dtype=np.uint64) * per + off[f]
                    present[j] = True
            lay = (off, per, per * n_layers + xa + 256, rel, present)
            if len(_IO_LAYOUT) > 64:
                _IO_LAYOUT.clear()
            _IO_LAYOUT[key] = lay
        self.off, self.per_layer, total, rel, present = lay
        self.buf = torch.empty(total, dtype=torch.uint8, device=device)
        ptr = self.buf.data_ptr()
        base = (ptr + 255) // 256 * 256
        self.base_off = base - ptr
        self.io_np = np.zeros((n_layers,), dtype=_io_dtype())
        ptrs = (rel + np.uint64(base)) * present.astype(np.uint64)
        for j, f in enumerate(self._FIELDS):
            self.io_np[f] = ptrs[:, j]
        self.io_ptr = self.io_np.ctypes.data
        self._n_out = None

    def results(self):
        """(k_out, k_used, n_out) lists, read once after the call."""
        if self._n_out is None:
            self._k_out = self.io_np["k_out"].tolist()
            self._k_used = self.io_np["k_used"].tolist()
            self._n_out = self.io_np["n_out"].tolist()
        return self._k_out, self._k_used, self._n_out

    def ptr(self, layer, field):
        return int(self.io_np[field][layer])

    def view(self, layer, field, dtype, *shape):
        """tensor view of a layer's buffer (leading elements)."""
        n = 1
        for d in shape:
            n *= d
        if field == "x_attn":
            start = self.base_off + self.per_layer * self.n_layers
        else:
            start = self.base_off + layer * self.per_layer + self.off[field]
        return self.buf[start:start + n * _ESZ[dtype]].view(dtype).view(*shape)

    def n_in(self, layer):
        return self.N0 if layer == 0 else self.results()[2][layer - 1]

    def info(self, layer, temperature):
        """last_prune record of a layer (None when the layer ran without pruning scores)."""
        if not temperature > 0:
            return None
        k_out, k_used_l, _ = self.results()
        k, k_used, n = k_out[layer], k_used_l[layer], self.n_in(layer) - 1

        def fill():
            d = {"k": k, "score": self.view(layer, "score", torch.float32, self.B, n),
                 "threshold": self.view(layer, "threshold", torch.float32, self.B),
                 "count": self.view(layer, "count", torch.int32, self.B), "pruned": k_used > 0, "indices": None,
                 "indices_sort": None}
            if k_used > 0:
                d["indices"] = self.view(layer, "indices", torch.int64, self.B, k_used)
                d["indices_sort"] = self.view(layer, "indices_sort", torch.int64, self.B, n)
            return d
        return LazyPruneInfo(fill)

    def output(self, layer):
        return self.view(layer, "y", torch.float32, self.B, self.results()[2][layer], self.D)


def _query_w(qargs):
    """qargs: dict(sd_w, sd_hi, sd_lo, split_dtype, sd_scale, K, sd_dim, att_ft, stats_ws) -> (QueryW or None, keepalive)"""
    if qargs is None:
        return None
    q = QueryW(_p(qargs["sd_w"]), _p(qargs.get("sd_hi")), _p(qargs.get("sd_lo")), qargs.get("split_dtype", BF16),
               float(qargs.get("sd_scale", 1.0)), qargs["K"], 1.0 / (qargs["sd_dim"] ** 0.5), _p(qargs.get("att_ft")),
               _p(qargs.get("stats_ws")))
    return q


def vit_encoder(weights, x, qargs, temperature, sync_free=False, enqueue_only=False):
    """VisionTransformer's block loop in ONE library call.  weights: (list of VitBlockW, ctypes array of their addresses)
    from runtime.EncoderWeights; x f32 [B,N,D] contiguous; qargs: query-model operands (see _query_w) or None.  -> EncoderRun."""
    B, N, D = x.shape
    lib = load()
    wstructs, arr = weights
    L = len(wstructs)
    w0 = wstructs[0]
    nbytes = lib.madtp_vit_block_workspace(B, N, w0.dim, w0.fc1.n, w0.heads, w0.dtype)
    ws = workspace(nbytes, x.device)
    prune = qargs is not None and temperature > 0
    run = EncoderRun(L, B, N, D, x.device, qargs is not None, prune, False, None)
    q = _query_w(qargs)
    if sync_free:
        # device-side lengths: the whole encoder is enqueued without a host read of k; one read of the layers' records at the end
        if not vit_encoder_sync_free_ok(B, N, prune, qargs):
            raise RuntimeError("sync-free encoder call: needs pruning with a deferred att_ft, B * N < 4096 token rows and N <= 256")
        dims_dev = torch.empty(((L + 2) * 4,), device=x.device, dtype=torch.int32)
        dims_host = (ctypes.c_int32 * ((L + 1) * 4))()
        # enqueue_only: no copy of the records and no wait (dims_host NULL) - run.dims_dev holds {N_l, k, k applied, N_l+1} per layer
        # once the stream has run; run.results() is NOT filled (stream capture / enqueue-ahead, tools/graph_replay_probe.py)
        _check(lib.madtp_vit_encoder_async(arr, L, ctypes.byref(q), _p(x), run.io_ptr, _p(ws), ws.numel(), B, N, float(temperature),
                                           _p(dims_dev), 0 if enqueue_only else ctypes.addressof(dims_host), _stream()),
               "madtp_vit_encoder_async")
        run.keep = (x, wstructs, qargs, dims_dev, dims_host, ws)
        run.dims_dev = dims_dev
        return run
    _check(lib.madtp_vit_encoder(arr, L, ctypes.byref(q) if q is not None else None, _p(x), run.io_ptr, _p(ws), ws.numel(), B, N,
                                 float(temperature if prune else 0.0), _stream()), "madtp_vit_encoder")
    run.keep = (x, wstructs, qargs)
    return run


def vit_encoder_sync_free_ok(B, N, prune, qargs):
    """shapes / options madtp_vit_encoder_async takes (the launch-bound regime: small-tile GEMMs, <= 256-key attention)"""
    return bool(prune and qargs is not None and qargs.get("att_ft") is None and B * N < 4096 and 3 <= N <= 256)


def bert_decode_step(weights, x, kv_cache, t, kv_pre, kv_index, kv_ld, Nk, group=1):
    """One incremental decoding step (madtp_bert_decode_step): x f32 [rows, D] = the embedded new tokens at position t; kv_cache
    [layers, rows, Lmax, 2 D] in the attention dtype (appended in place); kv_pre: the per-layer cached cross-attention [k|v]
    tensors, kv_index int32 [rows / group] (group consecutive rows - an item's beams - read one block).  -> y f32 [rows, D]."""
    rows, D = x.shape
    lib = load()
    wstructs, arr = weights
    L = len(wstructs)
    w0 = wstructs[0]
    nbytes = max(lib.madtp_bert_layer_workspace(rows, 1, Nk, w0.dim, w0.inter.n, w0.heads, w0.dtype),
                 lib.madtp_bert_layer_workspace(rows // group, group, Nk, w0.dim, w0.inter.n, w0.heads, w0.dtype))
    ws = workspace(nbytes, x.device)
    y = torch.empty_like(x)
    kv = (c_void_p * L)(*[_p(tn) for tn in kv_pre])
    if kv_index.numel() != rows // group:
        raise ValueError("bert_decode_step: kv_index must hold one entry per group of rows")
    _check(lib.madtp_bert_decode_step(arr, L, _p(x), _p(kv_cache), rows, int(t), kv_cache.shape[2], kv, _p(kv_index), int(kv_ld), int(Nk),
                                      int(group), _p(y), _p(ws), ws.numel(), _stream()), "madtp_bert_decode_step")
    return y


def kv_cache_reorder(src, dst, beam_src, t):
    """dst[l, r, :t] = src[l, beam_src[r], :t] for the decode step's cache [layers, rows, Lmax, 2 D] (madtp_kv_cache_reorder)."""
    L, rows, Lmax, W = src.shape
    _check(load().madtp_kv_cache_reorder(_p(src), _p(dst), _p(beam_src), L, rows, Lmax, int(t), W * src.element_size(), _stream()),
           "madtp_kv_cache_reorder")


def bert_encoder_sync_free_ok(B, L, Nk, prune, qargs, mask2d):
    """shapes / options madtp_bert_encoder_async takes (small-tile GEMMs, <= 256-key attention, a deferred att_ft)"""
    return bool(prune and qargs is not None and qargs.get("att_ft") is None and mask2d is not None and B * L < 4096
                and 3 <= L <= 256 and Nk <= 256)


def bert_encoder(weights, hidden, hidden_lp, mask2d, qargs, temperature, cross_mode, enc0, enc1, Nk, enc_mask0, enc_mask1,
                 kv_pre0=None, kv_pre1=None, kv_index=None, kv_ld=0, sync_free=False):
    """BertEncoder's layer loop in ONE library call -> EncoderRun (y_lp of layer l: run.view(l, 'y_lp', ...)).
    sync_free: madtp_bert_encoder_async (device-side lengths: no host read of k between the layers)."""
    B, Lq, D = hidden.shape
    lib = load()
    wstructs, arr = weights
    L = len(wstructs)
    w0 = wstructs[0]
    nbytes = lib.madtp_bert_layer_workspace(B, Lq, Nk, w0.dim, w0.inter.n, w0.heads, w0.dtype)
    ws = workspace(nbytes, hidden.device)
    prune = qargs is not None and temperature > 0
    lp_dtype = None if w0.dtype == F32 else (torch.float16 if w0.dtype == F16S else torch.bfloat16)
    run = EncoderRun(L, B, Lq, D, hidden.device, qargs is not None, prune, mask2d is not None and prune, lp_dtype)
    q = _query_w(qargs)
    kv0 = (c_void_p * L)(*[_p(t) for t in kv_pre0]) if kv_pre0 is not None else None
    kv1 = (c_void_p * L)(*[_p(t) for t in kv_pre1]) if kv_pre1 is not None else None
    if sync_free:
        if not bert_encoder_sync_free_ok(B, Lq, Nk, prune, qargs, mask2d):
            raise RuntimeError("sync-free encoder call: needs pruning with a deferred att_ft and a padding mask, B * L < 4096 rows, "
                               "L <= 256 and <= 256 keys")
        dims_dev = torch.empty(((L + 2) * 4,), device=hidden.device, dtype=torch.int32)
        dims_host = (ctypes.c_int32 * ((L + 1) * 4))()
        _check(lib.madtp_bert_encoder_async(arr, L, ctypes.byref(q), _p(hidden), _p(hidden_lp), _p(mask2d), run.io_ptr, _p(ws), ws.numel(),
                                            B, Lq, Nk, float(temperature), int(cross_mode), _p(enc0), _p(enc1), _p(enc_mask0),
                                            _p(enc_mask1), kv0, kv1, _p(kv_index), int(kv_ld), _p(dims_dev), ctypes.addressof(dims_host),
                                            _stream()), "madtp_bert_encoder_async")
        run.keep = (hidden, hidden_lp, mask2d, wstructs, qargs, enc0, enc1, enc_mask0, enc_mask1, kv_pre0, kv_pre1, kv_index, dims_dev,
                    dims_host)
        return run
    _check(lib.madtp_bert_encoder(arr, L, ctypes.byref(q) if q is not None else None, _p(hidden), _p(hidden_lp), _p(mask2d), run.io_ptr,
                                  _p(ws), ws.numel(), B, Lq, Nk, float(temperature if prune else 0.0), int(cross_mode), _p(enc0),
                                  _p(enc1), _p(enc_mask0), _p(enc_mask1), kv0, kv1, _p(kv_index), int(kv_ld), _stream()),
           "madtp_bert_encoder")
    run.keep = (hidden, hidden_lp, mask2d, wstructs, qargs, enc0, enc1, enc_mask0, enc_mask1, kv_pre0, kv_pre1, kv_index)
    return run
