"""ctypes binding of libmgx.so (the C-ABI declared in include/mgx.h).

There is no fallback: if the shared library is missing or a call returns a non-zero status the
caller gets an exception.  Nothing here computes; it only marshals device pointers and sizes.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgx.so")

MGX_OK, MGX_ERR_ARG, MGX_ERR_LAUNCH, MGX_ERR_WORKSPACE = 0, 1, 2, 3      # status codes of include/mgx.h
_ERR = {1: "bad argument", 2: "kernel launch failure", 3: "workspace too small"}


class MgxError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mgx_conv_desc of include/mgx.h."""
    _fields_ = [
        ("nimg", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32), ("cin", C.c_int32),
        ("hout", C.c_int32), ("wout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
        ("pt", C.c_int32), ("pl", C.c_int32),
        ("flip", C.c_int32),
        ("ngemm", C.c_int32),
        ("dkh", C.c_int32), ("dkw", C.c_int32), ("cout", C.c_int32),
        ("wsk", C.c_int32), ("wsn", C.c_int32),
        ("wtap", C.c_int64),
        ("nbias", C.c_int32),
        ("accumulate", C.c_int32),
        ("act_alpha", C.c_float),
    ]


class GBlockWeights(C.Structure):
    """mgx_gblock_weights of include/mgx.h (device pointers to one grow block's variables)."""
    _fields_ = [("k0", C.c_void_p), ("b0", C.c_void_p),
                ("k1", C.c_void_p * 6), ("b1", C.c_void_p * 6), ("d1", C.c_void_p * 6), ("c1", C.c_void_p * 6),
                ("k2", C.c_void_p * 6), ("b2", C.c_void_p * 6), ("d2", C.c_void_p * 6), ("c2", C.c_void_p * 6),
                ("w1", C.c_void_p), ("e1", C.c_void_p), ("w2", C.c_void_p), ("e2", C.c_void_p),
                ("w3", C.c_void_p), ("e3", C.c_void_p), ("w4", C.c_void_p), ("e4", C.c_void_p)]


class GHeadWeights(C.Structure):
    """mgx_ghead_weights / mgx_ghead_grads of include/mgx.h (device pointers to the variables, or to their
    gradients, of the stage-0 generator's head)."""
    _fields_ = [("k0", C.c_void_p), ("b0", C.c_void_p),
                ("k1", C.c_void_p * 4), ("b1", C.c_void_p * 4), ("d1", C.c_void_p * 4), ("c1", C.c_void_p * 4),
                ("k2", C.c_void_p * 4), ("b2", C.c_void_p * 4), ("d2", C.c_void_p * 4), ("c2", C.c_void_p * 4)]


class GHeadTensors(C.Structure):
    """mgx_ghead_tensors: the four per-branch tensors (nimg, 4, 50, 32)."""
    _fields_ = [("y", C.c_void_p * 4)]


_P = C.c_void_p
_I64 = C.c_int64
_I = C.c_int
_F = C.c_float
_DESC = C.POINTER(ConvDesc)

# name -> (restype, argtypes); mirrors include/mgx.h one to one
PROTOTYPES = {
    "mgx_version": (_I, []),
    "mgx_conv_gemm_f32": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P]),
    "mgx_conv_wgrad_workspace": (_I64, [_DESC]),
    "mgx_conv_wgrad_f32": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _I64, _P]),
    "mgx_conv_route": (_I, [_DESC, _I]),
    "mgx_ghead_fwd_f32": (_I, [C.POINTER(GHeadWeights), _P, _I64, C.POINTER(GHeadTensors), _P]),
    "mgx_ghead_bwd_workspace": (_I64, [_I64]),
    "mgx_ghead_bwd_f32": (_I, [C.POINTER(GHeadWeights), _P, _I64, C.POINTER(GHeadTensors), C.POINTER(GHeadWeights), _P, _P,
                               _I64, _P]),
    "mgx_clip_checksum_workspace": (_I64, [_I64]),
    "mgx_clip_checksum": (_I, [_P, _I64, _I64, _P, _P, _I64, _P]),
    "mgx_gblock_pack_floats": (_I64, []),
    "mgx_gblock_side_floats_n": (_I64, [_I64]),
    "mgx_gblock_side_floats": (_I64, [_I64, _I, _I]),
    "mgx_gblock_fwd_f32": (_I, [C.POINTER(GBlockWeights), _P, _P, _P, _P, _I64, _I, _I, _I64, _I64, _P]),
    "mgx_gblock_packt_floats": (_I64, []),
    "mgx_gblock_wgrad_workspace_n": (_I64, [_I64]),
    "mgx_gblock_wgrad_workspace": (_I64, [_I64, _I, _I]),
    "mgx_gblock_dgrad_f32": (_I, [C.POINTER(GBlockWeights), _P, _P, _P, _P, _I64, _I, _I, _I64, _I64, _P]),
    "mgx_gblock_wgrad_f32": (_I, [_P, _P, C.POINTER(GBlockWeights), _P, _I64, _I64, _P]),
    "mgx_upsample2x_fwd": (_I, [_P, _P, _I64, _I, _I, _I, _P]),
    "mgx_upsample2x_bwd": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "mgx_avgpool2x_fwd": (_I, [_P, _P, _I64, _I, _I, _I, _P]),
    "mgx_avgpool2x_bwd": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "mgx_add_n": (_I, [C.POINTER(_P), _I, _P, _I64, _P]),
    "mgx_add_inplace": (_I, [_P, _P, _I64, _P]),
    "mgx_gather_groups": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "mgx_collapse_table": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mgx_collapse_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "mgx_weighted_sum_fwd": (_I, [_P, _P, _P, _P, _I64, _P]),
    "mgx_weighted_sum_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I64, _I64, _P]),
    "mgx_reduce_workspace": (_I64, []),
    "mgx_layernorm_workspace": (_I64, [_I64, _I64]),
    "mgx_layernorm_fwd": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _F, _P, _I64, _P]),
    "mgx_layernorm_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I64, _I64, _P, _I64, _P]),
    "mgx_layernorm_fwd_static_max_samples": (_I64, []),
    "mgx_layernorm_fwd_static": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _F, _P]),
    "mgx_mbstd_workspace": (_I64, [_I64, _I64]),
    "mgx_mbstd_fwd": (_I, [_P, _P, _P, _I64, _I, _I, _I, _P, _I64, _P]),
    "mgx_mbstd_bwd": (_I, [_P, _P, _P, _I64, _I, _I, _I, _P, _I64, _P]),
    "mgx_softrect_fwd": (_I, [_P, _P, _I64, _F, _P]),
    "mgx_softmax_rows": (_I, [_P, _P, _I64, _I, _P]),
    "mgx_softrect_bwd": (_I, [_P, _P, _P, _I, _I64, _F, _P]),
    "mgx_dense1_fwd": (_I, [_P, _P, _P, _P, _I64, _I64, _P]),
    "mgx_dense1_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mgx_logit_sums": (_I, [_P, _P, _I64, _P, _I, _P]),
    "mgx_combine_grads": (_I, [_P, _P, _P, _P, _P, _I64, _P]),
    "mgx_wgan_loss": (_I, [_P, _I64, _I, _P, _P, _P, _I64, _P]),
    "mgx_adam_multi": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _F, _F, _F, _P]),
    "mgx_adam_multi_m": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _P]),
    "mgx_softmax_xent": (_I, [_P, _P, _P, _P, _P, _I64, _I, _P]),
    "mgx_adam_multi_devlr": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _F, _F, _P]),
    "mgx_conv_bf16_supported": (_I, [_DESC]),
    "mgx_conv_gemm_bf16": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P]),
    "mgx_conv1x1_blend_bf16": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P]),
    "mgx_conv_dgrad_to_audio_bf16": (_I, [_DESC, _P, _P, _P, _P, _P, _I, _P]),
    "mgx_conv1x1_blend_audio_bf16": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    "mgx_conv1x1_blend_supported": (_I, [_DESC]),
    "mgx_conv1x1_blend_f32": (_I, [_DESC, _P, _P, _P, _P, _P, _P, _P]),
    "mgx_conv_pair_bf16_supported": (_I, [_I64, _I, _I, _I]),
    "mgx_conv_pair_bf16": (_I, [_I64, _I, _I, _P, _P, _P, _F, _P, _P, _F, _I, _P, _P]),
    "mgx_audio_conv3_bf16_supported": (_I, [_I64, _I, _I]),
    "mgx_audio_conv3_bf16": (_I, [_I64, _I, _I, _P, _P, _P, _F, _P, _P, _F, _P, _P]),
    "mgx_conv_chain_bf16_supported": (_I, [_I64, _I, _I, _I, _P]),
    "mgx_conv_chain_bf16": (_I, [_I64, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mgx_conv_chain_bf16_taped": (_I, [_I64, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "mgx_conv_wgrad_bf16_workspace": (_I64, [_DESC]),
    "mgx_conv_wgrad_bf16": (_I, [_DESC, _P, _P, _P, _P, _P, _I64, _P]),
    "mgx_conv_bwd_bf16_supported": (_I, [_DESC]),
    "mgx_conv_bwd_bf16_workspace": (_I64, [_DESC]),
    "mgx_conv_bwd_bf16": (_I, [_DESC, _P, _P, _P, _P, _P, _P, C.c_float, _P, _I64, _P]),
    "mgx_pointwise2_bf16_fwd": (_I, [_P, _P, _P, _P, _P, _I64, _F, _P]),
    "mgx_pointwise2_bf16_wgrad_workspace": (_I64, [_I64]),
    "mgx_pointwise2_bf16_wgrad": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mgx_pointwise_to2_bf16": (_I, [_P, _P, _P, _I64, _I, _P]),
    "mgx_weighted_sum_bf16_fwd": (_I, [_P, _P, _P, _P, _I64, _P]),
    "mgx_weighted_sum_bf16_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mgx_weighted_sum_bf16_bwd_mul": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P]),
    "mgx_weighted_sum_bf16_bwd_act": (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _I64, _I64, _P]),
    "mgx_gblock_pack_bf16_bytes": (_I64, []),
    "mgx_gblock_packt_bf16_bytes": (_I64, []),
    "mgx_gblock_side_elems_bf16": (_I64, [_I64]),
    "mgx_gblock_fwd_bf16": (_I, [C.POINTER(GBlockWeights), _P, _P, _P, _P, _I64, _I, _I, _I64, _I64, _P]),
    "mgx_gblock_dgrad_bf16": (_I, [C.POINTER(GBlockWeights), _P, _P, _P, _P, _I64, _I, _I, _I64, _I64, _P]),
    "mgx_gblock_wgrad_bf16": (_I, [_P, _P, C.POINTER(GBlockWeights), _P, _I64, _I64, _P]),
    "mgx_gblock_side_elems_bf16_lean": (_I64, [_I64]),
    "mgx_gblock_fwd_bf16_lean": (_I, [C.POINTER(GBlockWeights), _P, _P, _P, _P, _I64, _I, _I, _I64, _I64, _P]),
    "mgx_gblock_dgrad_bf16_lean": (_I, [C.POINTER(GBlockWeights), _P, _P, _P, _P, _I64, _I, _I, _I64, _I64, _P]),
    "mgx_gblock_wgrad_bf16_lean": (_I, [C.POINTER(GBlockWeights), _P, _P, C.POINTER(GBlockWeights), _P, _I64, _I64, _P]),
    "mgx_mul_bf16": (_I, [_P, _I, _P, _P, _I64, _P]),
    "mgx_mul_dact_bf16": (_I, [_P, _I, _P, _F, _P, _I64, _P]),
    "mgx_cast_f32_to_bf16": (_I, [_P, _P, _I64, _P]),
    "mgx_cast_bf16_to_f32": (_I, [_P, _P, _I64, _P]),
}

_lib = None


def load():
    """Load libmgx.so (built by `make -C music_generator_amd/csrc`); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MGX_LIBRARY", LIB_PATH)      # override: diagnostic builds of the same ABI
    if not os.path.exists(path):
        raise MgxError(
            "libmgx.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU or PyTorch fallback for this path)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str):
    if status != MGX_OK:
        raise MgxError("%s failed: %s (status %d)" % (what, _ERR.get(status, "unknown"), status))
