"""ctypes binding of libskoots_hip.so (C ABI: include/skoots_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C skoots_amd/csrc``.  Loading fails loudly if it is absent: the product
has no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKOOTS_HIP_LIB") or os.path.join(_HERE, "libskoots_hip.so")  # override: A/B kernel builds

SK_F16, SK_F32, SK_I16, SK_I32, SK_U8 = 0, 1, 2, 3, 4

_DTYPE_CODE = {torch.float16: SK_F16, torch.float32: SK_F32, torch.int16: SK_I16,
               torch.int32: SK_I32, torch.uint8: SK_U8}


class SkootsHipError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C skoots_amd/csrc). "
            "skoots_amd has no CPU fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int32)


class ConvSrc(C.Structure):
    _fields_ = [("data", vp), ("affine", vp), ("c", i32), ("upsample", i32)]


_SIGS = {
    "sk_last_error": (C.c_char_p, []),
    "sk_abi_version": (i32, []),
    "sk_mfma_probe": (i32, [vp, sz, i32, i32, C.POINTER(C.c_double), vp]),
    "sk_debug_set_timing_buffer": (i32, [vp, sz]),
    "sk_stream_create_cu_mask": (i32, [C.POINTER(C.c_uint32), i32, C.POINTER(vp)]),
    "sk_stream_destroy": (i32, [vp]),
    "sk_debug_where": (i32, [vp, i32, i32, vp]),
    "sk_vec_interleave": (i32, [vp, vp, i64, vp]),
    "sk_vec_deinterleave": (i32, [vp, vp, i64, vp]),
    "sk_vector_to_embedding": (i32, [vp, i32, vp, i32, i32, i32, fp, i32, vp]),
    "sk_index_skeleton_by_embed": (i32, [vp, i32, i32, i32, i32, vp, i64, vp, vp]),
    "sk_follow_assign": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, i32, i32, fp,
                               i32, i32, i32, vp]),
    "sk_gate_dilate_scatter": (i32, [vp, i32, i32, C.POINTER(C.c_int64), i64, i64, i64, i32, i32, i32, ip, ip, ip,
                                     vp, vp, vp, i32, i32, i32, f32, f32, vp]),
    "sk_max_filter3d": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "sk_ccl_workspace_bytes": (sz, [i64]),
    "sk_ccl_crop": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp, vp]),
    "sk_seam_pairs": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp]),
    "sk_seam_components_host": (i32, [ip, i32, ip, ip, i32]),
    "sk_relabel_lut": (i32, [vp, i64, vp, i32, vp]),
    "sk_compact_nonzero": (i32, [vp, i64, vp, vp, i64, vp]),
    "sk_seam_union": (i32, [vp, i32, i32, i32, vp, vp, i64, vp]),
    "sk_relabel_lut_offset": (i32, [vp, i64, vp, i64, vp, vp]),
    "sk_renumber_workspace_bytes": (sz, [i64, i32]),
    "sk_first_seen": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sk_renumber": (i32, [vp, i64, i32, vp, sz, vp, vp]),
    "sk_conv3d": (i32, [C.POINTER(ConvSrc), i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "sk_conv3d_box": (i32, [C.POINTER(ConvSrc), i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, ip, vp]),
    "sk_conv3d_box_split": (i32, [C.POINTER(ConvSrc), i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, ip, vp]),
    "sk_conv3d_down_act": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "sk_conv3d_down_act_split": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "sk_conv3d_num_blocks": (i32, [i32, i32, i32, i32, i32, i32]),
    "sk_conv3d_pack_weight_host": (i64, [fp, i32, i32, i32, vp]),
    "sk_conv3d_upfold": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "sk_conv3d_upfold_split": (i32, [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "sk_conv3d_pack_weight_upfold_split_host": (i64, [fp, i32, i32, i32, vp]),
    "sk_conv3d_upfold_num_blocks": (i32, [i32, i32, i32, i32]),
    "sk_conv3d_pack_weight_upfold_host": (i64, [fp, i32, i32, i32, vp]),
    "sk_conv3d_stem": (i32, [vp, i32, i32, i32, ip, i32, i32, i32, i32, f32, f32, vp, vp, i32, vp, vp, sz, vp]),
    "sk_conv3d_stem_raw": (i32, [vp, i32, i32, i32, ip, i32, i32, i32, i32, f32, f32, vp, vp, i32, vp, vp, vp, sz, vp]),
    "sk_conv3d_stem_apply": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "sk_conv3d_stem_num_blocks": (i32, [i32, i32, i32]),
    "sk_conv3d_stem_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "sk_groupnorm_finalize": (i32, [vp, i32, i32, i32, i32, i64, vp, vp, f32, vp, vp]),
    "sk_groupnorm_silu": (i32, [vp, vp, i32, i64, i32, vp]),
    "sk_heads": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, ip, ip, vp]),
    "sk_conv3d_split": (i32, [C.POINTER(ConvSrc), i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "sk_conv3d_pack_weight_split_host": (i64, [fp, i32, i32, i32, vp]),
    "sk_conv3d_stem_apply_split": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "sk_conv3d_down_act_mix8": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "sk_conv3d_upfold_mix8": (i32, [vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "sk_conv3d_pack_weight_upfold_mix8_host": (i64, [fp, i32, i32, i32, vp, ip]),
    "sk_conv3d_stem_apply_mix8": (i32, [i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]),
    "sk_groupnorm_silu_mix8": (i32, [vp, vp, i32, i64, i32, vp]),
    "sk_conv3d_mix8": (i32, [C.POINTER(ConvSrc), i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, ip, vp]),
    "sk_conv3d_pack_weight_mix8_host": (i64, [fp, i32, i32, vp, ip]),
    "sk_groupnorm_silu_split": (i32, [vp, vp, i32, i64, i32, vp]),
    "sk_heads_split": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, ip, ip, vp]),
    "sk_conv3d_f32": (i32, [C.POINTER(ConvSrc), i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sk_conv3d_f32_num_blocks": (i32, [i32, i32, i32]),
    "sk_groupnorm_silu_f32": (i32, [vp, vp, i32, i64, i32, vp]),
    "sk_groupnorm_finalize_stats": (i32, [vp, i32, i32, i32, i32, i64, vp, vp, C.c_float, vp, vp, vp]),
    "sk_train_gn_silu": (i32, [vp, vp, vp, i32, i64, i32, vp]),
    "sk_train_gn_silu_bwd": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp]),
    "sk_train_gn_bwd_workspace_floats": (i64, [i32, i64, i32]),
    "sk_train_gn_bwd_num_blocks": (i32, [i64]),
    "sk_train_loss": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, fp, fp, fp, vp, vp, vp, vp]),
    "sk_baked_embed_to_prob": (i32, [vp, vp, vp, i32, i64, fp, C.c_float, vp]),
    "sk_train_tversky": (i32, [vp, vp, i32, i64, C.c_float, C.c_float, C.c_float, vp, vp, vp]),
    "sk_train_loss_workspace_floats": (i64, [i32, i64]),
    "sk_train_loss_num_blocks": (i32, [i64]),
    "sk_train_conv_dgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "sk_train_conv_wgrad": (i32, [C.POINTER(ConvSrc), i32, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "sk_train_conv_wgrad_workspace_floats": (i64, [i32, i32, i32, i32, i32, i32, i32]),
    "sk_train_pack_weight": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "sk_train_interleave2": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "sk_train_gn_silu_f16": (i32, [vp, vp, vp, vp, i32, i64, i32, vp]),
    "sk_train_gn_silu_bwd_f16": (i32, [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp, vp]),
    "sk_train_gn_bwd_f16_workspace_floats": (i64, [i32, i64, i32]),
    "sk_train_absmax_scale": (i32, [vp, i64, vp, vp]),
    "sk_train_cast_f32_f16": (i32, [vp, vp, i64, vp, vp]),
    "sk_train_cast_f16_f32": (i32, [vp, vp, i64, vp, i32, vp]),
    "sk_train_conv_wgrad_f16": (i32, [C.POINTER(ConvSrc), i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "sk_train_sumpool2": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "sk_train_gn_silu_bwd_f16h": (i32, [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp, vp]),
    "sk_train_sumpool2_f16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "sk_train_interleave2_add16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "sk_train_interleave2_h": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "sk_train_sumpool2_hh": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "sk_train_heads_dgrad_f16": (i32, [vp, vp, vp, vp, vp, i64, i32, vp]),
    "sk_train_heads_fwd_f16": (i32, [vp, vp, vp, vp, i64, vp]),
    "sk_train_heads_wgrad_workspace_floats": (i64, [i64]),
    "sk_train_heads_wgrad_f16": (i32, [vp, vp, vp, vp, i64, vp, vp]),
    "sk_train_stem_fwd_f16": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, C.c_size_t, vp]),
    "sk_train_stem_wgrad_f16": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "sk_bake_skeleton": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, fp, vp, vp, vp]),
    "sk_average_baked_skeletons": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "sk_mask_iou_workspace_bytes": (sz, [i32, i32]),
    "sk_mask_iou": (i32, [vp, vp, i64, vp, i32, i32, vp, i32, i32, vp, vp, sz, vp]),
    "sk_train_adamw": (i32, [vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]),
}

# bf16 twins (include/skoots_hip_bf16.h): the training-path sources are compiled a second time on bf16 storage and
# exported with the suffix _bf16, same signatures
BF16_TWINS = (
    "sk_conv3d", "sk_conv3d_box", "sk_conv3d_box_split", "sk_conv3d_down_act", "sk_conv3d_down_act_split", "sk_conv3d_num_blocks", "sk_conv3d_pack_weight_host", "sk_conv3d_pack_weight_split_host",
    "sk_conv3d_mix8", "sk_conv3d_down_act_mix8", "sk_conv3d_pack_weight_mix8_host", "sk_conv3d_stem_apply_mix8", "sk_groupnorm_silu_mix8",
    "sk_conv3d_split", "sk_conv3d_stem", "sk_conv3d_stem_raw", "sk_conv3d_stem_apply", "sk_conv3d_stem_apply_split",
    "sk_conv3d_stem_num_blocks", "sk_conv3d_stem_workspace_bytes", "sk_groupnorm_finalize",
    "sk_groupnorm_finalize_stats", "sk_groupnorm_silu", "sk_groupnorm_silu_split", "sk_heads", "sk_heads_split",
    "sk_baked_embed_to_prob", "sk_train_absmax_scale", "sk_train_adamw", "sk_train_cast_f16_f32",
    "sk_train_cast_f32_f16", "sk_train_conv_wgrad", "sk_train_conv_wgrad_f16",
    "sk_train_conv_wgrad_workspace_floats", "sk_train_gn_bwd_f16_workspace_floats", "sk_train_gn_bwd_num_blocks",
    "sk_train_gn_bwd_workspace_floats", "sk_train_gn_silu", "sk_train_gn_silu_bwd", "sk_train_gn_silu_bwd_f16",
    "sk_train_gn_silu_bwd_f16h", "sk_train_gn_silu_f16", "sk_train_heads_fwd_f16", "sk_train_heads_wgrad_f16",
    "sk_train_heads_wgrad_workspace_floats", "sk_train_interleave2", "sk_train_interleave2_add16", "sk_train_interleave2_h",
    "sk_train_sumpool2_hh", "sk_train_heads_dgrad_f16", "sk_train_loss",
    "sk_train_loss_num_blocks", "sk_train_loss_workspace_floats", "sk_train_pack_weight", "sk_train_stem_fwd_f16",
    "sk_train_stem_wgrad_f16", "sk_train_sumpool2", "sk_train_sumpool2_f16", "sk_train_tversky",
)
for _name in BF16_TWINS:
    _SIGS[_name + "_bf16"] = _SIGS[_name]

EXPORTS = tuple(_SIGS)
_missing = []
for _name, (_res, _args) in _SIGS.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        _missing.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args
if _missing:
    raise ImportError(f"{LIB_PATH} lacks symbols declared in include/skoots_hip*.h: {_missing}")


def last_error() -> str:
    return lib.sk_last_error().decode()


def check(rc: int) -> None:
    """Map the C status convention onto Python exceptions."""
    if rc == 0:
        return
    msg = last_error()
    if rc == -1:
        raise ValueError(msg)
    raise SkootsHipError(f"[{rc}] {msg}")


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise ValueError(f"unsupported dtype {t.dtype}") from None


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} must live on the MI355X (got device {t.device}); skoots_amd runs HIP kernels "
            "only and has no CPU fallback")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


def float_array(values):
    arr = (C.c_float * len(values))(*values)
    return arr
