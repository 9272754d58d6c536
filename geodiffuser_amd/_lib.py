"""ctypes binding of libgeodiff_hip.so — the C ABI declared in include/geodiff_hip.h.

There is deliberately NO fallback: if the library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libgeodiff_hip.so")

GD_F16, GD_BF16, GD_F32 = 0, 1, 2
GD_TOKEN_MAJOR, GD_CHANNEL_MAJOR = 0, 1
GD_ATTN_MAX_SEGS = 12
GD_ATTN_MAX_PAIRS = 8
GD_ABI_VERSION = 6


class GdAttnSeg(Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("lse", c_void_p),
                ("bh", c_int32), ("heads", c_int32),
                ("warp_idx", c_void_p), ("warp_w", c_void_p), ("warp_m", c_void_p), ("warp_K", c_int32), ("q_scaled", c_int32),
                ("q_rows", c_void_p), ("q_rows_n", c_void_p), ("q_rows_len", c_int32)]


class GdAttnCfg(Structure):                  # gd_attn_cfg_t (defaults = GD_ATTN_CFG_DEFAULT)
    _fields_ = [("even_split", c_int32), ("qb", c_int32), ("ks", c_int32), ("nsplit", c_int32), ("handoff", c_int32)]


class GdConv3x3Cfg(Structure):               # gd_conv3x3_cfg_t (defaults = GD_CONV3X3_CFG_DEFAULT)
    _fields_ = [("pi", c_int32), ("ki", c_int32), ("ksplit", c_int32), ("dma", c_int32)]


class GdProbs(Structure):                    # gd_probs_t
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("lse", c_void_p), ("rows", c_void_p), ("n_valid", c_void_p), ("P", c_void_p),
                ("BH", c_int32), ("N", c_int32), ("R", c_int32), ("M", c_int32), ("Mpad", c_int32)]


class GdEditLosses(Structure):               # gd_edit_losses_t
    _fields_ = [("eo", c_void_p), ("ro", c_void_p), ("tgt", c_void_p), ("m_wo", c_void_p), ("m_edit", c_void_p), ("w_am", c_void_p),
                ("m_amodal", c_void_p),
                ("best", c_void_p), ("rows", c_void_p), ("n_valid", c_void_p),
                ("p_in", c_void_p), ("j_in", c_void_p), ("p_wo", c_void_p), ("j_wo", c_void_p), ("wgt", c_void_p),
                ("inv5", c_void_p), ("inv_rm", c_void_p), ("wv", c_void_p), ("inv5_bwd", c_void_p),
                ("out12", c_void_p), ("workspace", c_void_p), ("ticket", c_void_p),
                ("log_acc", c_void_p), ("loss_in", c_void_p), ("loss_out", c_void_p),
                ("H", c_int32), ("S", c_int32), ("D", c_int32), ("R", c_int32), ("use_amodal", c_int32)]


class GdRemovalBwd(Structure):               # gd_removal_bwd_t
    _fields_ = [("Pe", c_void_p), ("Pb", c_void_p), ("q", c_void_p), ("k", c_void_p), ("rows", c_void_p),
                ("p_in", c_void_p), ("j_in", c_void_p), ("p_wo", c_void_p), ("j_wo", c_void_p), ("wgt", c_void_p),
                ("m_inp", c_void_p), ("m_wo", c_void_p), ("gscale", c_void_p), ("gscale2", c_void_p), ("n_valid", c_void_p),
                ("dk_f32", c_void_p), ("workspace", c_void_p),
                ("coef", c_float), ("scale", c_float),
                ("H", c_int32), ("R", c_int32), ("N", c_int32), ("M", c_int32), ("Mpad", c_int32), ("D", c_int32), ("variant", c_int32)]


class GdHeadsSplit(Structure):               # gd_heads_split_t
    _fields_ = [("src", c_void_p * 6), ("dst", c_void_p * 6), ("rows", c_int32 * 6), ("n", c_int32), ("B", c_int32), ("heads", c_int32), ("D", c_int32)]


class GdHeadsMerge(Structure):               # gd_heads_merge_t
    _fields_ = [("src", c_void_p * 16), ("blend_b", c_void_p * 16), ("m", c_void_p * 16), ("out", c_void_p),
                ("src_f32", c_int32), ("B", c_int32), ("rows", c_int32), ("heads", c_int32), ("D", c_int32)]


class GeodiffError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol of include/geodiff_hip.h must be listed here
SIGNATURES = {
    "gd_version": (c_int, []),
    "gd_stream_capture_id": (c_int, [c_void_p, c_void_p]),
    "gd_copy_rows": (c_int, [c_void_p, c_int, c_int, ctypes.c_int64, c_void_p]),
    "gd_last_error": (c_char_p, []),
    "gd_error_string": (c_char_p, [c_int]),
    "gd_rasterize_workspace_bytes": (c_size_t, [c_int, c_int, c_float]),
    "gd_rasterize_points": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_size_t, c_void_p]),
    "gd_splat_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p]),
    "gd_splat_composite": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_int, c_void_p]),
    "gd_mesh_coverage": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gd_attn_fwd": (c_int, [POINTER(GdAttnSeg), c_int, c_int, c_int, c_int, c_float, POINTER(GdAttnCfg), c_void_p, c_size_t, c_int, c_void_p]),
    "gd_attn_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gd_attn_fwd_plan": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "gd_attn_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "gd_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_float, c_void_p, c_void_p, c_void_p, c_size_t, POINTER(c_int), POINTER(c_void_p), c_int, c_int, c_void_p]),
    "gd_attn_bwd_dkv_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "gd_attn_bwd_dkv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "gd_fp8_absmax_heads": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "gd_fp8_quant_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p]),
    "gd_fp8_quant_vt": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_fp8_quantize_qkv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p]),
    "gd_attn_fwd_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                c_void_p, c_void_p, c_int, c_void_p]),
    "gd_attn_probs": (c_int, [POINTER(GdProbs), POINTER(GdProbs), c_int, c_float, c_void_p, c_size_t, c_int, c_void_p]),
    "gd_removal_corr_max": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_void_p, c_int, c_int, c_int, c_void_p]),
    "gd_removal_loss_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_removal_bwd": (c_int, [POINTER(GdRemovalBwd), c_void_p, c_void_p, c_int, c_void_p]),
    "gd_nn_table": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_amodal_target": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                 c_int, c_void_p]),
    "gd_loss_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "gd_edit_losses_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gd_edit_losses_fwd": (c_int, [POINTER(GdEditLosses), c_int, c_void_p]),
    "gd_removal_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "gd_edit_losses_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, POINTER(GdRemovalBwd), c_int, c_void_p]),
    "gd_blend_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_attn_fwd_pair": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "gd_heads_split": (c_int, [c_void_p, c_int, c_void_p]),
    "gd_heads_merge": (c_int, [c_void_p, c_int, c_void_p]),
    "gd_edit_dq_fold": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_group_norm_nhwc_scratch_floats": (c_int64, [c_int, c_int, c_int]),
    "gd_group_norm_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p,
                                   c_void_p, c_int, c_void_p]),
    "gd_group_norm_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_bias_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "gd_geglu": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "gd_geglu_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "gd_layer_norm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_int, c_void_p]),
    "gd_add_layer_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "gd_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, POINTER(GdConv3x3Cfg)]),
    "gd_conv3x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(GdConv3x3Cfg),
                           c_void_p, c_size_t, c_int, c_void_p]),
    "gd_softsplat_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "gd_softsplat_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gd_hist_match": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gd_ddim_step": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int64, c_int, c_void_p]),
    "gd_ddim_step_v": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_void_p, c_int64, c_int, c_void_p]),
    "gd_masked_latent_update": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p]),
    "gd_sumsq": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "gd_norm_rescale": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
}

_lib = None


def load(path: str = LIB_PATH) -> ctypes.CDLL:
    """Load the library and bind every declared symbol (raises if the .so or a symbol is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise GeodiffError(
            f"{path} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m geodiffuser_amd.build` (needs hipcc).")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GeodiffError(f"libgeodiff_hip.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.gd_version() != GD_ABI_VERSION:
        raise GeodiffError(f"ABI version mismatch: library {lib.gd_version()} != binding {GD_ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        lib = load()
        raise GeodiffError(f"{what} failed ({code}: {lib.gd_error_string(code).decode()}): "
                           f"{lib.gd_last_error().decode()}")
