"""ctypes binding of libtfmq_hip.so (C ABI in include/tfmq_hip.h).

The product path has no CPU fallback: if the shared library cannot be loaded, or a GPU is
required and absent, every entry point raises.  PyTorch is used only as the owner of device
memory (``tensor.data_ptr()``) and of the HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFMQ_LIB_PATH") or os.path.join(_HERE, "libtfmq_hip.so")     # override: same-box A/B of two builds (scratch/)

c_void_p, c_int, c_size_t, c_float, c_double = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double


class TfmqError(RuntimeError):
    pass


class QSel(C.Structure):
    """tfmq_qsel: selects {delta, zero_point} of one activation quantizer for the current FSC group."""
    _fields_ = [("qtable", c_void_p), ("step", c_void_p), ("q_stride", C.c_int32), ("qid", C.c_int32)]


class ConvDesc(C.Structure):
    """tfmq_conv_desc"""
    _fields_ = [
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Cout", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
        ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("up2x", C.c_int32),
        ("x", c_void_p), ("w", c_void_p), ("wmeta", c_void_p), ("wscale", c_void_p), ("bias", c_void_p),
        ("aq", QSel),
        ("rowadd", c_void_p), ("rowadd_step", c_void_p), ("rowadd_ld", C.c_int32), ("rowadd_step_stride", C.c_int32),
        ("residual", c_void_p), ("y", c_void_p),
        ("ldy", C.c_int32), ("y_coff", C.c_int32),
        ("stats", c_void_p), ("stats_seg", C.c_int32), ("out_mode", C.c_int32),
        ("oq", QSel), ("yq", c_void_p), ("yt", c_void_p), ("t_col0", C.c_int32), ("x_f16", C.c_int32), ("tile", C.c_int32), ("res_f16", C.c_int32),
        ("x2", c_void_p), ("cin1", C.c_int32), ("w64", c_void_p), ("ksplit", C.c_int32),
    ]


class GnDesc(C.Structure):
    """tfmq_gn_desc"""
    _fields_ = [
        ("B", C.c_int32), ("HW", C.c_int32), ("C1", C.c_int32), ("C2", C.c_int32),
        ("x1", c_void_p), ("x2", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("eps", c_float), ("groups", C.c_int32), ("silu", C.c_int32),
        ("aq", QSel),
        ("yq", c_void_p), ("yf", c_void_p), ("xcat", c_void_p),
        ("half_out", C.c_int32), ("x_f16", C.c_int32),
    ]


class FfDesc(C.Structure):
    """tfmq_ff_desc"""
    _fields_ = [
        ("M", C.c_int32), ("C", C.c_int32), ("inner", C.c_int32),
        ("x", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
        ("aq0", QSel),
        ("w1", c_void_p), ("wmeta1", c_void_p), ("wscale1", c_void_p), ("bias1", c_void_p),
        ("aq2", QSel),
        ("w2", c_void_p), ("wmeta2", c_void_p), ("wscale2", c_void_p), ("bias2", c_void_p),
        ("y", c_void_p), ("oq", QSel), ("yq", c_void_p), ("ws", c_void_p),
        ("xq_pre", c_void_p), ("w0", c_void_p), ("wmeta0", c_void_p), ("wscale0", c_void_p), ("bias0", c_void_p), ("aq_pre", QSel),
        ("res_pre", c_void_p), ("y_pre", c_void_p),
        ("w3", c_void_p), ("wmeta3", c_void_p), ("wscale3", c_void_p), ("bias3", c_void_p), ("res_post", c_void_p), ("y_post", c_void_p),
        ("stats", c_void_p), ("stats_seg", C.c_int32),
    ]


class ChainGemm(C.Structure):
    """tfmq_chain_gemm"""
    _fields_ = [
        ("w", c_void_p), ("wmeta", c_void_p), ("wscale", c_void_p), ("bias", c_void_p), ("N", C.c_int32),
        ("aq", QSel), ("residual", c_void_p), ("y", c_void_p), ("ldy", C.c_int32), ("yt", c_void_p), ("t_col0", C.c_int32), ("next", C.c_int32),
    ]


class ChainDesc(C.Structure):
    """tfmq_chain_desc"""
    _fields_ = [
        ("M", C.c_int32), ("C", C.c_int32), ("T", C.c_int32), ("in_mode", C.c_int32),
        ("x", c_void_p), ("gn_a", c_void_p), ("gn_b", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
        ("n_gemm", C.c_int32), ("g", ChainGemm * 3), ("ws", c_void_p),
    ]


# name -> (restype, argtypes); every symbol declared in include/tfmq_hip.h
_SIGS = {
    "tfmq_abi_version": (c_int, []),
    "tfmq_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "tfmq_destroy": (c_int, [c_void_p]),
    "tfmq_last_error": (C.c_char_p, [c_void_p]),
    "tfmq_device_info": (c_int, [c_void_p, C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_size_t)]),
    "tfmq_quantize_act": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, QSel, c_int, c_void_p]),
    "tfmq_fake_quant_sel": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, QSel, c_int, C.c_float, c_void_p]),
    "tfmq_set_gemm_precision": (c_int, [c_void_p, c_int]),
    "tfmq_bins_to_grid": (c_int, [c_void_p, c_void_p, QSel, c_void_p, c_int, c_size_t, c_void_p]),
    "tfmq_scale_by_qdelta": (c_int, [c_void_p, c_void_p, QSel, c_void_p, c_int, c_void_p]),
    "tfmq_quantize_act_h": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, QSel, c_int, c_void_p]),
    "tfmq_fake_quant": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_int, c_void_p]),
    "tfmq_fake_quant_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "tfmq_minmax_ws_bytes": (c_size_t, [c_size_t, c_size_t]),
    "tfmq_minmax": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),
    "tfmq_minmax_to_qparam": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_act_range_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_void_p]),
    "tfmq_mse_ws_bytes": (c_size_t, [c_size_t, c_size_t]),
    "tfmq_mse_search": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tfmq_pack_w4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tfmq_unpack_w4": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_expand_w4": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_expand_w4_k64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_pack_w_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_conv2d_w4a8": (c_int, [c_void_p, C.POINTER(ConvDesc), c_void_p]),
    "tfmq_conv2d_f16": (c_int, [c_void_p, C.POINTER(ConvDesc), c_void_p]),
    "tfmq_ff_fused": (c_int, [c_void_p, C.POINTER(FfDesc), c_void_p]),
    "tfmq_row_chain": (c_int, [c_void_p, C.POINTER(ChainDesc), c_void_p]),
    "tfmq_gn_finalize": (c_int, [c_void_p, C.POINTER(GnDesc), c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "tfmq_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_linear_small_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tfmq_linear_small_w4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, QSel, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tfmq_groupnorm": (c_int, [c_void_p, C.POINTER(GnDesc), c_void_p]),
    "tfmq_groupnorm_from_stats": (c_int, [c_void_p, C.POINTER(GnDesc), c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "tfmq_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, C.c_long, c_int, QSel, c_void_p, c_void_p, c_void_p]),
    "tfmq_layernorm_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, C.c_long, c_int, QSel, c_void_p, c_void_p, c_void_p]),
    "tfmq_geglu": (c_int, [c_void_p, c_void_p, C.c_long, c_int, QSel, c_void_p, c_void_p, c_void_p]),
    "tfmq_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, QSel,
                               c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "tfmq_attention_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, QSel, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "tfmq_attention_q8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, QSel, QSel, QSel, QSel, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_float, c_void_p]),
    "tfmq_transpose_i8": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tfmq_ddim_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "tfmq_dpm_x0": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "tfmq_dpm_update": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p, c_size_t, c_void_p]),
    "tfmq_cfg_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "tfmq_plms_combine": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tfmq_ddim_update_cfg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                     c_void_p, c_void_p, c_void_p]),
    "tfmq_step_advance": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "tfmq_f32_to_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tfmq_row_broadcast_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, C.c_long, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_hw_selftest": (c_int, [c_void_p, c_void_p]),
    "tfmq_np_histogram": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, C.c_double, C.c_double, c_void_p, c_int, c_void_p, c_void_p]),
    "tfmq_np_histogram_rows": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "tfmq_silu": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tfmq_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "tfmq_nhwc_to_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "tfmq_adaround_init": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "tfmq_adaround_soft_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p]),
    "tfmq_adaround_bwd_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t,
                                       c_int, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "tfmq_adaround_scalars": (c_int, [c_float, c_float, c_float, c_int, c_void_p]),
    "tfmq_adaround_bwd_adam_dyn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t,
                                           c_int, c_void_p, c_void_p, c_void_p]),
    "tfmq_recon_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "tfmq_gemm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, C.c_long, C.c_long, C.c_long, C.c_long,
                              C.c_long, c_int, C.c_long, C.c_long, C.c_long, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p,
                              c_int, c_void_p]),
    "tfmq_gemm_f32_heads": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, C.c_long, C.c_long, C.c_long,
                                    C.c_long, C.c_long, c_int, C.c_long, C.c_long, C.c_long, c_int, C.c_long, C.c_long, C.c_long,
                                    c_float, c_int, c_void_p]),
    "tfmq_im2col": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "tfmq_col2im": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "tfmq_im2col_f16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    "tfmq_tap_gather_sum": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_void_p]),
    "tfmq_w_relayout": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "tfmq_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, C.c_long, c_int, c_void_p, c_void_p]),
    "tfmq_geglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, C.c_long, c_int, c_void_p, c_void_p]),
    "tfmq_silu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tfmq_groupnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                   c_int, c_void_p]),
    "tfmq_softmax_rows": (c_int, [c_void_p, c_void_p, c_void_p, C.c_long, c_int, c_float, c_void_p]),
    "tfmq_softmax_bwd_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_long, c_int, c_float, c_void_p]),
    "tfmq_attention_f32_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                       c_int, c_int, c_int, c_float, c_void_p]),
    "tfmq_attention_f32_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "tfmq_axpy": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_size_t, c_void_p]),
    "tfmq_upsample2x": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tfmq_upsample2x_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "tfmq_kl_softmax_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, C.c_long, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tfmq_fisher_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_size_t, c_void_p, c_void_p, c_void_p]),
    "tfmq_graph_begin": (c_int, [c_void_p, c_void_p]),
    "tfmq_graph_end": (c_int, [c_void_p, c_void_p, C.POINTER(c_int)]),
    "tfmq_graph_launch": (c_int, [c_void_p, c_int, c_void_p]),
    "tfmq_graph_destroy": (c_int, [c_void_p, c_int]),
    "tfmq_event_create": (c_int, [c_void_p, C.POINTER(c_int)]),
    "tfmq_event_record": (c_int, [c_void_p, c_int, c_void_p]),
    "tfmq_event_elapsed_ms": (c_int, [c_void_p, c_int, c_int, C.POINTER(c_float)]),
    "tfmq_stream_sync": (c_int, [c_void_p, c_void_p]),
    "tfmq_comm_unique_id": (c_int, [C.POINTER(C.c_uint8)]),
    "tfmq_comm_init": (c_int, [c_void_p, C.POINTER(C.c_uint8), c_int, c_int]),
    "tfmq_comm_info": (c_int, [c_void_p, C.POINTER(c_int), C.POINTER(c_int)]),
    "tfmq_allreduce_sum_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tfmq_comm_destroy": (c_int, [c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (no GPU needed for this step) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TfmqError(
            f"{LIB_PATH} is missing: build it with `python tfmq-dm_amd/build.py` (or __graft_entry__.build()). "
            "There is no CPU fallback for the TFMQ hot path.")
    # torch owns the device memory and streams handed to the kernels, so both must share ONE HIP
    # runtime: import torch first so libtfmq_hip.so binds to the libamdhip64 torch already loaded
    # (loading ours first pulls a second runtime from /opt/rocm and tfmq_create then sees no device).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Handle:
    """One tfmq_handle bound to a device.  `h.call("conv2d_w4a8", ...)` raises TfmqError on failure."""

    def __init__(self, device: int = 0):
        self.lib = load()
        hp = c_void_p()
        rc = self.lib.tfmq_create(int(device), C.byref(hp))
        if rc != 0 or not hp.value:
            raise TfmqError(f"tfmq_create(device={device}) failed with {rc}: no usable HIP device "
                            "(the TFMQ hot path has no CPU fallback)")
        self.h = hp
        self.device = device
        self.comm_world = 0     # > 0 once linklink.init_comm bound an RCCL communicator to this handle
        rep = C.c_uint32(0)
        self.call("hw_selftest", C.byref(rep))      # instruction semantics the epilogues rely on: fail loudly, never mis-quantise

    def call(self, name: str, *args):
        fn = getattr(self.lib, "tfmq_" + name)
        rc = fn(self.h, *args)
        if rc != 0:
            msg = self.lib.tfmq_last_error(self.h)
            raise TfmqError(f"tfmq_{name} failed ({rc}): {msg.decode() if msg else '?'}")
        return rc

    def device_info(self):
        cu, khz, mem = c_int(), c_int(), c_size_t()
        self.call("device_info", C.byref(cu), C.byref(khz), C.byref(mem))
        return cu.value, khz.value, mem.value

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.tfmq_destroy(self.h)
            self.h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_handles = {}           # device -> the exact-fp32 handle (the one the communicator and the samplers are bound to)
_prec_handles = {}      # (device, precision != 0) -> a handle whose fp32 GEMMs run at that operand precision


def handle(device: int = 0, gemm_precision: int = 0) -> Handle:
    """The process's tfmq_handle of `device` whose fp32 GEMMs run at `gemm_precision` (0 exact fp32, 1 bf16x3, 2 fp16).  The precision is a
    property a handle gets ONCE, right after tfmq_create, and keeps: ops.gemm_precision selects the handle, nothing toggles
    tfmq_set_gemm_precision between launches."""
    if not gemm_precision:
        if device not in _handles:
            _handles[device] = Handle(device)
        return _handles[device]
    key = (device, int(gemm_precision))
    if key not in _prec_handles:
        h = Handle(device)
        h.call("set_gemm_precision", key[1])
        h.gemm_precision = key[1]
        _prec_handles[key] = h
    return _prec_handles[key]
