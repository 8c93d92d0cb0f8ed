"""ctypes binding of libpgt_hip.so (the C-ABI declared in include/pgt_hip.h).

The library is loaded lazily on first use and the product path FAILS LOUDLY when it is missing:
there is no CPU or PyTorch fallback anywhere in `pgtformer_amd`.
"""
import ctypes as C
import os

_LIB = None
# PGT_LIB_PATH: load another build of the library (same-box A/B of two builds: tools/gpu/ab_env.sh "" "PGT_LIB_PATH=..." "")
LIB_PATH = os.environ.get("PGT_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpgt_hip.so")

PGT_F32, PGT_BF16, PGT_F16X3, PGT_F16 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU, ACT_LEAKY02, ACT_SIGMOID = range(6)
EPI_PLAIN, EPI_SFT = 0, 1

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class ConvDesc(C.Structure):
    """Mirror of `pgt_conv_desc` (include/pgt_hip.h)."""
    _fields_ = [(n, i32) for n in ("dtype", "N", "H", "W", "Cin", "ldx", "ups", "KH", "KW", "stride", "pad_t",
                                   "pad_l", "Ho", "Wo", "Cout", "ldy", "act", "post_relu", "ldr", "epi",
                                   "ld_dec", "ld_shift")] + [("sft_w", f32)] + \
               [(n, i32) for n in ("out_f32", "force_bm", "force_bn", "scalar_epilogue", "kernel", "splitk", "stages",
                                   "orow_mul", "orow_xmul", "orow_off", "x_lo", "y_lo", "r_lo", "gn_groups", "gn_sub",
                                   "gn_nsub", "gn_img0", "gn_nimg", "res_f32", "x3_fold", "dec_lo", "shift_lo", "bias_rows", "out_split", "w2")]


# name -> argtypes (restype is int32 unless listed in _RESTYPES); must cover every symbol of pgt_hip.h
SIGNATURES = {
    "pgt_version": [],
    "pgt_last_error": [],
    "pgt_conv2d": [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp],
    "pgt_conv2d_workspace_bytes": [C.POINTER(ConvDesc)],
    "pgt_conv2d_ws": [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, sz, vp],
    "pgt_conv_gn_workspace_bytes": [i32, i32, i32, i32],
    "pgt_conv2d_gn": [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp],
    "pgt_conv2d_affine_in_ok": [C.POINTER(ConvDesc)],
    "pgt_conv2d_affine_in": [C.POINTER(ConvDesc), vp, vp, vp, i32, vp, vp, vp, vp, vp],
    "pgt_groupnorm_from_partials": [vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp],
    "pgt_groupnorm_workspace_bytes": [i32, i32, i32, i32],
    "pgt_groupnorm_affine": [i32, vp, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, sz, vp],
    "pgt_affine_act": [i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp],
    "pgt_layernorm": [i32, vp, i32, i32, i32, vp, vp, f32, vp, i32, vp, i32, vp, i32, vp],
    "pgt_channel_stats": [i32, vp, i32, i32, i32, i32, vp, vp, vp],
    "pgt_adain_affine": [vp, vp, vp, vp, f32, vp, vp, i32, vp],
    "pgt_sampled_channel_mean": [i32, vp, i32, i32, i32, i32, vp, vp],
    "pgt_sampled_pixel": [i32, i32],
    "pgt_weight_defect": [i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp],
    "pgt_sampled_rownorm_workspace_bytes": [i32, i32, i32],
    "pgt_sampled_rownorm_mean": [i32, vp, i32, i32, i32, i32, f32, vp, vp, vp],
    "pgt_fold_layernorm": [vp, vp, vp, vp, i32, i32, vp, vp, vp],
    "pgt_ln_linear": [i32, vp, i32, i32, i32, f32, vp, vp, i32, i32, vp, i32, vp],
    "pgt_ln_linear_x3": [vp, i32, i32, i32, i32, f32, vp, vp, i32, vp, i32, i32, vp],
    "pgt_ln_mlp_x3": [vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, i32, i32, vp],
    "pgt_attn_proj_mlp": [i32, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, i32, f32, vp, i32, vp],
    "pgt_attn_proj_mlp_sample_workspace_bytes": [i32, i32],
    "pgt_attn_proj_mlp_sample": [i32, vp, i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, f32, vp, vp, vp, vp],
    "pgt_frame_bias_workspace_bytes": [i32, i32, i32],
    "pgt_frame_bias": [i32, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp, vp],
    "pgt_sampled_pixel_cells": [i32, i32, i32],
    "pgt_mean_field_bias": [vp, vp, vp, i32, i32, i32, vp, vp],
    "pgt_window_attention": [i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "pgt_window_attention3d": [i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "pgt_mha": [i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, f32, vp],
    "pgt_groupnorm_affine_x3": [vp, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp, sz, vp],
    "pgt_affine_act_x3": [vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp],
    "pgt_layernorm_x3": [vp, i32, i32, i32, i32, vp, vp, f32, vp, i32, i32, vp, i32, i32, vp, i32, i32, vp],
    "pgt_window_attention_x3": [vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "pgt_mha_x3": [vp, i32, i32, vp, i32, i32, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, f32, vp],
    "pgt_x3_split": [vp, i32, vp, i32, i32, i64, i32, vp],
    "pgt_x3_merge": [vp, i32, i32, vp, i32, i64, i32, vp],
    "pgt_x3_to_half": [vp, i32, i32, vp, i32, i64, i32, vp],
    "pgt_sample_rows": [vp, i32, i32, i32, vp, vp, vp],
    "pgt_packed_weight_bytes": [i32, i32, i32, i32, i32, i32],
    "pgt_pack_conv_weight": [i32, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp],
    "pgt_fold_batchnorm": [vp, vp, vp, vp, f32, vp, i32, vp, vp, vp],
    "pgt_argmax_rows": [vp, i32, i32, i32, vp, vp],
    "pgt_rq_argmin": [vp, i32, vp, vp, i32, i32, vp, vp],
    "pgt_rq_nearest": [i32, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp],
    "pgt_rq_soft_codes": [vp, i32, vp, vp, i32, i32, f32, vp, vp, vp],
    "pgt_commit_loss_workspace_bytes": [],
    "pgt_commit_loss": [i32, vp, i32, vp, i32, i64, i32, vp, f32, i32, vp, sz, vp],
    "pgt_straight_through": [i32, vp, i32, vp, i32, vp, i32, i64, i32, vp],
    "pgt_vq_cluster_stats": [vp, i32, vp, i32, i32, i32, vp, vp],
    "pgt_vq_ema_update": [vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, vp],
    "pgt_embed_rows": [i32, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp],
    "pgt_row_sumsq": [i32, vp, i32, i32, i32, vp, vp],
    "pgt_maxpool3x3s2": [i32, vp, i32, i32, i32, i32, vp, vp],
    "pgt_gate_add": [i32, vp, i32, i32, i32, i32, vp, vp, vp, i32, vp, i32, vp],
    "pgt_resize_bilinear_ac": [i32, vp, i32, i32, i32, i32, i32, vp, i32, i32, i32, vp],
    "pgt_copy2d": [i32, vp, i32, i32, vp, i32, i64, i32, vp],
    "pgt_gather_frames": [vp, i64, vp, i64, vp, i32, i64, i32, vp],
    "pgt_zero2d": [vp, i64, i64, i32, vp],
    "pgt_count_saturated": [vp, i64, i64, i32, vp, vp],
    "pgt_prep_input": [i32, vp, i32, i32, i32, i32, vp, vp, vp],
    "pgt_nhwc_to_nchw_f32": [i32, vp, i32, i32, i32, i32, i32, vp, vp],
    "pgt_frame_to_u8": [i32, vp, i32, i32, i32, vp, vp],
    "pgt_program_load": [C.c_char_p, C.POINTER(C.c_void_p)],
    "pgt_program_destroy": [vp],
    "pgt_program_workspace_bytes": [vp],
    "pgt_program_io_bytes": [vp, C.POINTER(sz), C.POINTER(sz)],
    "pgt_program_info": [vp],
    "pgt_program_run": [vp, vp, vp, vp, sz, vp],
}
_RESTYPES = {"pgt_version": C.c_char_p, "pgt_last_error": C.c_char_p, "pgt_groupnorm_workspace_bytes": sz,
             "pgt_commit_loss_workspace_bytes": sz, "pgt_conv_gn_workspace_bytes": sz,
             "pgt_conv2d_workspace_bytes": sz, "pgt_packed_weight_bytes": sz, "pgt_attn_proj_mlp_sample_workspace_bytes": sz, "pgt_sampled_rownorm_workspace_bytes": sz,
             "pgt_frame_bias_workspace_bytes": sz, "pgt_program_workspace_bytes": sz, "pgt_program_info": C.c_char_p,
             "pgt_program_destroy": None}


class PgtError(RuntimeError):
    pass


# ---- call tracing (pgtformer_amd.export: the launch schedule of a forward as a tape a non-Python host replays) ------------------
# TRACE = None, or a list that receives (function name, [argument objects as passed]) for every call made through lib().
TRACE = None
TRACE_TENSORS = None       # {data_ptr: (storage base address, storage bytes, storage object id)} of the tensors ops._p handed out while tracing


class _TracedLib:
    """lib() while TRACE is a list: attribute access yields the real function wrapped in a recorder"""

    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)

        def call(*args):
            if TRACE is not None:
                rec = []
                for a in args:      # pointer arguments are resolved to their storage NOW: the allocator may hand the address out again later
                    if isinstance(a, C.c_void_p):
                        v = a.value or 0
                        rec.append(("ptr", v) + tuple(TRACE_TENSORS.get(v, (None, None, 0))))
                    elif a is None:
                        rec.append(("ptr", 0, None, None))
                    elif hasattr(a, "_obj"):           # C.byref(struct)
                        rec.append(("blob", bytes(a._obj)))
                    else:
                        rec.append(("val", a))
                TRACE.append((name, rec))
            return fn(*args)
        return call


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise PgtError(f"{LIB_PATH} is missing: build it with `python -m pgtformer_amd.build` "
                           "(there is no fallback path)")
        # torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's): import torch FIRST so this
        # library binds to the HIP runtime instance that owns torch's device context and streams.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, i32)
        _LIB = h
    if TRACE is not None:
        return _TracedLib(_LIB)
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().pgt_last_error()
        raise PgtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
