"""ctypes binding of libpp_hip.so (the C ABI declared in include/pp_hip.h).

The product path has NO fallback: if the shared library is missing this module raises at import of the symbols
(`lib()`), and every op raises on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpp_hip.so")
if os.environ.get("PP_LAB") == "1" and os.environ.get("PP_LIB"):      # lab: A/B against a variant build (csrc/Makefile)
    LIB_PATH = os.path.abspath(os.environ["PP_LIB"])

PP_X_PLAIN, PP_X_CONV3X3 = 0, 1
PP_ACT_NONE, PP_ACT_GEGLU, PP_ACT_SILU, PP_ACT_SOFTMAX80 = 0, 1, 2, 3
PP_TILE_AUTO, PP_TILE_128x160, PP_TILE_64x160, PP_TILE_256x160 = 0, 1, 2, 3
PP_DT_F32, PP_DT_BF16, PP_DT_F16 = 0, 1, 2      # dtype codes of the C ABI (include/pp_hip.h)
PP_ATTN_AUTO, PP_ATTN_PHASED, PP_ATTN_PIPE_Q32, PP_ATTN_PIPE_Q64, PP_ATTN_PIPE_LOG2 = 0, 1, 2, 3, 4   # pp_attention_fwd_variant
ABI_VERSION = 22                                  # PP_ABI_VERSION of include/pp_hip.h this binding was written against
PP_ERR = {0: "PP_OK", -1: "PP_ERR_BAD_ARG", -2: "PP_ERR_UNSUPPORTED", -3: "PP_ERR_LAUNCH", -4: "PP_ERR_WORKSPACE"}

vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t


class PPGemmArgs(C.Structure):
    _fields_ = [
        ("M", i32), ("N", i32), ("K", i32), ("x_mode", i32),
        ("x1", vp), ("x2", vp),
        ("c1", i32), ("c2", i32), ("ldx1", i32), ("ldx2", i32),
        ("batch", i32), ("hin", i32), ("win", i32), ("hout", i32), ("wout", i32), ("stride", i32), ("up", i32),
        ("w", vp), ("bias", vp), ("rowvec", vp),
        ("ld_rowvec", i32), ("rows_per_batch", i32),
        ("res1", vp), ("ldres1", i32),
        ("res2", vp), ("ldres2", i32),
        ("scale", f32), ("act", i32),
        ("out", vp), ("ldo", i32), ("out_f32", i32),
        ("out_vt", vp), ("vt_col0", i32), ("vt_ld", i32),
        ("splitk", i32), ("tile", i32),
        ("workspace", vp),
        ("dbg", i32), ("dtype", i32), ("out_dup_rows", i32), ("res1_wrap_rows", i32),
        ("row_stats_out", vp), ("ln_stats", vp), ("ln_colsum", vp),
        ("ln_tiles", i32), ("ln_dim", i32), ("ln_eps", f32), ("w_batch_stride", i32),
        ("gn_acc", vp * 2), ("gn_cg", i32 * 2), ("gn_c0", i32 * 2), ("gn_groups", i32 * 2),
        ("x3", vp), ("x4", vp), ("c3", i32), ("c4", i32),
        ("gn_in_acc", vp), ("gn_in_gb", vp), ("gn_in_groups", i32), ("gn_in_silu", i32), ("gn_in_eps", f32),
        ("gn_dup_batch", i32),
        ("gn_next_out", vp), ("gn_next_gamma", vp), ("gn_next_beta", vp), ("gn_next_eps", f32), ("gn_next_silu", i32),
        ("gn_next_sub", i32), ("gn_dup_mask", i32), ("vec_batch_stride", i32),
        ("tile_ctr", vp), ("combine_fault", vp),
    ]


# name -> (restype, argtypes); must list every symbol of include/pp_hip.h (tests/test_abi.py checks the header)
SIGNATURES = {
    "pp_abi_version": (C.c_int, []),
    "pp_last_error": (C.c_char_p, []),
    "pp_build_id": (C.c_char_p, []),
    "pp_gemm_bf16": (C.c_int, [C.POINTER(PPGemmArgs), vp]),
    "pp_gemm_workspace_bytes": (sz, [C.POINTER(PPGemmArgs)]),
    "pp_linear_skinny": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "pp_timestep_embedding": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "pp_groupnorm_workspace_bytes": (sz, [C.c_int, C.c_int, C.c_int]),
    "pp_gemm_gn_stats_ok": (C.c_int, [C.POINTER(PPGemmArgs)]),
    "pp_conv_gn_supported": (C.c_int, [C.POINTER(PPGemmArgs)]),
    "pp_conv_gn_preferred": (C.c_int, [C.POINTER(PPGemmArgs)]),
    "pp_gemm_gn_next_ok": (C.c_int, [C.POINTER(PPGemmArgs), C.c_int]),
    "pp_gemm_combine_ctr_bytes": (sz, [C.POINTER(PPGemmArgs)]),
    "pp_gemm_combine_fused": (C.c_int, [C.POINTER(PPGemmArgs)]),
    "pp_xcd_placement_ok": (C.c_int, []),
    "pp_groupnorm_apply_acc": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, C.c_int, vp,
                                         C.c_int, vp]),
    "pp_zero_u64": (C.c_int, [vp, C.c_longlong, vp]),
    "pp_embed_splice": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_longlong, vp]),
    "pp_attention_small": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, f32, C.c_int, C.c_int, vp]),
    "pp_softmax_rows": (C.c_int, [vp, C.c_longlong, C.c_int, C.c_int, f32, vp, C.c_longlong, C.c_int, vp]),
    "pp_groupnorm_stats": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "pp_groupnorm_apply": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, f32, vp, vp, vp, C.c_int, vp,
                                     C.c_int, vp]),
    "pp_layernorm": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, f32, vp, C.c_int, vp]),
    "pp_attention_fwd": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, f32, C.c_int, vp]),
    "pp_attention_fwd_variant": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, f32, C.c_int, C.c_int, vp]),
    "pp_transpose_v": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "pp_conv3x3_direct": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp,
                                    vp, C.c_int, vp]),
    "pp_conv3x3_smallcout": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, vp]),
    "pp_nchw_to_nhwc": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]),
    "pp_nhwc_to_nchw": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]),
    "pp_add_bf16": (C.c_int, [vp, vp, vp, C.c_longlong, C.c_int, vp]),
    "pp_cfg_sched_step": (C.c_int, [vp, C.c_int, f32, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "pp_step_select_t": (C.c_int, [vp, vp, vp, vp]),
    "pp_ddim_variance_noise": (C.c_int, [vp, vp, C.c_int, vp, vp, vp]),
    "pp_latent_blend": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "pp_gn_conv3x3_smallcout_supported": (C.c_int, [C.c_int] * 3),
    "pp_gn_conv3x3_smallcout": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp,
                                          C.c_int, vp, C.c_int, vp]),
    "pp_tfront_supported": (C.c_int, [C.c_int] * 4),
    "pp_tfront": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, vp, vp, vp, vp, vp, C.c_float, vp, C.c_int, vp, C.c_int,
                            vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    "pp_attention_log2_ok": (C.c_int, [C.c_int] * 3),
    "pp_xattn_block_supported": (C.c_int, [C.c_int] * 5),
    "pp_xattn_fold": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_float,
                                vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "pp_xattn_block": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_float, vp, vp, vp, vp, vp, vp, C.c_int, vp,
                                 C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    "pp_ff_fused_supported": (C.c_int, [C.c_int] * 3),
    "pp_ff_fused": (C.c_int, [C.POINTER(PPGemmArgs), vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp]),
    "pp_step_head": (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp,
                              C.c_longlong, vp]),
    "pp_step_advance": (C.c_int, [vp, vp]),
    "pp_mask_prep": (C.c_int, [C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
}

_lib = None


class PPError(RuntimeError):
    pass


def lib():
    """Load libpp_hip.so (once).  Raises loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PPError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C powerpaint_amd/csrc`). There is no CPU/PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.pp_abi_version() != ABI_VERSION:
            raise PPError("libpp_hip.so ABI version mismatch")
        _lib = l
    return _lib


def dtype_code(dtype) -> int:
    """torch 16-bit dtype -> PP_DT_* (the HIP path stores activations and matrix weights in bf16 or fp16)."""
    import torch
    if dtype == torch.bfloat16:
        return PP_DT_BF16
    if dtype == torch.float16:
        return PP_DT_F16
    raise PPError(f"the HIP path computes in bf16 or fp16 (fp32 accumulate), not {dtype}")


def build_id() -> str:
    """What the loaded library was built from (pp_build_id: digest of sources + headers + extra flags)."""
    return lib().pp_build_id().decode()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().pp_last_error().decode() if rc == -3 else ""
        raise PPError(f"{what} failed: {PP_ERR.get(rc, rc)} {msg}")
