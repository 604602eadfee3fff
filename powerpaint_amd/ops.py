"""Tensor-level convenience wrappers over the C ABI (one call = one kernel launch family).

These take/return torch CUDA tensors (bf16 NHWC / row-major activations) and exist for tests, micro-benchmarks and
users who want a single op; the networks themselves go through `engine.Builder`, which bakes raw pointers into plans.
Every wrapper raises `PPError` on a non-zero return code -- there is no fallback.
"""
import ctypes as C
from typing import Optional

import torch

from . import _lib as L

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


def _set_gn(a, gn, rows_per_batch):
    """gn: up to two (acc int64 [B][groups][2] (zeroed by the caller), channels_per_group, channel_offset, groups)."""
    if not gn:
        return
    a.rows_per_batch = rows_per_batch
    for k, (acc, cg, c0, groups) in enumerate(gn):
        a.gn_acc[k], a.gn_cg[k], a.gn_c0[k], a.gn_groups[k] = acc.data_ptr(), cg, c0, groups


# (ABI v21) the in-kernel split-K combine: `fuse_combine=True` on gemm / conv3x3 hands the launch zeroed tile counters where
# pp_gemm_combine_ctr_bytes() asks for them.  `last_combine` describes the most recent such call: fused (did pp_gemm_bf16
# combine in-kernel), ctr (the counters: all zero again after the launch), and combine_faults() reads the fault counter.
last_combine = {"fused": False, "ctr": None}
_fault_words = {}


def _fault_word(device) -> torch.Tensor:
    key = torch.device(device)
    if key not in _fault_words:
        _fault_words[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _fault_words[key]


def combine_faults(device) -> int:
    """Split-K tiles whose splits did NOT share an XCD since the process started (synchronises; must stay 0)."""
    return int(_fault_word(device).item())


def _attach_combine(a, device, enable):
    """enable: False / None -- no counters; True -- where pp_gemm_combine_ctr_bytes() advises them; "force" -- also where the
    library advises the separate combine (large tiles); a tensor -- the caller's own (zeroed or re-armed) counters."""
    last_combine["fused"], last_combine["ctr"] = False, None
    if enable is None or enable is False:
        return
    bound = ((a.M + 127) // 128) * ((a.N + 159) // 160) * 16      # 16 bytes per 128 x 160 tile always suffice
    if torch.is_tensor(enable):
        ctr = enable
        assert ctr.dtype == torch.int64 and ctr.numel() * 8 >= bound
    else:
        n = L.lib().pp_gemm_combine_ctr_bytes(C.byref(a))
        if enable == "force":
            n = max(n, bound)
        if not n:
            return
        ctr = torch.zeros(n // 8, dtype=torch.int64, device=device)
    a.tile_ctr, a.combine_fault = ctr.data_ptr(), _fault_word(device).data_ptr()
    last_combine["ctr"] = ctr
    last_combine["fused"] = bool(L.lib().pp_gemm_combine_fused(C.byref(a)))


def groupnorm_apply_acc(x: torch.Tensor, acc: torch.Tensor, gamma, beta, eps: float, silu: bool, groups: int = 32,
                        x2=None):
    """GroupNorm(+SiLU) of concat(x, x2) from statistics accumulated by the producers (PPGemmArgs.gn_acc)."""
    B, H, W, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    y = torch.empty(B, H, W, C1 + C2, dtype=x.dtype, device=x.device)
    L.check(L.lib().pp_groupnorm_apply_acc(_p(x), C1, _p(x2), C2, B, H * W, groups, eps, _p(gamma), _p(beta), _p(acc),
                                           int(silu), _p(y), L.dtype_code(x.dtype), _s()), "pp_groupnorm_apply_acc")
    return y


def gemm(x: torch.Tensor, w: torch.Tensor, bias=None, x2=None, res1=None, res2=None, scale: float = 1.0, act: int = 0,
         rowvec=None, rows_per_batch: int = 0, out_f32: bool = False, vt_col0: int = 0, tile: int = 0,
         splitk: int = 0, row_stats: bool = False, ln_stats=None, ln_colsum=None, ln_dim: int = 0,
         ln_eps: float = 1e-5, gn=None, res1_wrap: int = 0, fuse_combine: bool = False):
    """x [M,K1] (+ x2 [M,K2]) bf16, w [N,K1+K2] bf16 -> out [M,N] (or [M,N/2] for GEGLU; (out, vt) when vt_col0).
    w [nb, N, K]: one matrix per batch item of rows_per_batch rows (PPGemmArgs.w_batch_stride); with act =
    L.PP_ACT_SOFTMAX80 bias / ln_colsum may then be [nb, N] as well (vec_batch_stride).
    res1_wrap: res1 holds that many rows only, row m adds res1[m mod res1_wrap] (PPGemmArgs.res1_wrap_rows).
    row_stats=True additionally returns the per-row (sum, sumsq) partials [M, ceil(N/160), 2] fp32;
    ln_stats (that layout) + ln_colsum [N] fp32 apply the folded-LayerNorm correction (see include/pp_hip.h)."""
    lib = L.lib()
    M, K1 = x.shape
    K2 = x2.shape[1] if x2 is not None else 0
    N = w.shape[-2]
    a = L.PPGemmArgs()
    if w.dim() == 3:
        a.w_batch_stride = w.stride(0)
        if bias is not None and bias.dim() == 2:
            a.vec_batch_stride = bias.stride(0)
    a.M, a.N, a.K, a.x_mode = M, N, K1 + K2, L.PP_X_PLAIN
    a.x1, a.x2, a.c1, a.c2, a.ldx1, a.ldx2 = _p(x), _p(x2), K1, K2, x.stride(0), (x2.stride(0) if x2 is not None else 0)
    a.w, a.bias = _p(w), _p(bias)
    a.rowvec = _p(rowvec)
    a.ld_rowvec = rowvec.stride(0) if (rowvec is not None and rowvec.dim() == 2 and rowvec.shape[0] > 1) else 0
    a.rows_per_batch = rows_per_batch
    a.res1, a.ldres1 = _p(res1), (res1.stride(0) if res1 is not None else N)
    a.res1_wrap_rows = res1_wrap
    a.res2, a.ldres2 = _p(res2), (res2.stride(0) if res2 is not None else N)
    a.scale, a.act = scale, act
    n_out = N // 2 if act == L.PP_ACT_GEGLU else (vt_col0 if vt_col0 else N)
    out = torch.empty(M, n_out, dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    a.dtype = L.dtype_code(x.dtype)
    a.out, a.ldo, a.out_f32 = _p(out), n_out, int(out_f32)
    vt = None
    if vt_col0:
        nb = M // rows_per_batch
        vt = torch.zeros(nb, N - vt_col0, rows_per_batch, dtype=x.dtype, device=x.device)
        a.out_vt, a.vt_col0, a.vt_ld = _p(vt), vt_col0, rows_per_batch
    a.tile, a.splitk = tile, splitk
    _set_gn(a, gn, rows_per_batch)
    stats = None
    if row_stats:
        stats = torch.zeros(M, (N + 159) // 160, 2, dtype=torch.float32, device=x.device)
        a.row_stats_out = _p(stats)
    if ln_stats is not None:
        a.ln_stats, a.ln_colsum, a.ln_tiles, a.ln_dim, a.ln_eps = _p(ln_stats), _p(ln_colsum), ln_stats.shape[1], ln_dim, ln_eps
    ws = lib.pp_gemm_workspace_bytes(C.byref(a))
    wsb = torch.empty(max(ws, 4) // 4, dtype=torch.float32, device=x.device) if ws else None
    a.workspace = _p(wsb)
    _attach_combine(a, x.device, fuse_combine)
    L.check(lib.pp_gemm_bf16(C.byref(a), _s()), "pp_gemm_bf16")
    if row_stats:
        return out, stats
    return (out, vt) if vt_col0 else out


def gn_gamma_beta(gamma: torch.Tensor, beta: torch.Tensor) -> torch.Tensor:
    """(gamma, beta) interleaved per channel, fp32 [C][2]: PPGemmArgs.gn_in_gb."""
    return torch.stack([gamma.float(), beta.float()], 1).contiguous()


def conv_gn_supported(x: torch.Tensor, cout: int, x2=None, x3=None, x4=None, groups: int = 32) -> bool:
    """Would conv3x3(..., gn_in=...) run as the fused GroupNorm + SiLU + conv launch for these shapes?"""
    a, _ = _conv_args(x, cout, 1, False, x2, x3, x4)
    a.gn_in_acc, a.gn_in_gb, a.gn_in_groups, a.gn_in_silu, a.gn_in_eps = 1, 1, groups, 1, 1e-5   # (non-null placeholders)
    return bool(L.lib().pp_conv_gn_supported(C.byref(a)))


def conv_halo_routed(x: torch.Tensor, cout: int, x2=None, x3=None, x4=None, stride: int = 1, up: bool = False,
                     tile: int = 0) -> bool:
    """Does a PLAIN conv3x3 of these shapes run on the halo-tile loop (pp_conv_gn_supported() == 2) rather than the tap-major
    implicit GEMM?"""
    a, _ = _conv_args(x, cout, stride, up, x2, x3, x4)
    a.tile = tile
    return L.lib().pp_conv_gn_supported(C.byref(a)) == 2


def _conv_args(x, cout, stride, up, x2, x3, x4):
    B, H, W, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    hv, wv = (2 * H, 2 * W) if up else (H, W)
    ho, wo = (hv - 1) // stride + 1, (wv - 1) // stride + 1
    a = L.PPGemmArgs()
    a.dtype = L.dtype_code(x.dtype)
    C3 = x3.shape[3] if x3 is not None else 0
    C4 = x4.shape[3] if x4 is not None else 0
    a.M, a.N, a.K, a.x_mode = B * ho * wo, cout, 9 * (C1 + C2) + C3 + C4, L.PP_X_CONV3X3
    a.x1, a.x2, a.c1, a.c2 = _p(x), _p(x2), C1, C2
    a.x3, a.x4, a.c3, a.c4 = _p(x3), _p(x4), C3, C4     # 1x1 tail over concat(x3, x4) at the output pixel
    a.batch, a.hin, a.win, a.hout, a.wout, a.stride, a.up = B, H, W, ho, wo, stride, int(up)
    a.rows_per_batch, a.ldres1, a.ldres2, a.ldo, a.scale = ho * wo, cout, cout, cout, 1.0
    return a, (B, ho, wo)


def conv3x3(x: torch.Tensor, w: torch.Tensor, bias=None, stride: int = 1, up: bool = False, x2=None, rowvec=None,
            res1=None, res2=None, scale: float = 1.0, tile: int = 0, splitk: int = 0, gn=None, x3=None, x4=None,
            gn_in=None, gn_next=None, dup: bool = False, gn_dup_mask: int = 0, res1_wrap: int = 0,
            fuse_combine: bool = False):
    """x NHWC bf16 [B,H,W,C1] (+x2 [B,H,W,C2]); w bf16 [Cout, 9*(C1+C2)] (k = (ky*3+kx)*C + c) -> NHWC bf16.
    dup: every output row is stored twice -> out [2B, ...] (PPGemmArgs.out_dup_rows: the CFG twin prefix); gn_dup_mask:
    which of the `gn` subscriptions hold [2B][groups][2] accumulators that receive both halves' sums.
    gn_in = (acc int64 [B][groups][2], gamma_beta fp32 [C1+C2][2], groups, eps): GroupNorm + SiLU of concat(x, x2) fused
    into the loader (x, x2 are then the RAW tensors); raises PPError(PP_ERR_UNSUPPORTED) where conv_gn_supported() is
    False.  gn_next = (gamma, beta, eps, silu, sub): the GroupNorm that consumes the OUTPUT (its statistics subscription
    is gn[sub]) applied by the split-K combine (PPGemmArgs.gn_next_*) -> returns (out, normalised)."""
    lib = L.lib()
    B, H, W, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    cout = w.shape[0]
    hv, wv = (2 * H, 2 * W) if up else (H, W)
    ho, wo = (hv - 1) // stride + 1, (wv - 1) // stride + 1
    out = torch.empty(2 * B if dup else B, ho, wo, cout, dtype=x.dtype, device=x.device)
    a = L.PPGemmArgs()
    a.dtype = L.dtype_code(x.dtype)
    C3 = x3.shape[3] if x3 is not None else 0
    C4 = x4.shape[3] if x4 is not None else 0
    a.M, a.N, a.K, a.x_mode = B * ho * wo, cout, 9 * (C1 + C2) + C3 + C4, L.PP_X_CONV3X3
    a.x1, a.x2, a.c1, a.c2 = _p(x), _p(x2), C1, C2
    a.x3, a.x4, a.c3, a.c4 = _p(x3), _p(x4), C3, C4     # 1x1 tail over concat(x3, x4) at the output pixel
    a.batch, a.hin, a.win, a.hout, a.wout, a.stride, a.up = B, H, W, ho, wo, stride, int(up)
    a.w, a.bias, a.rowvec, a.ld_rowvec, a.rows_per_batch = _p(w), _p(bias), _p(rowvec), 0, ho * wo
    if rowvec is not None and rowvec.dim() == 2 and rowvec.shape[0] > 1:
        a.ld_rowvec = rowvec.stride(0)
    a.res1, a.ldres1, a.res2, a.ldres2 = _p(res1), cout, _p(res2), cout
    a.res1_wrap_rows = res1_wrap
    if dup:
        a.out_dup_rows = a.M
        if gn_dup_mask:
            a.gn_dup_batch, a.gn_dup_mask = B, gn_dup_mask
    a.scale, a.act, a.out, a.ldo = scale, 0, _p(out), cout
    a.tile, a.splitk = tile, splitk
    _set_gn(a, gn, ho * wo)
    ynext = None
    if gn_next is not None:
        g_, b_, eps_, silu_, sub_ = gn_next
        ynext = torch.empty_like(out)
        a.gn_next_out, a.gn_next_gamma, a.gn_next_beta = _p(ynext), _p(g_), _p(b_)
        a.gn_next_eps, a.gn_next_silu, a.gn_next_sub = eps_, int(silu_), sub_
    if gn_in is not None:
        acc, gb, groups, eps = gn_in
        assert gb.dtype == torch.float32 and gb.shape == (C1 + C2, 2) and gb.is_contiguous()
        a.gn_in_acc, a.gn_in_gb, a.gn_in_groups, a.gn_in_silu, a.gn_in_eps = _p(acc), _p(gb), groups, 1, eps
    ws = lib.pp_gemm_workspace_bytes(C.byref(a))
    wsb = torch.empty(max(ws, 4) // 4, dtype=torch.float32, device=x.device) if ws else None
    a.workspace = _p(wsb)
    if gn_next is not None and not lib.pp_gemm_gn_next_ok(C.byref(a), a.gn_next_sub):
        raise L.PPError("pp_gemm_gn_next_ok() = 0 for this launch (PP_ERR_UNSUPPORTED)")
    _attach_combine(a, x.device, fuse_combine)
    L.check(lib.pp_gemm_bf16(C.byref(a), _s()), "pp_gemm_bf16(conv)")
    return (out, ynext) if gn_next is not None else out


def groupnorm(x: torch.Tensor, gamma, beta, eps: float, silu: bool, groups: int = 32, x2=None):
    """x NHWC bf16 [B,H,W,C1] (+x2) -> NHWC bf16 [B,H,W,C1+C2]; gamma/beta fp32."""
    lib = L.lib()
    B, H, W, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    Ct = C1 + C2
    ws = torch.empty(lib.pp_groupnorm_workspace_bytes(B, H * W, Ct) // 4, dtype=torch.float32, device=x.device)
    y = torch.empty(B, H, W, Ct, dtype=x.dtype, device=x.device)
    dt = L.dtype_code(x.dtype)
    L.check(lib.pp_groupnorm_stats(_p(x), C1, _p(x2), C2, B, H * W, groups, _p(ws), dt, _s()), "pp_groupnorm_stats")
    L.check(lib.pp_groupnorm_apply(_p(x), C1, _p(x2), C2, B, H * W, groups, eps, _p(gamma), _p(beta), _p(ws),
                                   int(silu), _p(y), dt, _s()), "pp_groupnorm_apply")
    return y


def layernorm(x: torch.Tensor, gamma, beta, eps: float = 1e-5):
    rows, Cc = x.shape
    y = torch.empty_like(x)
    L.check(L.lib().pp_layernorm(_p(x), rows, Cc, _p(gamma), _p(beta), eps, _p(y), L.dtype_code(x.dtype), _s()),
            "pp_layernorm")
    return y


def transpose_v(v: torch.Tensor, batch: int, nk: int, ldvt: Optional[int] = None):
    """v [batch*nk, cols] (row stride v.stride(0)) -> vt [batch, cols, ldvt]."""
    cols = v.shape[1]
    ldvt = ldvt or (nk + 7) // 8 * 8
    vt = torch.empty(batch, cols, ldvt, dtype=v.dtype, device=v.device)
    L.check(L.lib().pp_transpose_v(_p(v), v.stride(0), batch, nk, cols, _p(vt), ldvt, _s()), "pp_transpose_v")
    return vt


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, heads: int, nq: int, nk: int,
                    causal: bool = False, scale: Optional[float] = None):
    """q [batch*nq, >=heads*64], k / v [batch*nk, ...] row-major bf16 (row strides from the tensors) -> o [batch*nq, heads*64]."""
    d = 64
    o = torch.empty(batch * nq, heads * d, dtype=q.dtype, device=q.device)
    L.check(L.lib().pp_attention_small(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), heads * d,
                                       batch, heads, nq, nk, d, scale if scale is not None else d ** -0.5, int(causal),
                                       L.dtype_code(q.dtype), _s()), "pp_attention_small")
    return o


def softmax_rows(s: torch.Tensor, scale: float = 1.0, dtype=torch.bfloat16):
    """fp32 logits [rows, n] (row stride s.stride(0)) -> 16-bit softmax(scale * s) [rows, n]."""
    rows, n = s.shape
    p = torch.empty(rows, n, dtype=dtype, device=s.device)
    L.check(L.lib().pp_softmax_rows(_p(s), s.stride(0), rows, n, float(scale), _p(p), n, L.dtype_code(dtype), _s()),
            "pp_softmax_rows")
    return p


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, batch: int, heads: int, nq: int, nk: int, d: int,
              scale: Optional[float] = None, variant: int = L.PP_ATTN_AUTO):
    """q [batch*nq, >=heads*d] / k [batch*nk, ...] row-major bf16 (row strides taken from the tensors);
    vt [batch, heads*d, ldvt].  Returns o [batch*nq, heads*d].  `variant` names the kernel (L.PP_ATTN_*): AUTO is what
    the pipelines run; a named kernel raises PP_ERR_UNSUPPORTED on a shape it does not cover."""
    o = torch.empty(batch * nq, heads * d, dtype=q.dtype, device=q.device)
    L.check(L.lib().pp_attention_fwd_variant(_p(q), q.stride(0), _p(k), k.stride(0), _p(vt), vt.stride(1), _p(o),
                                             heads * d, batch, heads, nq, nk, d,
                                             scale if scale is not None else d ** -0.5, L.dtype_code(q.dtype),
                                             int(variant), _s()), "pp_attention_fwd")
    return o


def xattn_fold(k: torch.Tensor, vt: torch.Tensor, batch: int, nctx: int, heads: int, wq: torch.Tensor, wo: torch.Tensor,
               q_colsum=None, q_bias=None, scale: Optional[float] = None, kperm: bool = False):
    """Once per prompt (pp_xattn_fold): k [batch*nctx, >=c], vt [batch, c, ldvt], wq / wo [c, c] ([out][in]) ->
    (gt [batch, heads*80, c], gcs [batch, heads*80] fp32, gbias [batch, heads*80] fp32, ht [batch, c, heads*80])."""
    c = wq.shape[0]
    dev, dt = k.device, k.dtype
    gt = torch.empty(batch, heads * 80, c, dtype=dt, device=dev)
    ht = torch.empty(batch, c, heads * 80, dtype=dt, device=dev)
    gcs = torch.empty(batch, heads * 80, dtype=torch.float32, device=dev)
    gb = torch.empty(batch, heads * 80, dtype=torch.float32, device=dev)
    L.check(L.lib().pp_xattn_fold(_p(k), k.stride(0), _p(vt), vt.stride(1), batch, nctx, heads, c, _p(wq), _p(q_colsum),
                                  _p(q_bias), _p(wo), scale if scale is not None else (c // heads) ** -0.5, _p(gt),
                                  _p(gcs), _p(gb), _p(ht), int(kperm), L.dtype_code(dt), _s()), "pp_xattn_fold")
    return gt, gcs, gb, ht


def xattn_block(x: torch.Tensor, folded, bias_o=None, res=None, ln_stats=None, ln_eps: float = 1e-5,
                rows_per_batch: int = 0, row_stats: bool = False, twin: bool = False, pre_w=None, pre_b=None,
                ln_fold: bool = False):
    """out = softmax_per_head(LNfold(x) gt^T) ht^T + bias_o + res  (pp_xattn_block); x [M, c], folded = xattn_fold(...).
    twin: x / res / ln_stats hold ONE half of a CFG pair (M rows) whose other half is identical; the output has 2 M rows
    (src_wrap_rows = M) and `folded` is per batch item of the full batch."""
    gt, gcs, gb, ht = folded
    wrap = x.shape[0] if twin else 0
    M, c = x.shape[0] * (2 if twin else 1), x.shape[1]
    out = torch.empty(M, c, dtype=x.dtype, device=x.device)
    st = torch.zeros(M, c // 160, 2, dtype=torch.float32, device=x.device) if row_stats else None
    # pre_w / pre_b: the Linear in front (h = x pre_w^T + pre_b + res) in the same launch; `folded` must come from
    # xattn_fold(kperm=True); ln_fold = a LayerNorm of h is folded into `folded` (row moments are computed in the kernel)
    tiles = ln_stats.shape[1] if ln_stats is not None else (c // 160 if (pre_w is not None and ln_fold) else 0)
    L.check(L.lib().pp_xattn_block(_p(x), x.stride(0), _p(res), res.stride(0) if res is not None else 0, _p(ln_stats),
                                   tiles, ln_eps, _p(gt), _p(gcs), _p(gb), _p(ht),
                                   _p(bias_o), _p(out), c, _p(st), M, c, rows_per_batch or M, wrap, _p(pre_w), _p(pre_b),
                                   L.dtype_code(x.dtype), _s()), "pp_xattn_block")
    return (out, st) if row_stats else out


def ff_fused(hs: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2kp: torch.Tensor, bias2=None, cs1=None, ln_stats=None,
             ln_eps: float = 1e-5, res1=None, res2=None, res1_wrap: int = 0, rows_per_batch: int = 0, gn=None,
             w2_kperm: bool = True):
    """pp_ff_fused: out = [h (.) gelu(g) | hs] w2kp^T + bias2 + res1 + res2 with h | g = FF1(LayerNorm-folded hs), one launch.
    hs [M, 320]; w1 [2560, 320] (GEGLU-interleaved rows, gamma folded); w2kp [320, 1600]: w2_kperm=True -> hidden index
    permuted (engine._kperm_geglu; the 4-wave kernel that chains the GEGLU in registers), False -> natural order (the 8-wave
    kernel, activations exchanged through LDS); ln_stats [M, 2, 2] row moments of hs or None; gn: as ops.gemm takes them."""
    M, Cc = hs.shape
    a = L.PPGemmArgs()
    a.M, a.N, a.K, a.x_mode = M, Cc, 5 * Cc, L.PP_X_PLAIN
    a.x1, a.x2, a.c1, a.c2, a.ldx1, a.ldx2 = _p(hs), _p(hs), 4 * Cc, Cc, 4 * Cc, hs.stride(0)
    a.w, a.bias = _p(w2kp), _p(bias2)
    a.res1, a.ldres1, a.res1_wrap_rows = _p(res1), (res1.stride(0) if res1 is not None else Cc), res1_wrap
    a.res2, a.ldres2 = _p(res2), (res2.stride(0) if res2 is not None else Cc)
    a.scale, a.act = 1.0, 0
    out = torch.empty(M, Cc, dtype=hs.dtype, device=hs.device)
    a.out, a.ldo, a.dtype = _p(out), Cc, L.dtype_code(hs.dtype)
    a.rows_per_batch = rows_per_batch
    _set_gn(a, gn, rows_per_batch)
    L.check(L.lib().pp_ff_fused(C.byref(a), _p(w1), _p(b1), _p(cs1), _p(ln_stats),
                                ln_stats.shape[1] if ln_stats is not None else 0, ln_eps, int(w2_kperm), _s()), "pp_ff_fused")
    return out


def gn_conv3x3_smallcout(x, acc, gamma, beta, eps: float, w, bias, groups: int = 32):
    """conv_norm_out + SiLU + conv_out in one launch: x NHWC 16-bit [B,H,W,Cin], acc int64 [B,groups,2] (the producers'
    GroupNorm accumulators), w [4, 9*Cin] -> fp32 NCHW [B,4,H,W]."""
    B, H, W, Cin = x.shape
    out = torch.empty(B, w.shape[0], H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib().pp_gn_conv3x3_smallcout(_p(x), B, H, W, Cin, groups, eps, _p(gamma), _p(beta), _p(acc), _p(w), _p(bias),
                                            w.shape[0], _p(out), L.dtype_code(x.dtype), _s()), "pp_gn_conv3x3_smallcout")
    return out


def conv3x3_direct(x, w, bias, stride: int = 1, silu: bool = False, add=None):
    """x NHWC bf16 [B,H,W,Cin]; w bf16 [3,3,Cin,Cout]."""
    B, H, W, Cin = x.shape
    cout = w.shape[3]
    ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty(B, ho, wo, cout, dtype=x.dtype, device=x.device)
    L.check(L.lib().pp_conv3x3_direct(_p(x), B, H, W, Cin, _p(w), _p(bias), cout, stride, int(silu), _p(add), _p(out),
                                      L.dtype_code(x.dtype), _s()), "pp_conv3x3_direct")
    return out


def conv3x3_smallcout(x, w, bias):
    """x NHWC bf16 [B,H,W,Cin]; w bf16 [4, 9*Cin] -> fp32 NCHW [B,4,H,W]."""
    B, H, W, Cin = x.shape
    out = torch.empty(B, w.shape[0], H, W, dtype=torch.float32, device=x.device)
    L.check(L.lib().pp_conv3x3_smallcout(_p(x), B, H, W, Cin, _p(w), _p(bias), w.shape[0], _p(out),
                                         L.dtype_code(x.dtype), _s()), "pp_conv3x3_smallcout")
    return out


def nchw_to_nhwc(src: torch.Tensor, batch: Optional[int] = None, ldc: Optional[int] = None, c0: int = 0, dst=None,
                 dtype=torch.bfloat16):
    B, Cc, H, W = src.shape
    nb = batch or B
    ldc = ldc or Cc
    if dst is None:
        dst = torch.zeros(nb, H, W, ldc, dtype=dtype, device=src.device)
    src = src.contiguous()
    L.check(L.lib().pp_nchw_to_nhwc(_p(src), _DT[src.dtype], nb, Cc, H * W, B if nb != B else 0, _p(dst), ldc, c0,
                                    L.dtype_code(dst.dtype), _s()), "pp_nchw_to_nhwc")
    return dst


def nhwc_to_nchw(src: torch.Tensor, dtype=torch.float32):
    B, H, W, Cc = src.shape
    dst = torch.empty(B, Cc, H, W, dtype=dtype, device=src.device)
    L.check(L.lib().pp_nhwc_to_nchw(_p(src), B, Cc, H * W, _p(dst), _DT[dtype], L.dtype_code(src.dtype), _s()),
            "pp_nhwc_to_nchw")
    return dst


def add(a: torch.Tensor, b: torch.Tensor):
    out = torch.empty_like(a)
    L.check(L.lib().pp_add_bf16(_p(a), _p(b), _p(out), a.numel(), L.dtype_code(a.dtype), _s()), "pp_add_bf16")
    return out


def timestep_embedding(t: torch.Tensor, rows: int, dim: int):
    out = torch.empty(rows, dim, dtype=torch.float32, device=t.device)
    L.check(L.lib().pp_timestep_embedding(_p(t), rows, dim, _p(out), _s()), "pp_timestep_embedding")
    return out


def linear_skinny(x: torch.Tensor, w: torch.Tensor, bias=None, act_in: int = 0, act_out: int = 0):
    rows, K = x.shape
    N = w.shape[0]
    out = torch.empty(rows, N, dtype=torch.float32, device=x.device)
    L.check(L.lib().pp_linear_skinny(_p(x), rows, K, _p(w), _p(bias), N, _p(out), N, act_in, act_out,
                                     L.dtype_code(w.dtype), _s()), "pp_linear_skinny")
    return out


def tfront(x: torch.Tensor, acc: torch.Tensor, gamma, beta, w1: torch.Tensor, b1, w2p: torch.Tensor, cs2, b2, rows_per_batch: int,
           gn_eps: float = 1e-6, ln_eps: float = 1e-5, groups: int = 32, q_scale: float = 1.0):
    """pp_tfront: hs = proj_in(GroupNorm(x)); q | k | v = QKV(LayerNorm1(hs)), V transposed.  x [M, 320] raw rows, acc the
    int64 GroupNorm accumulators of x, w2p the gamma-folded QKV weight with engine._kperm applied.  -> (hs, qk [M, 640], vt).
    q_scale multiplies the Q third before its rounding (head_dim^-0.5 * log2 e for L.PP_ATTN_PIPE_LOG2)."""
    M, c = x.shape
    nb = M // rows_per_batch
    hs = torch.empty(M, c, dtype=x.dtype, device=x.device)
    qk = torch.empty(M, 2 * c, dtype=x.dtype, device=x.device)
    vt = torch.empty(nb, c, rows_per_batch, dtype=x.dtype, device=x.device)
    L.check(L.lib().pp_tfront(_p(x), x.stride(0), _p(acc), _p(gamma), _p(beta), gn_eps, groups, _p(w1), _p(b1), _p(w2p), _p(cs2),
                              _p(b2), ln_eps, _p(hs), c, _p(qk), 2 * c, _p(vt), rows_per_batch, M, c, rows_per_batch,
                              float(q_scale), L.dtype_code(x.dtype), _s()), "pp_tfront")
    return hs, qk, vt
