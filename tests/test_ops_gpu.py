"""-m gpu: every HIP kernel, called through the C ABI, against a plain PyTorch fp32 reference of the same op.

Inputs are rounded to bf16 first so the reference sees exactly the bits the kernel sees; the reference itself runs in
fp32 (on the GPU via torch, which is only the checker here).  Tolerances are stated per test: bf16 output rounding is
2^-9 relative, fp32 accumulation order differences are ~1e-6 * sqrt(K).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd import ops  # noqa: E402

DEV = "cuda"


def bf(t):
    return t.to(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def check(out, ref, atol, rtol, what=""):
    out, ref = out.float(), ref.float()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    err = (out - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} off; max abs err {float(err.max()):.4g} "
                           f"(ref max {float(ref.abs().max()):.4g}) at {np.unravel_index(int(err.argmax()), err.shape)}")


# ------------------------------------------------------------------------------------------------ GEMM
PP_TILES = [53, 44, 54]      # ping-pong 8-wave tiles (256x160x3, 128x160x3, 128x160x4)
ALL_TILES = [1, 2, 3, 21, 31, 22, 32, 42, 23, 33, 24] + PP_TILES


@pytest.mark.parametrize("tile", ALL_TILES)
@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (616, 640, 768), (1024, 960, 320), (2048, 320, 1280)])
def test_gemm_plain(tile, M, N, K):
    x, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    res = bf(rnd(M, N, seed=4))
    out = ops.gemm(x, w, bias=bias, res1=res, tile=tile, splitk=1)
    ref = x.float() @ w.float().t() + bias + res.float()
    check(out, ref, 2e-2, 1e-2, f"gemm tile{tile}")


@pytest.mark.parametrize("tile", [33, 31] + PP_TILES)
@pytest.mark.parametrize("K", [64, 128, 192])
def test_gemm_short_k_pipeline_edges(tile, K):
    """K of one to three 64-deep tiles: fewer tiles than pipeline stages (the prologue's look-ahead refills are all
    past the end) and, for the ping-pong tiles, a loop of 1-3 iterations around the group stagger; ragged M and N."""
    M, N = 300, 200
    x, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = rnd(N, seed=3)
    out = ops.gemm(x, w, bias=bias, tile=tile, splitk=1)
    check(out, x.float() @ w.float().t() + bias, 2e-2, 1e-2, f"gemm K={K} tile{tile}")
    B, H, C = 1, 8, 64                                   # conv with 9 single-tile taps + a one-tile 1x1 tail
    xc, x3 = bf(rnd(B, H, H, C, seed=4)), bf(rnd(B, H, H, 64, seed=5))
    wc = bf(rnd(160, 9 * C + 64, seed=6, scale=(9 * C + 64) ** -0.5))
    oc = ops.conv3x3(xc, wc, None, x3=x3, tile=tile, splitk=1)
    ref = conv_ref(xc, wc[:, :9 * C].contiguous(), None) + x3.float() @ wc[:, 9 * C:].float().t()
    check(oc, ref, 2e-2, 1e-2, f"conv + tail, single-tile segments, tile{tile}")


def test_gemm_transpose_detect():
    """A = I check with asymmetric W (guide rule: a symmetric B hides a row<->col swap)."""
    K = 320
    x = bf(torch.eye(K, device=DEV))
    w = bf(rnd(640, K, seed=5))
    out = ops.gemm(x, w, tile=1, splitk=1)
    check(out, w.float().t(), 0, 0, "gemm identity")


@pytest.mark.parametrize("tile", [2, 22, 42, 31, 53, 54])
@pytest.mark.parametrize("splitk", [2, 4, 3])
def test_gemm_splitk(splitk, tile):
    M, N, K = 512, 1280, 2560
    x, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, res = rnd(N, seed=3), bf(rnd(M, N, seed=4))
    out = ops.gemm(x, w, bias=bias, res1=res, res2=res, scale=0.5, tile=tile, splitk=splitk)
    ref = (x.float() @ w.float().t() + bias) * 0.5 + 2 * res.float()
    check(out, ref, 2e-2, 1e-2, "gemm splitk")


def test_gemm_concat_and_rowvec():
    M, K1, K2, N = 512, 640, 320, 320
    x1, x2 = bf(rnd(M, K1, seed=1)), bf(rnd(M, K2, seed=2))
    w = bf(rnd(N, K1 + K2, seed=3, scale=(K1 + K2) ** -0.5))
    rv = rnd(2, N, seed=4)
    out = ops.gemm(x1, w, x2=x2, rowvec=rv, rows_per_batch=256, out_f32=True)
    ref = torch.cat([x1, x2], 1).float() @ w.float().t() + rv.repeat_interleave(256, 0)
    check(out, ref, 2e-3, 1e-3, "gemm concat+rowvec")


def test_gemm_geglu():
    from powerpaint_amd.engine import _geglu_interleave
    M, C = 512, 320
    x = bf(rnd(M, C, seed=1))
    w = bf(rnd(8 * C, C, seed=2, scale=C ** -0.5))
    b = rnd(8 * C, seed=3)
    out = ops.gemm(x, _geglu_interleave(w).contiguous(), bias=_geglu_interleave(b).contiguous(), act=L.PP_ACT_GEGLU)
    y = x.float() @ w.float().t() + b
    h, g = y.chunk(2, -1)
    check(out, h * F.gelu(g), 2e-2, 1e-2, "gemm geglu")


def test_gemm_vt_epilogue():
    B, hw, C = 2, 256, 320
    x = bf(rnd(B * hw, C, seed=1))
    w = bf(rnd(3 * C, C, seed=2, scale=C ** -0.5))
    out, vt = ops.gemm(x, w, vt_col0=2 * C, rows_per_batch=hw)
    y = x.float() @ w.float().t()
    check(out, y[:, :2 * C], 2e-2, 1e-2, "qk part")
    check(vt, y[:, 2 * C:].reshape(B, hw, C).transpose(1, 2), 2e-2, 1e-2, "v^T part")


V2_TILES = [21, 31, 22, 32, 42, 23, 33, 24] + PP_TILES


@pytest.mark.parametrize("tile", V2_TILES + [0])
@pytest.mark.parametrize("M,N", [(520, 320), (1024, 640), (192, 1280)])
def test_gemm_row_stats(tile, M, N):
    """Producer side of the folded LayerNorm: per-row (sum, sumsq) of the STORED bf16 output, per 160-column tile."""
    K = 320
    x, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, res = rnd(N, seed=3), bf(rnd(M, N, seed=4))
    out, st = ops.gemm(x, w, bias=bias, res1=res, tile=tile, row_stats=True)
    o = out.float()
    tiles = N // 160
    ref = torch.stack([o.reshape(M, tiles, 160).sum(-1), (o * o).reshape(M, tiles, 160).sum(-1)], -1)
    check(st, ref, 1e-3, 1e-5, "row moments")


@pytest.mark.parametrize("tile", [1, 2, 3] + V2_TILES + [0])
@pytest.mark.parametrize("kind", ["plain", "geglu", "vt"])
def test_gemm_folded_layernorm(tile, kind):
    """Consumer side: Linear(LayerNorm(h)) == rstd * (h (g.W)^T - mean * colsum) + W b  with moments from the producer."""
    from powerpaint_amd.engine import _geglu_interleave
    B, hw, C = 2, 264, 320
    M = B * hw
    N = {"plain": C, "geglu": 8 * C, "vt": 3 * C}[kind]
    h = bf(rnd(M, C, seed=1, scale=3.0) + 0.7)
    g, b = rnd(C, seed=5) * 0.3 + 1.0, rnd(C, seed=6) * 0.2
    w = rnd(N, C, seed=2, scale=C ** -0.5)
    bias = rnd(N, seed=3) if kind != "vt" else None
    true = F.layer_norm(h.float(), (C,), g, b, 1e-5) @ w.t()          # what the network means (fp32 weights)
    if bias is not None:
        true = true + bias
    hf = h.float()
    tiles = C // 160
    st = torch.stack([hf.reshape(M, tiles, 160).sum(-1), (hf * hf).reshape(M, tiles, 160).sum(-1)], -1).contiguous()
    wf = bf(w * g[None, :])
    cs = wf.float().sum(1)
    t = w @ b if bias is None else w @ b + bias
    # what the kernel is asked to compute, with the bf16 weights it actually multiplies
    mean = hf.mean(-1, keepdim=True)
    rstd = torch.rsqrt((hf * hf).mean(-1, keepdim=True) - mean * mean + 1e-5)
    ref = rstd * (hf @ wf.float().t() - mean * cs) + t
    check(ref, true, 6e-2, 2e-2, "fold algebra vs LayerNorm -> Linear")
    kw = dict(ln_stats=st, ln_colsum=cs.contiguous(), ln_dim=C, tile=tile)
    if kind == "plain":
        check(ops.gemm(h, wf, bias=t.contiguous(), **kw), ref, 3e-2, 1e-2, "folded LN")
    elif kind == "geglu":
        kw["ln_colsum"] = _geglu_interleave(cs).contiguous()
        out = ops.gemm(h, _geglu_interleave(wf).contiguous(), bias=_geglu_interleave(t).contiguous(),
                       act=L.PP_ACT_GEGLU, **kw)
        a_, g_ = ref.chunk(2, -1)
        check(out, a_ * F.gelu(g_), 3e-2, 1e-2, "folded LN + GEGLU")
    else:
        out, vt = ops.gemm(h, wf, bias=t.contiguous(), vt_col0=2 * C, rows_per_batch=hw, **kw)
        check(out, ref[:, :2 * C], 3e-2, 1e-2, "folded LN qk")
        check(vt, ref[:, 2 * C:].reshape(B, hw, C).transpose(1, 2), 3e-2, 1e-2, "folded LN v^T")


# ------------------------------------------------------------------------------------------------ conv3x3 (implicit GEMM)
def conv_ref(x_nhwc, w_igemm, bias, stride=1, up=False):
    B, H, W, Cin = x_nhwc.shape
    cout = w_igemm.shape[0]
    w = w_igemm.float().reshape(cout, 3, 3, Cin).permute(0, 3, 1, 2)
    x = x_nhwc.float().permute(0, 3, 1, 2)
    if up:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    y = F.conv2d(x, w, bias, stride=stride, padding=1)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("tile", ALL_TILES)
@pytest.mark.parametrize("stride,up", [(1, False), (2, False), (1, True)])
def test_conv3x3(tile, stride, up):
    B, H, W, Cin, Cout = 2, 16, 16, 320, 320
    x = bf(rnd(B, H, W, Cin, seed=1))
    w = bf(rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5))
    bias = rnd(Cout, seed=3)
    out = ops.conv3x3(x, w, bias, stride=stride, up=up, tile=tile, splitk=1)
    check(out, conv_ref(x, w, bias, stride, up), 2e-2, 1e-2, f"conv s{stride} up{up} tile{tile}")


def test_conv3x3_concat_temb_res_splitk():
    B, H, W, C1, C2, Cout = 2, 8, 8, 640, 320, 640
    x1, x2 = bf(rnd(B, H, W, C1, seed=1)), bf(rnd(B, H, W, C2, seed=2))
    w = bf(rnd(Cout, 9 * (C1 + C2), seed=3, scale=(9 * (C1 + C2)) ** -0.5))
    bias, temb = rnd(Cout, seed=4), rnd(1, Cout, seed=5)
    r1, r2 = bf(rnd(B, H, W, Cout, seed=6)), bf(rnd(B, H, W, Cout, seed=7))
    ref = conv_ref(torch.cat([x1, x2], -1), w, bias) + temb.view(1, 1, 1, -1) + r1.float() + r2.float()
    for tile in (2, 32, 31, 33, 53, 44):
        for sk in (1, 4):
            out = ops.conv3x3(x1, w, bias, x2=x2, rowvec=temb, res1=r1, res2=r2, tile=tile, splitk=sk)
            check(out, ref, 3e-2, 1e-2, f"conv concat tile{tile} splitk{sk}")


def test_conv3x3_halo_exact():
    """All-ones input/weights: every output equals the number of valid taps * Cin (catches halo / tap-order bugs)."""
    B, H, W, Cin, Cout = 1, 8, 8, 64, 320
    x = torch.ones(B, H, W, Cin, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(Cout, 9 * Cin, dtype=torch.bfloat16, device=DEV) / 64
    cnt = F.conv2d(torch.ones(1, 1, H, W, device=DEV), torch.ones(1, 1, 3, 3, device=DEV), padding=1)[0, 0]
    for tile in (2, 22, 21, 33, 53, 54):
        out = ops.conv3x3(x, w, None, tile=tile, splitk=1).float()
        assert torch.equal(out[0, :, :, 0], cnt), (tile, out[0, :, :, 0], cnt)
        assert torch.equal(out[0, :, :, 319], cnt)


@pytest.mark.parametrize("tile,splitk", [(0, 0), (21, 1), (22, 2), (33, 1), (24, 1), (31, 4), (53, 1), (53, 3), (54, 2), (44, 1)])
@pytest.mark.parametrize("two", [False, True])
def test_conv3x3_with_1x1_tail(tile, splitk, two):
    """conv2(h) + conv_shortcut(concat(x, skip)) as ONE implicit GEMM: K = 9 C_h + C_x (+ C_skip)."""
    B, H, Ch, Cx, Cs, Cout = 2, 16, 320, 640, 320, 320
    h = bf(rnd(B, H, H, Ch, seed=1))
    x = bf(rnd(B, H, H, Cx, seed=2))
    sk = bf(rnd(B, H, H, Cs, seed=3)) if two else None
    w2 = bf(rnd(Cout, 9 * Ch, seed=4, scale=(9 * Ch) ** -0.5))
    cin = Cx + (Cs if two else 0)
    wsc = bf(rnd(Cout, cin, seed=5, scale=cin ** -0.5))
    bias = rnd(Cout, seed=6)
    out = ops.conv3x3(h, torch.cat([w2, wsc], 1).contiguous(), bias, x3=x, x4=sk, tile=tile, splitk=splitk)
    xin = torch.cat([x, sk], -1) if two else x
    ref = conv_ref(h, w2, bias) + xin.float() @ wsc.float().t()
    check(out, ref, 3e-2, 1e-2, "conv + 1x1 tail")


# ------------------------------------------------------------------------------------------------ GroupNorm statistics in the epilogue
def _gn_ref(t, cg, c0, groups):
    """(sum, sumsq) per (batch, group) of tensor t [B, hw, C] placed at channel offset c0 of a `groups`-group norm."""
    B, hw, C = t.shape
    out = torch.zeros(B, groups, 2, dtype=torch.float64, device=t.device)
    tf = t.double()
    for c in range(C):
        g = (c0 + c) // cg
        out[:, g, 0] += tf[:, :, c].sum(1)
        out[:, g, 1] += (tf[:, :, c] ** 2).sum(1)
    return out


@pytest.mark.parametrize("tile,splitk", [(0, 0), (21, 1), (22, 1), (24, 1), (33, 1), (31, 4), (32, 2), (33, 8), (53, 1), (54, 1), (53, 4)])
def test_conv_epilogue_groupnorm_stats(tile, splitk):
    """The conv epilogue (or the split-K combine) accumulates the consumer GroupNorms' (sum, sumsq): two consumers with
    different groupings (own 32-group norm; a 1920-channel concat norm where this tensor sits at channel offset 1280,
    so 60-channel groups straddle the 160-column tiles), bit-reproducible, and the apply kernel consumes them."""
    B, H, Cin, Cout = 2, 16, 320, 640
    x = bf(rnd(B, H, H, Cin, seed=1))
    w = bf(rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5))
    bias, res = rnd(Cout, seed=3), bf(rnd(B, H, H, Cout, seed=4))
    acc1 = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    acc2 = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    gn = [(acc1, Cout // 32, 0, 32), (acc2, 1920 // 32, 1280, 32)]
    out = ops.conv3x3(x, w, bias, res1=res, tile=tile, splitk=splitk, gn=gn)
    plain = ops.conv3x3(x, w, bias, res1=res, tile=tile, splitk=splitk)
    assert torch.equal(out, plain)
    o = out.float().reshape(B, H * H, Cout)
    for acc, cg, c0 in ((acc1, 20, 0), (acc2, 60, 1280)):
        ref = _gn_ref(o, cg, c0, 32)
        got = torch.stack([acc[..., 0].double() / 2 ** 24, acc[..., 1].double() / 2 ** 20], -1)
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-2), (cg, (got - ref).abs().max())
    again1 = torch.zeros_like(acc1)
    ops.conv3x3(x, w, bias, res1=res, tile=tile, splitk=splitk, gn=[(again1, 20, 0, 32)])
    assert torch.equal(again1, acc1)                       # integer accumulation: order-independent
    g, b = rnd(Cout, seed=5), rnd(Cout, seed=6)
    y = ops.groupnorm_apply_acc(out, acc1, g, b, 1e-5, True)
    ref = F.silu(F.group_norm(out.float().permute(0, 3, 1, 2), 32, g, b, 1e-5)).permute(0, 2, 3, 1)
    check(y, ref, 2e-2, 1e-2, "groupnorm from accumulated statistics")


def test_gemm_epilogue_groupnorm_stats_plain():
    B, hw, K, N = 2, 256, 1600, 320
    x, w = bf(rnd(B * hw, K, seed=1)), bf(rnd(N, K, seed=2, scale=K ** -0.5))
    acc = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    out = ops.gemm(x, w, rows_per_batch=hw, gn=[(acc, 10, 0, 32)])
    ref = _gn_ref(out.float().reshape(B, hw, N), 10, 0, 32)
    got = torch.stack([acc[..., 0].double() / 2 ** 24, acc[..., 1].double() / 2 ** 20], -1)
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-2)


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("C1,C2,H", [(320, 0, 16), (640, 320, 8), (1280, 1280, 8), (1920, 0, 4), (320, 0, 64)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm(C1, C2, H, silu):
    B = 2
    x1 = bf(rnd(B, H, H, C1, seed=1) * 2 + 0.5)
    x2 = bf(rnd(B, H, H, C2, seed=2)) if C2 else None
    C = C1 + C2
    g, b = rnd(C, seed=3), rnd(C, seed=4)
    eps = 1e-5 if silu else 1e-6
    out = ops.groupnorm(x1, g, b, eps, silu, x2=x2)
    x = torch.cat([x1, x2], -1) if C2 else x1
    ref = F.group_norm(x.float().permute(0, 3, 1, 2), 32, g, b, eps)
    if silu:
        ref = F.silu(ref)
    check(out, ref.permute(0, 2, 3, 1), 2e-2, 1e-2, "groupnorm")


@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layernorm(C):
    x = bf(rnd(1000, C, seed=1) * 3 + 1)
    g, b = rnd(C, seed=2), rnd(C, seed=3)
    check(ops.layernorm(x, g, b), F.layer_norm(x.float(), (C,), g, b, 1e-5), 2e-2, 1e-2, "layernorm")


# ------------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("d,nq,nk", [(40, 256, 256), (80, 256, 256), (160, 64, 64), (40, 1024, 77), (80, 200, 77),
                                     (160, 256, 77), (40, 4096, 4096), (40, 200, 192), (40, 128, 128), (40, 64, 64),
                                     (40, 1000, 960), (80, 1024, 1024), (80, 1000, 960),
                                     # key-tail paths of attn_fwd_kernel: a last tile with <= 32 live keys computes only
                                     # its first 32-key block (77 = 64 + 13, 96 = 64 + exactly 32, 16 / 32: first tile),
                                     # 33 / 97 / 100 keep the masked full tile
                                     (40, 96, 96), (80, 64, 16), (160, 64, 32), (40, 64, 33), (80, 128, 97),
                                     (40, 128, 100)])
def test_attention(d, nq, nk):
    B, Hh = 2, 8
    C = Hh * d
    q, k, v = bf(rnd(B * nq, C, seed=1)), bf(rnd(B * nk, C, seed=2)), bf(rnd(B * nk, C, seed=3))
    vt = ops.transpose_v(v, B, nk)
    out = ops.attention(q, k, vt, B, Hh, nq, nk, d)
    qh = q.float().view(B, nq, Hh, d).transpose(1, 2)
    kh = k.float().view(B, nk, Hh, d).transpose(1, 2)
    vh = v.float().view(B, nk, Hh, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * nq, C)
    check(out, ref, 2e-2, 2e-2, f"attention d{d}")


def test_attention_strided_qkv_and_spike():
    """q/k taken as column slices of a fused [M, 2C] buffer; one huge score forces the online-softmax rescale path."""
    B, Hh, d, n = 1, 8, 40, 512
    C = Hh * d
    qk = bf(rnd(B * n, 2 * C, seed=1))
    qk[300, C:C + d] = 30.0          # key 300 of head 0 dominates for queries aligned with it
    qk[5, :d] = 4.0
    v = bf(rnd(B * n, C, seed=3))
    vt = ops.transpose_v(v, B, n)
    out = ops.attention(qk[:, :C], qk[:, C:], vt, B, Hh, n, n, d)
    qh = qk[:, :C].float().view(B, n, Hh, d).transpose(1, 2)
    kh = qk[:, C:].float().view(B, n, Hh, d).transpose(1, 2)
    vh = v.float().view(B, n, Hh, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * n, C)
    check(out, ref, 2e-2, 2e-2, "attention strided+spike")


LOG2E = 1.4426950408889634


def _log2_ref(qs, k, v, B, Hh, nq, nk, d):
    """softmax over exp2(q' . k) in fp32: what PP_ATTN_PIPE_LOG2 computes from a pre-multiplied q'."""
    qh = qs.float().view(B, nq, Hh, d).transpose(1, 2)
    kh = k.float().view(B, nk, Hh, d).transpose(1, 2)
    vh = v.float().view(B, nk, Hh, d).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(2, 3)) * math.log(2.0)
    return torch.matmul(torch.softmax(s, dim=-1), vh).transpose(1, 2).reshape(B * nq, Hh * d)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,nq,nk", [(2, 256, 256),       # 32-queries-per-wave kernel, the minimum of four 64-key tiles
                                     (1, 200, 320),       # ragged last query block, odd tile count
                                     (2, 1024, 1024),
                                     (8, 2048, 1024),     # 64-queries-per-wave kernel (batch * heads * ceil(nq / 256) >= 512)
                                     (4, 4096, 4096)])    # the twin-prefix launch of the headline step
def test_attention_log2_form(B, nq, nk, dtype):
    """PP_ATTN_PIPE_LOG2 (round 5): q arrives as Q * d^-0.5 * log2(e), the running reference enters the QK^T MFMA as its
    initial accumulator.  Against fp32 torch on the same 16-bit operands, and against the plain kernel on the unscaled Q
    (the two differ by Q's second rounding here; in the product pp_tfront rounds Q once, after the multiplication)."""
    Hh, d = 8, 40
    C = Hh * d
    q, k, v = rnd(B * nq, C, seed=1), rnd(B * nk, C, seed=2).to(dtype), rnd(B * nk, C, seed=3).to(dtype)
    qs = (q * (d ** -0.5 * LOG2E)).to(dtype)
    vt = ops.transpose_v(v, B, nk)
    assert L.lib().pp_attention_log2_ok(nq, nk, d) == 1
    out = ops.attention(qs, k, vt, B, Hh, nq, nk, d, variant=L.PP_ATTN_PIPE_LOG2)
    check(out, _log2_ref(qs, k, v, B, Hh, nq, nk, d), 2e-2 if dtype == torch.bfloat16 else 3e-3, 2e-2, "attention LOG2 vs fp32")
    plain = ops.attention(q.to(dtype), k, vt, B, Hh, nq, nk, d)
    check(out, plain, 3e-2 if dtype == torch.bfloat16 else 6e-3, 2e-2, "attention LOG2 vs the plain kernel")


@pytest.mark.parametrize("B", [1, 8])
@pytest.mark.parametrize("case", ["spike", "cold_start", "ramp"])
def test_attention_log2_reference_updates(case, B):
    """The rare paths of the LOG2 form: a late dominant key (the reference jumps, scores already baked with the old one are
    fixed up), a first tile far BELOW zero for every query (the first tile always replaces the initial reference 0), and
    scores that climb through every tile (an update every few stages)."""
    Hh, d, n = 8, 40, 1024
    C = Hh * d
    q, k, v = rnd(B * n, C, seed=1), rnd(B * n, C, seed=2), rnd(B * n, C, seed=3)
    if case == "spike":
        k[300, :d] = 30.0
        q[5, :d] = 4.0
        k[n - 1, d:2 * d] = -25.0
        q[n - 7, d:2 * d] = -5.0
    elif case == "cold_start":            # every score of head 0 strongly negative, those of the first tile most of all
        q[:, :d] = q[:, :d].abs() + 1.0
        k[:, :d] = -(k[:, :d].abs() + 1.0)
        k[:64, :d] *= 6.0
    else:                                  # key t scores ~ t / 16 for every query of head 1
        q[:, d:2 * d] = 1.0
        k[:, d:2 * d] = (torch.arange(B * n, device=DEV) % n)[:, None].float() / 16.0 / d * (d ** 0.5) / LOG2E
    qs = (q * (d ** -0.5 * LOG2E)).to(torch.bfloat16)
    k, v = bf(k), bf(v)
    vt = ops.transpose_v(v, B, n)
    out = ops.attention(qs, k, vt, B, Hh, n, n, d, variant=L.PP_ATTN_PIPE_LOG2)
    check(out, _log2_ref(qs, k, v, B, Hh, n, n, d), 2e-2, 2e-2, f"attention LOG2 {case}")


def test_attention_log2_refuses_what_it_does_not_cover():
    B, Hh, n = 1, 8, 256
    for d, nk in ((80, 256), (40, 192), (40, 128)):
        assert L.lib().pp_attention_log2_ok(n, nk, d) == 0
        q = torch.zeros(B * n, Hh * d, dtype=torch.bfloat16, device=DEV)
        kk = torch.zeros(B * nk, Hh * d, dtype=torch.bfloat16, device=DEV)
        with pytest.raises(L.PPError, match="UNSUPPORTED"):
            ops.attention(q, kk, ops.transpose_v(kk, B, nk), B, Hh, n, nk, d, variant=L.PP_ATTN_PIPE_LOG2)


def test_transpose_v():
    B, nk, cols = 2, 77, 320
    v = bf(rnd(B * nk, cols, seed=1))
    vt = ops.transpose_v(v, B, nk)
    assert vt.shape == (B, cols, 80)
    assert torch.equal(vt[:, :, :nk], v.view(B, nk, cols).transpose(1, 2))
    assert torch.count_nonzero(vt[:, :, nk:]) == 0


# ------------------------------------------------------------------------------------------------ small kernels
def test_timestep_embedding_and_skinny():
    t = torch.tensor([981.0], device=DEV)
    e = ops.timestep_embedding(t, 1, 320)
    k = torch.arange(160, device=DEV, dtype=torch.float32)
    f = torch.exp(-math.log(10000.0) * k / 160)
    ref = torch.cat([torch.cos(t * f), torch.sin(t * f)])[None]
    check(e, ref, 2e-4, 0, "timestep embedding")
    w, b = bf(rnd(1280, 320, seed=1, scale=320 ** -0.5)), rnd(1280, seed=2)
    y = ops.linear_skinny(e, w, b, act_out=L.PP_ACT_SILU)
    check(y, F.silu(e @ w.float().t() + b), 1e-4, 1e-4, "skinny 1")
    x8 = rnd(8, 1280, seed=3)
    w2 = bf(rnd(333, 1280, seed=4, scale=1280 ** -0.5))
    y2 = ops.linear_skinny(x8, w2, None, act_in=L.PP_ACT_SILU)
    check(y2, F.silu(x8) @ w2.float().t(), 1e-4, 1e-4, "skinny 8 rows")


@pytest.mark.parametrize("cin,cout,stride", [(9, 320, 1), (4, 320, 1), (3, 16, 1), (16, 32, 2), (256, 320, 1)])
def test_conv_direct(cin, cout, stride):
    B, H = 2, 16
    x = bf(rnd(B, H, H, cin, seed=1))
    w = bf(rnd(cout, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=3)
    add = bf(rnd(B, (H - 1) // stride + 1, (H - 1) // stride + 1, cout, seed=4))
    out = ops.conv3x3_direct(x, w.permute(2, 3, 1, 0).contiguous(), b, stride, True, add)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, stride=stride, padding=1).permute(0, 2, 3, 1)
    check(out, F.silu(ref + add.float()), 2e-2, 1e-2, "conv direct")


@pytest.mark.parametrize("B,H,cin", [(2, 16, 320),      # UNet conv_out: the MFMA kernel (cin % 32 == 0)
                                     (1, 6, 128),       # 36 pixels: a ragged last 16-pixel group; VAE decoder width
                                     (3, 5, 32),        # one 32-deep step: three of the four waves contribute zeros
                                     (1, 8, 416),       # 13 steps: a second load round with one live step
                                     (2, 8, 24)])       # cin % 32 != 0: the scalar kernel
def test_conv_smallcout(B, H, cin):
    x = bf(rnd(B, H, H, cin, seed=1))
    w = bf(rnd(4, cin, 3, 3, seed=2, scale=(9 * cin) ** -0.5))
    b = rnd(4, seed=3)
    out = ops.conv3x3_smallcout(x, w.permute(0, 2, 3, 1).reshape(4, -1).contiguous(), b)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b, padding=1)
    check(out, ref, 1e-3, 1e-3, "conv_out")


def test_layout_and_add():
    x = rnd(2, 4, 8, 8, seed=1)
    y = ops.nchw_to_nhwc(x, batch=4, ldc=9, c0=0)
    assert torch.equal(y[:2, :, :, :4], bf(x).permute(0, 2, 3, 1)) and torch.equal(y[2:, :, :, :4], y[:2, :, :, :4])
    assert torch.count_nonzero(y[..., 4:]) == 0
    z = bf(rnd(2, 8, 8, 320, seed=2))
    assert torch.equal(ops.nhwc_to_nchw(z), z.float().permute(0, 3, 1, 2))
    a, b = bf(rnd(4096, seed=3)), bf(rnd(4096, seed=4))
    assert torch.equal(ops.add(a, b), bf(a.float() + b.float()))


def test_mask_prep_bit_exact():
    from powerpaint_amd.pipelines._base import hip_mask_prep
    g = torch.Generator("cpu").manual_seed(0)
    mask = torch.rand(2, 1, 64, 64, generator=g)
    mask[0, 0, 0, :8] = torch.tensor([0.5, 0.49999997, 0.50000006, 0.0, 1.0, 0.25, 0.75, 0.5])
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    m = mask.clone(); m[m < 0.5] = 0; m[m >= 0.5] = 1
    out = hip_mask_prep(0, mask.to(DEV), None, mask.shape, 2, 1, 64, 64)
    assert torch.equal(out.cpu(), m)
    mi = hip_mask_prep(1, img.to(DEV), out, img.shape, 2, 3, 64, 64)
    assert torch.equal(mi.cpu(), img * (m < 0.5))
    dn = hip_mask_prep(2, out, None, (2, 1, 8, 8), 2, 1, 64, 64, 8, 8)
    assert torch.equal(dn.cpu(), F.interpolate(m, size=(8, 8)))
    dn2 = hip_mask_prep(2, out, None, (2, 1, 24, 40), 2, 1, 64, 64, 24, 40)
    assert torch.equal(dn2.cpu(), F.interpolate(m, size=(24, 40)))
    rgb = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    om = hip_mask_prep(3, rgb.to(DEV), None, (2, 1, 64, 64), 2, 3, 64, 64)
    assert torch.equal(om.cpu(), (rgb.sum(1)[:, None] < 0).float())


@pytest.mark.parametrize("kind", ["ddim", "dpm"])
@pytest.mark.parametrize("N", [10, 50])
def test_scheduler_step_kernel(kind, N):
    """Fused CFG + step kernel over a whole schedule vs the float64 closed form (oracle)."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    sch = (PS.DDIMScheduler if kind == "ddim" else PS.DPMSolverMultistepScheduler)()
    sch.set_timesteps(N, device=DEV)
    g = torch.Generator("cpu").manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(4, 4, 8, 8, generator=g) for _ in range(N)]
    gs = 7.5
    x = x0.to(DEV).clone()
    mp = torch.zeros_like(x)
    lib = L.lib()
    sch.reset()
    ticket = torch.zeros(1, dtype=torch.int32, device=DEV)
    for i in range(N):
        e = eps[i].to(DEV)
        # even steps: the counter moves on inside the launch (advance_ticket, ABI v20); odd steps: pp_step_advance behind it
        L.check(lib.pp_cfg_sched_step(e.data_ptr(), 1, gs, x.data_ptr(), mp.data_ptr(), x.numel(), sch.kind,
                                      sch.coef_table().data_ptr(), sch.step_counter().data_ptr(),
                                      ticket.data_ptr() if i % 2 == 0 else None,
                                      torch.cuda.current_stream().cuda_stream), "step")
        if i % 2:
            L.check(lib.pp_step_advance(sch.step_counter().data_ptr(), torch.cuda.current_stream().cuda_stream), "adv")
        assert int(sch.step_counter()) == i + 1 and int(ticket) == 0
    comb = [(e[:2] + gs * (e[2:] - e[:2])).double().numpy() for e in eps]
    if kind == "ddim":
        ref = x0.double().numpy()
        for i, t in enumerate(OS.ddim_timesteps(N)):
            ref = OS.ddim_step_f64(ref, comb[i], int(t), N)
    else:
        ref = OS.dpm_run_f64(x0.double().numpy(), comb, N)
    check(x.cpu(), torch.from_numpy(ref).float(), 2e-3, 2e-3, f"{kind} {N} steps")
    assert int(sch.step_counter()) == N


@pytest.mark.parametrize("N", [4, 20])
def test_pndm_step_kernel_vs_oracle(N):
    """scheduler.step of the product PNDM (HIP kernel, kind 2: history ring + saved sample + table-driven multistep
    weights) against the oracle's diffusers-protocol class on the same eps sequence."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    o, h = OS.PNDMScheduler(), PS.PNDMScheduler()
    o.set_timesteps(N)
    h.set_timesteps(N, device=DEV)
    assert h.timesteps.cpu().tolist() == o.timesteps.tolist()
    g = torch.Generator("cpu").manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(N + 1)]
    xo, xh = x0, x0.to(DEV)
    for k, t in enumerate(o.timesteps):
        xo = o.step(eps[k], t, xo)[0]
        xh = h.step(eps[k].to(DEV), t, xh, return_dict=False)[0]
        check(xh.cpu(), xo, 1e-4 * max(1.0, float(xo.abs().max())), 1e-4, f"pndm evaluation {k}")


def test_ddim_step_with_eta_vs_oracle():
    """Stochastic DDIM through `scheduler.step(..., eta, generator)` (pp_cfg_sched_step + pp_ddim_variance_noise): same
    CPU generator state -> same variance noise as the oracle's diffusers-protocol step; eta = 0 afterwards is
    deterministic again."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    o, h = OS.DDIMScheduler(), PS.DDIMScheduler()
    o.set_timesteps(6)
    h.set_timesteps(6, device=DEV)
    g = torch.Generator("cpu").manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(6)]
    go, gh = torch.Generator("cpu").manual_seed(7), torch.Generator("cpu").manual_seed(7)
    xo, xh = x0, x0.to(DEV)
    for k, t in enumerate(o.timesteps):
        eta = 0.8 if k < 4 else 0.0
        xo = o.step(eps[k], t, xo, eta=eta, generator=go)[0]
        xh = h.step(eps[k].to(DEV), t, xh, eta=eta, generator=gh, return_dict=False)[0]
        check(xh.cpu(), xo, 1e-4 * max(1.0, float(xo.abs().max())), 1e-4, f"ddim eta step {k}")
    assert torch.equal(torch.randn(3, generator=go), torch.randn(3, generator=gh))      # same number of draws
    with pytest.raises(ValueError):
        h.set_eta(1.5)


@pytest.mark.parametrize("name,kw", [("DDIMScheduler", {}), ("DPMSolverMultistepScheduler", {}), ("PNDMScheduler", {}),
                                     ("UniPCMultistepScheduler", dict(solver_order=3))])
@pytest.mark.parametrize("N,begin", [(10, 4), (6, 5)])
def test_schedulers_enter_the_schedule_late(name, kw, N, begin):
    """`strength < 1` (pipeline_PowerPaint.py:713-720): the loop runs over `scheduler.timesteps[t_start:]`.  The oracle's
    diffusers-protocol classes find their place from the first timestep they see and restart their warm-up there; the
    product tables do the same after `set_begin_index` (first-order first step, PLMS start-up, UniPC order ramp)."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    o, h = getattr(OS, name)(**kw), getattr(PS, name)(**kw)
    o.set_timesteps(N)
    h.set_timesteps(N, device=DEV)
    h.set_begin_index(begin)
    assert h.begin_index == begin and int(h.step_counter()) == begin
    assert h.timesteps.cpu().tolist() == o.timesteps.tolist()          # the scheduler keeps the whole list
    g = torch.Generator("cpu").manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    ts = o.timesteps[begin:]
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in ts]
    xo, xh = x0, x0.to(DEV)
    h.reset()
    for k, t in enumerate(ts):
        xo = o.step(eps[k], t, xo)[0]
        xh = h.step(eps[k].to(DEV), t, xh, return_dict=False)[0]
        check(xh.cpu(), xo, 1e-4 * max(1.0, float(xo.abs().max())), 1e-4, f"{name} entered at {begin}: step {k}")
    h.set_timesteps(N, device=DEV)                                      # ... and set_timesteps starts from the top again
    assert h.begin_index == 0 and int(h.step_counter()) == 0


@pytest.mark.parametrize("K,N,spacing,off", [(2, 10, "leading", 1), (3, 6, "linspace", 0), (1, 3, "trailing", 0)])
def test_unipc_step_kernel_vs_oracle(K, N, spacing, off):
    """scheduler.step of the product UniPC (HIP kernel, kind 3: corrector + predictor as one linear form over the
    [last, m1, m2, m3] state) against the oracle's diffusers-protocol class on the same eps sequence."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    kw = dict(solver_order=K, timestep_spacing=spacing, steps_offset=off)
    o, h = OS.UniPCMultistepScheduler(**kw), PS.UniPCMultistepScheduler(**kw)
    o.set_timesteps(N)
    h.set_timesteps(N, device=DEV)
    assert h.timesteps.cpu().tolist() == o.timesteps.tolist()
    g = torch.Generator("cpu").manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(N)]
    xo, xh = x0, x0.to(DEV)
    for k, t in enumerate(o.timesteps):
        xo = o.step(eps[k], t, xo)[0]
        xh = h.step(eps[k].to(DEV), t, xh, return_dict=False)[0]
        check(xh.cpu(), xo, 1e-4 * max(1.0, float(xo.abs().max())), 1e-4, f"unipc step {k}")


def test_latent_blend_kernel():
    """pp_latent_blend = the 4-channel-UNet branch of the v1 loop body (pipeline_PowerPaint.py:1025-1036): row `step` of
    the re-noise table, first image / first mask broadcast over the batch, clean latents on the (1, 0) row."""
    B, Cc, h, w = 3, 4, 16, 24
    lat, nz = rnd(B, Cc, h, w, seed=1), rnd(B, Cc, h, w, seed=2)
    x0 = rnd(1, Cc, h, w, seed=3)
    mk = (rnd(1, 1, h, w, seed=4) > 0).float()
    tab = torch.tensor([[0.3, 0.95], [0.8, 0.6], [1.0, 0.0]], device=DEV)
    lib = L.lib()
    for r in range(3):
        step = torch.tensor([r], dtype=torch.int32, device=DEV)
        out = lat.clone()
        rc = lib.pp_latent_blend(out.data_ptr(), x0.data_ptr(), mk.data_ptr(), nz.data_ptr(), tab.data_ptr(), step.data_ptr(),
                                 B, Cc, h * w, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        proper = x0 if r == 2 else tab[r, 0] * x0 + tab[r, 1] * nz
        ref = (1 - mk) * proper + mk * lat
        check(out, ref, 1e-6, 1e-6, f"latent blend row {r}")
        if r == 2:
            assert torch.equal(out[:, :, mk[0, 0] == 0], x0.expand(B, -1, -1, -1)[:, :, mk[0, 0] == 0])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("B,hw,nctx,C", [(2, 1024, 77, 320), (3, 256, 77, 320), (1, 128, 80, 320), (2, 128, 5, 320),
                                         (2, 1024, 77, 640), (3, 64, 77, 640), (1, 128, 5, 640),
                                         (2, 256, 77, 1280), (3, 64, 80, 1280), (1, 192, 7, 1280)])
def test_fused_cross_attention_block(B, hw, nctx, C, fold, dtype):
    """pp_xattn_fold + pp_xattn_block (norm2 -> attn2 -> + residual of BasicTransformerBlock in one launch: 128-row tiles
    at C = 320, 64-row tiles x 320-column groups at C = 640 / 1280 -- the 32x32, 16x16 and 8x8 levels) against (a) fp32
    torch of the same sub-block and (b) the three-launch chain it replaces (to_q GEMM with the folded LayerNorm,
    pp_attention_fwd over the 77 keys, to_out GEMM + residual + row moments)."""
    heads = 8
    d = C // heads
    M = B * hw
    tol = 4.0 if dtype == torch.bfloat16 else 1.0                       # (fp16: 8x finer mantissa; gates 4x tighter)
    h = (rnd(M, C, seed=1, scale=1.5) + 0.4).to(dtype)
    ctx = rnd(B * nctx, 768, seed=2).to(dtype)
    g2, b2 = rnd(C, seed=3) * 0.3 + 1.0, rnd(C, seed=4) * 0.2
    wq, wk, wv = (rnd(C, C, seed=5, scale=C ** -0.5), rnd(C, 768, seed=6, scale=768 ** -0.5),
                  rnd(C, 768, seed=7, scale=768 ** -0.5))
    wo, bo = rnd(C, C, seed=8, scale=C ** -0.5), rnd(C, seed=9) * 0.1
    # step-invariant K and V^T exactly as the setup plan makes them (one GEMM, V transposed by the epilogue)
    ldvt = (nctx + 7) // 8 * 8
    wkv = torch.cat([wk, wv]).to(dtype).contiguous()
    k, vt = ops.gemm(ctx, wkv, vt_col0=C, rows_per_batch=nctx)
    assert vt.shape == (B, C, nctx)
    vtp = torch.zeros(B, C, ldvt, dtype=dtype, device=DEV)
    vtp[:, :, :nctx] = vt
    hf = h.float()
    if fold:
        tiles = C // 160
        st = torch.stack([hf.reshape(M, tiles, 160).sum(-1), (hf * hf).reshape(M, tiles, 160).sum(-1)], -1).contiguous()
        wqf = (wq * g2[None, :]).to(dtype).contiguous()
        cs, tq = wqf.float().sum(1).contiguous(), (wq @ b2).contiguous()
        x_in, kw_q, kw_f, ln = h, dict(ln_stats=st, ln_colsum=cs, ln_dim=C, bias=tq), dict(q_colsum=cs, q_bias=tq), st
        mean = hf.mean(-1, keepdim=True)
        rstd = torch.rsqrt((hf * hf).mean(-1, keepdim=True) - mean * mean + 1e-5)
        q_ref = rstd * (hf @ wqf.float().t() - mean * cs) + tq
    else:
        x_in = F.layer_norm(hf, (C,), g2, b2, 1e-5).to(dtype)
        wqf = wq.to(dtype).contiguous()
        kw_q, kw_f, ln = {}, {}, None
        q_ref = x_in.float() @ wqf.float().t()
    wod = wo.to(dtype).contiguous()
    folded = ops.xattn_fold(k, vtp, B, nctx, heads, wqf, wod, **kw_f)
    out, rs = ops.xattn_block(x_in, folded, bias_o=bo, res=h, ln_stats=ln, rows_per_batch=hw, row_stats=True)
    # (a) fp32 reference from the same 16-bit operands
    kf, vf = k.float().reshape(B, nctx, heads, d), vt.float().reshape(B, heads, d, nctx)
    qh = q_ref.reshape(B, hw, heads, d)
    p = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", qh, kf) * d ** -0.5, -1)
    o = torch.einsum("bhqk,bhdk->bqhd", p, vf).reshape(M, C)
    ref = o @ wod.float().t() + bo + hf
    check(out, ref, 1.0e-2 * tol, 4e-3 * tol, "fused cross-attention block vs fp32")
    # (b) the chain it replaces
    q = ops.gemm(x_in, wqf, **kw_q)
    ao = ops.attention(q, k, vtp, B, heads, hw, nctx, d)
    old, rs_old = ops.gemm(ao, wod, bias=bo, res1=h, row_stats=True)
    check(out, old, 1.2e-2 * tol, 4e-3 * tol, "fused cross-attention block vs the three-launch chain")
    ofl = out.float()
    rs_ref = torch.stack([ofl.reshape(M, C // 160, 160).sum(-1), (ofl * ofl).reshape(M, C // 160, 160).sum(-1)], -1)
    check(rs, rs_ref, 2e-3, 1e-5, "row moments of the stored values")
    # the folded matrices themselves
    gt, gcs, gb, ht = folded
    if fold:      # the mean term of the folded LayerNorm is the column sum of G^T AS STORED
        check(gcs, gt.float().sum(-1), 2e-3, 1e-5, "logit colsum = sum over c of the rounded G^T")
    G = torch.einsum("bkhd,hdc->bhkc", kf, wqf.float().reshape(heads, d, C)) * (d ** -0.5 * 1.4426950408889634)
    check(gt.reshape(B, heads, 80, C)[:, :, :nctx], G, 2e-3 * tol, 4e-3 * tol, "G^T")
    assert torch.all(gt.reshape(B, heads, 80, C)[:, :, nctx:] == 0)
    assert torch.all(torch.isinf(gb.reshape(B, heads, 80)[:, :, nctx:])) and torch.isfinite(gb.reshape(B, heads, 80)[:, :, :nctx]).all()
    H = torch.einsum("nhd,bhdk->bnhk", wod.float().reshape(C, heads, d), vf)                 # [B, C, heads, nctx]
    kp = torch.arange(640)
    s32, kg, j = kp // 32, (kp // 8) % 4, kp % 8
    kk = 32 * s32 + 16 * (j // 4) + 4 * kg + (j % 4)                    # contraction index stored at position kp
    Hfull = torch.zeros(B, C, heads, 80, device=DEV)
    Hfull[..., :nctx] = H
    check(ht, Hfull.reshape(B, C, 640)[:, :, kk.to(DEV)], 2e-3 * tol, 4e-3 * tol, "H^T (k-permuted)")
    # (c) round 5, ABI v20: the same folded sub-block as TWO plain GEMMs with one weight matrix per batch item -- logits +
    # per-head softmax in the first one's epilogue (PP_ACT_SOFTMAX80), probabilities x H^T (natural order: kperm = 2) +
    # bias + residual + row moments in the second.  What the engine runs at C = 1280 (whole 64-row tiles per batch item).
    if hw % 64 == 0:
        gt2, gcs2, gb2, ht2 = ops.xattn_fold(k, vtp, B, nctx, heads, wqf, wod, kperm=2, **kw_f)
        assert torch.equal(gt2, gt) and torch.equal(gb2, gb) and torch.equal(gcs2, gcs)
        check(ht2, Hfull.reshape(B, C, 640), 2e-3 * tol, 4e-3 * tol, "H^T (natural order)")
        prob = ops.gemm(x_in, gt2, bias=gb2, act=L.PP_ACT_SOFTMAX80, rows_per_batch=hw, ln_stats=ln,
                        ln_colsum=gcs2 if fold else None, ln_dim=C)
        assert prob.shape == (M, 640)
        xin_f = x_in.float().reshape(B, hw, C)
        lg = torch.einsum("bmc,bnc->bmn", xin_f, gt2.float())
        if fold:
            lg = rstd.reshape(B, hw, 1) * (lg - mean.reshape(B, hw, 1) * gcs2[:, None, :])
        lg = lg + gb2[:, None, :]
        p_ref = torch.softmax(lg.reshape(B, hw, heads, 80) * math.log(2.0), -1).reshape(M, 640)
        check(prob, p_ref, 4e-3 * tol, 8e-3 * tol, "PP_ACT_SOFTMAX80 probabilities")
        assert torch.all(prob.reshape(M, heads, 80)[:, :, nctx:] == 0)
        out2, rs2 = ops.gemm(prob, ht2, bias=bo, res1=h, row_stats=True, rows_per_batch=hw)
        check(out2, ref, 1.0e-2 * tol, 4e-3 * tol, "two-GEMM cross-attention vs fp32")
        check(out2, old, 1.2e-2 * tol, 4e-3 * tol, "two-GEMM cross-attention vs the three-launch chain")
        o2 = out2.float()
        check(rs2, torch.stack([o2.reshape(M, C // 160, 160).sum(-1), (o2 * o2).reshape(M, C // 160, 160).sum(-1)], -1),
              2e-3, 1e-5, "row moments of the two-GEMM output")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 320), (1, 24, 40, 320), (2, 13, 9, 128), (1, 8, 8, 64)])
def test_fused_groupnorm_silu_conv_out(B, H, W, C, dtype):
    """pp_gn_conv3x3_smallcout (conv_norm_out + SiLU + conv_out in one launch) against the two launches it replaces
    (pp_groupnorm_apply_acc -> pp_conv3x3_smallcout: same rounding point of the normalised activation, so only the fp32
    summation order differs) and against fp32 torch; ragged sizes exercise the patch borders."""
    groups = 32
    cg = C // groups
    x = (rnd(B, H, W, C, seed=1, scale=2.0) + 0.5).to(dtype)
    xf = x.float().reshape(B, H * W, groups, cg)
    acc = torch.stack([(xf.sum((1, 3)).double() * 2 ** 24).round().long(),
                       ((xf * xf).sum((1, 3)).double() * 2 ** 20).round().long()], -1).contiguous()
    g, b = rnd(C, seed=2) * 0.3 + 1.0, rnd(C, seed=3) * 0.2
    w = rnd(4, 9 * C, seed=4, scale=(9 * C) ** -0.5).to(dtype).contiguous()
    bias = rnd(4, seed=5)
    out = ops.gn_conv3x3_smallcout(x, acc, g, b, 1e-5, w, bias)
    y = ops.groupnorm_apply_acc(x, acc, g, b, 1e-5, True)
    old = ops.conv3x3_smallcout(y, w, bias)
    tol = 1.0 if dtype == torch.bfloat16 else 0.25
    check(out, old, 2e-4, 1e-4, "fused conv_out vs groupnorm_apply_acc + conv3x3_smallcout")
    yn = F.silu(F.group_norm(x.float().permute(0, 3, 1, 2), groups, g, b, 1e-5))
    ref = F.conv2d(yn, w.float().reshape(4, 3, 3, C).permute(0, 3, 1, 2), bias, padding=1)
    check(out, ref, 3e-2 * tol, 1e-2 * tol, "fused conv_out vs fp32")




@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,hw", [(2, 1024), (1, 4096), (3, 128), (8, 4096)])
def test_transformer_front_end_in_one_launch(B, hw, dtype):
    """pp_tfront (csrc/tfront.hip): Transformer2DModel.norm -> proj_in -> LayerNorm1-folded QKV at C = 320 in one launch
    against (a) the three launches it replaces (pp_groupnorm_apply_acc, pp_gemm_bf16 with row moments, pp_gemm_bf16 with the
    folded LayerNorm and the transposed V) and (b) fp32 torch of the same modules on the same 16-bit operands."""
    from powerpaint_amd.engine import _kperm
    C, groups = 320, 32
    M = B * hw
    tol = 4.0 if dtype == torch.bfloat16 else 1.0
    x = (rnd(M, C, seed=1, scale=1.4) + 0.3).to(dtype)
    xf = x.float().reshape(B, hw, groups, C // groups)
    acc = torch.stack([(xf.sum((1, 3)).double() * 2 ** 24).round().long(),
                       ((xf * xf).sum((1, 3)).double() * 2 ** 20).round().long()], -1).contiguous()
    gg, gb = rnd(C, seed=2) * 0.3 + 1.0, rnd(C, seed=3) * 0.2
    w1, b1 = rnd(C, C, seed=4, scale=C ** -0.5).to(dtype).contiguous(), rnd(C, seed=5) * 0.1
    g1, be1 = rnd(C, seed=6) * 0.3 + 1.0, rnd(C, seed=7) * 0.2
    wqkv = rnd(3 * C, C, seed=8, scale=C ** -0.5)
    wf = (wqkv * g1[None, :]).to(dtype).contiguous()
    cs, tb = wf.float().sum(1).contiguous(), (wqkv @ be1).contiguous()
    hs, qk, vt = ops.tfront(x, acc, gg, gb, w1, b1, _kperm(wf).contiguous(), cs, tb, hw)
    # (a) the chain
    n = ops.groupnorm_apply_acc(x.view(B, hw, 1, C), acc, gg, gb, 1e-6, False).view(M, C)
    hs_old, st = ops.gemm(n, w1, b1, row_stats=True)
    qk_old, vt_old = ops.gemm(hs_old, wf, bias=tb, ln_stats=st, ln_colsum=cs, ln_dim=C, vt_col0=2 * C, rows_per_batch=hw)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    check(hs, hs_old, 2 * ulp, 1.5 * ulp, "hs vs groupnorm_apply + proj_in")
    check(qk, qk_old, 1.2e-2 * tol, 4e-3 * tol, "Q | K vs the chain")
    check(vt, vt_old, 1.2e-2 * tol, 4e-3 * tol, "V^T vs the chain")
    # (b) fp32 torch
    nt = F.group_norm(x.float().reshape(B, hw, C).permute(0, 2, 1), groups, gg, gb, 1e-6).permute(0, 2, 1).reshape(M, C)
    hst = nt @ w1.float().t() + b1
    check(hs, hst, 1.5e-2 * tol, 6e-3 * tol, "hs vs fp32")
    qkvt = F.layer_norm(hs.float(), (C,), g1, be1, 1e-5) @ wqkv.t()        # (from the kernel's own 16-bit hs: isolates GEMM 2)
    check(qk, qkvt[:, :2 * C], 1.5e-2 * tol, 6e-3 * tol, "Q | K vs fp32")
    check(vt, qkvt[:, 2 * C:].reshape(B, hw, C).permute(0, 2, 1), 1.5e-2 * tol, 6e-3 * tol, "V^T vs fp32")
    # (c) q_scale (ABI v20): only the Q third changes -- multiplied in fp32 before its one rounding
    qs = 40 ** -0.5 * LOG2E
    hs2, qk2, vt2 = ops.tfront(x, acc, gg, gb, w1, b1, _kperm(wf).contiguous(), cs, tb, hw, q_scale=qs)
    assert torch.equal(hs2, hs) and torch.equal(vt2, vt) and torch.equal(qk2[:, C:], qk[:, C:])
    check(qk2[:, :C], qkvt[:, :C] * qs, 1.5e-2 * tol * qs, 6e-3 * tol, "pre-multiplied Q vs fp32")
    check(qk2[:, :C], qk[:, :C].float() * qs, 2 * ulp, 2 * ulp, "pre-multiplied Q vs the rounded plain Q")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_step_head_is_the_three_launches_it_replaces(dtype):
    """pp_step_head = pp_embed_splice (one row of the time-embedding table, indexed by the device step counter) +
    pp_nchw_to_nhwc (fp32 NCHW latents -> 16-bit NHWC input buffer with the CFG duplication) + pp_zero_u64, bit for bit."""
    B, C, h, ldc, rows, steps = 4, 4, 24, 64, 20160, 7
    table = rnd(steps, rows, seed=1)
    lat = rnd(B, C, h, h, seed=2)
    step = torch.tensor([5], dtype=torch.int32, device=DEV)
    temb = torch.full((rows,), -1.0, device=DEV)
    x = torch.full((2 * B, h * h, ldc), 3.0, device=DEV).to(dtype)
    acc = torch.full((999,), 7, dtype=torch.int64, device=DEV)
    L.check(L.lib().pp_step_head(table.data_ptr(), step.data_ptr(), temb.data_ptr(), rows, lat.data_ptr(), 2 * B, C, h * h, B,
                                 x.data_ptr(), ldc, 0, L.dtype_code(dtype), acc.data_ptr(), acc.numel(),
                                 torch.cuda.current_stream().cuda_stream), "pp_step_head")
    assert torch.equal(temb, table[5]) and int(acc.abs().sum()) == 0
    ref = torch.cat([lat, lat]).permute(0, 2, 3, 1).reshape(2 * B, h * h, C).to(dtype)
    assert torch.equal(x[:, :, :C], ref) and bool((x[:, :, C:] == 3.0).all())          # (pad channels untouched)
