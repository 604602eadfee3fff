"""-m gpu: pp_ff_fused (csrc/ff_fused.hip) -- FeedForward (GEGLU) + FF2 . proj_out of a C = 320 transformer in one launch,
hidden dimension streamed -- against fp32 torch and against the two-launch chain it replaces
(pp_gemm_bf16(act = GEGLU, folded LayerNorm) -> pp_gemm_bf16 over [g | hs]).

Reference: diffusers 0.27 FeedForward(GEGLU) / BasicTransformerBlock.forward `ff(norm3(h)) + h`, Transformer2DModel.proj_out
(ctor site /root/reference/powerpaint/models/unet_2d_blocks.py:1289-1300).  Tolerances as in tests/test_ops_gpu.py: 16-bit
output rounding is 2^-9 (bf16) / 2^-11 (fp16) relative; the fused launch rounds the GEGLU values to 16 bits exactly where the
chain stores them, so the two differ only by the fp32 summation order of the second GEMM.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd import ops  # noqa: E402
from powerpaint_amd.engine import _geglu_interleave, _kperm_geglu  # noqa: E402

DEV = "cuda"
C = 320


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


W8 = [True, False]      # the 8-wave kernel (W2' natural) and the 4-wave kernel (hidden index permuted)


def _w2(k, w8):
    return dict(w2kp=k["w2"] if w8 else k["w2kp"], w2_kperm=not w8)


def _case(M, dtype, fold, seed=0):
    """Weights of one transformer's feed-forward in the layouts the engine packs, plus the fp32 meaning of the op."""
    hs = (rnd(M, C, seed=seed + 1, scale=2.0) + 0.5).to(dtype)
    gam, bet = rnd(C, seed=seed + 2) * 0.3 + 1.0, rnd(C, seed=seed + 3) * 0.2
    w_ff1, b_ff1 = rnd(8 * C, C, seed=seed + 4, scale=C ** -0.5), rnd(8 * C, seed=seed + 5) * 0.1
    w_ff2, b_ff2 = rnd(C, 4 * C, seed=seed + 6, scale=(4 * C) ** -0.5), rnd(C, seed=seed + 7) * 0.1
    w_po, b_po = rnd(C, C, seed=seed + 8, scale=C ** -0.5), rnd(C, seed=seed + 9) * 0.1
    hf = hs.float()
    if fold:
        w1 = _geglu_interleave(w_ff1 * gam[None, :]).to(dtype).contiguous()
        b1 = _geglu_interleave(w_ff1 @ bet + b_ff1).contiguous()
        cs1 = w1.float().sum(1).contiguous()
        st = torch.stack([hf.reshape(M, 2, 160).sum(-1), (hf * hf).reshape(M, 2, 160).sum(-1)], -1).contiguous()
        mean = hf.mean(-1, keepdim=True)
        rstd = torch.rsqrt((hf * hf).mean(-1, keepdim=True) - mean * mean + 1e-5)
        s = rstd * (hf @ w1.float().t() - mean * cs1) + b1
    else:
        w1 = _geglu_interleave(w_ff1).to(dtype).contiguous()
        b1 = _geglu_interleave(b_ff1).contiguous()
        cs1 = st = None
        s = hf @ w1.float().t() + b1
    # interleaved quads (h0, h1, g0, g1) -> units (2q, 2q + 1)
    q = s.reshape(M, 2 * C, 4)
    act = torch.stack([q[..., 0] * F.gelu(q[..., 2]), q[..., 1] * F.gelu(q[..., 3])], -1).reshape(M, 4 * C)
    w2 = torch.cat([w_po @ w_ff2, w_po], 1).to(dtype)
    bias2 = (w_po @ b_ff2 + b_po).contiguous()
    ref = act.to(dtype).float() @ w2[:, :4 * C].float().t() + hf @ w2[:, 4 * C:].float().t() + bias2
    w2kp = torch.cat([_kperm_geglu(w2[:, :4 * C]), w2[:, 4 * C:]], 1).contiguous()
    return dict(hs=hs, w1=w1, b1=b1, cs1=cs1, st=st, w2=w2.contiguous(), w2kp=w2kp, bias2=bias2, ref=ref)


def _chain(k, res1=None, res2=None, rows_per_batch=0, gn=None):
    kw = dict(ln_stats=k["st"], ln_colsum=k["cs1"], ln_dim=C) if k["st"] is not None else {}
    g = ops.gemm(k["hs"], k["w1"], bias=k["b1"], act=L.PP_ACT_GEGLU, **kw)
    return ops.gemm(g, k["w2"], bias=k["bias2"], x2=k["hs"], res1=res1, res2=res2, rows_per_batch=rows_per_batch, gn=gn)


def _close(out, ref, dtype, what):
    out, ref = out.float(), ref.float()
    assert torch.isfinite(out).all(), what
    atol, rtol = (2e-2, 1e-2) if dtype == torch.bfloat16 else (5e-3, 2.5e-3)
    err = (out - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), (what, int(bad.sum()), float(err.max()), float(ref.abs().max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("M", [128, 640, 4096])
@pytest.mark.parametrize("w8", W8)
def test_fused_feed_forward_vs_fp32_and_vs_the_chain(w8, M, fold, dtype):
    k = _case(M, dtype, fold, seed=M)
    res1 = rnd(M, C, seed=91).to(dtype)
    res2 = rnd(M, C, seed=92).to(dtype)
    out = ops.ff_fused(k["hs"], k["w1"], k["b1"], bias2=k["bias2"], cs1=k["cs1"], ln_stats=k["st"], res1=res1, res2=res2,
                       **_w2(k, w8))
    ref = k["ref"] + res1.float() + res2.float()
    _close(out, ref, dtype, "fused feed-forward vs fp32")
    chain = _chain(k, res1, res2)
    _close(chain, ref, dtype, "two-launch chain vs fp32")
    # the two paths differ by fp32 summation order only: a small share of 1-ulp flips of the 16-bit output
    d = (out.float() - chain.float()).abs()
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert float((d > 0).float().mean()) < 0.10 and bool((d <= 2 * ulp * chain.float().abs().clamp(min=1.0)).all()), \
        (float((d > 0).float().mean()), float(d.max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("w8", W8)
def test_fused_feed_forward_epilogue_statistics_and_wrap(w8, dtype):
    """The second GEMM's epilogue as the engine uses it: GroupNorm statistics of the output for two consumers (own 32-group
    norm; a 640-channel concat norm with this tensor at channel offset 320) and the half-batch residual of the CFG twin
    prefix (res1_wrap_rows) -- both bit-equal to what the chain's second launch produces from the same stored values."""
    B, hw = 4, 256
    M = B * hw
    k = _case(M, dtype, True, seed=7)
    res_half = rnd(M // 2, C, seed=93).to(dtype)
    a1 = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    a2 = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    out = ops.ff_fused(k["hs"], k["w1"], k["b1"], bias2=k["bias2"], cs1=k["cs1"], ln_stats=k["st"], res1=res_half,
                       res1_wrap=M // 2, rows_per_batch=hw, gn=[(a1, 10, 0, 32), (a2, 20, 320, 32)], **_w2(k, w8))
    ref = k["ref"] + torch.cat([res_half, res_half]).float()
    _close(out, ref, dtype, "fused feed-forward with a wrapped residual")
    # statistics = moments of the values AS STORED
    o = out.float().reshape(B, hw, C)
    for acc, cg, c0 in ((a1, 10, 0), (a2, 20, 320)):
        g0 = c0 // cg
        exp_s = torch.zeros(B, 32, device=DEV, dtype=torch.float64)
        exp_q = torch.zeros(B, 32, device=DEV, dtype=torch.float64)
        for gi in range(C // cg):
            sl = o[:, :, gi * cg:(gi + 1) * cg].double()
            exp_s[:, g0 + gi] = sl.sum((1, 2))
            exp_q[:, g0 + gi] = (sl * sl).sum((1, 2))
        got_s, got_q = acc[..., 0].double() / 2 ** 24, acc[..., 1].double() / 2 ** 20
        assert torch.allclose(got_s, exp_s, rtol=1e-4, atol=1e-2) and torch.allclose(got_q, exp_q, rtol=1e-4, atol=1e-2)
    again = torch.zeros_like(a1)
    out2 = ops.ff_fused(k["hs"], k["w1"], k["b1"], bias2=k["bias2"], cs1=k["cs1"], ln_stats=k["st"], res1=res_half,
                        res1_wrap=M // 2, rows_per_batch=hw, gn=[(again, 10, 0, 32)], **_w2(k, w8))
    assert torch.equal(out, out2) and torch.equal(again, a1)            # deterministic, order-independent accumulation


def test_fused_feed_forward_refuses_what_it_does_not_implement():
    k = _case(128, torch.bfloat16, True)
    with pytest.raises(L.PPError):
        ops.ff_fused(k["hs"][:64], k["w1"], k["b1"], k["w2kp"], bias2=k["bias2"], cs1=k["cs1"], ln_stats=k["st"][:64])   # M % 128
    assert L.lib().pp_ff_fused_supported(256, 640, 256) == 0 and L.lib().pp_ff_fused_supported(32768, 320, 4096) == 1
