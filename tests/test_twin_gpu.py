"""-m gpu: the CFG-twin prefix (round 5, ABI v19).

The two halves of `torch.cat([latents] * 2)` (/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:990-996) over
CFG-duplicated mask / masked-image latents are identical until the prompt enters at the first cross-attention
(/root/reference/powerpaint/models/unet_2d_condition.py:1183-1236), so the loop's networks run conv_in, the first resnet and
the first self-attention on ONE half.  Everything here is an exactness statement: the half-batch launches compute the same
rows as the full-batch ones (same kernels, the K walk of a row does not depend on the row count unless the split-K
factor changes), the twice-stored tensor and the wrap-addressed reads are bit copies.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd import ops  # noqa: E402
from powerpaint_amd.engine import SDNet  # noqa: E402
from powerpaint_amd.runtime import NetRuntime  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,cin,tile", [(2, 16, 64, 0), (4, 64, 64, 0), (1, 16, 128, 54), (3, 8, 64, 32)])
def test_conv_stores_its_rows_twice_and_feeds_both_halves_statistics(B, H, cin, tile, dtype):
    """PPGemmArgs.out_dup_rows / gn_dup_batch / gn_dup_mask: conv_in of the twin prefix.  Subscription 0 = the next layer
    of the prefix (half batch), subscription 1 = the up-block concat norm that sees the full, twice-stored tensor."""
    cout = 320
    x = rnd(B, H, H, cin, seed=1, dtype=dtype)
    w = rnd(cout, 9 * cin, seed=2, scale=(9 * cin) ** -0.5, dtype=dtype)
    bias = rnd(cout, seed=3, dtype=torch.float32)
    res = rnd(B, H, H, cout, seed=4, dtype=dtype)
    a_half = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    a_full = torch.zeros(2 * B, 32, 2, dtype=torch.int64, device=DEV)
    out = ops.conv3x3(x, w, bias, res2=res, tile=tile, gn=[(a_half, 10, 0, 32), (a_full, 20, 320, 32)], dup=True,
                      gn_dup_mask=2)
    r_half = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    r_full = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    ref = ops.conv3x3(x, w, bias, res2=res, tile=tile, gn=[(r_half, 10, 0, 32), (r_full, 20, 320, 32)])
    assert out.shape[0] == 2 * B
    assert torch.equal(out[:B], out[B:]) and torch.equal(a_full[:B], a_full[B:])
    if tile == 0 and ops.conv_halo_routed(x, cout):
        # (round 6) the plain reference runs on the halo-tile loop (other tiles, maybe split-K: another summation order), the
        # twin store on the tap-major kernel's single-pass epilogue: equal to rounding, not bit for bit
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        err = (out[:B].float() - ref.float()).abs()
        assert not (err > 2 * ulp + 1.5 * ulp * ref.float().abs()).any(), float(err.max())
        for got, want in ((a_half, r_half), (a_full[:B], r_full)):
            rel = ((got.double() - want.double()).abs() / (want.double().abs() + 2.0 ** 20)).max().item()
            assert rel < 2e-3, rel
        return
    assert torch.equal(out[:B], ref)
    assert torch.equal(a_half, r_half)
    assert torch.equal(a_full[:B], r_full)


def test_dup_is_refused_where_the_single_pass_epilogue_does_not_run():
    x = rnd(1, 16, 16, 64, seed=1)
    w = rnd(320, 9 * 64, seed=2, scale=0.05)
    with pytest.raises(L.PPError):
        ops.conv3x3(x, w, splitk=2, tile=32, dup=True)           # split-K: the combine writes no twins
    with pytest.raises(L.PPError):
        ops.conv3x3(x, w, tile=2, dup=True)                      # the register-staged v1 kernel


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile,splitk", [(0, 0), (53, 1), (54, 1), (24, 1), (32, 2), (1, 1)])
def test_gemm_reads_a_half_batch_residual_with_wrap(tile, splitk, dtype):
    """PPGemmArgs.res1_wrap_rows: the last GEMM of the first transformer (proj_out's residual is the transformer input,
    which exists for one half only) -- every epilogue family incl. the split-K combines and the v1 kernel."""
    Mh, N, K = 512, 320, 640
    x = rnd(2 * Mh, K, seed=1, dtype=dtype)
    w = rnd(N, K, seed=2, scale=K ** -0.5, dtype=dtype)
    res = rnd(Mh, N, seed=3, dtype=dtype)
    acc = torch.zeros(2, 32, 2, dtype=torch.int64, device=DEV)
    gn = [(acc, 10, 0, 32)] if tile not in (1,) else None
    out = ops.gemm(x, w, res1=res, res1_wrap=Mh, tile=tile, splitk=splitk, rows_per_batch=Mh, gn=gn)
    acc2 = torch.zeros_like(acc)
    ref = ops.gemm(x, w, res1=torch.cat([res, res]), tile=tile, splitk=splitk, rows_per_batch=Mh,
                   gn=[(acc2, 10, 0, 32)] if gn else None)
    assert torch.equal(out, ref)
    assert torch.equal(acc, acc2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fold", [True, False])
def test_cross_attention_block_reads_the_half_batch_with_wrap(fold, dtype):
    """pp_xattn_block(src_wrap_rows): x, res and the LayerNorm row moments hold one half, the folded K / V operands are per
    batch item of the full batch (the prompt differs between the halves): bit-equal to the launch on the explicit copy."""
    Bh, hw, nctx, C, heads = 2, 256, 77, 320, 8
    B = 2 * Bh
    x = rnd(Bh * hw, C, seed=1, dtype=dtype)
    res = rnd(Bh * hw, C, seed=2, dtype=dtype)
    k = rnd(B * nctx, C, seed=3, dtype=dtype)
    vt = rnd(B, C, 80, seed=4, dtype=dtype)
    wq = rnd(C, C, seed=5, scale=C ** -0.5, dtype=dtype)
    wo = rnd(C, C, seed=6, scale=C ** -0.5, dtype=dtype)
    bo = rnd(C, seed=7, dtype=torch.float32)
    st = None
    qcs = qb = None
    if fold:
        xf = x.float()
        st = torch.stack([torch.stack([xf[:, :160].sum(1), (xf[:, :160] ** 2).sum(1)], 1),
                          torch.stack([xf[:, 160:].sum(1), (xf[:, 160:] ** 2).sum(1)], 1)], 1).contiguous()
        qcs = wq.float().sum(1).contiguous()
        qb = rnd(C, seed=8, dtype=torch.float32)
    folded = ops.xattn_fold(k, vt, B, nctx, heads, wq, wo, q_colsum=qcs, q_bias=qb)
    out, rs = ops.xattn_block(x, folded, bias_o=bo, res=res, ln_stats=st, rows_per_batch=hw, row_stats=True, twin=True)
    ref, rs_ref = ops.xattn_block(torch.cat([x, x]), folded, bias_o=bo, res=torch.cat([res, res]),
                                  ln_stats=torch.cat([st, st]) if fold else None, rows_per_batch=hw, row_stats=True)
    assert out.shape == (B * hw, C)
    assert torch.equal(out, ref) and torch.equal(rs, rs_ref)
    assert not torch.equal(out[:Bh * hw], out[Bh * hw:])           # (the halves do differ: different prompts)


NETS = {
    "unet": dict(kind="unet", cin=9, tot=9, kw={}),
    "brushnet": dict(kind="brushnet", cin=4, tot=9, kw=dict(conditioning_channels=5)),
    "controlnet": dict(kind="controlnet", cin=4, tot=4, kw=dict(conditioning_channels=3)),
}
SMALL = dict(block_out_channels=(320, 640), layers_per_block=1,
             down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))


def _run(net, twin, B, H, nctx, tot, x, ehs, cond, t=500.0):
    rt = NetRuntime(net, DEV)
    rt.ensure(B, H, H, nctx, tot, ("plain",), cond_hw=(8 * H, 8 * H), twin=twin)
    if net.kind == "controlnet":
        rt.set_cond(cond)
    rt.set_context(ehs)
    rt.load_input([(x, 0)])
    rt.set_timestep(t)
    rt.run_step()
    torch.cuda.synchronize()
    if net.kind == "unet":
        outs = [rt.eps_tensor().clone()]
    else:
        o = rt.outputs
        outs = [rt.act_as_nchw(a).clone() for a in o["down"] + [o["mid"]] + o.get("up", [])]
    return rt, outs


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", list(NETS))
@pytest.mark.parametrize("B,H", [(2, 16), (4, 32)])
def test_twin_plan_computes_what_the_full_batch_plan_computes(name, B, H, dtype):
    """NetRuntime.ensure(twin=True) against twin=False on a network input whose halves are identical and a context that
    differs between the halves: same weights, same inputs.  The prefix runs fewer rows per launch, which may change a
    split-K factor and with it the fp32 summation order, so the gate is 'equal up to 16-bit rounding of a few values':
    cosine, max-abs and the share of differing values are all checked (achieved: bit-equal in most cases)."""
    spec = NETS[name]
    kw = dict(SMALL)
    if name == "controlnet":
        kw.pop("up_block_types")
    net = SDNet(spec["kind"], spec["cin"], dtype=dtype, **kw, **spec["kw"])
    net.load_state_dict(net.synthetic_state_dict(seed=3), DEV)
    nctx = 77
    xh = rnd(B // 2, spec["tot"], H, H, seed=11, dtype=torch.float32)
    x = torch.cat([xh, xh])
    ehs = rnd(B, nctx, 768, seed=12, dtype=dtype)
    ch = rnd(B // 2, 3, 8 * H, 8 * H, seed=13, dtype=torch.float32)
    cond = torch.cat([ch, ch])
    rt0, full = _run(net, False, B, H, nctx, spec["tot"], x, ehs, cond)
    rt1, twin = _run(net, True, B, H, nctx, spec["tot"], x, ehs, cond)
    assert rt1.twin and not rt0.twin
    # the twin plan really runs the prefix on half the rows
    assert rt1.step_plan.flops < rt0.step_plan.flops
    dup = [a for a in rt1.step_plan.keep if getattr(a, "out_dup_rows", 0)]
    wrap = [a for a in rt1.step_plan.keep if getattr(a, "res1_wrap_rows", 0)]
    assert len(dup) == 1 and len(wrap) == 1 and dup[0].M == (B // 2) * H * H and wrap[0].res1_wrap_rows == dup[0].M
    assert not [a for a in rt0.step_plan.keep if getattr(a, "out_dup_rows", 0) or getattr(a, "res1_wrap_rows", 0)]
    for a, b in zip(full, twin):
        assert a.shape == b.shape and torch.isfinite(b.float()).all()
        af, bf_ = a.float().flatten(), b.float().flatten()
        cos = torch.nn.functional.cosine_similarity(af, bf_, dim=0).item()
        err = (af - bf_).abs().max().item()
        frac = (af != bf_).float().mean().item()
        assert cos > 0.999999 and err <= 4e-3 * max(1.0, af.abs().max().item()), (name, cos, err, frac)
    # the halves of the (last) output differ -- the prompts do -- i.e. the test would notice a plan that copied one half;
    # the first residual of a side network (zero conv of conv_in's output) is prompt-independent: its halves ARE equal
    assert not torch.equal(twin[-1][: B // 2], twin[-1][B // 2:])
    if name != "unet":
        assert torch.equal(twin[0][: B // 2], twin[0][B // 2:])


def test_twin_is_ignored_where_the_prefix_cannot_be_split():
    """No fused cross-attention block at the first level (odd latent size for its 128-row tiles) or BrushNet adds inside the
    down path: the flag changes nothing -- never a half-computed batch."""
    net = SDNet("unet", 9, **SMALL)
    net.load_state_dict(net.synthetic_state_dict(seed=3), DEV)
    rt = NetRuntime(net, DEV)
    rt.ensure(2, 8, 8, 77, 9, ("plain",), twin=True)               # hw = 64 < 128
    assert not [a for a in rt.step_plan.keep if getattr(a, "out_dup_rows", 0)]
    shapes = rt._residual_shapes(2, 16, 16, True)
    rt.ensure(2, 16, 16, 77, 9, ("brushnet", {k: [0] * len(v) for k, v in shapes.items()}), twin=True)
    assert not [a for a in rt.step_plan.keep if getattr(a, "out_dup_rows", 0)]
    rt.ensure(2, 16, 16, 77, 9, ("plain",), twin=True)
    assert [a for a in rt.step_plan.keep if getattr(a, "out_dup_rows", 0)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("twin", [False, True])
@pytest.mark.parametrize("Bh,hw,nctx", [(2, 256, 77), (1, 1024, 80), (3, 128, 5)])
def test_attn1_to_out_in_front_of_the_cross_attention_block(Bh, hw, nctx, twin, dtype):
    """pp_xattn_block(pre_w, pre_b): BasicTransformerBlock `attn1(norm1(h0)) + h0` (the to_out Linear + residual) in the
    launch of the cross-attention sub-block, G^T folded with kperm = 1, against the two launches it replaces -- pp_gemm_bf16
    (to_out + residual + row moments) -> pp_xattn_block -- with and without the CFG-twin wrap addressing.  h is rounded to
    16 bits in both paths; only the fp32 order of the LayerNorm row moments differs."""
    C, heads = 320, 8
    B = 2 * Bh if twin else Bh
    Mh = Bh * hw
    ao = rnd(Mh, C, seed=1, dtype=dtype)                          # self-attention output (one half when twin)
    h0 = rnd(Mh, C, seed=2, scale=1.5, dtype=dtype)
    w1 = rnd(C, C, seed=3, scale=C ** -0.5, dtype=dtype)
    b1 = rnd(C, seed=4, scale=0.1, dtype=torch.float32)
    k = rnd(B * nctx, C, seed=5, dtype=dtype)
    vt = rnd(B, C, 80, seed=6, dtype=dtype)
    wq = rnd(C, C, seed=7, scale=C ** -0.5, dtype=dtype)
    wo = rnd(C, C, seed=8, scale=C ** -0.5, dtype=dtype)
    bo = rnd(C, seed=9, scale=0.1, dtype=torch.float32)
    qcs = wq.float().sum(1).contiguous()
    qb = rnd(C, seed=10, scale=0.1, dtype=torch.float32)
    plain = ops.xattn_fold(k, vt, B, nctx, heads, wq, wo, q_colsum=qcs, q_bias=qb)
    perm = ops.xattn_fold(k, vt, B, nctx, heads, wq, wo, q_colsum=qcs, q_bias=qb, kperm=True)
    h1, st = ops.gemm(ao, w1, bias=b1, res1=h0, row_stats=True)
    ref, rs_ref = ops.xattn_block(h1, plain, bias_o=bo, res=h1, ln_stats=st, rows_per_batch=hw, row_stats=True, twin=twin)
    out, rs = ops.xattn_block(ao, perm, bias_o=bo, res=h0, rows_per_batch=hw, row_stats=True, twin=twin, pre_w=w1, pre_b=b1,
                              ln_fold=True)
    assert out.shape == ref.shape == (B * hw, C) and torch.isfinite(out.float()).all()
    d = (out.float() - ref.float()).abs()
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert float((d > 0).float().mean()) < 0.05 and bool((d <= 2 * ulp * ref.float().abs().clamp(min=1.0)).all()), \
        (float((d > 0).float().mean()), float(d.max()))
    assert torch.allclose(rs, rs_ref, rtol=2e-3, atol=0.5)
    # the permuted G^T is the plain one with its channel index shuffled inside every group of 32; everything else equal
    kp = torch.arange(C)
    s32, kg, j = kp // 32, (kp // 8) % 4, kp % 8
    src = (32 * s32 + 16 * (j // 4) + 4 * kg + (j % 4)).to(DEV)
    assert torch.equal(perm[0], plain[0][:, :, src]) and torch.equal(perm[2], plain[2]) and torch.equal(perm[3], plain[3])
    assert torch.allclose(perm[1], plain[1], rtol=1e-5, atol=1e-5)       # (column sums of the stored G^T: another fp32 order)
