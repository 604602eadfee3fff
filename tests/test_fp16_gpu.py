"""-m gpu: the fp16 compute path (BASELINE config 5 is fp16; fp16 is the reference's default dtype,
/root/reference/app.py:548,559: `weight_dtype = torch.float16`, `torch_dtype=weight_dtype` at :92,145-176).

Same kernels, second instantiation: `v_mfma_f32_*_f16`, fp16 pack / unpack, fp32 accumulation, statistics, softmax and
latents.  Checked against fp32 torch / the fp32 CPU oracle on fp16-rounded inputs and weights at a tolerance FOUR TIMES
TIGHTER than the bf16 gates (10 mantissa bits against 7) -- "within fp16 tolerance" (BASELINE.json north_star):
  op level      : atol 5e-3, rtol 2.5e-3      (bf16: 2e-2 / 1e-2)
  network level : cosine >= 0.99999, max-abs <= 7.5e-3 * max(1, max|ref|)      (bf16: 0.999 / 3e-2)
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import loops as OL  # noqa: E402
from oracle import schedulers as OS  # noqa: E402
from oracle import sd_modules as OM  # noqa: E402
from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd import models as PM  # noqa: E402
from powerpaint_amd import ops  # noqa: E402
from powerpaint_amd import pipelines as PP  # noqa: E402
from powerpaint_amd import schedulers as PS  # noqa: E402

from test_models_gpu import TINY, close, gen  # noqa: E402
from test_ops_gpu import check, conv_ref, rnd  # noqa: E402

DEV = "cuda"
H16 = torch.float16


def hf(t):
    return t.to(H16)


def fp16_weights_(m):
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(p.to(H16).float())
    return m


# ------------------------------------------------------------------------------------------------ ops
@pytest.mark.parametrize("tile", [0, 2, 21, 31, 32, 24, 33, 53, 54])
def test_gemm_fp16(tile):
    M, N, K = 512, 640, 1280
    x, w = hf(rnd(M, K, seed=1)), hf(rnd(N, K, seed=2, scale=K ** -0.5))
    bias, res = rnd(N, seed=3), hf(rnd(M, N, seed=4))
    out = ops.gemm(x, w, bias=bias, res1=res, tile=tile, splitk=1 if tile else 0)
    assert out.dtype == H16
    check(out, x.float() @ w.float().t() + bias + res.float(), 5e-3, 2.5e-3, f"fp16 gemm tile{tile}")
    # the same bits through the bf16 path differ (the kernels really interpret the storage format)
    assert not torch.equal(out.view(torch.int16), ops.gemm(x.view(torch.bfloat16), w.view(torch.bfloat16), tile=tile,
                                                           splitk=1 if tile else 0).view(torch.int16))


@pytest.mark.parametrize("tile,splitk", [(0, 0), (22, 1), (33, 2), (53, 1), (54, 2)])
@pytest.mark.parametrize("stride,up", [(1, False), (2, False), (1, True)])
def test_conv3x3_fp16(tile, splitk, stride, up):
    B, H, W, Cin, Cout = 2, 16, 16, 320, 320
    x = hf(rnd(B, H, W, Cin, seed=1))
    w = hf(rnd(Cout, 9 * Cin, seed=2, scale=(9 * Cin) ** -0.5))
    bias = rnd(Cout, seed=3)
    out = ops.conv3x3(x, w, bias, stride=stride, up=up, tile=tile, splitk=splitk)
    check(out, conv_ref(x, w, bias, stride, up), 5e-3, 2.5e-3, f"fp16 conv s{stride} up{up} tile{tile}")


def test_geglu_folded_layernorm_and_gn_stats_fp16():
    from powerpaint_amd.engine import _geglu_interleave
    M, C = 512, 320
    x = hf(rnd(M, C, seed=1))
    w = hf(rnd(8 * C, C, seed=2, scale=C ** -0.5))
    b = rnd(8 * C, seed=3)
    out = ops.gemm(x, _geglu_interleave(w).contiguous(), bias=_geglu_interleave(b).contiguous(), act=L.PP_ACT_GEGLU)
    y = x.float() @ w.float().t() + b
    h, g = y.chunk(2, -1)
    check(out, h * F.gelu(g), 5e-3, 2.5e-3, "fp16 geglu")
    # producer-side row moments of the STORED fp16 values
    w2, bias = hf(rnd(320, C, seed=4, scale=C ** -0.5)), rnd(320, seed=5)
    o, st = ops.gemm(x, w2, bias=bias, row_stats=True)
    of = o.float()
    ref = torch.stack([of.reshape(M, 2, 160).sum(-1), (of * of).reshape(M, 2, 160).sum(-1)], -1)
    assert torch.allclose(st, ref, rtol=1e-4, atol=1e-3)
    # GroupNorm statistics in the conv epilogue (integer accumulators) and the apply kernel on fp16
    B, Hh = 2, 16
    xc = hf(rnd(B, Hh, Hh, 320, seed=6))
    wc = hf(rnd(320, 9 * 320, seed=7, scale=(9 * 320) ** -0.5))
    acc = torch.zeros(B, 32, 2, dtype=torch.int64, device=DEV)
    oc = ops.conv3x3(xc, wc, None, gn=[(acc, 10, 0, 32)])
    gamma, beta = rnd(320, seed=8).abs() + 0.5, rnd(320, seed=9)
    got = ops.groupnorm_apply_acc(oc, acc, gamma, beta, 1e-5, True)
    ref = F.silu(F.group_norm(oc.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    check(got, ref, 5e-3, 2.5e-3, "fp16 groupnorm from epilogue statistics")
    check(ops.groupnorm(oc, gamma, beta, 1e-5, True), ref, 5e-3, 2.5e-3, "fp16 groupnorm (stats + apply)")
    check(ops.layernorm(x, gamma, beta), F.layer_norm(x.float(), (C,), gamma, beta), 5e-3, 2.5e-3, "fp16 layernorm")


@pytest.mark.parametrize("d,nq,nk", [(40, 4096, 4096), (40, 1024, 77), (80, 1024, 1024), (160, 256, 77)])
def test_attention_fp16(d, nq, nk):
    B, heads = 1, 8
    Cc = heads * d
    q, k, v = hf(rnd(B * nq, Cc, seed=1)), hf(rnd(B * nk, Cc, seed=2)), hf(rnd(B * nk, Cc, seed=3))
    vt = ops.transpose_v(v, B, nk)
    o = ops.attention(q, k, vt, B, heads, nq, nk, d)
    qf = q.float().view(B, nq, heads, d).transpose(1, 2)
    kf = k.float().view(B, nk, heads, d).transpose(1, 2)
    vf = v.float().view(B, nk, heads, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(B * nq, Cc)
    check(o, ref, 5e-3, 2.5e-3, f"fp16 attention d{d} {nq}x{nk}")


# ------------------------------------------------------------------------------------------------ networks
def make_tiny16(kind, seed=0, **extra):
    torch.manual_seed(seed)
    cfg = TINY if kind != "controlnet" else {k: v for k, v in TINY.items() if k != "up_block_types"}
    if kind == "unet":
        o = OM.UNet2DConditionModel(in_channels=extra.pop("in_channels", 9), **cfg)
        h = PM.UNet2DConditionModel(in_channels=o.config.in_channels, device=DEV, dtype=H16, **cfg)
    elif kind == "brushnet":
        o = OM.randomize_zero_convs(OM.BrushNetModel(in_channels=4, conditioning_channels=5, **cfg))
        h = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device=DEV, dtype=H16, **cfg)
    else:
        o = OM.randomize_zero_convs(OM.ControlNetModel(in_channels=4, **cfg))
        h = PM.ControlNetModel(in_channels=4, device=DEV, dtype=H16, **cfg)
    fp16_weights_(o).eval()
    h.load_state_dict(o.state_dict())
    return o, h


def close16(out, ref, what, cos_min=0.99999, rel=7.5e-3):
    cos, err = close(out, ref, what, cos_min=cos_min, rel=rel)
    print(f"[fp16 parity] {what}: cosine {cos:.7f}  max-abs {err:.4g}  (max|ref| {float(ref.abs().max()):.4g})")


def test_unet_brushnet_controlnet_forward_fp16():
    o, h = make_tiny16("unet", in_channels=9)
    assert h.dtype == H16
    x, e = gen(2, 9, 16, 16, seed=1), gen(2, 77, 768, seed=2)
    with torch.no_grad():
        ref = o(x, 500, e)[0]
    close16(h(x.to(DEV), 500, e.to(DEV), return_dict=False)[0], ref, "fp16 unet tiny")
    ob, hb = make_tiny16("brushnet")
    ou, hu = make_tiny16("unet", seed=1, in_channels=4)
    x4, eu, cond = gen(2, 4, 16, 16, seed=1), gen(2, 77, 768, seed=3), gen(2, 5, 16, 16, seed=4)
    with torch.no_grad():
        dn, md, up = ob(x4, 321, e, cond, conditioning_scale=0.8)
        ref = ou(x4, 321, eu, down_block_add_samples=list(dn), mid_block_add_sample=md, up_block_add_samples=list(up))[0]
    hdn, hmd, hup = hb(x4.to(DEV), 321, e.to(DEV), cond.to(DEV), conditioning_scale=0.8, return_dict=False)
    assert hdn[0].dtype == H16
    for i, (a, b) in enumerate(zip(hdn + [hmd] + hup, list(dn) + [md] + list(up))):
        close(a, b, f"fp16 brushnet residual {i}", cos_min=0.9999, rel=7.5e-3)
    out = hu(x4.to(DEV), 321, eu.to(DEV), down_block_add_samples=list(hdn), mid_block_add_sample=hmd,
             up_block_add_samples=list(hup), return_dict=False)[0]
    close16(out, ref, "fp16 brushnet -> unet tiny")
    oc, hc = make_tiny16("controlnet")
    img = torch.rand(2, 3, 128, 128, generator=torch.Generator("cpu").manual_seed(3))
    with torch.no_grad():
        dn, md = oc(x4, 700, e, img, conditioning_scale=0.5)
    hdn, hmd = hc(x4.to(DEV), 700, e.to(DEV), img.to(DEV), conditioning_scale=0.5, return_dict=False)
    for i, (a, b) in enumerate(zip(hdn + [hmd], list(dn) + [md])):
        close(a, b, f"fp16 controlnet residual {i}", cos_min=0.9999, rel=7.5e-3)


def test_full_unet_fp16_32x32_vs_oracle():
    """The real SD-1.5 inpainting architecture in fp16 at 32x32, CFG pair."""
    torch.manual_seed(0)
    o = fp16_weights_(OM.UNet2DConditionModel(in_channels=9)).eval()
    h = PM.UNet2DConditionModel(in_channels=9, device=DEV, dtype=H16).load_state_dict(o.state_dict())
    x, e = gen(2, 9, 32, 32, seed=1), gen(2, 77, 768, seed=2)
    with torch.no_grad():
        ref = o(x, 981, e)[0]
    close16(h(x.to(DEV), 981, e.to(DEV), return_dict=False)[0], ref, "fp16 SD-1.5 UNet 32x32")


def test_pipeline_v2_fp16_dpm_loop():
    """ppt-v2 (BrushNet + UNet, DPM-Solver++) in fp16: free-running 4 steps against the oracle loop, graph == eager."""
    ob, hb = make_tiny16("brushnet")
    ou, hu = make_tiny16("unet", seed=1, in_channels=4)
    B, hh, N = 2, 16, 4
    lat = gen(B, 4, hh, hh, seed=0)
    mask = torch.zeros(B, 1, hh, hh)
    mask[:, :, 4:12, 4:12] = 1.0
    cl = torch.cat([gen(B, 4, hh, hh, seed=1, scale=0.5), mask], 1)
    pe, peU = gen(2 * B, 77, 768, seed=2), gen(2 * B, 77, 768, seed=3)
    ref = OL.loop_v2(ou, ob, OS.DPMSolverMultistepScheduler(), lat, torch.cat([cl] * 2), pe, peU, N, 7.5, 1.0)
    pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(unet=hu, brushnet=hb, scheduler=PS.DPMSolverMultistepScheduler())
    kw = dict(prompt_embeds=pe[B:].to(DEV), negative_prompt_embeds=pe[:B].to(DEV), prompt_embedsU=peU[B:].to(DEV),
              negative_prompt_embedsU=peU[:B].to(DEV), conditioning_latents=cl.to(DEV), num_inference_steps=N,
              guidance_scale=7.5, latents=lat.to(DEV), output_type="latent", return_dict=False)
    out = pipe(**kw)[0]
    close16(out, ref, "fp16 v2 free-running, 4 DPM-Solver++ steps", cos_min=0.9999, rel=2.5e-2)   # (bf16 gate: 0.9997 / 4.5e-2)
    pipe.use_graph = False
    assert torch.equal(out, pipe(**kw)[0])


def test_mixing_dtypes_is_refused():
    with pytest.raises(L.PPError):
        PM.UNet2DConditionModel(in_channels=4, device=DEV, dtype=torch.float32, **TINY)
