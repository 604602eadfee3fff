"""Golden vectors produced by the REFERENCE'S OWN model code (tests/golden/make_ref_wiring.py, through oracle/ref_shim.py).

CPU: the oracle reproduces them (pins the oracle's wiring -- BrushNet residual routing, from_unet, skip capture order).
GPU (-m gpu): the HIP path reproduces them within bf16 tolerance.
"""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ref_wiring as G  # noqa: E402

from oracle import sd_modules as OM  # noqa: E402

GOLD = torch.load(os.path.join(HERE, "golden", "ref_wiring.pt"), weights_only=False)


def _oracle_outputs():
    u9, u4 = G.oracle_models()
    inp = GOLD["inputs"]
    with torch.no_grad():
        eps9 = u9(inp["x9"], inp["t"], inp["ehs"])[0]
        ob = OM.randomize_zero_convs(OM.BrushNetModel.from_unet(u4).eval(), seed=11)
        dn, md, up = ob(inp["x4"], inp["t"], inp["ehs_b"], inp["cond"], conditioning_scale=inp["scale"])
        res = list(dn) + [md] + list(up)
        eps4b = u4(inp["x4"], inp["t"], inp["ehs"], down_block_add_samples=list(dn), mid_block_add_sample=md,
                   up_block_add_samples=list(up))[0]
        eps4p = u4(inp["x4"], inp["t"], inp["ehs"])[0]
    return u9, u4, ob, eps9, res, eps4b, eps4p


def test_oracle_reproduces_reference_outputs():
    _, _, _, eps9, res, eps4b, eps4p = _oracle_outputs()
    assert (GOLD["n_down"], GOLD["n_up"]) == (8, 11)
    assert [tuple(t.shape) for t in res] == GOLD["res_shapes"]
    assert torch.allclose(eps9, GOLD["eps9"], atol=2e-6, rtol=1e-5)
    assert torch.allclose(eps4p, GOLD["eps4_plain"], atol=2e-6, rtol=1e-5)
    assert torch.allclose(eps4b, GOLD["eps4_brush"], atol=2e-6, rtol=1e-5)
    assert not torch.allclose(eps4b, eps4p, atol=1e-3)          # the residuals really act
    summ = torch.stack([G.summary(t) for t in res])
    assert torch.allclose(summ, GOLD["res_summary"], atol=2e-6, rtol=1e-5)


@pytest.mark.gpu
def test_hip_path_reproduces_reference_outputs():
    from powerpaint_amd import models as PM
    u9, u4, ob, *_ = _oracle_outputs()
    inp = GOLD["inputs"]
    dev = "cuda"
    cfg = dict(GOLD["cfg"])
    cfg.pop("attention_head_dim")

    def close(out, ref, what):
        out, ref = out.float().cpu(), ref.float()
        cos = torch.nn.functional.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item()
        err = (out - ref).abs().max().item()
        assert cos >= 0.999 and err <= 3e-2 * max(1.0, ref.abs().max().item()), f"{what}: cos {cos:.6f} err {err:.4g}"

    h9 = PM.UNet2DConditionModel(in_channels=9, device=dev, **cfg).load_state_dict(u9.state_dict())
    close(h9(inp["x9"].to(dev), inp["t"], inp["ehs"].to(dev), return_dict=False)[0], GOLD["eps9"], "eps9")
    h4 = PM.UNet2DConditionModel(in_channels=4, device=dev, **cfg).load_state_dict(u4.state_dict(), keep_state_dict=True)
    close(h4(inp["x4"].to(dev), inp["t"], inp["ehs"].to(dev), return_dict=False)[0], GOLD["eps4_plain"], "eps4 plain")
    # from_unet on the HIP side, then the randomised zero-convs of the fixture
    hb0 = PM.BrushNetModel.from_unet(h4)
    d0, m0, u0 = hb0(inp["x4"].to(dev), inp["t"], inp["ehs_b"].to(dev), inp["cond"].to(dev), return_dict=False)
    assert all(float(t.float().abs().max()) == 0.0 for t in d0 + [m0] + u0)        # zero-convs -> exact zeros
    hb = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device=dev, **cfg).load_state_dict(ob.state_dict())
    dn, md, up = hb(inp["x4"].to(dev), inp["t"], inp["ehs_b"].to(dev), inp["cond"].to(dev),
                    conditioning_scale=inp["scale"], return_dict=False)
    assert (len(dn), len(up)) == (GOLD["n_down"], GOLD["n_up"])
    summ = torch.stack([G.summary(t.float().cpu()) for t in dn + [md] + up])
    assert torch.allclose(summ[:, :3], GOLD["res_summary"][:, :3], atol=2e-2, rtol=5e-2)
    out = h4(inp["x4"].to(dev), inp["t"], inp["ehs"].to(dev), down_block_add_samples=dn, mid_block_add_sample=md,
             up_block_add_samples=up, return_dict=False)[0]
    close(out, GOLD["eps4_brush"], "eps4 + brushnet residuals")


@pytest.mark.gpu
def test_mask_prep_matches_reference_function():
    """Row a20: `prepare_mask_and_masked_image` (binarise + mask the image in the HIP kernel) against the reference's
    own function (tests/golden/ref_mask_prep.json, lifted out of pipeline_PowerPaint.py:39-153 by AST): bit-exact mask,
    masked image and image for PIL, ndarray and tensor inputs."""
    import hashlib
    import json
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_ref_mask_prep import inputs
    from powerpaint_amd.pipelines._base import prepare_mask_and_masked_image

    def sha(t):
        a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
        return dict(shape=list(a.shape), sha=hashlib.sha256(a.tobytes()).hexdigest())

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mask_prep.json")) as f:
        G = json.load(f)
    for c in G["cases"]:
        img, msk = inputs(c["kind"], c["seed"], c["h"], c["w"], c["batch"])
        m, mi, im = prepare_mask_and_masked_image(img, msk, c["h"], c["w"], "cuda", return_image=True)
        assert m.is_cuda and sha(m) == c["mask"] and int(m.sum()) == c["ones"], c["kind"]
        assert sha(mi) == c["masked"] and sha(im) == c["image"], c["kind"]


def _ref_call_fixture():
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call.pt"), weights_only=False)
    return M, gold


def test_oracle_loop_reproduces_the_reference_call():
    """Rows a1 / a18 / a19: the oracle's restated loop (oracle/loops.py), fed by the oracle VAE and the transformers
    text encoder, against the final and per-step latents of the reference's OWN `StableDiffusionInpaintPipeline.__call__`
    (tests/golden/ref_pipeline_call.pt, produced through oracle/ref_pipeline.py) -- fp32 on CPU, same operation order."""
    from oracle import loops as OL, schedulers as OS
    M, gold = _ref_call_fixture()
    tok, enc, unet, vae = M.components()
    img, mask, lat = M.inputs()
    c = M.CALL

    def emb(p):
        ids = tok(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        return enc(ids)[0]

    with torch.no_grad():
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        mil = vae.encode(img * (mask < 0.5)).latent_dist.sample(torch.Generator().manual_seed(5)) * vae.config.scaling_factor
        m = torch.nn.functional.interpolate(mask, size=(16, 16))
        rec = []
        out = OL.loop_v1(unet, OS.DDIMScheduler(), lat, torch.cat([m] * 2), torch.cat([mil] * 2), torch.cat([neg, pos]),
                         c["num_inference_steps"], c["guidance_scale"],
                         eps_hook=lambda i, t, l, e: rec.append((i, int(t), l.clone())))
    assert [r[:2] for r in rec] == [s[:2] for s in gold["steps"]]
    for (i, t, l_in), (_, _, l_prev) in zip(rec[1:], gold["steps"][:-1]):      # latents entering step i = leaving step i-1
        assert torch.allclose(l_in, l_prev, atol=1e-4, rtol=1e-4), i
    assert torch.allclose(out, gold["latents"], atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
def test_hip_pipeline_reproduces_the_reference_call():
    """The product (HIP UNet / VAE / CLIP tower, fused DDIM + CFG loop) on the call the reference's own `__call__` was
    frozen on: strings and pixels in, latents out, bf16 against fp32."""
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    M, gold = _ref_call_fixture()
    tok, enc, unet, vae = M.components()
    img, mask, lat = M.inputs()
    hu = PM.UNet2DConditionModel(in_channels=9, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    pipe = PP.StableDiffusionInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu, scheduler=PS.DDIMScheduler())
    out = pipe(image=img, mask=mask, latents=lat.cuda(), generator=torch.Generator().manual_seed(5), output_type="latent",
               return_dict=False, **M.CALL)[0]
    want = gold["latents"]
    _close_latents(out, want, "v1 pipeline vs the reference's own __call__")


def test_oracle_loop_v2_reproduces_the_reference_brushnet_call():
    """Rows a2 / a16: the oracle's ppt-v2 loop (BrushNet residuals into the UNet, DPM-Solver++, two prompt encoders)
    against the final latents of the reference's OWN `StableDiffusionPowerPaintBrushNetPipeline.__call__`
    (tests/golden/ref_pipeline_call_v2.pt).  Reproduces its data flow exactly: the CFG twin of the image is VAE-encoded
    too (prepare_image duplicates before the encoder), with posterior noise from the global RNG."""
    from oracle import loops as OL, schedulers as OS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_v2.pt"), weights_only=False)
    tok, enc, unet, bn, vae = M.components_v2()
    img, mask3, lat = M.inputs_v2()
    c = M.CALL_V2

    def emb(p):
        ids = tok(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        return enc(ids)[0]

    with torch.no_grad():
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        peU = torch.cat([emb(c["negative_promptU"]), emb(c["promptU"])])
        torch.manual_seed(9)
        cl = vae.encode(torch.cat([img] * 2)).latent_dist.sample() * vae.config.scaling_factor
        keep = (torch.cat([mask3] * 2).sum(1)[:, None] < 0).float()
        cond = torch.cat([cl, torch.nn.functional.interpolate(keep, size=cl.shape[-2:])], 1)
        out = OL.loop_v2(unet, bn, OS.DPMSolverMultistepScheduler(), lat, cond, torch.cat([neg, pos]), peU,
                         c["num_inference_steps"], c["guidance_scale"], c["brushnet_conditioning_scale"])
    assert torch.allclose(out, gold["latents"], atol=2e-4, rtol=1e-4)


def test_oracle_loop_reproduces_the_reference_controlnet_call():
    """Rows a3 / a17: the oracle's v1 loop with a ControlNet against the final latents of the reference's OWN
    `StableDiffusionControlNetInpaintPipeline.__call__` (tests/golden/ref_pipeline_call_cn.pt)."""
    from oracle import loops as OL, schedulers as OS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_cn.pt"), weights_only=False)
    tok, enc, unet, cn, vae = M.components_cn()
    img, mask, lat = M.inputs()
    c = M.CALL_CN

    def emb(p):
        ids = tok(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        return enc(ids)[0]

    with torch.no_grad():
        pe = torch.cat([emb(c["negative_promptA"]), emb(c["promptA"])])           # tradoff = 1: promptA only
        mil = vae.encode(img * (mask < 0.5)).latent_dist.sample(torch.Generator().manual_seed(5)) * vae.config.scaling_factor
        m = torch.nn.functional.interpolate(mask, size=(16, 16))
        out = OL.loop_v1(unet, OS.DDIMScheduler(), lat, torch.cat([m] * 2), torch.cat([mil] * 2), pe,
                         c["num_inference_steps"], c["guidance_scale"], controlnet=cn,
                         control_image=torch.cat([M.control_image()] * 2),
                         controlnet_conditioning_scale=c["controlnet_conditioning_scale"])
    assert torch.allclose(out, gold["latents"], atol=2e-4, rtol=1e-4)


def _close_latents(out, want, what, cos_min=0.9997, rel=4.5e-2):
    """Free-running HIP pipeline against a frozen reference `__call__`: gates at twice the worst achieved error
    (profiles/r03_parity_achieved.txt: cosine 0.99987, max-abs 2.2e-2 of max|ref|)."""
    cos = torch.nn.functional.cosine_similarity(out.float().cpu().flatten(), want.flatten(), dim=0).item()
    err = (out.float().cpu() - want).abs().max().item()
    from test_models_gpu import _record_achieved
    _record_achieved("golden: " + what, cos, err, want.abs().max().item(), cos_min, rel * max(1.0, want.abs().max().item()))
    assert cos >= cos_min and err <= rel * max(1.0, want.abs().max().item()), (what, cos, err)


@pytest.mark.gpu
def test_hip_controlnet_pipeline_reproduces_the_reference_call():
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_cn.pt"), weights_only=False)
    tok, enc, unet, cn, vae = M.components_cn()
    img, mask, lat = M.inputs()
    no_up = {k: v for k, v in M.TINY.items() if k != "up_block_types"}
    hu = PM.UNet2DConditionModel(in_channels=9, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hc = PM.ControlNetModel(in_channels=4, device="cuda", **no_up).load_state_dict(cn.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    pipe = PP.StableDiffusionControlNetInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu, controlnet=hc,
                                                       scheduler=PS.DDIMScheduler())
    out = pipe(image=img, mask=mask, control_image=M.control_image(), latents=lat.cuda(),
               generator=torch.Generator().manual_seed(5), output_type="latent", return_dict=False, **M.CALL_CN)[0]
    _close_latents(out, gold["latents"], "controlnet pipeline vs the reference's own __call__")


@pytest.mark.gpu
def test_hip_brushnet_pipeline_reproduces_the_reference_call():
    """The reference samples the conditioning latents from the global CPU RNG; the same noise is applied here to the
    HIP VAE's posterior (mean, std), everything downstream is the product pipeline."""
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_v2.pt"), weights_only=False)
    tok, enc, unet, bn, vae = M.components_v2()
    img, mask3, lat = M.inputs_v2()
    hu = PM.UNet2DConditionModel(in_channels=4, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hb = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device="cuda", **M.TINY).load_state_dict(bn.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(vae=hv, text_encoder=he, text_encoder_brushnet=he, tokenizer=tok,
                                                        unet=hu, brushnet=hb, scheduler=PS.DPMSolverMultistepScheduler())
    dist = hv.encode(torch.cat([img] * 2).cuda()).latent_dist
    torch.manual_seed(9)
    noise = torch.randn(dist.mean.shape)                                       # CPU global RNG, as in the reference run
    cl = (dist.mean + dist.std * noise.cuda()) * hv.config.scaling_factor
    keep = (torch.cat([mask3] * 2).sum(1)[:, None] < 0).float()
    cond = torch.cat([cl, torch.nn.functional.interpolate(keep, size=cl.shape[-2:]).cuda()], 1)
    c = dict(M.CALL_V2)
    out = pipe(conditioning_latents=cond, latents=lat.cuda(), output_type="latent", return_dict=False, **c)[0]
    _close_latents(out, gold["latents"], "BrushNet pipeline vs the reference's own __call__")


# ------------------------------------------------------------------------------------------------------------------
# signature-visible variants of the loops: strength < 1 (enter the schedule late from the noised init image) and
# guess_mode (side network on the conditional half only), each frozen on the reference's OWN `__call__`
# ------------------------------------------------------------------------------------------------------------------
def _emb(tok, enc):
    def emb(p):
        ids = tok(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        return enc(ids)[0]
    return emb


def test_oracle_loop_reproduces_the_reference_call_with_strength():
    """`strength = 0.6`, 5 DPM-Solver++ steps -> entries 2..4 of the schedule, initial latents = add_noise(VAE posterior
    sample of the init image, noise, t = 499) (pipeline_PowerPaint.py:604-655,713-720); draw order posterior sample,
    noise, masked-image posterior sample."""
    from oracle import loops as OL, schedulers as OS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_strength.pt"), weights_only=False)
    tok, enc, unet, vae = M.components()
    img, mask, _ = M.inputs()
    c, emb = M.CALL_STRENGTH, _emb(tok, enc)
    with torch.no_grad():
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        sch = OS.DPMSolverMultistepScheduler(**M.DPM_SD15)
        sch.set_timesteps(c["num_inference_steps"])
        t_start = c["num_inference_steps"] - int(c["num_inference_steps"] * c["strength"])
        assert sch.timesteps[t_start:].tolist() == [s[1] for s in gold["steps"]] == [499, 333, 167]
        g = torch.Generator().manual_seed(5)
        il = vae.encode(img).latent_dist.sample(g) * vae.config.scaling_factor
        noise = torch.randn(1, 4, 16, 16, generator=g)
        lat = sch.add_noise(il, noise, sch.timesteps[t_start:t_start + 1])
        mil = vae.encode(img * (mask < 0.5)).latent_dist.sample(g) * vae.config.scaling_factor
        m = torch.nn.functional.interpolate(mask, size=(16, 16))
        out = OL.loop_v1(unet, sch, lat, torch.cat([m] * 2), torch.cat([mil] * 2), torch.cat([neg, pos]),
                         c["num_inference_steps"], c["guidance_scale"], t_start=t_start)
    assert torch.allclose(out, gold["latents"], atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
def test_hip_pipeline_reproduces_the_reference_call_with_strength():
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_strength.pt"), weights_only=False)
    tok, enc, unet, vae = M.components()
    img, mask, _ = M.inputs()
    hu = PM.UNet2DConditionModel(in_channels=9, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    pipe = PP.StableDiffusionInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu,
                                             scheduler=PS.DPMSolverMultistepScheduler(**M.DPM_SD15))
    seen = []
    out = pipe(image=img, mask=mask, generator=torch.Generator().manual_seed(5), output_type="latent", return_dict=False,
               callback=lambda i, t, l: seen.append(int(t)), **M.CALL_STRENGTH)[0]
    assert seen == [499, 333, 167]
    _close_latents(out, gold["latents"], "v1 pipeline, strength 0.6, vs the reference's own __call__")
    # a DUCK-TYPED scheduler (the oracle's class: not one of powerpaint_amd.schedulers, so `scheduler.step` runs as that
    # object's own code) enters the schedule late too: the networks must see timesteps[t_start:], the same values
    # `scheduler.step` receives (the loop's device timestep table held the HEAD of the schedule)
    from oracle import schedulers as OS
    duck = PP.StableDiffusionInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu,
                                             scheduler=OS.DPMSolverMultistepScheduler(**M.DPM_SD15))
    for use_graph in (True, False):
        duck.use_graph = use_graph
        seen2 = []
        out2 = duck(image=img, mask=mask, generator=torch.Generator().manual_seed(5), output_type="latent",
                    return_dict=False, callback=lambda i, t, l: seen2.append(int(t)), **M.CALL_STRENGTH)[0]
        assert seen2 == [499, 333, 167]
        assert [int(v) for v in duck._loop._f_ts[:3].cpu()] == [499, 333, 167]      # what the UNet was given
        _close_latents(out2, gold["latents"], "v1 pipeline, duck-typed scheduler, strength 0.6")
        cos = torch.nn.functional.cosine_similarity(out2.float().flatten(), out.float().flatten(), dim=0).item()
        # (same network program on both paths; a one-ulp flip of a 16-bit network input after ~1e-6 of scheduler
        # rounding difference is amplified by CFG: 0.99988 measured, see profiles/r03_duck_typed_rounding_flip.txt)
        assert cos >= 0.9997, ("duck-typed vs fused at strength 0.6", cos)
    # ... and the next full-strength call starts from the top of the schedule again
    _, _, lat = M.inputs()
    gold1 = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call.pt"), weights_only=False)
    pipe.scheduler = PS.DDIMScheduler()
    out = pipe(image=img, mask=mask, latents=lat.cuda(), generator=torch.Generator().manual_seed(5), output_type="latent",
               return_dict=False, **M.CALL)[0]
    _close_latents(out, gold1["latents"], "v1 pipeline after a strength < 1 call")


def test_oracle_loops_reproduce_the_reference_calls_in_guess_mode():
    """guess_mode under CFG (pipeline_PowerPaint_Brushnet_CA.py:1394-1425, pipeline_PowerPaint_ControlNet.py:1669-1702)."""
    from oracle import loops as OL, schedulers as OS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_v2_guess.pt"), weights_only=False)
    tok, enc, unet, bn, vae = M.components_v2()
    img, mask3, lat = M.inputs_v2()
    c, emb = M.CALL_V2, _emb(tok, enc)
    with torch.no_grad():
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        peU = torch.cat([emb(c["negative_promptU"]), emb(c["promptU"])])
        torch.manual_seed(9)
        cl = vae.encode(img).latent_dist.sample() * vae.config.scaling_factor         # NOT duplicated (:949)
        keep = (mask3.sum(1)[:, None] < 0).float()
        cond = torch.cat([cl, torch.nn.functional.interpolate(keep, size=cl.shape[-2:])], 1)
        out = OL.loop_v2(unet, bn, OS.DPMSolverMultistepScheduler(), lat, cond, torch.cat([neg, pos]), peU,
                         c["num_inference_steps"], c["guidance_scale"], c["brushnet_conditioning_scale"], guess_mode=True)
    assert torch.allclose(out, gold["latents"], atol=3e-4, rtol=1e-4)
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_cn_guess.pt"), weights_only=False)
    tok, enc, unet, cn, vae = M.components_cn()
    img, mask, lat = M.inputs()
    c, emb = M.CALL_CN, _emb(tok, enc)
    with torch.no_grad():
        pe = torch.cat([emb(c["negative_promptA"]), emb(c["promptA"])])
        mil = vae.encode(img * (mask < 0.5)).latent_dist.sample(torch.Generator().manual_seed(5)) * vae.config.scaling_factor
        m = torch.nn.functional.interpolate(mask, size=(16, 16))
        out = OL.loop_v1(unet, OS.DDIMScheduler(), lat, torch.cat([m] * 2), torch.cat([mil] * 2), pe,
                         c["num_inference_steps"], c["guidance_scale"], controlnet=cn, control_image=M.control_image(),
                         controlnet_conditioning_scale=c["controlnet_conditioning_scale"], guess_mode=True)
    assert torch.allclose(out, gold["latents"], atol=2e-4, rtol=1e-4)


@pytest.mark.gpu
def test_hip_side_pipelines_reproduce_the_reference_calls_in_guess_mode():
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    import make_ref_pipeline_call as M
    # ---- BrushNet
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_v2_guess.pt"), weights_only=False)
    tok, enc, unet, bn, vae = M.components_v2()
    img, mask3, lat = M.inputs_v2()
    hu = PM.UNet2DConditionModel(in_channels=4, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hb = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device="cuda", **M.TINY).load_state_dict(bn.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(vae=hv, text_encoder=he, text_encoder_brushnet=he, tokenizer=tok,
                                                        unet=hu, brushnet=hb, scheduler=PS.DPMSolverMultistepScheduler())
    dist = hv.encode(img.cuda()).latent_dist
    torch.manual_seed(9)
    noise = torch.randn(dist.mean.shape)
    cl = (dist.mean + dist.std * noise.cuda()) * hv.config.scaling_factor
    keep = (mask3.sum(1)[:, None] < 0).float()
    cond = torch.cat([cl, torch.nn.functional.interpolate(keep, size=cl.shape[-2:]).cuda()], 1)
    out = pipe(conditioning_latents=cond, latents=lat.cuda(), output_type="latent", return_dict=False, guess_mode=True,
               **M.CALL_V2)[0]
    _close_latents(out, gold["latents"], "BrushNet pipeline, guess_mode, vs the reference's own __call__")
    gold0 = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_v2.pt"), weights_only=False)
    assert (gold["latents"] - gold0["latents"]).abs().max() > 1.0        # (the two modes are far apart: a real check)
    # ---- ControlNet
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_cn_guess.pt"), weights_only=False)
    tok, enc, unet, cn, vae = M.components_cn()
    img, mask, lat = M.inputs()
    no_up = {k: v for k, v in M.TINY.items() if k != "up_block_types"}
    hu = PM.UNet2DConditionModel(in_channels=9, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hc = PM.ControlNetModel(in_channels=4, device="cuda", **no_up).load_state_dict(cn.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    pipe = PP.StableDiffusionControlNetInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu, controlnet=hc,
                                                       scheduler=PS.DDIMScheduler())
    out = pipe(image=img, mask=mask, control_image=M.control_image(), latents=lat.cuda(),
               generator=torch.Generator().manual_seed(5), output_type="latent", return_dict=False, guess_mode=True,
               **M.CALL_CN)[0]
    _close_latents(out, gold["latents"], "ControlNet pipeline, guess_mode, vs the reference's own __call__")


def test_oracle_loop_reproduces_the_reference_call_with_eta():
    """Stochastic DDIM: the loop's per-step variance noise comes from the generator that sampled the masked image's
    posterior (tests/golden/ref_pipeline_call_eta.pt, eta = 0.7)."""
    from oracle import loops as OL, schedulers as OS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_eta.pt"), weights_only=False)
    gold0 = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call.pt"), weights_only=False)
    assert (gold["latents"] - gold0["latents"]).abs().max() > 0.5
    tok, enc, unet, vae = M.components()
    img, mask, lat = M.inputs()
    c, emb = M.CALL_ETA, _emb(tok, enc)
    with torch.no_grad():
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        g = torch.Generator().manual_seed(5)
        mil = vae.encode(img * (mask < 0.5)).latent_dist.sample(g) * vae.config.scaling_factor
        m = torch.nn.functional.interpolate(mask, size=(16, 16))
        out = OL.loop_v1(unet, OS.DDIMScheduler(), lat, torch.cat([m] * 2), torch.cat([mil] * 2), torch.cat([neg, pos]),
                         c["num_inference_steps"], c["guidance_scale"], eta=c["eta"], generator=g)
    assert torch.allclose(out, gold["latents"], atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
def test_hip_pipeline_reproduces_the_reference_call_with_eta():
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    import make_ref_pipeline_call as M
    gold = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_eta.pt"), weights_only=False)
    tok, enc, unet, vae = M.components()
    img, mask, lat = M.inputs()
    hu = PM.UNet2DConditionModel(in_channels=9, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    pipe = PP.StableDiffusionInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu, scheduler=PS.DDIMScheduler())
    for use_graph in (True, False):                 # the noise buffer is refilled before every graph replay / eager step
        pipe.use_graph = use_graph
        out = pipe(image=img, mask=mask, latents=lat.cuda(), generator=torch.Generator().manual_seed(5),
                   output_type="latent", return_dict=False, **M.CALL_ETA)[0]
        _close_latents(out, gold["latents"], f"v1 pipeline, eta 0.7 (graph={use_graph}), vs the reference's own __call__")
    # eta back to 0 on the same pipeline object: the deterministic golden again (table and step program rebuilt)
    gold0 = torch.load(os.path.join(HERE, "golden", "ref_pipeline_call.pt"), weights_only=False)
    out = pipe(image=img, mask=mask, latents=lat.cuda(), generator=torch.Generator().manual_seed(5), output_type="latent",
               return_dict=False, **M.CALL)[0]
    _close_latents(out, gold0["latents"], "v1 pipeline, eta back to 0")


def test_product_prepare_latents_for_strength_matches_the_reference_draw_order():
    """CPU half of the `strength < 1` path: the product's `get_timesteps` + `_initial_latents` over the oracle VAE (duck
    typed) reproduce the initial latents the reference's `prepare_latents` built inside the frozen call -- posterior
    sample of the init image first, then the noise, `scheduler.add_noise` at the first timestep of the shortened schedule
    -- and leave the generator where the reference leaves it (the masked image's posterior sample comes next)."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    from powerpaint_amd.pipelines._base import PipelineBase
    import make_ref_pipeline_call as M
    _, _, _, vae = M.components()
    img, mask, _ = M.inputs()
    c = M.CALL_STRENGTH
    pb = PipelineBase.__new__(PipelineBase)
    pb.vae, pb.scheduler = vae, PS.DPMSolverMultistepScheduler(**M.DPM_SD15)
    pb.scheduler.set_timesteps(c["num_inference_steps"])
    ts, n = pb.get_timesteps(c["num_inference_steps"], c["strength"], "cpu")
    assert (n, ts.tolist(), pb.scheduler.begin_index) == (3, [499, 333, 167], 2)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        lat = pb._initial_latents((1, 4, 16, 16), c["strength"], ts, None, img, g, "cpu", torch.float32)
        after = torch.randn(3, generator=g)
        # the reference's own sequence on the oracle parts
        g2 = torch.Generator().manual_seed(5)
        il = vae.encode(img).latent_dist.sample(g2) * vae.config.scaling_factor
        noise = torch.randn(1, 4, 16, 16, generator=g2)
        o = OS.DPMSolverMultistepScheduler(**M.DPM_SD15)
        o.set_timesteps(c["num_inference_steps"])
        ref = o.add_noise(il, noise, o.timesteps[2:3])
    assert torch.allclose(lat, ref, atol=1e-5, rtol=1e-5) and torch.equal(after, torch.randn(3, generator=g2))
    # latents handed in: the reference treats them as the noise whatever the strength (:646-648)
    given = torch.randn(1, 4, 16, 16)
    assert torch.equal(pb._initial_latents((1, 4, 16, 16), 0.6, ts, given, img, g, "cpu", torch.float32), given)
    with pytest.raises(ValueError):
        pb._initial_latents((1, 4, 16, 16), 0.6, ts, None, None, g, "cpu", torch.float32)


def _four_channel_fixture():
    import make_ref_pipeline_call as M
    return M, torch.load(os.path.join(HERE, "golden", "ref_pipeline_call_4ch.pt"), weights_only=False)


def test_oracle_loop_reproduces_the_reference_call_with_a_4_channel_unet():
    """The `num_channels_unet == 4` branch (pipeline_PowerPaint.py:927-928,1025-1036; ControlNet pipeline :1612-1613,
    1725-1736): the oracle loop's restatement against the reference's own `__call__`s, per step and final.  Draw order
    from the one generator: init-image posterior sample, noise, masked-image posterior sample."""
    from oracle import loops as OL, schedulers as OS
    M, gold = _four_channel_fixture()
    tok, enc, unet, vae = M.components_4ch()
    img, mask, _ = M.inputs()
    emb = _emb(tok, enc)
    m = torch.nn.functional.interpolate(mask, size=(16, 16))
    with torch.no_grad():
        c = M.CALL_4CH
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        g = torch.Generator().manual_seed(5)
        il = vae.encode(img).latent_dist.sample(g) * vae.config.scaling_factor
        noise = torch.randn(1, 4, 16, 16, generator=g)
        rec = []
        out = OL.loop_v1(unet, OS.DDIMScheduler(), noise, torch.cat([m] * 2), None, torch.cat([neg, pos]),
                         c["num_inference_steps"], c["guidance_scale"], image_latents=il, noise=noise,
                         eps_hook=lambda i, t, l, e: rec.append((i, int(t), l.clone())))
        assert [r[:2] for r in rec] == [s[:2] for s in gold["steps"]] == [(0, 751), (1, 501), (2, 251), (3, 1)]
        for (i, t, l_in), (_, _, l_prev) in zip(rec[1:], gold["steps"][:-1]):
            assert torch.allclose(l_in, l_prev, atol=1e-4, rtol=1e-4), i
        assert torch.allclose(out, gold["latents"], atol=1e-4, rtol=1e-4)
        # the last step puts the CLEAN init latents back outside the mask
        keep = (m[0, 0] == 0)
        assert torch.equal(out[0][:, keep], il[0][:, keep])
        # ControlNet pipeline, DPM-Solver++, strength 0.8: schedule entered at entry 1 of 4
        c = M.CALL_4CH_CN
        _, _, _, cn, _ = M.components_cn()
        pos = emb(c["promptA"]) * c["tradoff"] + (1 - c["tradoff"]) * emb(c["promptB"])
        neg = emb(c["negative_promptA"]) * c["tradoff_nag"] + (1 - c["tradoff_nag"]) * emb(c["negative_promptB"])
        sch = OS.DPMSolverMultistepScheduler(**M.DPM_SD15)
        sch.set_timesteps(c["num_inference_steps"])
        t_start = c["num_inference_steps"] - int(c["num_inference_steps"] * c["strength"])
        assert sch.timesteps[t_start:].tolist() == [s[1] for s in gold["steps_cn"]] == [601, 401, 201]
        g = torch.Generator().manual_seed(6)
        il = vae.encode(img).latent_dist.sample(g) * vae.config.scaling_factor
        noise = torch.randn(1, 4, 16, 16, generator=g)
        lat = sch.add_noise(il, noise, sch.timesteps[t_start:t_start + 1])
        out = OL.loop_v1(unet, sch, lat, torch.cat([m] * 2), None, torch.cat([neg, pos]), c["num_inference_steps"],
                         c["guidance_scale"], controlnet=cn, control_image=torch.cat([M.control_image()] * 2),
                         controlnet_conditioning_scale=c["controlnet_conditioning_scale"], t_start=t_start,
                         image_latents=il, noise=noise)
        assert torch.allclose(out, gold["latents_cn"], atol=1e-4, rtol=1e-4)


@pytest.mark.gpu
def test_hip_pipelines_reproduce_the_reference_call_with_a_4_channel_unet():
    """VERDICT round 2, missing #5: the product's v1 and ControlNet pipelines with a plain 4-channel UNet (known region
    re-noised and blended back after every step: pp_latent_blend inside the captured step; a duck-typed scheduler takes
    the torch path with its own add_noise) against the reference's own `__call__`s."""
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    from oracle import schedulers as OS
    M, gold = _four_channel_fixture()
    tok, enc, unet, vae = M.components_4ch()
    img, mask, _ = M.inputs()
    hu = PM.UNet2DConditionModel(in_channels=4, device="cuda", **M.TINY).load_state_dict(unet.state_dict())
    hv = PM.AutoencoderKL(device="cuda", **M.VAE_CFG).load_state_dict(vae.state_dict())
    he = PM.CLIPTextModel(device="cuda", vocab_size=enc.config.vocab_size, num_hidden_layers=1,
                          eos_token_id=enc.config.eos_token_id)
    he.load_state_dict(enc.state_dict())
    outs = []
    for sch in (PS.DDIMScheduler(), OS.DDIMScheduler()):
        pipe = PP.StableDiffusionInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu, scheduler=sch)
        for use_graph in (True, False):
            pipe.use_graph = use_graph
            seen = []
            out = pipe(image=img, mask=mask, generator=torch.Generator().manual_seed(5), output_type="latent",
                       return_dict=False, callback=lambda i, t, l: seen.append((int(t), l.clone())), **M.CALL_4CH)[0]
            assert [s[0] for s in seen] == [751, 501, 251, 1]
            what = f"v1 pipeline, 4-channel UNet, {type(sch).__module__.split('.')[0]} scheduler, graph={use_graph}"
            _close_latents(seen[0][1], gold["steps"][0][2], what + " (after step 0)")
            # (four free-running CFG-7.5 steps of a random tiny network with the known region re-imposed every step:
            # achieved cosine 0.99980, profiles/r03_parity_achieved.txt -- gate at twice that error, like the others)
            _close_latents(out, gold["latents"], what, cos_min=0.9996)
            outs.append(out)
    # latent-space inputs cannot feed this branch (there is no init image to put back)
    with pytest.raises(ValueError, match="4-channel UNet needs the init image"):
        pipe(promptA="a", promptB="b", mask_latents=torch.zeros(1, 1, 16, 16), masked_image_latents=torch.zeros(1, 4, 16, 16),
             height=128, width=128, num_inference_steps=2, output_type="latent")
    _, _, _, cn, _ = M.components_cn()
    no_up = {k: v for k, v in M.TINY.items() if k != "up_block_types"}
    hc = PM.ControlNetModel(in_channels=4, device="cuda", **no_up).load_state_dict(cn.state_dict())
    pipe = PP.StableDiffusionControlNetInpaintPipeline(vae=hv, text_encoder=he, tokenizer=tok, unet=hu, controlnet=hc,
                                                       scheduler=PS.DPMSolverMultistepScheduler(**M.DPM_SD15))
    seen = []
    out = pipe(image=img, mask=mask, control_image=M.control_image(), generator=torch.Generator().manual_seed(6),
               output_type="latent", return_dict=False, callback=lambda i, t, l: seen.append(int(t)), **M.CALL_4CH_CN)[0]
    assert seen == [601, 401, 201]
    _close_latents(out, gold["latents_cn"], "controlnet pipeline, 4-channel UNet, strength 0.8")
