"""-m gpu: parity of the HEADLINE configurations over their whole schedule ("missing" 2 and 3).

  * config 2 (the configuration the benchmark's metric is quoted on): FIFTY free-running DDIM steps (CFG 7.5) of the full
    9-channel SD-1.5 UNet at 64x64 latents through the product's fused loop (hipGraph replay) against the CPU oracle's
    restatement of /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:988-1041 -- cosine, max-abs and the
    drift of the latents after steps 10 / 20 / 30 / 40 / 50;
  * config 3: TEN teacher-forced DPM-Solver++(2M) steps of full BrushNet + full UNet at 64x64
    (pipeline_PowerPaint_Brushnet_CA.py:1384-1466): per-step epsilon and scheduler output.

The oracle costs ~4 s (UNet) / ~9 s (BrushNet + UNet) of host time per step: 9 minutes for both cases, which does not fit
beside the rest of the suite in the driver's GPU test step.  By default the expected values therefore come from
tests/golden/headline_ref.pt -- the SAME oracle run once by tests/golden/make_headline_ref.py (inputs and oracle calls
shared through tests/headline_cases.py) -- and the file runs in seconds; PP_HEADLINE_LIVE=1 runs the oracle live as well
and checks it against the fixture (done on the GPU box in round 4: profiles/r04_parity_achieved.txt).
Achieved numbers are appended to gpurun_out/parity.txt; gates at ~2x achieved.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from powerpaint_amd import models as PM  # noqa: E402
from powerpaint_amd import pipelines as PP  # noqa: E402
from powerpaint_amd import schedulers as PS  # noqa: E402

import headline_cases as HC  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
LIVE = os.environ.get("PP_HEADLINE_LIVE", "0") == "1"
KEEP = (9, 19, 29, 39, 49)


def fixture():
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "headline_ref.pt"), weights_only=False)
    # the fixture matches only if torch.manual_seed + the nn initialisers reproduce the oracle's weights on THIS torch build
    # (ADVICE round 4): say so instead of failing an opaque parity gate after an upgrade
    tv = fx.get("torch_version") if isinstance(fx, dict) else None
    if tv is not None and str(tv).split("+")[0] != torch.__version__.split("+")[0]:
        pytest.skip(f"tests/golden/headline_ref.pt was generated with torch {tv}, this is {torch.__version__}: regenerate it "
                    f"(tests/golden/make_headline_ref.py) or run with PP_HEADLINE_LIVE=1")
    return fx


def record(line: str):
    print(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def measure(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    return F.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item(), (out - ref).abs().max().item()


def test_config2_fifty_free_running_ddim_steps_64x64():
    fx = fixture()
    o = HC.config2_oracle_model()          # (the seeded weights; no oracle forward unless LIVE)
    h = PM.UNet2DConditionModel(in_channels=9, device=DEV).load_state_dict(o.state_dict())
    inp = HC.config2_inputs()
    B, hh, N = 1, HC.HH, HC.CFG2_STEPS
    lat_after, scale = fx["cfg2_lat_after"], fx["cfg2_scale"]
    src = "committed oracle fixture"
    if LIVE:
        with torch.no_grad():
            live = HC.config2_oracle_run(o, inp)
        for k in KEEP:      # the fixture IS this oracle (CPU summation order aside)
            assert (live[k] - lat_after[k]).abs().max().item() <= 1e-3 * scale[k], k
        lat_after, src = {k: live[k] for k in KEEP}, "live oracle (= fixture within 1e-3)"
    pipe = PP.StableDiffusionInpaintPipeline(unet=h, scheduler=PS.DDIMScheduler())
    drift = {}

    def watch(i, t, latents):
        if i in lat_after:
            drift[i] = (latents.float().cpu() - lat_after[i]).abs().max().item()

    pe, mask, mil = inp["pe"], inp["mask"], inp["mil"]
    out = pipe(prompt_embeds=pe[B:].to(DEV), negative_prompt_embeds=pe[:B].to(DEV), height=hh * 8, width=hh * 8,
               num_inference_steps=N, guidance_scale=7.5, latents=inp["lat"].to(DEV), mask_latents=mask.to(DEV),
               masked_image_latents=mil.to(DEV), output_type="latent", return_dict=False, callback=watch,
               callback_steps=1)[0]
    assert sorted(drift) == list(KEEP)
    record(f"[headline parity] config 2, 50 free-running DDIM steps, 64x64 ({src}): max-abs latent drift after steps "
           "10/20/30/40/50 = " + " / ".join(f"{drift[k]:.4g} (on {scale[k]:.3g})" for k in KEEP))
    ref_final = lat_after[N - 1]
    cos, err = measure(out, ref_final)
    mx = ref_final.abs().max().item()
    record(f"[headline parity] config 2, 50 free-running DDIM steps, 64x64, final latents: cosine {cos:.7f}  max-abs {err:.4g}  "
           f"(max|ref| {mx:.4g}, relative {err / mx:.3g}); worst relative drift at the checkpoints "
           f"{max(drift[k] / scale[k] for k in KEEP):.3g}")
    assert torch.isfinite(out).all()
    # achieved on MI355X (profiles/r04_parity_achieved.txt): cosine 0.9999866, relative max-abs 7.2e-3, drift growing
    # 0.090 -> 0.41 on latents growing 11 -> 57 -- gates at ~2x achieved
    assert cos >= 0.99995 and err <= 0.015 * mx, (cos, err, mx)
    assert max(drift[k] / scale[k] for k in KEEP) <= 0.017


def test_config3_ten_teacher_forced_dpm_steps_64x64():
    fx = fixture()
    ou, ob = HC.config3_oracle_models()
    hu = PM.UNet2DConditionModel(in_channels=4, device=DEV).load_state_dict(ou.state_dict())
    hb = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device=DEV).load_state_dict(ob.state_dict())
    inp = HC.config3_inputs()
    B, N = 1, HC.CFG3_STEPS
    eps_ref, t_ref, lat_after = fx["cfg3_eps"], fx["cfg3_t"], fx["cfg3_lat_after"]
    if LIVE:
        with torch.no_grad():
            e_l, t_l, l_l = HC.config3_oracle_run(ou, ob, inp)
        assert t_l == t_ref
        assert (torch.stack(e_l) - eps_ref).abs().max().item() <= 1e-4 and \
            (torch.stack(l_l) - lat_after).abs().max().item() <= 1e-3 * lat_after.abs().max().item()
        eps_ref, lat_after = torch.stack(e_l), torch.stack(l_l)
    pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(unet=hu, brushnet=hb, scheduler=PS.DPMSolverMultistepScheduler())
    worst = {"eps_cos": 1.0, "eps_err": 0.0, "lat_cos": 1.0, "lat_rel": 0.0}

    def teacher(i, t, latents):
        assert int(t) == t_ref[i]
        ec, ee = measure(pipe._loop.rt.eps_tensor(), eps_ref[i])
        lc, le = measure(latents, lat_after[i])
        lm = lat_after[i].abs().max().item()
        record(f"[headline parity] config 3 teacher-forced DPM step {i} (t={int(t)}): eps cosine {ec:.7f} max-abs {ee:.4g} "
               f"(max|ref| {eps_ref[i].abs().max().item():.3g}); latents cosine {lc:.7f} max-abs {le:.4g} (max|ref| {lm:.3g})")
        worst["eps_cos"], worst["eps_err"] = min(worst["eps_cos"], ec), max(worst["eps_err"], ee)
        worst["lat_cos"], worst["lat_rel"] = min(worst["lat_cos"], lc), max(worst["lat_rel"], le / lm)
        if i + 1 < N:
            latents.copy_(lat_after[i].to(latents.device))          # teacher-force the next step of the fused loop

    pe, peU = inp["pe"], inp["peU"]
    out = pipe(prompt_embeds=pe[B:].to(DEV), negative_prompt_embeds=pe[:B].to(DEV), prompt_embedsU=peU[B:].to(DEV),
               negative_prompt_embedsU=peU[:B].to(DEV), conditioning_latents=inp["cl"].to(DEV), num_inference_steps=N,
               guidance_scale=7.5, latents=inp["lat"].to(DEV), output_type="latent", return_dict=False, callback=teacher,
               callback_steps=1)[0]
    assert torch.isfinite(out).all()
    record(f"[headline parity] config 3, 10 teacher-forced DPM-Solver++ steps, 64x64: worst eps cosine {worst['eps_cos']:.7f} / "
           f"max-abs {worst['eps_err']:.4g}; worst latents cosine {worst['lat_cos']:.7f} / relative max-abs {worst['lat_rel']:.3g}")
    # achieved on MI355X (profiles/r04_parity_achieved.txt): eps cosine 0.9999803 / max-abs 0.016; latents cosine
    # 0.9999458 / relative 0.0115 -- gates at ~2x achieved.  Teacher forcing replaces the latents only: from the second
    # step on the product's multistep history (its own previous x0 prediction) differs from the oracle's by its eps error
    assert worst["eps_cos"] >= 0.99995 and worst["eps_err"] <= 3.5e-2, worst
    assert worst["lat_cos"] >= 0.9998 and worst["lat_rel"] <= 2.5e-2, worst
