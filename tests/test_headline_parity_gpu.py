"""-m gpu: parity of the HEADLINE configurations over their whole schedule ("missing" 2 and 3).

  * config 2 (the configuration the benchmark's metric is quoted on): FIFTY free-running DDIM steps (CFG 7.5) of the full
    9-channel SD-1.5 UNet at 64x64 latents through the product's fused loop (hipGraph replay) against the CPU oracle's
    restatement of /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:988-1041 -- cosine, max-abs and the
    per-step drift of the latents;
  * config 3: TEN teacher-forced DPM-Solver++(2M) steps of full BrushNet + full UNet at 64x64
    (pipeline_PowerPaint_Brushnet_CA.py:1384-1466): per-step epsilon and scheduler output.

The oracle costs ~4 s (UNet) / ~9 s (BrushNet + UNet) of host time per step for one CFG pair: ~5 minutes for this file.
Achieved numbers are appended to gpurun_out/parity_r04.txt (-> profiles/r04_parity_achieved.txt); gates at ~2x achieved.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import loops as OL  # noqa: E402
from oracle import schedulers as OS  # noqa: E402
from oracle import sd_modules as OM  # noqa: E402
from powerpaint_amd import models as PM  # noqa: E402
from powerpaint_amd import pipelines as PP  # noqa: E402
from powerpaint_amd import schedulers as PS  # noqa: E402

from test_models_gpu import DEV, bf16_weights_, gen  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(line: str):
    print(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_r04.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def measure(out, ref):
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    return F.cosine_similarity(out.flatten(), ref.flatten(), dim=0).item(), (out - ref).abs().max().item()


def test_config2_fifty_free_running_ddim_steps_64x64():
    torch.manual_seed(8)
    o = bf16_weights_(OM.UNet2DConditionModel(in_channels=9)).eval()
    h = PM.UNet2DConditionModel(in_channels=9, device=DEV).load_state_dict(o.state_dict())
    B, hh, N = 1, 64, 50
    lat = gen(B, 4, hh, hh, seed=81)
    mask = torch.zeros(B, 1, hh, hh)
    mask[:, :, 16:48, 16:48] = 1.0
    mil = gen(B, 4, hh, hh, seed=82, scale=0.5)
    pe = gen(2 * B, 77, 768, seed=83)
    rec = []
    ref_final = OL.loop_v1(o, OS.DDIMScheduler(), lat, torch.cat([mask] * 2), torch.cat([mil] * 2), pe, N, 7.5,
                           eps_hook=lambda i, t, l, e: rec.append(l.clone()))
    lat_after = [rec[i + 1] for i in range(N - 1)] + [ref_final]
    pipe = PP.StableDiffusionInpaintPipeline(unet=h, scheduler=PS.DDIMScheduler())
    drift, scale = [], []

    def watch(i, t, latents):
        drift.append((latents.float().cpu() - lat_after[i]).abs().max().item())
        scale.append(lat_after[i].abs().max().item())

    out = pipe(prompt_embeds=pe[B:].to(DEV), negative_prompt_embeds=pe[:B].to(DEV), height=hh * 8, width=hh * 8,
               num_inference_steps=N, guidance_scale=7.5, latents=lat.to(DEV), mask_latents=mask.to(DEV),
               masked_image_latents=mil.to(DEV), output_type="latent", return_dict=False, callback=watch,
               callback_steps=1)[0]
    assert len(drift) == N
    record("[headline parity] config 2, 50 free-running DDIM steps, 64x64: max-abs latent drift after steps 10/20/30/40/50 = "
           + " / ".join(f"{drift[k]:.4g} (on {scale[k]:.3g})" for k in (9, 19, 29, 39, 49)))
    cos, err = measure(out, ref_final)
    mx = ref_final.abs().max().item()
    record(f"[headline parity] config 2, 50 free-running DDIM steps, 64x64, final latents: cosine {cos:.7f}  max-abs {err:.4g}  "
           f"(max|ref| {mx:.4g}, relative {err / mx:.3g}); worst relative drift over the schedule "
           f"{max(d / s for d, s in zip(drift, scale)):.3g}")
    assert torch.isfinite(out).all()
    # gates at ~2x achieved (profiles/r04_parity_achieved.txt)
    assert cos >= 0.9995 and err <= 0.06 * mx, (cos, err, mx)


def test_config3_ten_teacher_forced_dpm_steps_64x64():
    torch.manual_seed(4)
    ou = bf16_weights_(OM.UNet2DConditionModel(in_channels=4)).eval()
    hu = PM.UNet2DConditionModel(in_channels=4, device=DEV).load_state_dict(ou.state_dict())
    ob = bf16_weights_(OM.randomize_zero_convs(OM.BrushNetModel(in_channels=4, conditioning_channels=5))).eval()
    hb = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device=DEV).load_state_dict(ob.state_dict())
    B, hh, N = 1, 64, 10
    lat = gen(B, 4, hh, hh, seed=41)
    mask = torch.zeros(B, 1, hh, hh)
    mask[:, :, 16:48, 16:48] = 1.0
    cl = torch.cat([gen(B, 4, hh, hh, seed=42, scale=0.5), mask], 1)
    pe, peU = gen(2 * B, 77, 768, seed=43), gen(2 * B, 77, 768, seed=44)
    rec = []
    ref_final = OL.loop_v2(ou, ob, OS.DPMSolverMultistepScheduler(), lat, torch.cat([cl] * 2), pe, peU, N, 7.5, 1.0,
                           eps_hook=lambda i, t, l, e: rec.append((l.clone(), e.clone(), int(t))))
    lat_after = [rec[i + 1][0] for i in range(N - 1)] + [ref_final]
    pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(unet=hu, brushnet=hb, scheduler=PS.DPMSolverMultistepScheduler())
    worst = {"eps_cos": 1.0, "eps_err": 0.0, "lat_cos": 1.0, "lat_rel": 0.0}

    def teacher(i, t, latents):
        assert int(t) == rec[i][2]
        ec, ee = measure(pipe._loop.rt.eps_tensor(), rec[i][1])
        lc, le = measure(latents, lat_after[i])
        lm = lat_after[i].abs().max().item()
        record(f"[headline parity] config 3 teacher-forced DPM step {i} (t={int(t)}): eps cosine {ec:.7f} max-abs {ee:.4g} "
               f"(max|ref| {rec[i][1].abs().max().item():.3g}); latents cosine {lc:.7f} max-abs {le:.4g} (max|ref| {lm:.3g})")
        worst["eps_cos"], worst["eps_err"] = min(worst["eps_cos"], ec), max(worst["eps_err"], ee)
        worst["lat_cos"], worst["lat_rel"] = min(worst["lat_cos"], lc), max(worst["lat_rel"], le / lm)
        if i + 1 < N:
            latents.copy_(lat_after[i].to(latents.device))          # teacher-force the next step of the fused loop

    out = pipe(prompt_embeds=pe[B:].to(DEV), negative_prompt_embeds=pe[:B].to(DEV), prompt_embedsU=peU[B:].to(DEV),
               negative_prompt_embedsU=peU[:B].to(DEV), conditioning_latents=cl.to(DEV), num_inference_steps=N,
               guidance_scale=7.5, latents=lat.to(DEV), output_type="latent", return_dict=False, callback=teacher,
               callback_steps=1)[0]
    assert torch.isfinite(out).all()
    record(f"[headline parity] config 3, 10 teacher-forced DPM-Solver++ steps, 64x64: worst eps cosine {worst['eps_cos']:.7f} / "
           f"max-abs {worst['eps_err']:.4g}; worst latents cosine {worst['lat_cos']:.7f} / relative max-abs {worst['lat_rel']:.3g}")
    # gates at ~2x achieved (profiles/r04_parity_achieved.txt).  Teacher forcing replaces the latents only: from the second
    # step on the product's multistep history (its own previous x0 prediction) differs from the oracle's by its eps error
    assert worst["eps_cos"] >= 0.9999 and worst["eps_err"] <= 4e-2, worst
    assert worst["lat_cos"] >= 0.9998 and worst["lat_rel"] <= 4e-2, worst
