"""Inputs of the two headline-parity cases (tests/test_headline_parity_gpu.py) and the oracle runs that define their
expected values -- shared by the test (live oracle: PP_HEADLINE_LIVE=1) and by tests/golden/make_headline_ref.py, which
runs the same oracle once and commits what it produced (tests/golden/headline_ref.pt), so that the default `-m gpu` run
checks 50 free-running steps in seconds instead of ~9 minutes of host time.

config 2: /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:988-1041 x 50 (DDIM, CFG 7.5), full 9-channel UNet
config 3: /root/reference/powerpaint/pipelines/pipeline_PowerPaint_Brushnet_CA.py:1384-1466 x 10 (DPM-Solver++ 2M),
          full BrushNet + full UNet, teacher-forced on the oracle's latents
"""
import torch

from oracle import loops as OL
from oracle import schedulers as OS
from oracle import sd_modules as OM


def gen(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator("cpu").manual_seed(seed)) * scale


def bf16_weights_(m):
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    return m


CFG2_STEPS, CFG3_STEPS, HH = 50, 10, 64


def config2_oracle_model():
    torch.manual_seed(8)
    return bf16_weights_(OM.UNet2DConditionModel(in_channels=9)).eval()


def config2_inputs():
    B, hh = 1, HH
    mask = torch.zeros(B, 1, hh, hh)
    mask[:, :, 16:48, 16:48] = 1.0
    return dict(lat=gen(B, 4, hh, hh, seed=81), mask=mask, mil=gen(B, 4, hh, hh, seed=82, scale=0.5),
                pe=gen(2 * B, 77, 768, seed=83))


def config2_oracle_run(o, inp, steps=CFG2_STEPS):
    """-> list of the latents AFTER each step (the last one = the loop's result)."""
    rec = []
    final = OL.loop_v1(o, OS.DDIMScheduler(), inp["lat"], torch.cat([inp["mask"]] * 2), torch.cat([inp["mil"]] * 2), inp["pe"],
                       steps, 7.5, eps_hook=lambda i, t, l, e: rec.append(l.clone()))
    return [rec[i + 1] for i in range(steps - 1)] + [final]


def config3_oracle_models():
    torch.manual_seed(4)
    ou = bf16_weights_(OM.UNet2DConditionModel(in_channels=4)).eval()
    ob = bf16_weights_(OM.randomize_zero_convs(OM.BrushNetModel(in_channels=4, conditioning_channels=5))).eval()
    return ou, ob


def config3_inputs():
    B, hh = 1, HH
    mask = torch.zeros(B, 1, hh, hh)
    mask[:, :, 16:48, 16:48] = 1.0
    return dict(lat=gen(B, 4, hh, hh, seed=41), cl=torch.cat([gen(B, 4, hh, hh, seed=42, scale=0.5), mask], 1),
                pe=gen(2 * B, 77, 768, seed=43), peU=gen(2 * B, 77, 768, seed=44))


def config3_oracle_run(ou, ob, inp, steps=CFG3_STEPS):
    """-> (eps [2B,4,h,w] per step, timestep per step, latents after each step)."""
    rec = []
    final = OL.loop_v2(ou, ob, OS.DPMSolverMultistepScheduler(), inp["lat"], torch.cat([inp["cl"]] * 2), inp["pe"], inp["peU"],
                       steps, 7.5, 1.0, eps_hook=lambda i, t, l, e: rec.append((l.clone(), e.clone(), int(t))))
    lat_after = [rec[i + 1][0] for i in range(steps - 1)] + [final]
    return [r[1] for r in rec], [r[2] for r in rec], lat_after
