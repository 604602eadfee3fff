"""CPU: pins for the oracle itself (it has no reference golden vectors to lean on -- see oracle/__init__.py)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loops as OL
from oracle import schedulers as OS
from oracle import sd_modules as OM


def test_parameter_counts_match_sd15():
    """UNet 859.5 M is the well-known SD-1.5 figure; BrushNet_CA / ControlNet from the constructor loops."""
    with torch.device("meta"):
        assert OM.count_params(OM.UNet2DConditionModel(in_channels=4)) == 859_520_964
        assert OM.count_params(OM.UNet2DConditionModel(in_channels=9)) == 859_535_364
        assert round(OM.count_params(OM.BrushNetModel()) / 1e6, 1) == 886.1
        assert round(OM.count_params(OM.ControlNetModel()) / 1e6, 1) == 361.3


def test_state_dict_keys_follow_diffusers_naming():
    with torch.device("meta"):
        sd = OM.BrushNetModel().state_dict()
    for k in ["conv_in_condition.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.norm1.weight",
              "down_blocks.0.attentions.1.transformer_blocks.0.attn2.to_k.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
              "down_blocks.2.downsamplers.0.conv.weight", "mid_block.attentions.0.proj_out.bias",
              "up_blocks.1.resnets.2.conv_shortcut.weight", "up_blocks.2.upsamplers.0.conv.bias",
              "brushnet_down_blocks.11.weight", "brushnet_mid_block.bias", "brushnet_up_blocks.14.weight"]:
        assert k in sd, k
    assert "down_blocks.3.downsamplers.0.conv.weight" not in sd and "up_blocks.3.upsamplers.0.conv.weight" not in sd


def test_timestep_arrays():
    d = OS.DDIMScheduler(); d.set_timesteps(50)
    assert d.timesteps[:3].tolist() == [981, 961, 941] and d.timesteps[-1].item() == 1
    assert np.array_equal(d.timesteps.numpy(), OS.ddim_timesteps(50))
    p = OS.DPMSolverMultistepScheduler(); p.set_timesteps(50)
    assert p.timesteps[:3].tolist() == [999, 979, 959] and p.timesteps[-1].item() == 20
    p.set_timesteps(30)
    assert np.array_equal(p.timesteps.numpy(), OS.dpm_timesteps(30))


@pytest.mark.parametrize("N", [10, 30, 50])
def test_ddim_torch_vs_float64_closed_form(N):
    s = OS.DDIMScheduler(); s.set_timesteps(N)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 8, generator=g)
    ref = x.double().numpy()
    for t in s.timesteps:
        e = torch.randn(2, 4, 8, 8, generator=g)
        x = s.step(e, t, x)[0]
        ref = OS.ddim_step_f64(ref, e.double().numpy(), int(t), N)
    assert np.allclose(x.numpy(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("N", [10, 30, 50])
def test_dpm_torch_vs_float64_closed_form(N):
    s = OS.DPMSolverMultistepScheduler(); s.set_timesteps(N)
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(N)]
    x = x0
    for t, e in zip(s.timesteps, eps):
        x = s.step(e, t, x)[0]
    ref = OS.dpm_run_f64(x0.double().numpy(), [e.double().numpy() for e in eps], N)
    assert np.allclose(x.numpy(), ref, rtol=5e-4, atol=5e-4)
    # DPM-Solver++ is exact for a constant data prediction: x0 fixed -> final latent == x0
    s.set_timesteps(N)
    x, target = x0, torch.randn(2, 4, 8, 8, generator=g)
    for i, t in enumerate(s.timesteps):
        a, sg = s._alpha_sigma(s.sigmas[i])
        x = s.step((x - a * target) / sg, t, x)[0]
    assert torch.allclose(x, target, atol=1e-4)


@pytest.mark.parametrize("N", [4, 10, 50])
def test_pndm_torch_vs_float64_closed_form(N):
    """PLMS (PNDMScheduler(skip_prk_steps=True), the SD-1.5 checkpoint's scheduler): N steps = N + 1 evaluations, the
    second timestep repeats; the diffusers-protocol class against an independent float64 derivation."""
    s = OS.PNDMScheduler(); s.set_timesteps(N)
    ts = s.timesteps.tolist()
    assert len(ts) == N + 1 and ts[1] == ts[2] and ts == OS.pndm_timesteps(N).tolist()
    assert ts[0] == (N - 1) * (1000 // N) + 1 and ts[-1] == 1
    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    eps = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(N + 1)]
    x = x0
    for t, e in zip(s.timesteps, eps):
        x = s.step(e, t, x)[0]
    ref = OS.pndm_run_f64(x0.double().numpy(), [e.double().numpy() for e in eps], N)
    assert np.allclose(x.numpy(), ref, rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(ref).max())))
    # a constant noise prediction makes every multistep combination that prediction and every transfer deterministic DDIM
    s.set_timesteps(N)
    e = torch.randn(2, 4, 8, 8, generator=g)
    x = x0
    for t in s.timesteps:
        x = s.step(e, t, x)[0]
    d = OS.DDIMScheduler(); d.set_timesteps(N)
    y = x0
    for t in d.timesteps:
        y = d.step(e, t, y)[0]
    assert torch.allclose(x, y, rtol=1e-3, atol=1e-3)


def test_leaf_modules_against_functional():
    torch.manual_seed(0)
    a = OM.Attention(320, 768, 8, 40)
    x, c = torch.randn(2, 64, 320), torch.randn(2, 77, 768)
    q = a.to_q(x).view(2, 64, 8, 40).transpose(1, 2)
    k = a.to_k(c).view(2, 77, 8, 40).transpose(1, 2)
    v = a.to_v(c).view(2, 77, 8, 40).transpose(1, 2)
    ref = a.to_out[0](F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(2, 64, 320))
    assert torch.allclose(a(x, c), ref, atol=1e-5)
    e = OM.timestep_embedding(torch.tensor([981]), 320)
    f = torch.exp(-math.log(10000) * torch.arange(160) / 160)
    assert torch.allclose(e[0, :160], torch.cos(981 * f), atol=1e-6) and torch.allclose(e[0, 160:], torch.sin(981 * f), atol=1e-6)
    ff = OM.FeedForward(320)
    y = ff.net[0].proj(x)
    assert torch.allclose(ff(x), ff.net[2](y[..., :1280] * F.gelu(y[..., 1280:])), atol=1e-6)


TINY = dict(block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))


def test_brushnet_routing_properties():
    torch.manual_seed(0)
    u = OM.UNet2DConditionModel(in_channels=4, **TINY).eval()
    b = OM.BrushNetModel.from_unet(u).eval()
    x, e, cond = torch.randn(1, 4, 16, 16), torch.randn(1, 77, 768), torch.randn(1, 5, 16, 16)
    with torch.no_grad():
        dn, md, up = b(x, 10, e, cond)
        assert len(dn) == 4 and len(up) == 5
        assert all(float(t.abs().max()) == 0 for t in dn + [md] + up)           # zero-convs: exact zeros (:955-958)
        base = u(x, 10, e)[0]
        same = u(x, 10, e, down_block_add_samples=list(dn), mid_block_add_sample=md, up_block_add_samples=list(up))[0]
        assert torch.equal(base, same)
        # from_unet copies conv_in into channels 0-3 and 4-7, zero for channel 8 (BrushNet_CA.py:525-540)
        w = b.conv_in_condition.weight
        assert torch.equal(w[:, :4], u.conv_in.weight) and torch.equal(w[:, 4:8], u.conv_in.weight)
        assert float(w[:, 8].abs().max()) == 0
        # the residual list is consumed destructively and completely
        OM.randomize_zero_convs(b)
        dn, md, up = b(x, 10, e, cond, conditioning_scale=2.0)
        d2, _, _ = b(x, 10, e, cond, conditioning_scale=1.0)
        assert torch.allclose(dn[1], 2 * d2[1], atol=1e-6)
        ld, lu = list(dn), list(up)
        u(x, 10, e, down_block_add_samples=ld, mid_block_add_sample=md, up_block_add_samples=lu)
        assert ld == [] and lu == []


def test_bit_exact_prep_and_cfg_order():
    g = torch.Generator().manual_seed(0)
    m = torch.rand(2, 1, 16, 16, generator=g)
    mb = OL.binarize_mask(m)
    assert set(mb.unique().tolist()) <= {0.0, 1.0} and torch.equal(mb, (m >= 0.5).float())
    img = torch.rand(2, 3, 16, 16, generator=g) * 2 - 1
    assert torch.equal(OL.masked_image(img, mb), img * (1 - mb))
    assert torch.equal(OL.mask_to_latent(mb, 2, 2), mb[:, :, ::8, ::8])
    rgb = torch.stack([-torch.ones(16, 16), -torch.ones(16, 16), -torch.ones(16, 16)])[None]
    assert OL.brushnet_original_mask(rgb).min() == 1.0 and OL.brushnet_original_mask(-rgb).max() == 0.0
    a, b = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    assert torch.equal(OL.blend_prompt_embeds(a, b, 1.0), a * 1.0 + 0.0 * b)


# ------------------------------------------------------------------------------------------------ UniPC
def _gaussian_ode_errors(make, Ns, c2=0.25, frac=0.6):
    """Data ~ N(0, c2 I): the optimal eps model is sigma_t x / (alpha_t^2 c2 + sigma_t^2) and the probability-flow ODE
    has the closed-form solution x_t = x_T sqrt(alpha_t^2 c2 + sigma_t^2) / sqrt(alpha_T^2 c2 + sigma_T^2).  The error
    is read after `frac` of the steps: the last steps towards t = 0 have log-SNR steps that do not shrink with N."""
    g = torch.Generator().manual_seed(0)
    x_T = torch.randn(4, 4, 8, 8, generator=g, dtype=torch.float64)
    errs = []
    for N in Ns:
        s = make()
        s.set_timesteps(N)
        x = x_T.clone()
        sig = s.sigmas.double()
        stop = int(N * frac)
        for i, t in enumerate(s.timesteps[:stop]):
            a, sg = s._alpha_sigma(sig[i])
            x = s.step(sg * x / (a * a * c2 + sg * sg), t, x)[0]
        a0, s0 = s._alpha_sigma(sig[0])
        a1, s1 = s._alpha_sigma(sig[stop])
        exact = x_T * torch.sqrt(a1 * a1 * c2 + s1 * s1) / torch.sqrt(a0 * a0 * c2 + s0 * s0)
        errs.append(float((x - exact).abs().max()))
    return errs


def test_unipc_structure_pins():
    """diffusers is not installable, so the UniPC restatement is pinned to the paper: UniP-1 is DDIM, the grid matches
    the library's spacing rules, and the order of convergence on a closed-form ODE is what UniPC-p promises."""
    u = OS.UniPCMultistepScheduler(timestep_spacing="leading", steps_offset=1)
    u.set_timesteps(10)
    assert u.timesteps.tolist() == [901, 811, 721, 631, 541, 451, 361, 271, 181, 91]
    u2 = OS.UniPCMultistepScheduler()
    u2.set_timesteps(10)
    assert u2.timesteps.tolist() == [999, 899, 799, 699, 599, 500, 400, 300, 200, 100]
    assert abs(float(u.sigmas[-1]) - ((1 - 0.99915) / 0.99915) ** 0.5) < 1e-4            # sigma of alphas_cumprod[0]
    g = torch.Generator().manual_seed(1)
    x, e = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    x1 = u.step(e, u.timesteps[0], x)[0]
    a0, s0 = u._alpha_sigma(u.sigmas[0])
    a1, s1 = u._alpha_sigma(u.sigmas[1])
    assert torch.allclose(x1, a1 * (x - s0 * e) / a0 + s1 * e, atol=1e-5)                 # first step == DDIM
    # order of convergence (error ratio when the step is halved): DDIM 2, UniPC-1 (UniP-1 + UniC-1) 4, UniPC-2 8 --
    # "UniPC-p has order p + 1"; and UniP-2 without the corrector IS DPM-Solver++(2M), whose oracle is pinned on its own
    Ns = [10, 20, 40]
    off = range(1000)
    ddim = _gaussian_ode_errors(lambda: OS.UniPCMultistepScheduler(solver_order=1, disable_corrector=off), Ns)
    uni1 = _gaussian_ode_errors(lambda: OS.UniPCMultistepScheduler(solver_order=1), Ns)
    uni2 = _gaussian_ode_errors(lambda: OS.UniPCMultistepScheduler(solver_order=2), Ns)
    uni2p = _gaussian_ode_errors(lambda: OS.UniPCMultistepScheduler(solver_order=2, disable_corrector=off), Ns)
    dpm = _gaussian_ode_errors(lambda: OS.DPMSolverMultistepScheduler(), Ns)
    for e, lo, hi in ((ddim, 1.8, 2.2), (uni1, 3.2, 4.6), (uni2p, 3.5, 4.4), (uni2, 6.5, 9.5)):
        assert lo < e[0] / e[1] < hi and lo < e[1] / e[2] < hi, e
    assert np.allclose(uni2p, dpm, rtol=1e-3)
    uni3 = _gaussian_ode_errors(lambda: OS.UniPCMultistepScheduler(solver_order=3), [10, 20])
    assert uni3[0] < uni2[0] / 5 and uni3[1] < uni2[1] / 20


@pytest.mark.parametrize("K", [1, 2, 3])
@pytest.mark.parametrize("spacing,off,N,dc", [("leading", 1, 10, ()), ("linspace", 0, 7, ()), ("trailing", 0, 5, ()),
                                              ("leading", 1, 4, (1,)), ("linspace", 0, 2, ()), ("linspace", 0, 1, ())])
def test_unipc_product_table_vs_oracle(K, spacing, off, N, dc):
    """The product's per-step linear form (float64 solve on the host, 16-float rows for pp_cfg_sched_step kind 3),
    evaluated here in NumPy exactly as the kernel does, against the oracle's list-based class."""
    from powerpaint_amd import schedulers as PS
    kw = dict(solver_order=K, timestep_spacing=spacing, steps_offset=off, disable_corrector=dc)
    o, p = OS.UniPCMultistepScheduler(**kw), PS.UniPCMultistepScheduler(**kw)
    o.set_timesteps(N)
    p.set_timesteps(N)
    assert torch.equal(o.timesteps, p.timesteps) and torch.equal(o.sigmas, p.sigmas)
    g = torch.Generator().manual_seed(K * 100 + N)
    x = torch.randn(2, 4, 6, 6, generator=g).double()
    eps = [torch.randn(2, 4, 6, 6, generator=g).double() for _ in range(N)]
    xs, ref = x.clone(), []
    for t, e in zip(o.timesteps, eps):
        xs = o.step(e, t, xs)[0]
        ref.append(xs.numpy())
    coef = p._coef.numpy().astype(np.float64)
    xv = x.numpy()
    last = m1 = m2 = m3 = np.zeros_like(xv)
    for i, e in enumerate(eps):
        c = coef[i]
        x0 = (xv - c[0] * e.numpy()) / c[1]
        xc = c[3] * last + c[4] * m1 + c[5] * m2 + c[6] * m3 + c[7] * x0 if c[2] else xv
        xv = c[8] * xc + c[9] * x0 + c[10] * m1 + c[11] * m2
        last, m1, m2, m3 = xc, x0, m1, m2
        assert np.abs(xv - ref[i]).max() <= 2e-6 * max(1.0, np.abs(ref[i]).max()), i
    q = PS.UniPCMultistepScheduler.from_config(dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                                                    timestep_spacing="leading", steps_offset=1, skip_prk_steps=True,
                                                    set_alpha_to_one=False, clip_sample=False))
    assert q.config.timestep_spacing == "leading" and q.config.steps_offset == 1 and q.config.solver_order == 2
