"""The pin that turns "parity unpinned" into "pinned" on a box that has the wheel (last row).

The leaf arithmetic of the path -- ResnetBlock2D, Transformer2DModel, the samplers, the time embedding, the whole stock
UNet2DConditionModel / ControlNetModel / AutoencoderKL and the four schedulers -- is `diffusers==0.27.0` code that is
NOT in /root/reference and cannot be installed offline, so `oracle/sd_modules.py`, `oracle/schedulers.py` and
`oracle/vae.py` restate it from its published semantics.  Whenever `diffusers` IS importable (a developer machine, a
future image), this module compares the restatement with the real classes on the same weights and inputs; here it
skips.  CPU only, tiny configurations (seconds).
"""
import pytest
import torch

diffusers = pytest.importorskip("diffusers")

from oracle import schedulers as OS  # noqa: E402
from oracle import sd_modules as OM  # noqa: E402
from oracle import vae as OV  # noqa: E402

TINY = dict(block_out_channels=(64, 128), layers_per_block=1, norm_num_groups=32, cross_attention_dim=96,
            attention_head_dim=8, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))


def gen(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator("cpu").manual_seed(seed))


def same(a, b, what, tol=2e-5):
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), f"{what}: max-abs {err:.3g}"


def test_unet_forward_matches_diffusers():
    torch.manual_seed(0)
    ref = diffusers.UNet2DConditionModel(sample_size=16, in_channels=9, out_channels=4, **TINY).eval()
    o = OM.UNet2DConditionModel(in_channels=9, block_out_channels=TINY["block_out_channels"], layers_per_block=1,
                                cross_attention_dim=96, down_block_types=TINY["down_block_types"],
                                up_block_types=TINY["up_block_types"]).eval()
    missing, unexpected = o.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x, e = gen(2, 9, 16, 16, seed=1), gen(2, 77, 96, seed=2)
    with torch.no_grad():
        same(o(x, 481, e)[0], ref(x, 481, e).sample, "UNet2DConditionModel.forward")


def test_controlnet_forward_matches_diffusers():
    torch.manual_seed(1)
    kw = {k: v for k, v in TINY.items() if k != "up_block_types"}
    ref = diffusers.ControlNetModel(in_channels=4, conditioning_embedding_out_channels=(16, 32, 96, 256), **kw).eval()
    o = OM.ControlNetModel(in_channels=4, block_out_channels=kw["block_out_channels"], layers_per_block=1,
                           cross_attention_dim=96, down_block_types=kw["down_block_types"]).eval()
    with torch.no_grad():
        for p in ref.parameters():                       # the zero convs are zero-initialised: give them values
            if p.abs().max() == 0:
                p.normal_(0, 0.02)
    missing, unexpected = o.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x, e, img = gen(2, 4, 16, 16, seed=3), gen(2, 77, 96, seed=4), torch.rand(2, 3, 128, 128)
    with torch.no_grad():
        dn, md = o(x, 700, e, img, conditioning_scale=0.5)
        rdn, rmd = ref(x, 700, e, img, conditioning_scale=0.5, return_dict=False)
    for i, (a, b) in enumerate(zip(list(dn) + [md], list(rdn) + [rmd])):
        same(a, b, f"ControlNet residual {i}")
    with torch.no_grad():                                # guess mode: logspace(-1, 0, 13) * scale
        dn, md = o(x, 700, e, img, conditioning_scale=0.5, guess_mode=True)
        rdn, rmd = ref(x, 700, e, img, conditioning_scale=0.5, guess_mode=True, return_dict=False)
    for i, (a, b) in enumerate(zip(list(dn) + [md], list(rdn) + [rmd])):
        same(a, b, f"ControlNet guess-mode residual {i}")


def test_autoencoder_kl_matches_diffusers():
    torch.manual_seed(2)
    kw = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(32, 64), layers_per_block=1,
              norm_num_groups=32, down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2)
    ref = diffusers.AutoencoderKL(**kw).eval()
    o = OV.AutoencoderKL(block_out_channels=(32, 64), layers_per_block=1).eval()
    missing, unexpected = o.load_state_dict(ref.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    img, z = gen(1, 3, 32, 32, seed=5), gen(1, 4, 16, 16, seed=6)
    with torch.no_grad():
        same(o.encode(img).latent_dist.mean, ref.encode(img).latent_dist.mean, "AutoencoderKL.encode (mean)")
        same(o.decode(z).sample, ref.decode(z).sample, "AutoencoderKL.decode")


@pytest.mark.parametrize("name,kw", [
    ("DDIMScheduler", dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False,
                           set_alpha_to_one=False, steps_offset=1)),
    ("PNDMScheduler", dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True,
                           set_alpha_to_one=False, steps_offset=1)),
    ("DPMSolverMultistepScheduler", dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")),
    ("UniPCMultistepScheduler", dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear")),
])
@pytest.mark.parametrize("N", [10, 30, 50])
def test_schedulers_match_diffusers(name, kw, N):
    """Same timesteps and the same trajectory on a fixed epsilon sequence (fp64 inputs cast to fp32 on both sides)."""
    ref = getattr(diffusers, name)(**kw)
    o = getattr(OS, name)()
    ref.set_timesteps(N)
    o.set_timesteps(N)
    assert [int(t) for t in ref.timesteps] == [int(t) for t in o.timesteps], name
    x = xr = gen(1, 4, 8, 8, seed=7)
    for i, t in enumerate(ref.timesteps):
        eps = gen(1, 4, 8, 8, seed=100 + i)
        x = o.step(eps, o.timesteps[i], x)[0]
        xr = ref.step(eps, t, xr, return_dict=False)[0]
        same(x, xr, f"{name} N={N} step {i}", tol=5e-5)


def test_leaf_modules_match_diffusers():
    from diffusers.models.embeddings import get_timestep_embedding
    t = torch.tensor([1.0, 481.0, 999.0])
    same(OM.timestep_embedding(t, 320), get_timestep_embedding(t, 320, flip_sin_to_cos=True, downscale_freq_shift=0),
         "Timesteps(320, flip_sin_to_cos=True, shift 0)")
    from diffusers.models.resnet import ResnetBlock2D
    torch.manual_seed(3)
    ref = ResnetBlock2D(in_channels=64, out_channels=128, temb_channels=256, groups=32, eps=1e-5).eval()
    o = OM.ResnetBlock2D(64, 128, 256).eval()
    o.load_state_dict(ref.state_dict())
    x, temb = gen(2, 64, 8, 8, seed=8), gen(2, 256, seed=9)
    with torch.no_grad():
        same(o(x, temb), ref(x, temb), "ResnetBlock2D")
