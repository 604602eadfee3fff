"""SURVEY.md §8f-1 -- AutoencoderKL (VAE encode / decode either side of the denoising loop).

The oracle (oracle/vae.py) restates diffusers' AutoencoderKL; diffusers is not installable here, so parity is
**unpinned** beyond published facts (parameter count, state-dict key names).  CPU: those facts, the 180-degree
rotation identity the HIP encoder relies on, host logic (spec, packing, plan shapes).  GPU: the row-softmax kernel, and
encode / decode against the oracle in fp32 with the bf16-rounded weights the HIP path stores.
"""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import vae as OV
from powerpaint_amd.engine import Act, Arena, Builder
from powerpaint_amd.vae import VAENet

SMALL = dict(block_out_channels=(64, 128, 256, 256), layers_per_block=1)


def bf16_weights(sd):
    """Matrix weights as the HIP path stores them (bf16), vectors in fp32."""
    return {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v.clone()) for k, v in sd.items()}


def oracle_from(net: VAENet, sd, **cfg):
    o = OV.AutoencoderKL(**cfg).eval()
    o.load_state_dict(bf16_weights(sd))
    return o


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_matches_published_facts():
    o = OV.AutoencoderKL()
    assert sum(p.numel() for p in o.parameters()) == 83_653_863          # the SD-1.5 VAE ("sd-vae-ft-*", 83.7 M)
    keys = set(o.state_dict())
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.downsamplers.0.conv.weight",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "encoder.mid_block.attentions.0.group_norm.weight",
              "encoder.mid_block.attentions.0.to_out.0.bias", "decoder.up_blocks.2.upsamplers.0.conv.bias",
              "decoder.up_blocks.3.resnets.2.norm2.weight", "decoder.conv_norm_out.weight", "quant_conv.weight",
              "post_quant_conv.bias"):
        assert k in keys
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in keys and len(keys) == 248
    assert o.config.scaling_factor == 0.18215
    with torch.no_grad():
        z = o.encode(torch.zeros(1, 3, 64, 64)).latent_dist
        assert z.mean.shape == (1, 4, 8, 8) and o.decode(z.mode(), return_dict=False)[0].shape == (1, 3, 64, 64)


def test_product_spec_equals_oracle_state_dict():
    for cfg in ({}, SMALL):
        net = VAENet(**cfg)
        sp, osd = net.state_dict_spec(), OV.AutoencoderKL(**cfg).state_dict()
        assert set(sp) == set(osd)
        assert all(tuple(osd[k].shape) == sp[k] for k in sp)


def test_rotation_identity_of_the_encoder():
    """Downsample2D(padding=0) on the image == symmetric pad-1 stride-2 conv, rotated filters, on the rotated image;
    holds through the whole encoder (GroupNorm, attention and the 1x1 convs commute with the rotation)."""
    x, w, b = torch.randn(2, 5, 12, 16), torch.randn(7, 5, 3, 3), torch.randn(7)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    alt = F.conv2d(x.flip(2, 3), w.flip(2, 3), b, stride=2, padding=1).flip(2, 3)
    assert torch.allclose(ref, alt, atol=1e-5)

    torch.manual_seed(0)
    o = OV.AutoencoderKL(block_out_channels=(32, 32, 64, 64), layers_per_block=1).eval()
    r = copy.deepcopy(o)
    with torch.no_grad():
        for m in r.encoder.modules():
            if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3):
                m.weight.copy_(m.weight.flip(2, 3))
        for blk in r.encoder.down_blocks:
            for d in getattr(blk, "downsamplers", []):
                d.forward = (lambda conv: lambda t: F.conv2d(t, conv.weight, conv.bias, stride=2, padding=1))(d.conv)
        img = torch.randn(2, 3, 64, 48)
        want = o.moments(img)
        got = r.moments(img.flip(2, 3)).flip(2, 3)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4)


def test_plans_build_on_a_dry_arena():
    net = VAENet()
    net.load_state_dict(net.synthetic_state_dict(seed=1), "cpu")
    dry = Arena()
    pb = Builder(dry)
    shape = net.build_decode(pb, Act(dry.alloc(1 * 64 * 64 * 16), 1, 64, 64, 8), dry.alloc(4 * 512 * 512 * 4))
    assert shape == (1, 4, 512, 512)
    names = [c[2] for c in pb.plan.calls]
    assert names.count("conv3x3") == 31 and names.count("groupnorm_apply") == 30 and names.count("softmax_rows") == 1
    assert abs(pb.plan.flops / 2.5e12 - 1.0) < 0.02                        # 1.24 TMAC per 512x512 image
    dry = Arena()
    pb = Builder(dry)
    m = net.build_encode(pb, Act(dry.alloc(512 * 512 * 16), 1, 512, 512, 8))
    assert (m.B, m.H, m.W, m.C) == (1, 64, 64, 8)
    assert [c[2] for c in pb.plan.calls].count("conv3x3") == 23
    with pytest.raises(Exception):
        VAENet(block_out_channels=(100, 200, 400, 400))
    legacy = {}
    for k, v in net.synthetic_state_dict(seed=1).items():
        for new_name, old_name in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            k = k.replace(f".attentions.0.{new_name}.", f".attentions.0.{old_name}.")
        legacy[k] = v
    assert "decoder.mid_block.attentions.0.proj_attn.weight" in legacy
    VAENet().load_state_dict(legacy, "cpu")                                # pre-0.15 diffusers attention key names
    bad = net.synthetic_state_dict(seed=1)
    bad.pop("decoder.conv_out.bias")
    with pytest.raises(Exception):
        VAENet().load_state_dict(bad, "cpu")


def test_distribution_matches_oracle():
    from powerpaint_amd.models.autoencoder_kl import DiagonalGaussianDistribution
    p = torch.randn(2, 8, 4, 4) * 20
    a, b = DiagonalGaussianDistribution(p), OV.DiagonalGaussianDistribution(p)
    assert torch.equal(a.mean, b.mean) and torch.equal(a.std, b.std) and torch.equal(a.mode(), b.mode())
    ga, gb = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    assert torch.equal(a.sample(ga), b.sample(gb))


# ------------------------------------------------------------------------------------------------ GPU
def _close(got, want, cos_min, rel_max):
    got, want = got.float().cpu().flatten(), want.float().cpu().flatten()
    cos = F.cosine_similarity(got, want, dim=0).item()
    rel = ((got - want).abs().max() / want.abs().max()).item()
    assert cos >= cos_min and rel <= rel_max, f"cos {cos:.6f} (>= {cos_min}), max-abs/max {rel:.4f} (<= {rel_max})"
    return cos, rel


@pytest.mark.gpu
@pytest.mark.parametrize("rows,n,scale", [(64, 4096, 512 ** -0.5), (7, 256, 1.0), (3, 1030, 0.3), (5, 64, 2.0)])
def test_softmax_rows_gpu(rows, n, scale):
    from powerpaint_amd import ops
    torch.manual_seed(rows * n)
    n_ld = (n + 3) // 4 * 4
    s = (torch.randn(rows, n_ld, device="cuda") * 30)[:, :n]
    if n % 4:
        with pytest.raises(Exception):                                      # row strides must be multiples of 4
            ops.softmax_rows(s.contiguous(), scale)
        return
    p = ops.softmax_rows(s, scale)
    want = torch.softmax(s.double() * scale, dim=-1)
    assert p.dtype == torch.bfloat16
    assert torch.allclose(p.double(), want, atol=2e-3, rtol=2 ** -7)
    assert torch.allclose(p.float().sum(-1), torch.ones(rows, device="cuda"), atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,hw,batch", [(SMALL, 16, 2), ({}, 32, 1)])
def test_vae_decode_matches_oracle_gpu(cfg, hw, batch):
    from powerpaint_amd.models import AutoencoderKL
    vae = AutoencoderKL(device="cuda", **cfg)
    sd = vae.net.synthetic_state_dict(seed=11)
    vae.load_state_dict(sd)
    o = oracle_from(vae.net, sd, **cfg)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(batch, 4, hw, hw, generator=g) * 3.0
    with torch.no_grad():
        want = o.decode(z.to(torch.bfloat16).float(), return_dict=False)[0]
    got = vae.decode(z.cuda(), return_dict=False)[0]
    assert got.shape == want.shape == (batch, 3, 8 * hw, 8 * hw) and got.dtype == torch.float32
    _close(got, want, 0.999, 0.05)
    assert torch.equal(vae.decode(z.cuda()).sample, got)                    # return_dict form, deterministic replay


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,side,batch", [(SMALL, 128, 2), ({}, 256, 1)])
def test_vae_encode_matches_oracle_gpu(cfg, side, batch):
    from powerpaint_amd.models import AutoencoderKL
    vae = AutoencoderKL(device="cuda", **cfg)
    sd = vae.net.synthetic_state_dict(seed=12)
    vae.load_state_dict(sd)
    o = oracle_from(vae.net, sd, **cfg)
    g = torch.Generator().manual_seed(6)
    img = torch.rand(batch, 3, side, side + 64, generator=g) * 2 - 1       # non-square: the rotation must be per axis
    with torch.no_grad():
        want = o.moments(img.to(torch.bfloat16).float())
    dist = vae.encode(img.cuda()).latent_dist
    assert dist.mean.shape == (batch, 4, side // 8, (side + 64) // 8)
    _close(torch.cat([dist.mean, dist.logvar], 1), torch.cat([want[:, :4], want[:, 4:].clamp(-30, 20)], 1), 0.999, 0.05)
    ga, gb = torch.Generator().manual_seed(9), torch.Generator().manual_seed(9)
    smp = dist.sample(ga)
    assert torch.equal(smp, dist.mean + dist.std * torch.randn(dist.mean.shape, generator=gb).cuda())


@pytest.mark.gpu
def test_vae_batch_chunking_and_errors_gpu():
    from powerpaint_amd.models import AutoencoderKL
    from powerpaint_amd.models import autoencoder_kl as M
    vae = AutoencoderKL(device="cuda", **SMALL)
    with pytest.raises(Exception):
        vae.decode(torch.zeros(1, 4, 8, 8, device="cuda"))                  # no weights yet
    vae.load_state_dict(vae.net.synthetic_state_dict(seed=3))
    z = torch.randn(3, 4, 8, 8, device="cuda")
    whole = vae.decode(z, return_dict=False)[0]
    old = M.MAX_PIXELS
    try:
        M.MAX_PIXELS = 64 * 64                                              # one image per plan run
        parts = vae.decode(z, return_dict=False)[0]
    finally:
        M.MAX_PIXELS = old
    assert torch.equal(whole, parts)
    with pytest.raises(ValueError):
        vae.decode(torch.zeros(1, 3, 8, 8, device="cuda"))
    with pytest.raises(ValueError):
        vae.encode(torch.zeros(1, 3, 60, 64, device="cuda"))
    assert next(iter(vae.parameters())).dtype == torch.float32 and vae.config.block_out_channels == SMALL["block_out_channels"]
