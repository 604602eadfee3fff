"""Oracle (TEST INFRASTRUCTURE, never imported by the product): fp32 PyTorch restatement of the SD-1.5 AutoencoderKL.

The VAE is the step either side of the denoising loop:
    /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:657-669   vae.encode(image).latent_dist.sample(generator)
    /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:1051      vae.decode(latents / scaling_factor)[0]
The class itself lives in diffusers==0.27.0 (requirements/requirements.txt:3), which is neither vendored in
/root/reference nor installable here, so this file restates the published architecture
(diffusers.models.autoencoders.autoencoder_kl.AutoencoderKL with the SD-1.5 `vae/config.json`: 3 -> (128, 256, 512, 512)
-> 2x4 latent moments, layers_per_block 2, GroupNorm(32, eps 1e-6), one single-head attention in each mid block,
Downsample2D(padding=0) = F.pad((0, 1, 0, 1)) + stride-2 conv, nearest x2 + conv upsampling, 1x1 quant / post_quant
convs, scaling_factor 0.18215) with diffusers' parameter names.  **Parity unpinned**: there are no reference outputs
to check it against; it is pinned only to published facts (parameter count 83,653,863; state-dict key names) in
tests/test_vae.py.
"""
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

SD15_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215)
EPS = 1e-6


class Resnet(nn.Module):
    """ResnetBlock2D(temb_channels=None)."""

    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=EPS)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=EPS)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class MidAttention(nn.Module):
    """Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups=32, bias=True) of UNetMidBlock2D."""

    def __init__(self, c: int, groups: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=EPS)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).view(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class _Conv(nn.Module):
    def __init__(self, c: int, stride: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=stride, padding=0 if stride == 2 else 1)


class Down(_Conv):
    def __init__(self, c):
        super().__init__(c, 2)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))            # Downsample2D(padding=0): pad right / bottom only


class Up(_Conv):
    def __init__(self, c):
        super().__init__(c, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Block(nn.Module):
    def __init__(self, cin, cout, n, groups, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        if down:
            self.downsamplers = nn.ModuleList([Down(cout)])
        if up:
            self.upsamplers = nn.ModuleList([Up(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        for s in list(getattr(self, "downsamplers", [])) + list(getattr(self, "upsamplers", [])):
            x = s(x)
        return x


class Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.attentions = nn.ModuleList([MidAttention(c, groups)])
        self.resnets = nn.ModuleList([Resnet(c, c, groups), Resnet(c, c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cin, latent, boc: Sequence[int], n, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([Block(boc[max(i - 1, 0)], c, n, groups, down=i != len(boc) - 1)
                                          for i, c in enumerate(boc)])
        self.mid_block = Mid(boc[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=EPS)
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class Decoder(nn.Module):
    def __init__(self, latent, cout, boc: Sequence[int], n, groups):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = Mid(rev[0], groups)
        self.up_blocks = nn.ModuleList([Block(rev[max(i - 1, 0)], c, n + 1, groups, up=i != len(boc) - 1)
                                        for i, c in enumerate(rev)])
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=EPS)
        self.conv_out = nn.Conv2d(rev[-1], cout, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution."""

    def __init__(self, parameters: torch.Tensor):
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype).to(self.mean.device)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      latent_channels=latent_channels, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor)

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.moments(x)))

    def decode(self, z, return_dict: bool = True, generator=None):
        img = self.decoder(self.post_quant_conv(z))
        return (img,) if not return_dict else SimpleNamespace(sample=img)
