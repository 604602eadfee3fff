"""AutoencoderKL on the MI355X HIP path.

Drop-in for the object the pipelines hold as `self.vae` (diffusers.AutoencoderKL; call sites
/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:283,657-669,926,1051 and the same lines of the BrushNet /
ControlNet pipelines): `encode(x).latent_dist.sample(generator)`, `decode(z, return_dict=False)[0]`,
`config.scaling_factor / block_out_channels / latent_channels`.  The network itself is `powerpaint_amd.vae.VAENet`.
"""
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from .. import _lib as L
from ..loaders import PretrainedMixin
from ..vae import VAENet, VAERuntime
from ._base import Output

MAX_PIXELS = 1 << 21      # image pixels per launch plan (8 x 512^2, 2 x 1024^2): keeps every tensor below 2^31 bytes


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution over the encoder's [B, 2*latent, h, w] moments."""

    def __init__(self, parameters: torch.Tensor, deterministic: bool = False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        from ..pipelines._base import randn_tensor
        noise = randn_tensor(self.mean.shape, generator=generator, device=self.parameters.device,
                             dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKL(PretrainedMixin):
    def __init__(self, in_channels: int = 3, out_channels: int = 3,
                 down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                 block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2, act_fn: str = "silu",
                 latent_channels: int = 4, norm_num_groups: int = 32, sample_size: int = 512,
                 scaling_factor: float = 0.18215, device="cuda", **unused):
        if act_fn != "silu" or any(t != "DownEncoderBlock2D" for t in down_block_types) or \
                any(t != "UpDecoderBlock2D" for t in up_block_types) or len(down_block_types) != len(block_out_channels):
            raise L.PPError("AutoencoderKL: only the SD-1.5 layout (DownEncoderBlock2D / UpDecoderBlock2D, SiLU) is built")
        self._device = torch.device(device)
        self.net = VAENet(in_channels, out_channels, latent_channels, block_out_channels, layers_per_block,
                          norm_num_groups)
        self._dec = VAERuntime(self.net, self._device, "decode")
        self._enc = VAERuntime(self.net, self._device, "encode")
        self.config = SimpleNamespace(
            in_channels=in_channels, out_channels=out_channels, down_block_types=tuple(down_block_types),
            up_block_types=tuple(up_block_types), block_out_channels=tuple(block_out_channels),
            layers_per_block=layers_per_block, act_fn=act_fn, latent_channels=latent_channels,
            norm_num_groups=norm_num_groups, sample_size=sample_size, scaling_factor=scaling_factor,
            force_upcast=False)

    # the module boundary is fp32 (inputs of any float dtype are accepted; the network computes in bf16 / fp32 acc)
    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self._device

    def parameters(self):
        yield torch.empty(0, dtype=torch.float32, device=self._device)

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        self.net.load_state_dict(sd, self._device)
        self._dec.key = self._enc.key = None
        return self

    def param_buffer(self) -> torch.Tensor:
        return self.net.params.buf

    def _chunks(self, n: int, pixels_per_item: int):
        step = max(1, MAX_PIXELS // max(pixels_per_item, 1))
        return [(i, min(i + step, n)) for i in range(0, n, step)]

    # ------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        if self.net.params is None:
            raise L.PPError("AutoencoderKL: load_state_dict first")
        B, Cc, H, W = x.shape
        if Cc != self.config.in_channels or H % 8 or W % 8:
            raise ValueError(f"encode expects [B, {self.config.in_channels}, 8k, 8m] images, got {tuple(x.shape)}")
        xr = torch.flip(x.to(self._device), dims=(2, 3))            # the encoder runs rotated by 180 degrees (vae.py)
        parts = [self._enc.run(xr[a:b]) for a, b in self._chunks(B, H * W)]
        mom = torch.flip(torch.cat(parts, 0), dims=(2, 3))[:, :2 * self.config.latent_channels].contiguous()
        dist = DiagonalGaussianDistribution(mom)
        return Output(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        if self.net.params is None:
            raise L.PPError("AutoencoderKL: load_state_dict first")
        B, Cc, h, w = z.shape
        if Cc != self.config.latent_channels:
            raise ValueError(f"decode expects [B, {self.config.latent_channels}, h, w] latents, got {tuple(z.shape)}")
        parts = [self._dec.run(z[a:b])[:, :self.config.out_channels].clone()
                 for a, b in self._chunks(B, 64 * h * w)]
        img = torch.cat(parts, 0) if len(parts) > 1 else parts[0]
        return Output(sample=img) if return_dict else (img,)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                generator: Optional[torch.Generator] = None):
        dist = self.encode(sample).latent_dist
        z = dist.sample(generator=generator) if sample_posterior else dist.mode()
        return self.decode(z, return_dict=return_dict)

    __call__ = forward
