"""Oracle (TEST INFRASTRUCTURE): fp32 PyTorch restatement of the SD-1.5 UNet family.

Parameter names follow the diffusers state-dict keys so that a single state_dict feeds
both this oracle and the HIP product (SURVEY.md section 8b "Weight naming").

Structure/wiring follows the reference fork:
  UNet2DConditionModel.forward   /root/reference/powerpaint/models/unet_2d_condition.py:1040-1363
  CrossAttnDownBlock2D.forward   /root/reference/powerpaint/models/unet_2d_blocks.py:1329-1402
  DownBlock2D.forward            /root/reference/powerpaint/models/unet_2d_blocks.py:1457-1500
  UNetMidBlock2DCrossAttn.fwd    /root/reference/powerpaint/models/unet_2d_blocks.py:850-899
  CrossAttnUpBlock2D.forward     /root/reference/powerpaint/models/unet_2d_blocks.py:2549-2643
  UpBlock2D.forward              /root/reference/powerpaint/models/unet_2d_blocks.py:2696-2770
  BrushNetModel                  /root/reference/powerpaint/models/BrushNet_CA.py:140-454,456-542,690-952
Leaf-module math restates diffusers==0.27.0 (not vendored in /root/reference; pinned at
/root/reference/requirements/requirements.txt:3): ResnetBlock2D, Transformer2DModel,
BasicTransformerBlock, Attention(AttnProcessor2_0), FeedForward(GEGLU), Downsample2D,
Upsample2D, Timesteps, TimestepEmbedding, ControlNetModel -- SURVEY.md Appendix B.
"""
import math
from types import SimpleNamespace
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# leaves  [diffusers-0.27.0 restatement]
# --------------------------------------------------------------------------------------
def timestep_embedding(timesteps: torch.Tensor, dim: int = 320, flip_sin_to_cos: bool = True,
                       freq_shift: float = 0.0, max_period: int = 10000) -> torch.Tensor:
    """diffusers.models.embeddings.get_timestep_embedding (Timesteps(320, True, 0))."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, groups: int = 32,
                 eps: float = 1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb, scale: float = 1.0):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2D(nn.Module):
    def __init__(self, channels: int, padding: int = 1):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x, scale: float = 1.0):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x, output_size=None, scale: float = 1.0):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, ctx=None):
        c = x if ctx is None else ctx
        B, N, _ = x.shape
        q = self.to_q(x).view(B, N, self.heads, -1).transpose(1, 2)
        k = self.to_k(c).view(B, c.shape[1], self.heads, -1).transpose(1, 2)
        v = self.to_v(c).view(B, c.shape[1], self.heads, -1).transpose(1, 2)
        # softmax is row-wise, so evaluating it per block of queries is the same arithmetic per row; it only keeps the
        # score matrix of the 16384-token self-attention (BASELINE config 5: 2 x 8 x 16384^2 fp32 = 17 GB) out of RAM
        step = N if N * c.shape[1] <= (1 << 24) else max(1, (1 << 24) // c.shape[1])
        outs = []
        for i in range(0, N, step):
            s = torch.matmul(q[:, :, i:i + step], k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
            outs.append(torch.matmul(torch.softmax(s, dim=-1), v))
        o = outs[0] if len(outs) == 1 else torch.cat(outs, 2)
        o = o.transpose(1, 2).reshape(B, N, -1)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)  # approximate="none" (erf)


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        x = x + self.ff(self.norm3(x))
        return x


class Transformer2DModel(nn.Module):
    """use_linear_projection=False (SD-1.5): 1x1-conv proj_in / proj_out."""

    def __init__(self, heads: int, dim_head: int, in_channels: int, cross_attention_dim: int,
                 num_layers: int = 1, groups: int = 32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states=None, **kw):
        B, C, H, W = x.shape
        r = x
        x = self.proj_in(self.norm(x))
        x = x.permute(0, 2, 3, 1).reshape(B, H * W, -1)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        return (self.proj_out(x) + r,)


# --------------------------------------------------------------------------------------
# blocks  [reference fork wiring: per-layer add-sample pop(0), return_res_samples]
# --------------------------------------------------------------------------------------
class DownBlock(nn.Module):
    """CrossAttnDownBlock2D (has_attn=True) / DownBlock2D."""

    def __init__(self, in_c, out_c, temb_c, num_layers, heads, ctx_dim, has_attn, add_downsample, groups, eps):
        super().__init__()
        self.has_cross_attention = has_attn
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_c if i == 0 else out_c, out_c, temb_c, groups, eps) for i in range(num_layers)])
        if has_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, out_c // heads, out_c, ctx_dim, 1, groups) for _ in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_c)]) if add_downsample else None

    def forward(self, h, temb, ctx=None, down_block_add_samples: Optional[list] = None):
        out = ()
        for i, resnet in enumerate(self.resnets):
            h = resnet(h, temb)
            if self.has_cross_attention:
                h = self.attentions[i](h, encoder_hidden_states=ctx)[0]
            if down_block_add_samples is not None:
                h = h + down_block_add_samples.pop(0)       # unet_2d_blocks.py:1388-1389
            out = out + (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            if down_block_add_samples is not None:
                h = h + down_block_add_samples.pop(0)       # unet_2d_blocks.py:1397-1398
            out = out + (h,)
        return h, out


class MidBlock(nn.Module):
    """UNetMidBlock2DCrossAttn."""

    def __init__(self, c, temb_c, heads, ctx_dim, groups, eps):
        super().__init__()
        self.has_cross_attention = True
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, c // heads, c, ctx_dim, 1, groups)])

    def forward(self, h, temb, ctx):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, encoder_hidden_states=ctx)[0]
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    """CrossAttnUpBlock2D (has_attn=True) / UpBlock2D."""

    def __init__(self, in_c, out_c, prev_c, temb_c, num_layers, heads, ctx_dim, has_attn, add_upsample, groups, eps):
        super().__init__()
        self.has_cross_attention = has_attn
        rs = []
        for i in range(num_layers):
            res_skip = in_c if i == num_layers - 1 else out_c
            res_in = prev_c if i == 0 else out_c
            rs.append(ResnetBlock2D(res_in + res_skip, out_c, temb_c, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if has_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(heads, out_c // heads, out_c, ctx_dim, 1, groups) for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_c)]) if add_upsample else None

    def forward(self, h, res_tuple, temb, ctx=None, upsample_size=None, return_res_samples=False,
                up_block_add_samples: Optional[list] = None):
        out = ()
        for i, resnet in enumerate(self.resnets):
            r = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            h = torch.cat([h, r], dim=1)
            h = resnet(h, temb)
            if self.has_cross_attention:
                h = self.attentions[i](h, encoder_hidden_states=ctx)[0]
            if return_res_samples:
                out = out + (h,)
            if up_block_add_samples is not None:
                h = h + up_block_add_samples.pop(0)         # unet_2d_blocks.py:2629-2630
        if self.upsamplers is not None:
            h = self.upsamplers[0](h, upsample_size)
            if return_res_samples:
                out = out + (h,)
            if up_block_add_samples is not None:
                h = h + up_block_add_samples.pop(0)         # unet_2d_blocks.py:2637-2638
        if return_res_samples:
            return h, out
        return h


SD15 = dict(
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, attention_head_dim=8,
    cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
)


def _build_trunk(self, in_conv_channels, cfg, with_up=True):
    """Shared constructor body of UNet / BrushNet / ControlNet (unet_2d_condition.py:256-481)."""
    boc = tuple(cfg["block_out_channels"])
    L = cfg["layers_per_block"]
    heads = cfg["attention_head_dim"]          # SD-1.5 quirk: "attention_head_dim" IS the head count
    ctx = cfg["cross_attention_dim"]
    g, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    temb_c = boc[0] * 4
    self.time_embedding = TimestepEmbedding(boc[0], temb_c)
    self.down_blocks = nn.ModuleList()
    out_c = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        final = i == len(boc) - 1
        self.down_blocks.append(DownBlock(in_c, out_c, temb_c, L, heads, ctx, t == "CrossAttnDownBlock2D",
                                          not final, g, eps))
    self.mid_block = MidBlock(boc[-1], temb_c, heads, ctx, g, eps)
    if with_up:
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, t in enumerate(cfg["up_block_types"]):
            final = i == len(boc) - 1
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(in_c, out_c, prev_c, temb_c, L + 1, heads, ctx,
                                          t == "CrossAttnUpBlock2D", not final, g, eps))


def _time_embed(self, sample, timestep):
    """unet_2d_condition.py:914-938 (get_time_embed) + time_embedding."""
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.int64 if isinstance(t, int) else torch.float64, device=sample.device)
    elif t.dim() == 0:
        t = t[None].to(sample.device)
    t = t.expand(sample.shape[0])
    t_emb = timestep_embedding(t, self.config.block_out_channels[0]).to(sample.dtype)
    return self.time_embedding(t_emb)


class UNet2DConditionModel(nn.Module):
    """SD-1.5 UNet; stock behaviour + the fork's three *_add_samples arguments + ControlNet residuals."""

    def __init__(self, in_channels: int = 4, out_channels: int = 4, sample_size: int = 64, **overrides):
        super().__init__()
        cfg = dict(SD15)
        cfg.update(overrides)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, sample_size=sample_size,
                                      time_cond_proj_dim=None, **cfg)
        boc = cfg["block_out_channels"]
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        _build_trunk(self, in_channels, cfg, with_up=True)
        self.conv_norm_out = nn.GroupNorm(cfg["norm_num_groups"], boc[0], eps=cfg["norm_eps"])
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep, encoder_hidden_states, timestep_cond=None, cross_attention_kwargs=None,
                added_cond_kwargs=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                down_block_add_samples: Optional[list] = None, mid_block_add_sample=None,
                up_block_add_samples: Optional[list] = None, return_dict: bool = False, **unused):
        emb = _time_embed(self, sample, timestep)
        sample = self.conv_in(sample)
        is_controlnet = mid_block_additional_residual is not None and down_block_additional_residuals is not None
        is_brushnet = (down_block_add_samples is not None and mid_block_add_sample is not None
                       and up_block_add_samples is not None)
        if is_brushnet:  # the reference consumes the caller's lists destructively; keep that contract
            pass
        res = (sample,)                                      # unet_2d_condition.py:1220 (captured BEFORE the add)
        if is_brushnet:
            sample = sample + down_block_add_samples.pop(0)  # unet_2d_condition.py:1222-1223
        for blk in self.down_blocks:
            adds = None
            if is_brushnet and len(down_block_add_samples) > 0:
                adds = [down_block_add_samples.pop(0)
                        for _ in range(len(blk.resnets) + (blk.downsamplers is not None))]
            sample, r = blk(sample, emb, encoder_hidden_states, down_block_add_samples=adds)
            res += r
        if is_controlnet:                                    # unet_2d_condition.py:1263-1272
            res = tuple(a + b for a, b in zip(res, down_block_additional_residuals))
        sample = self.mid_block(sample, emb, encoder_hidden_states)
        if is_controlnet:
            sample = sample + mid_block_additional_residual  # :1296-1297
        if is_brushnet:
            sample = sample + mid_block_add_sample           # :1299-1300
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            r, res = res[-n:], res[:-n]
            adds = None
            if is_brushnet and len(up_block_add_samples) > 0:
                adds = [up_block_add_samples.pop(0) for _ in range(n + (blk.upsamplers is not None))]
            sample = blk(sample, r, emb, encoder_hidden_states, up_block_add_samples=adds)
        sample = self.conv_out(F.silu(self.conv_norm_out(sample)))
        return (sample,)


class BrushNetModel(nn.Module):
    """BrushNet_CA.py:63-958.  Full second UNet (with cross-attention) + 12/1/15 zero 1x1 convs."""

    def __init__(self, in_channels: int = 4, conditioning_channels: int = 5, **overrides):
        super().__init__()
        cfg = dict(SD15)
        cfg.update(overrides)
        self.config = SimpleNamespace(in_channels=in_channels, conditioning_channels=conditioning_channels,
                                      global_pool_conditions=False, **cfg)
        boc = cfg["block_out_channels"]
        L = cfg["layers_per_block"]
        self.conv_in_condition = nn.Conv2d(in_channels + conditioning_channels, boc[0], 3, padding=1)
        _build_trunk(self, in_channels + conditioning_channels, cfg, with_up=True)

        def zc(c):
            m = nn.Conv2d(c, c, 1)
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
            return m

        self.brushnet_down_blocks = nn.ModuleList([zc(boc[0])])      # BrushNet_CA.py:330-332
        for i, c in enumerate(boc):
            for _ in range(L):
                self.brushnet_down_blocks.append(zc(c))
            if i != len(boc) - 1:
                self.brushnet_down_blocks.append(zc(c))
        self.brushnet_mid_block = zc(boc[-1])
        self.brushnet_up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        for i, c in enumerate(rev):                                    # BrushNet_CA.py:446-454
            for _ in range(L + 1):
                self.brushnet_up_blocks.append(zc(c))
            if i != len(boc) - 1:
                self.brushnet_up_blocks.append(zc(c))

    @property
    def dtype(self):
        return self.conv_in_condition.weight.dtype

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, conditioning_channels: int = 5, load_weights_from_unet=True):
        """BrushNet_CA.py:456-542."""
        cfg = {k: getattr(unet.config, k) for k in SD15}
        bn = cls(in_channels=unet.config.in_channels, conditioning_channels=conditioning_channels, **cfg)
        if load_weights_from_unet:
            w = torch.zeros_like(bn.conv_in_condition.weight)
            w[:, :4] = unet.conv_in.weight
            w[:, 4:8] = unet.conv_in.weight
            bn.conv_in_condition.weight = nn.Parameter(w)
            bn.conv_in_condition.bias = nn.Parameter(unet.conv_in.bias.detach().clone())
            bn.time_embedding.load_state_dict(unet.time_embedding.state_dict())
            bn.down_blocks.load_state_dict(unet.down_blocks.state_dict(), strict=False)
            bn.mid_block.load_state_dict(unet.mid_block.state_dict(), strict=False)
            bn.up_blocks.load_state_dict(unet.up_blocks.state_dict(), strict=False)
        return bn

    def forward(self, sample, timestep, encoder_hidden_states, brushnet_cond, conditioning_scale: float = 1.0,
                guess_mode: bool = False, return_dict: bool = False, **unused):
        emb = _time_embed(self, sample, timestep)
        sample = self.conv_in_condition(torch.cat([sample, brushnet_cond], 1))      # :822-823
        down = (sample,)
        for blk in self.down_blocks:
            sample, r = blk(sample, emb, encoder_hidden_states)
            down += r
        bdown = [z(s) for s, z in zip(down, self.brushnet_down_blocks)]             # :842-845
        sample = self.mid_block(sample, emb, encoder_hidden_states)
        bmid = self.brushnet_mid_block(sample)                                       # :861
        up = ()
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            r, down = down[-n:], down[:-n]
            upsample_size = down[-1].shape[2:] if i != len(self.up_blocks) - 1 else None
            sample, u = blk(sample, r, emb, encoder_hidden_states, upsample_size=upsample_size,
                            return_res_samples=True)
            up += u
        bup = [z(s) for s, z in zip(up, self.brushnet_up_blocks)]                   # :899-902
        if guess_mode and not self.config.global_pool_conditions:                    # :905-928
            scales = torch.logspace(-1, 0, len(bdown) + 1 + len(bup)) * conditioning_scale
            bdown = [s * sc for s, sc in zip(bdown, scales[:len(bdown)])]
            bmid = bmid * scales[len(bdown)]
            bup = [s * sc for s, sc in zip(bup, scales[len(bdown) + 1:])]
        else:                                                                        # :930-934
            bdown = [s * conditioning_scale for s in bdown]
            bmid = bmid * conditioning_scale
            bup = [s * conditioning_scale for s in bup]
        return bdown, bmid, bup


class ControlNetConditioningEmbedding(nn.Module):
    def __init__(self, out_c: int, cond_c: int = 3, block_out_channels: Sequence[int] = (16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(cond_c, block_out_channels[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(block_out_channels) - 1):
            a, b = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(nn.Conv2d(a, a, 3, padding=1))
            self.blocks.append(nn.Conv2d(a, b, 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(block_out_channels[-1], out_c, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)
        nn.init.zeros_(self.conv_out.bias)

    def forward(self, c):
        e = F.silu(self.conv_in(c))
        for b in self.blocks:
            e = F.silu(b(e))
        return self.conv_out(e)


class ControlNetModel(nn.Module):
    """[diffusers-0.27.0 ControlNetModel restatement] used at pipeline_PowerPaint_ControlNet.py:1686-1694."""

    def __init__(self, in_channels: int = 4, conditioning_channels: int = 3, **overrides):
        super().__init__()
        cfg = dict(SD15)
        cfg.update(overrides)
        self.config = SimpleNamespace(in_channels=in_channels, global_pool_conditions=False, **cfg)
        boc = cfg["block_out_channels"]
        L = cfg["layers_per_block"]
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        _build_trunk(self, in_channels, cfg, with_up=False)
        self.controlnet_cond_embedding = ControlNetConditioningEmbedding(boc[0], conditioning_channels)

        def zc(c):
            m = nn.Conv2d(c, c, 1)
            nn.init.zeros_(m.weight)
            nn.init.zeros_(m.bias)
            return m

        self.controlnet_down_blocks = nn.ModuleList([zc(boc[0])])
        for i, c in enumerate(boc):
            for _ in range(L):
                self.controlnet_down_blocks.append(zc(c))
            if i != len(boc) - 1:
                self.controlnet_down_blocks.append(zc(c))
        self.controlnet_mid_block = zc(boc[-1])

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale: float = 1.0,
                guess_mode: bool = False, return_dict: bool = False, **unused):
        emb = _time_embed(self, sample, timestep)
        sample = self.conv_in(sample) + self.controlnet_cond_embedding(controlnet_cond)
        down = (sample,)
        for blk in self.down_blocks:
            sample, r = blk(sample, emb, encoder_hidden_states)
            down += r
        sample = self.mid_block(sample, emb, encoder_hidden_states)
        cdown = [z(s) for s, z in zip(down, self.controlnet_down_blocks)]
        cmid = self.controlnet_mid_block(sample)
        if guess_mode and not self.config.global_pool_conditions:
            scales = torch.logspace(-1, 0, len(cdown) + 1) * conditioning_scale
            cdown = [s * sc for s, sc in zip(cdown, scales)]
            cmid = cmid * scales[-1]
        else:
            cdown = [s * conditioning_scale for s in cdown]
            cmid = cmid * conditioning_scale
        return cdown, cmid


def randomize_zero_convs(model: nn.Module, std: float = 0.02, seed: int = 7):
    """zero-convs must get non-zero synthetic weights or routing bugs hide."""
    g = torch.Generator("cpu").manual_seed(seed)
    for name, p in model.named_parameters():
        if ("brushnet_" in name or "controlnet_down_blocks" in name or "controlnet_mid_block" in name
                or "controlnet_cond_embedding.conv_out" in name):
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
