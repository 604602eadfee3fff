"""BrushNetModel ("_CA": keeps cross-attention) on the MI355X HIP path.

Drop-in for /root/reference/powerpaint/models/BrushNet_CA.py:690-704 (`forward`) and :456-542 (`from_unet`).
Returns `(list[12], Tensor, list[15])` like the reference (`return_dict=False`, BrushNet_CA.py:945-946).  The returned
tensors are zero-copy NCHW-logical views (channels_last strides, bf16) of the runtime arena: they stay valid until
the next `forward` of this model -- exactly how the pipeline consumes them (produced and eaten within one step).
"""
from types import SimpleNamespace
from typing import Union

import torch

from .. import _lib as L
from ._base import SD15_DOWN, SD15_UP, Output, _HipModel


class BrushNetModel(_HipModel):
    kind = "brushnet"

    def __init__(self, in_channels: int = 4, conditioning_channels: int = 5, down_block_types=SD15_DOWN,
                 up_block_types=SD15_UP, block_out_channels=(320, 640, 1280, 1280), layers_per_block: int = 2,
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 768,
                 attention_head_dim: int = 8, device="cuda", dtype=torch.bfloat16, **unused):
        self._extra_config = unused         # validated in the base constructor (check_fixed_config)
        super().__init__(in_channels, block_out_channels, layers_per_block, attention_head_dim, cross_attention_dim,
                         norm_num_groups, norm_eps, down_block_types, up_block_types, device, dtype,
                         conditioning_channels=conditioning_channels)
        self.config = SimpleNamespace(
            in_channels=in_channels, conditioning_channels=conditioning_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
            attention_head_dim=attention_head_dim, global_pool_conditions=False)

    @classmethod
    def from_unet(cls, unet, brushnet_conditioning_channel_order: str = "rgb",
                  conditioning_embedding_out_channels=(16, 32, 96, 256), load_weights_from_unet: bool = True,
                  conditioning_channels: int = 5):
        """BrushNet_CA.py:456-542: copy the UNet trunk, duplicate conv_in into channels 0-3 and 4-7, zero channel 8,
        zero-initialised 1x1 output convs."""
        sd = getattr(unet, "_sd", None)
        if sd is None:
            raise L.PPError("from_unet needs the UNet's original state dict: load it with keep_state_dict=True")
        c = unet.config
        bn = cls(in_channels=c.in_channels, conditioning_channels=conditioning_channels,
                 down_block_types=c.down_block_types, up_block_types=c.up_block_types,
                 block_out_channels=c.block_out_channels, layers_per_block=c.layers_per_block,
                 norm_num_groups=c.norm_num_groups, norm_eps=c.norm_eps, cross_attention_dim=c.cross_attention_dim,
                 attention_head_dim=c.attention_head_dim, device=unet.device, dtype=unet.dtype)
        new = {}
        for k, v in sd.items():
            if k.startswith(("down_blocks.", "mid_block.", "up_blocks.", "time_embedding.")):
                new[k] = v
        w = torch.zeros(sd["conv_in.weight"].shape[0], c.in_channels + conditioning_channels, 3, 3)
        if load_weights_from_unet:
            w[:, :4] = sd["conv_in.weight"]
            w[:, 4:8] = sd["conv_in.weight"]
        new["conv_in_condition.weight"] = w
        new["conv_in_condition.bias"] = sd["conv_in.bias"]
        for pre, ch in bn.net._zero_conv_specs():
            new[pre + ".weight"] = torch.zeros(ch, ch, 1, 1)
            new[pre + ".bias"] = torch.zeros(ch)
        return bn.load_state_dict(new)

    def prepare(self, sample_shape, encoder_hidden_states, conditioning_scale=1.0, guess_mode: bool = False,
                pad_uncond: bool = False, twin: bool = False):
        """pad_uncond: outputs laid out as `cat([zeros_like(d), d])` for a UNet running the CFG pair (the pipeline's guess
        mode, pipeline_PowerPaint_Brushnet_CA.py:1421-1425)."""
        B, Cin, H, W = sample_shape
        scale = conditioning_scale
        if guess_mode and not self.config.global_pool_conditions:       # BrushNet_CA.py:905-928
            n = len(self.net._zero_conv_specs())
            scale = [float(s) * conditioning_scale for s in torch.logspace(-1, 0, n)]
        self.rt.ensure(B, H, W, self._nctx(encoder_hidden_states), Cin + self.config.conditioning_channels,
                       ("plain",), scale=scale, pad_uncond=pad_uncond, twin=twin)
        self.rt.set_context(encoder_hidden_states)
        return self.rt

    def outputs(self):
        o = self.rt.outputs
        v = self.rt.act_as_nchw
        return [v(a) for a in o["down"]], v(o["mid"]), [v(a) for a in o["up"]]

    @torch.no_grad()
    def forward(self, sample: torch.FloatTensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, brushnet_cond: torch.FloatTensor,
                conditioning_scale: float = 1.0, class_labels=None, timestep_cond=None, attention_mask=None,
                added_cond_kwargs=None, cross_attention_kwargs=None, guess_mode: bool = False,
                return_dict: bool = True):
        for name, v in (("class_labels", class_labels), ("timestep_cond", timestep_cond),
                        ("attention_mask", attention_mask)):
            if v is not None:
                raise NotImplementedError(f"{name} is outside the PowerPaint hot path")
        if isinstance(conditioning_scale, (list, tuple)):
            conditioning_scale = conditioning_scale[0]
        rt = self.prepare(tuple(sample.shape), encoder_hidden_states, float(conditioning_scale), guess_mode)
        rt.load_input([(sample, 0), (brushnet_cond, self.config.in_channels)])
        rt.set_timestep(timestep)
        rt.run_step()
        down, mid, up = self.outputs()
        if not return_dict:
            return (down, mid, up)
        return Output(down_block_res_samples=down, mid_block_res_sample=mid, up_block_res_samples=up)

    __call__ = forward
