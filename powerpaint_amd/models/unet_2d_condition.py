"""UNet2DConditionModel on the MI355X HIP path.

Drop-in for the call surface of /root/reference/powerpaint/models/unet_2d_condition.py:1040-1058 (the fork, incl.
`down_block_add_samples` / `mid_block_add_sample` / `up_block_add_samples`) and of the stock diffusers class the
v1 / ControlNet pipelines import (/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:27,1009-1015;
pipeline_PowerPaint_ControlNet.py:1707-1715: `down_block_additional_residuals`, `mid_block_additional_residual`).
"""
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from .. import _lib as L
from ._base import SD15_DOWN, SD15_UP, Output, _HipModel


class UNet2DConditionModel(_HipModel):
    kind = "unet"

    def __init__(self, sample_size: Optional[int] = 64, in_channels: int = 4, out_channels: int = 4,
                 down_block_types=SD15_DOWN, up_block_types=SD15_UP, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block: int = 2, norm_num_groups: int = 32, norm_eps: float = 1e-5,
                 cross_attention_dim: int = 768, attention_head_dim: int = 8, device="cuda",
                 dtype=torch.bfloat16, **unused):
        self._extra_config = unused         # validated in the base constructor (check_fixed_config)
        super().__init__(in_channels, block_out_channels, layers_per_block, attention_head_dim, cross_attention_dim,
                         norm_num_groups, norm_eps, down_block_types, up_block_types, device, dtype,
                         out_channels=out_channels)
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
            norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
            attention_head_dim=attention_head_dim, time_cond_proj_dim=None, addition_embed_type=None,
            flip_sin_to_cos=True, freq_shift=0, act_fn="silu", only_cross_attention=False,
            use_linear_projection=False, class_embed_type=None, num_class_embeds=None, upcast_attention=False,
            resnet_time_scale_shift="default", mid_block_scale_factor=1, downsample_padding=1,
            num_attention_heads=None, projection_class_embeddings_input_dim=None, encoder_hid_dim=None,
            encoder_hid_dim_type=None, addition_time_embed_dim=None, transformer_layers_per_block=1)

    # ------------------------------------------------------------------
    def _wiring(self, down_add, mid_add, up_add, ctrl_down, ctrl_mid):
        def ptrs(lst):
            # zero-copy hand-off of a side network's NHWC arena tensor -- only when it is stored in THIS network's 16-bit
            # format (a bf16 BrushNet feeding an fp16 UNet would have its bits reinterpreted); any other tensor goes
            # through `load_residual`, which converts
            return [getattr(t, "_pp_nhwc_ptr", 0) if t.dtype == self._dtype else 0 for t in lst]

        if down_add is not None and mid_add is not None and up_add is not None:
            return ("brushnet", {"down": ptrs(down_add), "mid": ptrs([mid_add]), "up": ptrs(up_add)})
        if ctrl_down is not None and ctrl_mid is not None:
            return ("controlnet", {"down": ptrs(ctrl_down), "mid": ptrs([ctrl_mid])})
        return ("plain",)

    def prepare(self, sample_shape, encoder_hidden_states, down_block_add_samples=None, mid_block_add_sample=None,
                up_block_add_samples=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                twin: bool = False):
        """Compile (or reuse) the launch plan for this shape / residual wiring and bind the inputs.  twin: the caller (the
        fused denoising loop) vouches that the second half of the batch is a copy of the first (`torch.cat([latents] * 2)`
        over CFG-duplicated mask / masked-image latents, pipeline_PowerPaint.py:990-996): the prompt-independent prefix of
        the forward pass then runs on one half (NetRuntime.ensure).  `forward` never sets it."""
        B, Cin, H, W = sample_shape
        if Cin != self.config.in_channels:
            raise ValueError(f"sample has {Cin} channels, unet.config.in_channels = {self.config.in_channels}")
        wiring = self._wiring(down_block_add_samples, mid_block_add_sample, up_block_add_samples,
                              down_block_additional_residuals, mid_block_additional_residual)
        self.rt.ensure(B, H, W, self._nctx(encoder_hidden_states), Cin, wiring, twin=twin)
        if wiring[0] != "plain":
            groups = {"down": down_block_add_samples if wiring[0] == "brushnet" else down_block_additional_residuals,
                      "mid": [mid_block_add_sample if wiring[0] == "brushnet" else mid_block_additional_residual]}
            if wiring[0] == "brushnet":
                groups["up"] = up_block_add_samples
            for g, lst in groups.items():
                exp = len(self.rt.lay["slots"][g])
                if len(lst) != exp:
                    raise ValueError(f"{g} residual list has {len(lst)} tensors, expected {exp}")
                for i, (t, p) in enumerate(zip(lst, wiring[1][g])):
                    if not p:
                        self.rt.load_residual(g, i, t)
        self.rt.set_context(encoder_hidden_states)
        return self.rt

    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs: Optional[Dict[str, Any]] = None, added_cond_kwargs=None,
                down_block_additional_residuals: Optional[Tuple[torch.Tensor]] = None,
                mid_block_additional_residual: Optional[torch.Tensor] = None,
                down_intrablock_additional_residuals=None, encoder_attention_mask=None, return_dict: bool = True,
                down_block_add_samples: Optional[List[torch.Tensor]] = None,
                mid_block_add_sample: Optional[torch.Tensor] = None,
                up_block_add_samples: Optional[List[torch.Tensor]] = None, **kwargs):
        for name, v in (("class_labels", class_labels), ("timestep_cond", timestep_cond),
                        ("attention_mask", attention_mask), ("encoder_attention_mask", encoder_attention_mask),
                        ("down_intrablock_additional_residuals", down_intrablock_additional_residuals)):
            if v is not None:
                raise NotImplementedError(f"{name} is outside the PowerPaint hot path (never set by the pipelines)")
        if cross_attention_kwargs and cross_attention_kwargs.get("scale", 1.0) != 1.0:
            raise NotImplementedError("LoRA scale != 1 is outside the hot path")
        rt = self.prepare(tuple(sample.shape), encoder_hidden_states, down_block_add_samples, mid_block_add_sample,
                          up_block_add_samples, down_block_additional_residuals, mid_block_additional_residual)
        # the reference consumes the BrushNet lists destructively (.pop(0), unet_2d_condition.py:1223,1234,1318)
        for lst in (down_block_add_samples, up_block_add_samples):
            if isinstance(lst, list):
                del lst[:]
        rt.load_input([(sample, 0)])
        rt.set_timestep(timestep)
        rt.run_step()
        out = rt.eps_tensor().to(self._dtype)
        if not return_dict:
            return (out,)
        return Output(sample=out)

    __call__ = forward
