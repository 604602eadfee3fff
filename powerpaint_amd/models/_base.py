"""Shared host-side plumbing of the three model wrappers."""
from types import SimpleNamespace
from typing import Dict

import torch

from .. import _lib as L
from ..engine import SDNet
from ..loaders import PretrainedMixin
from ..runtime import NetRuntime

SD15_DOWN = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
SD15_UP = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")


# config.json keys of the diffusers SD-1.5 family that change the arithmetic and that the compiled networks fix at one
# value: a checkpoint whose config says otherwise must be REFUSED (a swallowed key would load, run and give wrong images)
_FIXED_CONFIG = {
    "act_fn": ("silu",), "use_linear_projection": (False,), "class_embed_type": (None,), "num_class_embeds": (None,),
    "upcast_attention": (False,), "resnet_time_scale_shift": ("default",), "only_cross_attention": (False,),
    "dual_cross_attention": (False,), "addition_embed_type": (None,), "time_cond_proj_dim": (None,),
    "time_embedding_type": ("positional",), "timestep_post_act": (None,), "time_embedding_act_fn": (None,),
    "time_embedding_dim": (None,), "conv_in_kernel": (3,), "conv_out_kernel": (3,),
    "mid_block_type": ("UNetMidBlock2DCrossAttn",), "transformer_layers_per_block": (1,),
    "reverse_transformer_layers_per_block": (None,), "encoder_hid_dim": (None,), "encoder_hid_dim_type": (None,),
    "flip_sin_to_cos": (True,), "freq_shift": (0,), "mid_block_scale_factor": (1, 1.0), "downsample_padding": (1,),
    "num_attention_heads": (None,), "center_input_sample": (False,), "dropout": (0, 0.0), "attention_type": ("default",),
    "cross_attention_norm": (None,), "class_embeddings_concat": (False,), "resnet_skip_time_act": (False,),
    "resnet_out_scale_factor": (1, 1.0), "mid_block_only_cross_attention": (None,), "addition_time_embed_dim": (None,),
    "projection_class_embeddings_input_dim": (None,), "controlnet_conditioning_channel_order": ("rgb",),
    "brushnet_conditioning_channel_order": ("rgb",), "global_pool_conditions": (False,),
    "addition_embed_type_num_heads": (64,),
}


def check_fixed_config(cls_name: str, extra: dict):
    """Constructor keyword arguments the model does not name (the rest of a diffusers config.json): keys of
    `_FIXED_CONFIG` must carry the value the HIP networks are compiled for; bookkeeping keys ("_class_name", ...) and
    unknown keys pass (newer diffusers versions add options whose defaults keep the SD-1.5 arithmetic)."""
    for k, v in extra.items():
        if k.startswith("_") or k not in _FIXED_CONFIG:
            continue
        if isinstance(v, list):
            v = tuple(v)
        if v not in _FIXED_CONFIG[k]:
            raise L.PPError(f"{cls_name}: config {k}={v!r} is not implemented on the HIP path "
                            f"(supported: {', '.join(repr(x) for x in _FIXED_CONFIG[k])})")


class Output(SimpleNamespace):
    """return_dict=True container (attribute access like diffusers BaseOutput)."""

    def __getitem__(self, i):
        return list(self.__dict__.values())[i]


class _HipModel(PretrainedMixin):
    """Base: config namespace, parameter loading into the packed device buffer, runtime handle."""
    kind = "unet"

    def __init__(self, in_channels, block_out_channels, layers_per_block, attention_head_dim, cross_attention_dim,
                 norm_num_groups, norm_eps, down_block_types, up_block_types, device, dtype, **net_kw):
        L.dtype_code(dtype)      # bf16 or fp16 storage (fp32 accumulate either way); raises PPError for anything else
        check_fixed_config(type(self).__name__, getattr(self, "_extra_config", {}))
        if not isinstance(attention_head_dim, int):
            raise L.PPError("per-block attention_head_dim tuples are not supported (SD-1.5 uses 8 everywhere)")
        self._device = torch.device(device)
        self._dtype = dtype
        self.net = SDNet(self.kind, in_channels, block_out_channels, layers_per_block, attention_head_dim,
                         cross_attention_dim, norm_num_groups, norm_eps, down_block_types, up_block_types, dtype=dtype,
                         **net_kw)
        self.rt = NetRuntime(self.net, self._device)
        self._sd = None

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    def to(self, *a, **k):
        """`.to(device)` / `.to(same dtype)` are no-ops (the packed parameter buffer lives on the device it was loaded
        to, in the model's 16-bit format).  A DIFFERENT dtype or device cannot be honoured after packing and is refused:
        silently ignoring `.to(torch.float16)` on a bf16 network would hand bf16 bits to an fp16 consumer."""
        want_dtype = k.get("dtype")
        want_dev = k.get("device")
        for v in a:
            if isinstance(v, torch.dtype):
                want_dtype = v
            elif isinstance(v, (str, torch.device)):
                want_dev = v
            elif torch.is_tensor(v):
                want_dtype, want_dev = v.dtype, v.device
        if want_dtype is not None and want_dtype != self._dtype:
            raise L.PPError(f"{type(self).__name__}.to({want_dtype}): the network was packed as {self._dtype}; build it "
                            f"with dtype={want_dtype} (from_pretrained(torch_dtype=...)) instead")
        if want_dev is not None:
            d = torch.device(want_dev)
            if d.type != self._device.type or (d.index is not None and self._device.index is not None
                                               and d.index != self._device.index):
                raise L.PPError(f"{type(self).__name__}.to({d}): the packed parameters live on {self._device}")
        return self

    def eval(self):
        return self

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, keep_state_dict: bool = False,
                        materialize: bool = True):
        """diffusers-format state dict (keys as in SURVEY.md section 8b)."""
        self.net.load_state_dict(sd, self._device, materialize)
        self.rt.key = None
        self._sd = sd if keep_state_dict else None
        return self

    def param_buffer(self) -> torch.Tensor:
        """The single contiguous device buffer holding every parameter (unit of the RCCL weight broadcast).  Whoever
        writes into it in place calls `params_changed()` afterwards (dist.broadcast_models does)."""
        return self.net.params.buf

    def params_changed(self):
        """The parameter values were rewritten in place: everything derived from them at run time (hoisted cross-attention
        K / V^T and the folded Wq / Wo, the per-schedule time-embedding table) is recomputed on the next call."""
        self.net.params.touch()
        self.rt._ctx_id = None
        return self

    def _nctx(self, ehs: torch.Tensor) -> int:
        if ehs is None or ehs.dim() != 3 or ehs.shape[-1] != self.net.ctx_dim:
            raise ValueError(f"encoder_hidden_states must be [B, L, {self.net.ctx_dim}]")
        return ehs.shape[1]
