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
        """The single contiguous device buffer holding every parameter (unit of the RCCL weight broadcast)."""
        return self.net.params.buf

    def _nctx(self, ehs: torch.Tensor) -> int:
        if ehs is None or ehs.dim() != 3 or ehs.shape[-1] != self.net.ctx_dim:
            raise ValueError(f"encoder_hidden_states must be [B, L, {self.net.ctx_dim}]")
        return ehs.shape[1]
