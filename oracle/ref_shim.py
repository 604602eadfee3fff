"""Oracle (TEST INFRASTRUCTURE): import the REFERENCE'S OWN model files without diffusers.

`/root/reference/powerpaint/models/{unet_2d_condition,unet_2d_blocks,BrushNet_CA}.py` import the un-vendored,
un-installable `diffusers==0.27.0`.  This module installs a stand-in `diffusers` package whose *leaf* modules
(ResnetBlock2D, Transformer2DModel, Downsample2D, Upsample2D, Timesteps, TimestepEmbedding) are the oracle's
restatements (oracle/sd_modules.py), and whose remaining names are inert placeholders.  Everything the fork itself
implements -- constructor loops, block wiring, BrushNet residual routing (`.pop(0)` order, "first skip excludes the
residual"), zero-conv placement, `from_unet` weight copy, `return_res_samples` -- then runs from the reference's own
source, unmodified, in this container.  tests/golden/make_ref_wiring.py uses it to generate golden tensors; the
fixtures travel, this module and /root/reference do not need to exist on the GPU box.

Nothing here is imported by the product.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import inspect
import os
import sys
import types
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import sd_modules as OM

REF_ROOT = "/root/reference"


# ---------------------------------------------------------------------------------------------- placeholders
class _Dummy:
    """Inert placeholder for every diffusers name the fork imports but never exercises on the SD-1.5 path."""

    def __init__(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__}: diffusers placeholder instantiated -- not on the SD-1.5 hot path")


def _make_dummy(name):
    return type(name, (_Dummy,), {})


class _ShimModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _make_dummy(name)
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname == "diffusers" or fullname.startswith("diffusers."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _ShimModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


# ---------------------------------------------------------------------------------------------- real stand-ins
class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def register_to_config(init):
    sig = inspect.signature(init)

    def wrapped(self, *args, **kwargs):
        ba = sig.bind(self, *args, **kwargs)
        ba.apply_defaults()
        cfg = {k: v for k, v in ba.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        object.__setattr__(self, "_pp_config", _Config(cfg))

    return wrapped


class ConfigMixin:
    config_name = None

    @property
    def config(self):
        return self._pp_config

    def register_to_config(self, **kw):
        self._pp_config.update(kw)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


@dataclass
class BaseOutput:
    pass


class _Logger:
    def __getattr__(self, k):
        return lambda *a, **kw: None


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _Logger()


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.n, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, t):
        return OM.timestep_embedding(t, self.n, self.flip, self.shift)


class TimestepEmbedding(OM.TimestepEmbedding):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None, cond_proj_dim=None,
                 sample_proj_bias=True):
        assert act_fn == "silu" and post_act_fn is None and cond_proj_dim is None and out_dim is None
        super().__init__(in_channels, time_embed_dim)

    def forward(self, sample, condition=None):
        assert condition is None
        return super().forward(sample)


class ResnetBlock2D(OM.ResnetBlock2D):
    def __init__(self, *, in_channels, out_channels=None, temb_channels=512, eps=1e-6, groups=32, dropout=0.0,
                 time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True, **kw):
        assert time_embedding_norm == "default" and non_linearity in ("silu", "swish") and output_scale_factor == 1.0
        assert dropout == 0.0
        super().__init__(in_channels, out_channels or in_channels, temb_channels, groups, eps)


class Downsample2D(OM.Downsample2D):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", **kw):
        assert use_conv and (out_channels in (None, channels))
        super().__init__(channels, padding)


class Upsample2D(OM.Upsample2D):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv", **kw):
        assert use_conv and not use_conv_transpose and (out_channels in (None, channels))
        super().__init__(channels)


class Transformer2DModel(OM.Transformer2DModel):
    def __init__(self, num_attention_heads=16, attention_head_dim=88, in_channels=None, num_layers=1,
                 cross_attention_dim=None, norm_num_groups=32, use_linear_projection=False, only_cross_attention=False,
                 upcast_attention=False, attention_type="default", **kw):
        assert not use_linear_projection and not only_cross_attention and not upcast_attention
        super().__init__(num_attention_heads, attention_head_dim, in_channels, cross_attention_dim, num_layers,
                         norm_num_groups)

    def forward(self, hidden_states, encoder_hidden_states=None, cross_attention_kwargs=None, attention_mask=None,
                encoder_attention_mask=None, return_dict=True, **kw):
        assert attention_mask is None and encoder_attention_mask is None
        return super().forward(hidden_states, encoder_hidden_states=encoder_hidden_states)


def get_activation(name):
    assert name in ("silu", "swish")
    return nn.SiLU()


_OVERRIDES = {
    "diffusers.configuration_utils": dict(ConfigMixin=ConfigMixin, register_to_config=register_to_config),
    "diffusers.loaders": dict(PeftAdapterMixin=type("PeftAdapterMixin", (), {}),
                              UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {})),
    "diffusers.models.activations": dict(get_activation=get_activation),
    "diffusers.models.embeddings": dict(Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding),
    "diffusers.models.modeling_utils": dict(ModelMixin=ModelMixin),
    "diffusers.models.resnet": dict(ResnetBlock2D=ResnetBlock2D, Downsample2D=Downsample2D, Upsample2D=Upsample2D),
    "diffusers.models.transformers.transformer_2d": dict(Transformer2DModel=Transformer2DModel),
    "diffusers.utils": dict(USE_PEFT_BACKEND=False, BaseOutput=BaseOutput, deprecate=lambda *a, **k: None,
                            logging=_Logging, scale_lora_layers=lambda *a, **k: None,
                            unscale_lora_layers=lambda *a, **k: None, is_torch_version=lambda *a, **k: True),
    "diffusers.utils.torch_utils": dict(apply_freeu=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("freeu"))),
}


def _populate(module):
    for k, v in _OVERRIDES.get(module.__name__, {}).items():
        setattr(module, k, v)


_installed = False


def install():
    global _installed
    if not _installed:
        sys.meta_path.insert(0, _Finder())
        _installed = True


def load_reference_models():
    """Returns (UNet2DConditionModel, BrushNetModel) classes defined by the reference's own source files."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("the reference tree is only available in the build container")
    install()
    pkg_name = "ref_powerpaint_models"
    if pkg_name not in sys.modules:
        pkg = types.ModuleType(pkg_name)
        pkg.__path__ = [os.path.join(REF_ROOT, "powerpaint", "models")]
        sys.modules[pkg_name] = pkg
    mods = {}
    for name in ("unet_2d_blocks", "unet_2d_condition", "BrushNet_CA"):
        full = f"{pkg_name}.{name}"
        if full not in sys.modules:
            spec = importlib.util.spec_from_file_location(full, os.path.join(REF_ROOT, "powerpaint", "models", name + ".py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[full] = m
            spec.loader.exec_module(m)
        mods[name] = sys.modules[full]
    return mods["unet_2d_condition"].UNet2DConditionModel, mods["BrushNet_CA"].BrushNetModel
