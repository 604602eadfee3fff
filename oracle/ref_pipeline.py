"""Oracle (TEST INFRASTRUCTURE, build container only): run the REFERENCE'S OWN pipeline `__call__` without diffusers.

/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py imports the un-installable diffusers, but the body of
`StableDiffusionInpaintPipeline` (its `__call__` loop, `_encode_prompt`, `prepare_latents`, `prepare_mask_latents`,
`_encode_vae_image`, `get_timesteps`, `prepare_extra_step_kwargs`, `check_inputs`, ...) only *uses* a handful of
diffusers names.  This module lifts the class and the module-level `prepare_mask_and_masked_image` out of the file by
AST, re-bases the class on a ten-line stand-in for `DiffusionPipeline` (component registry, progress bar, execution
device) and executes the reference source unmodified, with the oracle's UNet / scheduler / VAE and a transformers CLIP
text encoder as components.  tests/golden/make_ref_pipeline_call.py uses it to freeze what the reference's own loop
produces; nothing here is imported by the product or needed on the GPU box.
"""
import ast
import inspect
import types
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import PIL
import torch
from packaging import version

REF = "/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py"


class _PipeBase:
    """The few `DiffusionPipeline` services the reference class body calls."""

    def register_modules(self, **mods):
        for k, v in mods.items():
            setattr(self, k, v)

    def register_to_config(self, **kw):
        self.config = types.SimpleNamespace(**kw)

    @property
    def _execution_device(self):
        return torch.device("cpu")

    @property
    def device(self):
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        class _Bar:
            def __enter__(s):
                return s

            def __exit__(s, *a):
                return False

            def update(s, *a):
                pass
        return _Bar()

    def maybe_free_model_hooks(self):
        pass


class _PassThroughImageProcessor:
    """`VaeImageProcessor` stand-in for latent-space outputs: postprocess(output_type="latent") returns its input."""

    def __init__(self, vae_scale_factor=8, **kw):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image, height=None, width=None):
        """Tensor inputs only (already [B, C, H, W] in [-1, 1]): what diffusers' preprocess does to them is nothing."""
        if not torch.is_tensor(image) or image.ndim != 4:
            raise NotImplementedError("the harness feeds [B, C, H, W] tensors")
        return image

    def postprocess(self, image, output_type="latent", do_denormalize=None):
        if output_type != "latent":
            raise NotImplementedError("the harness freezes latents; decode with the oracle VAE separately")
        return image


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor for CPU generators (single or list)."""
    if isinstance(generator, list):
        shape1 = (1,) + tuple(shape[1:])
        return torch.cat([torch.randn(shape1, generator=g, dtype=dtype) for g in generator], dim=0).to(device)
    return torch.randn(shape, generator=generator, dtype=dtype).to(device)


def load_reference_pipeline_class():
    tree = ast.parse(open(REF).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_mask_and_masked_image"][0]
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "StableDiffusionInpaintPipeline"][0]
    cls.bases = [ast.Name(id="_PipeBase", ctx=ast.Load())]
    mod = ast.Module(body=[fn, cls], type_ignores=[])
    ast.fix_missing_locations(mod)
    quiet = types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    dummy = lambda name: type(name, (), {})                                   # noqa: E731
    ns = dict(inspect=inspect, Any=Any, Callable=Callable, Dict=Dict, List=List, Optional=Optional, Union=Union,
              np=np, PIL=PIL, torch=torch, version=version, _PipeBase=_PipeBase, FrozenDict=dict,
              VaeImageProcessor=_PassThroughImageProcessor, randn_tensor=randn_tensor, logger=quiet,
              deprecate=lambda *a, **k: None, is_accelerate_available=lambda: False,
              is_accelerate_version=lambda *a: False,
              StableDiffusionPipelineOutput=lambda images, nsfw_content_detected: types.SimpleNamespace(
                  images=images, nsfw_content_detected=nsfw_content_detected))
    for name in ("LoraLoaderMixin", "TextualInversionLoaderMixin", "FromSingleFileMixin", "AsymmetricAutoencoderKL",
                 "AutoencoderKL", "UNet2DConditionModel", "CLIPImageProcessor", "CLIPTextModel", "CLIPTokenizer",
                 "StableDiffusionSafetyChecker", "KarrasDiffusionSchedulers", "DiffusionPipeline"):
        ns[name] = dummy(name)
    exec(compile(mod, REF, "exec"), ns)
    return ns["StableDiffusionInpaintPipeline"], ns["prepare_mask_and_masked_image"]


REF_V2 = "/root/reference/powerpaint/pipelines/pipeline_PowerPaint_Brushnet_CA.py"


def load_reference_brushnet_pipeline_class(brushnet_cls):
    """`StableDiffusionPowerPaintBrushNetPipeline` (pipeline_PowerPaint_Brushnet_CA.py) the same way; `brushnet_cls` is
    the class the file's `isinstance(brushnet, BrushNetModel)` checks must recognise (the oracle's BrushNetModel)."""
    tree = ast.parse(open(REF_V2).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "retrieve_timesteps"][0]
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "StableDiffusionPowerPaintBrushNetPipeline"][0]
    cls.bases = [ast.Name(id="_PipeBase", ctx=ast.Load())]
    mod = ast.Module(body=[fn, cls], type_ignores=[])
    ast.fix_missing_locations(mod)
    quiet = types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    dummy = lambda name: type(name, (), {})                                   # noqa: E731
    import torch.nn.functional as F
    ns = dict(inspect=inspect, Any=Any, Callable=Callable, Dict=Dict, List=List, Optional=Optional, Union=Union,
              Tuple=__import__("typing").Tuple, np=np, PIL=PIL, torch=torch, F=F, _PipeBase=_PipeBase, FrozenDict=dict,
              VaeImageProcessor=_PassThroughImageProcessor, randn_tensor=randn_tensor, logger=quiet,
              deprecate=lambda *a, **k: None, is_compiled_module=lambda m: False, is_torch_version=lambda *a: False,
              USE_PEFT_BACKEND=False, adjust_lora_scale_text_encoder=lambda *a, **k: None,
              scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None,
              BrushNetModel=brushnet_cls, replace_example_docstring=lambda doc: (lambda f: f), EXAMPLE_DOC_STRING="",
              StableDiffusionPipelineOutput=lambda images, nsfw_content_detected: types.SimpleNamespace(
                  images=images, nsfw_content_detected=nsfw_content_detected))
    for name in ("LoraLoaderMixin", "TextualInversionLoaderMixin", "FromSingleFileMixin", "IPAdapterMixin",
                 "StableDiffusionMixin", "AutoencoderKL", "UNet2DConditionModel", "ImageProjection", "CLIPImageProcessor",
                 "CLIPTextModel", "CLIPTokenizer", "CLIPVisionModelWithProjection", "StableDiffusionSafetyChecker",
                 "KarrasDiffusionSchedulers", "DiffusionPipeline", "PipelineImageInput", "MultiControlNetModel"):
        ns[name] = dummy(name)
    exec(compile(mod, REF_V2, "exec"), ns)
    return ns["StableDiffusionPowerPaintBrushNetPipeline"]


REF_CN = "/root/reference/powerpaint/pipelines/pipeline_PowerPaint_ControlNet.py"


def load_reference_controlnet_pipeline_class(controlnet_cls):
    """`StableDiffusionControlNetInpaintPipeline` (pipeline_PowerPaint_ControlNet.py) the same way; `controlnet_cls` is
    the class its `isinstance(controlnet, ControlNetModel)` checks must recognise (the oracle's ControlNetModel)."""
    import warnings
    import torch.nn.functional as F
    tree = ast.parse(open(REF_CN).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_mask_and_masked_image"][0]
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "StableDiffusionControlNetInpaintPipeline"][0]
    cls.bases = [ast.Name(id="_PipeBase", ctx=ast.Load())]
    mod = ast.Module(body=[fn, cls], type_ignores=[])
    ast.fix_missing_locations(mod)
    quiet = types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None)
    dummy = lambda name: type(name, (), {})                                   # noqa: E731
    ns = dict(inspect=inspect, warnings=warnings, Any=Any, Callable=Callable, Dict=Dict, List=List, Optional=Optional,
              Union=Union, Tuple=__import__("typing").Tuple, np=np, PIL=PIL, torch=torch, F=F, _PipeBase=_PipeBase,
              VaeImageProcessor=_PassThroughImageProcessor, randn_tensor=randn_tensor, logger=quiet,
              deprecate=lambda *a, **k: None, is_compiled_module=lambda m: False,
              is_accelerate_available=lambda: False, is_accelerate_version=lambda *a: False,
              ControlNetModel=controlnet_cls, replace_example_docstring=lambda doc: (lambda f: f), EXAMPLE_DOC_STRING="",
              StableDiffusionPipelineOutput=lambda images, nsfw_content_detected: types.SimpleNamespace(
                  images=images, nsfw_content_detected=nsfw_content_detected))
    for name in ("LoraLoaderMixin", "TextualInversionLoaderMixin", "FromSingleFileMixin", "AsymmetricAutoencoderKL",
                 "AutoencoderKL", "UNet2DConditionModel", "CLIPImageProcessor", "CLIPTextModel", "CLIPTokenizer",
                 "StableDiffusionSafetyChecker", "KarrasDiffusionSchedulers", "DiffusionPipeline", "MultiControlNetModel"):
        ns[name] = dummy(name)
    exec(compile(mod, REF_CN, "exec"), ns)
    return ns["StableDiffusionControlNetInpaintPipeline"]
