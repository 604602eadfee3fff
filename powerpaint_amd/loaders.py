"""Checkpoint loading for the HIP models: the `from_pretrained` / `load_model` calls of
/root/reference/app.py:84-200 against directories in the diffusers / transformers layout.

    <dir>/[<subfolder>/]config.json                                   constructor arguments
    <dir>/[<subfolder>/]diffusion_pytorch_model.safetensors | .bin    UNet / ControlNet / BrushNet / VAE weights
    <dir>/[<subfolder>/]model.safetensors | pytorch_model.bin         CLIP text encoder weights

A name that is not a directory is looked up in the local Hugging Face cache (`huggingface_hub.snapshot_download(...,
local_files_only=True)`): nothing is ever downloaded.  `load_model(model, file)` is the stand-in for
`safetensors.torch.load_model` (app.py:110-111,188-191), which needs an nn.Module; it accepts .safetensors and
torch-pickled files and returns (missing, unexpected) like the original.
"""
import json
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L

WEIGHT_FILES = ("diffusion_pytorch_model.safetensors", "model.safetensors", "diffusion_pytorch_model.fp16.safetensors",
                "model.fp16.safetensors", "diffusion_pytorch_model.bin", "pytorch_model.bin")


def resolve_dir(name_or_path, subfolder: Optional[str] = None, local_files_only: bool = True, **hub_kw) -> str:
    """Directory holding config.json + weights.  No network access, ever."""
    path = str(name_or_path)
    if not os.path.isdir(path):
        try:
            from huggingface_hub import snapshot_download
            pattern = [f"{subfolder}/*"] if subfolder else None
            path = snapshot_download(path, local_files_only=True, allow_patterns=pattern,
                                     revision=hub_kw.get("revision"))
        except Exception as e:
            raise L.PPError(f"'{name_or_path}' is neither a directory nor in the local Hugging Face cache "
                            f"(downloads are disabled): {type(e).__name__}: {e}") from None
    d = os.path.join(path, subfolder) if subfolder else path
    if not os.path.isdir(d):
        raise L.PPError(f"checkpoint folder {d} does not exist")
    return d


def read_state_dict(file: str) -> Dict[str, torch.Tensor]:
    if file.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(file, device="cpu")
    sd = torch.load(file, map_location="cpu", weights_only=True)
    return sd.get("state_dict", sd) if isinstance(sd, dict) else sd


def find_weights(d: str) -> str:
    for f in WEIGHT_FILES:
        if os.path.isfile(os.path.join(d, f)):
            return os.path.join(d, f)
    raise L.PPError(f"no weight file in {d} (looked for {', '.join(WEIGHT_FILES)})")


def read_config(d: str) -> dict:
    f = os.path.join(d, "config.json")
    if not os.path.isfile(f):
        raise L.PPError(f"{f} not found")
    with open(f) as fh:
        cfg = json.load(fh)
    return {k: v for k, v in cfg.items() if not k.startswith("_")}


def load_model(model, filename: str, strict: bool = True, device="cpu") -> Tuple[list, list]:
    """`safetensors.torch.load_model` for the HIP models (and for nn.Modules): (missing, unexpected) keys."""
    sd = read_state_dict(filename)
    if isinstance(model, torch.nn.Module):
        r = model.load_state_dict(sd, strict=False)
        missing, unexpected = list(r.missing_keys), list(r.unexpected_keys)
    else:
        spec = model.state_dict_spec() if hasattr(model, "state_dict_spec") else model.net.state_dict_spec()
        missing = [k for k in spec if k not in sd]
        unexpected = [k for k in sd if k not in spec]
        if not missing:
            model.load_state_dict(sd)
    if strict and (missing or unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict for {type(model).__name__}: "
                           f"missing {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                           f"unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
    if missing and not isinstance(model, torch.nn.Module):
        raise RuntimeError(f"{type(model).__name__}: state dict misses {len(missing)} keys, e.g. {missing[:3]}")
    return missing, unexpected


class PretrainedMixin:
    """`Model.from_pretrained(name_or_dir, subfolder=..., torch_dtype=..., local_files_only=...)` for the HIP models."""

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder: Optional[str] = None, torch_dtype=None,
                        device="cuda", local_files_only: bool = True, revision=None, variant=None,
                        low_cpu_mem_usage=None, **kw):
        d = resolve_dir(pretrained_model_name_or_path, subfolder, revision=revision)
        cfg = read_config(d)
        cfg.update(kw)
        # torch_dtype = the 16-bit format the network computes in (the reference's default is fp16, app.py:548,559) for
        # the networks that take one (UNet / BrushNet / ControlNet); fp32 / None -> bf16; VAE and CLIP tower: bf16
        import inspect
        import torch
        if torch_dtype in (torch.float16, torch.bfloat16) and "dtype" in inspect.signature(cls.__init__).parameters:
            cfg["dtype"] = torch_dtype
        model = cls(device=device, **cfg)
        sd = read_state_dict(find_weights(d))
        model.load_state_dict(sd, keep_state_dict=True) if _keeps(model) else model.load_state_dict(sd)
        return model


def _keeps(model) -> bool:
    import inspect
    return "keep_state_dict" in inspect.signature(model.load_state_dict).parameters


def load_scheduler(d: str):
    """`<dir>/scheduler_config.json` -> the fused scheduler of the same `_class_name` (DDIM / PNDM / DPM-Solver++ / UniPC);
    other classes are refused by name -- pass a scheduler object (any duck-typed one works) instead."""
    from .schedulers import SCHEDULERS
    f = os.path.join(d, "scheduler_config.json")
    if not os.path.isfile(f):
        raise L.PPError(f"{f} not found")
    with open(f) as fh:
        cfg = json.load(fh)
    name = cfg.get("_class_name", "")
    if name not in SCHEDULERS:
        raise L.PPError(f"scheduler class '{name}' has no fused implementation here ({', '.join(SCHEDULERS)}); "
                        "construct the pipeline with scheduler=<object> instead")
    return SCHEDULERS[name].from_config({k: v for k, v in cfg.items() if not k.startswith("_")})


class PipelinePretrainedMixin:
    """`Pipeline.from_pretrained(dir_or_cached_repo, **component_overrides)` over a diffusers pipeline folder
    (`model_index.json` + one sub-folder per component), as app.py:90-92,168-176 call it.  Components: `unet`, `vae`,
    `text_encoder` -> the HIP models, `tokenizer` -> transformers' CLIPTokenizer, `scheduler` -> `load_scheduler`;
    `safety_checker` / `feature_extractor` / `image_encoder` are left out (None).  Keyword arguments naming a component
    (e.g. `brushnet=`, `text_encoder_brushnet=`, `unet=`) are used as given instead of being loaded."""

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, torch_dtype=None, device="cuda", local_files_only=True,
                        revision=None, low_cpu_mem_usage=None, **overrides):
        import inspect
        from . import models as PM
        root = resolve_dir(pretrained_model_name_or_path, None, revision=revision)
        f = os.path.join(root, "model_index.json")
        if not os.path.isfile(f):
            raise L.PPError(f"{f} not found: not a diffusers pipeline folder")
        with open(f) as fh:
            index = {k: v for k, v in json.load(fh).items() if not k.startswith("_")}
        loaders = {"unet": lambda: PM.UNet2DConditionModel.from_pretrained(root, subfolder="unet", device=device,
                                                                           torch_dtype=torch_dtype),
                   "vae": lambda: PM.AutoencoderKL.from_pretrained(root, subfolder="vae", device=device),
                   "text_encoder": lambda: PM.CLIPTextModel.from_pretrained(root, subfolder="text_encoder", device=device),
                   "scheduler": lambda: load_scheduler(os.path.join(root, "scheduler"))}

        def tokenizer():
            import transformers
            return transformers.CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))

        loaders["tokenizer"] = tokenizer
        wanted = [p for p in inspect.signature(cls.__init__).parameters if p != "self"]
        comps = {}
        for name in wanted:
            if name in overrides:
                comps[name] = overrides.pop(name)
            elif name in loaders and name in index and index[name] and index[name][0] is not None:
                comps[name] = loaders[name]()
        unknown = [k for k in overrides if k not in wanted]
        if unknown:
            raise TypeError(f"{cls.__name__}.from_pretrained got unexpected components {unknown}")
        return cls(**comps)
