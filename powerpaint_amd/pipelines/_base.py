"""Shared host plumbing of the three PowerPaint pipelines (everything either side of the denoising loop).

Restates, for the parts the hot path needs, the helpers of
/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py (check_inputs 553-602, _encode_prompt 317-518,
prepare_latents 604-655, prepare_mask_latents 671-710, get_timesteps 713-720, prepare_mask_and_masked_image 39-153).
VAE / CLIP / tokenizer are duck-typed collaborators passed in by the caller (out of scope of this build,
SURVEY.md section 8f); every entry point also accepts ready-made latents / embeddings so the loop can be driven
without them.
"""
import inspect
from types import SimpleNamespace
from typing import List, Optional, Union

import numpy as np
import torch

from .. import _lib as L
from ..loaders import PipelinePretrainedMixin


class StableDiffusionPipelineOutput(SimpleNamespace):
    def __getitem__(self, i):
        return (self.images, self.nsfw_content_detected)[i]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def hip_mask_prep(mode: int, a: torch.Tensor, b: Optional[torch.Tensor], out_shape, batch, c, h, w, ho=0, wo=0):
    """Bit-exact mask ops on the device (pp_mask_prep)."""
    a = a.contiguous().float()
    out = torch.empty(out_shape, dtype=torch.float32, device=a.device)
    bp = b.contiguous().float() if b is not None else None
    L.check(L.lib().pp_mask_prep(mode, a.data_ptr(), bp.data_ptr() if bp is not None else None, out.data_ptr(), batch,
                                 c, h, w, ho, wo, _stream()), "pp_mask_prep")
    return out


def prepare_mask_and_masked_image(image, mask, height, width, device, return_image: bool = False):
    """pipeline_PowerPaint.py:39-153.  Tensor and PIL/ndarray inputs; binarisation and masking run in the HIP kernel."""
    if image is None:
        raise ValueError("`image` input cannot be undefined.")
    if mask is None:
        raise ValueError("`mask_image` input cannot be undefined.")
    if isinstance(image, torch.Tensor):
        if not isinstance(mask, torch.Tensor):
            raise TypeError(f"`image` is a torch.Tensor but `mask` (type: {type(mask)} is not")
        if image.ndim == 3:
            assert image.shape[0] == 3, "Image outside a batch should be of shape (3, H, W)"
            image = image.unsqueeze(0)
        if mask.ndim == 2:
            mask = mask.unsqueeze(0).unsqueeze(0)
        if mask.ndim == 3:
            mask = mask.unsqueeze(0) if mask.shape[0] == 1 else mask.unsqueeze(1)
        assert image.ndim == 4 and mask.ndim == 4, "Image and Mask must have 4 dimensions"
        assert image.shape[-2:] == mask.shape[-2:], "Image and Mask must have the same spatial dimensions"
        assert image.shape[0] == mask.shape[0], "Image and Mask must have the same batch size"
        if image.min() < -1 or image.max() > 1:
            raise ValueError("Image should be in [-1, 1] range")
        if mask.min() < 0 or mask.max() > 1:
            raise ValueError("Mask should be in [0, 1] range")
        image = image.to(device=device, dtype=torch.float32)
        mask = mask.to(device=device, dtype=torch.float32)
    elif isinstance(mask, torch.Tensor):
        raise TypeError(f"`mask` is a torch.Tensor but `image` (type: {type(image)} is not")
    else:
        import PIL.Image
        if isinstance(image, (PIL.Image.Image, np.ndarray)):
            image = [image]
        if isinstance(image, list) and isinstance(image[0], PIL.Image.Image):
            image = [i.resize((width, height), resample=PIL.Image.LANCZOS) for i in image]
            image = np.concatenate([np.array(i.convert("RGB"))[None, :] for i in image], axis=0)
        elif isinstance(image, list) and isinstance(image[0], np.ndarray):
            image = np.concatenate([i[None, :] for i in image], axis=0)
        image = torch.from_numpy(image.transpose(0, 3, 1, 2)).to(dtype=torch.float32) / 127.5 - 1.0
        if isinstance(mask, (PIL.Image.Image, np.ndarray)):
            mask = [mask]
        if isinstance(mask, list) and isinstance(mask[0], PIL.Image.Image):
            mask = [i.resize((width, height), resample=PIL.Image.LANCZOS) for i in mask]
            mask = np.concatenate([np.array(m.convert("L"))[None, None, :] for m in mask], axis=0)
            mask = mask.astype(np.float32) / 255.0
        elif isinstance(mask, list) and isinstance(mask[0], np.ndarray):
            mask = np.concatenate([m[None, None, :] for m in mask], axis=0)
        image = image.to(device)
        mask = torch.from_numpy(np.ascontiguousarray(mask)).to(device=device, dtype=torch.float32)
    B, _, H, W = image.shape
    mask = hip_mask_prep(0, mask, None, mask.shape, mask.shape[0], 1, H, W)               # binarise  (:143-144)
    masked_image = hip_mask_prep(1, image, mask, image.shape, B, image.shape[1], H, W)    # image*(mask<0.5) (:147)
    if return_image:
        return mask, masked_image, image
    return mask, masked_image


class PipelineBase(PipelinePretrainedMixin):
    """Minimal DiffusionPipeline stand-in: component registry, device, progress bar, scheduler kwargs probing."""

    def register_modules(self, **mods):
        self._modules = mods
        for k, v in mods.items():
            setattr(self, k, v)

    @property
    def _execution_device(self):
        return self.unet.device

    @property
    def device(self):
        return self.unet.device

    def to(self, *a, **k):
        """Forwards the request to every component that has a `.to` (int arguments = cuda indices): the packed HIP
        networks are no-ops for their own device / dtype and REFUSE a different one (models/_base.py), so
        `pipe.to(torch.float16)` on a bf16 pipeline raises instead of being silently ignored."""
        a = tuple(torch.device("cuda", v) if isinstance(v, int) and not isinstance(v, bool) else v for v in a)
        if isinstance(k.get("device"), int):
            k = dict(k, device=torch.device("cuda", k["device"]))
        mods = [(name, m) for name, m in getattr(self, "_modules", {}).items()
                if m is not None and hasattr(m, "to") and name != "scheduler"]
        # the packed HIP networks only VALIDATE the request (and may refuse it): ask them before any nn.Module is converted,
        # so that a refused `pipe.to(torch.float16)` leaves the pipeline as it was
        mods.sort(key=lambda nm: isinstance(nm[1], torch.nn.Module))
        for name, m in mods:
            r = m.to(*a, **k)
            if isinstance(m, torch.nn.Module) and r is not None:
                setattr(self, name, r)
                self._modules[name] = r
        return self

    def progress_bar(self, iterable=None, total=None):
        try:
            from tqdm.auto import tqdm
            cfg = getattr(self, "_progress_bar_config", {"disable": True})
            return tqdm(iterable, total=total, **cfg)
        except Exception:  # pragma: no cover
            class _N:
                def __enter__(s): return s
                def __exit__(s, *a): return False
                def update(s, *a): pass
            return _N()

    def set_progress_bar_config(self, **kw):
        self._progress_bar_config = kw

    # Memory-saving switches of DiffusionPipeline that app.py flips (`enable_model_cpu_offload()`, app.py:199) or users
    # habitually call.  With 288 GB of HBM and static arenas there is nothing to offload or slice: accepted, no effect.
    def enable_model_cpu_offload(self, gpu_id=0, device="cuda"):
        return None

    def enable_sequential_cpu_offload(self, gpu_id=0, device="cuda"):
        return None

    def maybe_free_model_hooks(self):
        return None

    def enable_attention_slicing(self, slice_size="auto"):
        return None

    def disable_attention_slicing(self):
        return None

    def enable_vae_slicing(self):
        return None

    def disable_vae_slicing(self):
        return None

    def enable_vae_tiling(self):
        return None

    def disable_vae_tiling(self):
        return None

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        return None

    def disable_xformers_memory_efficient_attention(self):
        return None

    def prepare_extra_step_kwargs(self, generator, eta):
        """pipeline_PowerPaint.py:536-551."""
        kw = {}
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    @staticmethod
    def _noise_dtype(prompt_embeds):
        """The reference draws the initial noise in prompt_embeds.dtype (prepare_latents(..., prompt_embeds.dtype, ...),
        pipeline_PowerPaint.py:930-940) and only then runs the loop; the HIP loop keeps latents in fp32, so the draw
        happens in the reference's dtype (same generator state, same rounded values) and is widened afterwards."""
        dt = getattr(prompt_embeds, "dtype", torch.float32)
        return dt if dt in (torch.float16, torch.bfloat16, torch.float32) else torch.float32

    def get_timesteps(self, num_inference_steps, strength, device):
        """pipeline_PowerPaint.py:713-720.  The schedulers of this package additionally learn where the loop enters
        (`set_begin_index`): their per-step tables are indexed by a device-side step counter, not by timestep lookup."""
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        timesteps = self.scheduler.timesteps[t_start * self.scheduler.order:]
        if t_start and getattr(self.scheduler, "kind", -1) >= 0:
            self.scheduler.set_begin_index(t_start * self.scheduler.order)
        return timesteps, num_inference_steps - t_start

    def _initial_latents(self, shape, strength, timesteps, latents, init_image, generator, device, noise_dtype,
                         return_image_latents=False, return_all=False):
        """prepare_latents (pipeline_PowerPaint.py:604-655): pure noise * init_noise_sigma at strength 1 (or when the
        caller hands `latents`, which the reference then treats as the noise whatever the strength), otherwise the
        VAE latents of the init image noised to the first timestep of the shortened schedule.  Draw order as in the
        reference: the init image's posterior sample BEFORE the noise.  `return_image_latents` (a 4-channel UNet,
        :927-928): the init image is encoded in every case.  return_all -> (latents, noise, image_latents | None)."""
        image_latents = None
        if return_image_latents or (latents is None and strength != 1.0):
            if init_image is None:
                raise ValueError("Since strength < 1. initial latents are to be initialised as a combination of Image + "
                                 "Noise.However, either the image or the noise timestep has not been provided.")
            image_latents = self._vae_encode(init_image.to(device=device, dtype=noise_dtype), generator)
        if latents is not None:
            noise = latents.to(device=device)
            out = noise.to(torch.float32) * self.scheduler.init_noise_sigma
        else:
            noise = randn_tensor(shape, generator=generator, device=device, dtype=noise_dtype)
            if strength == 1.0:
                out = noise.to(torch.float32) * self.scheduler.init_noise_sigma
            else:
                t0 = timesteps[:1].repeat(shape[0])
                out = self.scheduler.add_noise(image_latents.to(noise.dtype), noise, t0).to(torch.float32)
        return (out, noise, image_latents) if return_all else out

    # ---- text: promptA/promptB blended by `tradoff` (pipeline_PowerPaint.py:317-518)
    def _text_embeds(self, text_encoder, prompt: Union[str, List[str]], device):
        tok = self.tokenizer
        ids = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True,
                  return_tensors="pt").input_ids
        out = text_encoder(ids.to(device))
        return out[0]

    def _encode_prompt(self, promptA, promptB, t, device, num_images_per_prompt, do_classifier_free_guidance,
                       negative_promptA=None, negative_promptB=None, t_nag=None, prompt_embeds=None,
                       negative_prompt_embeds=None, lora_scale=None, text_encoder=None):
        text_encoder = text_encoder or getattr(self, "text_encoder", None)
        # A call that passes the SAME embedding tensors as the previous one (unchanged storage and version counter) gets the
        # same result OBJECT: the networks then recognise their context and skip the per-prompt set-up (hoisted cross-attention
        # K / V, the folded projections: ~2 ms per call) instead of redoing it for an identical tensor.
        memo_key = None
        if prompt_embeds is not None and torch.is_tensor(prompt_embeds) and \
                (negative_prompt_embeds is None or torch.is_tensor(negative_prompt_embeds)) and \
                (not do_classifier_free_guidance or negative_prompt_embeds is not None):
            ident = lambda t: None if t is None else (id(t), t.data_ptr(), t._version, tuple(t.shape), t.dtype, str(t.device))  # noqa: E731
            memo_key = (ident(prompt_embeds), ident(negative_prompt_embeds), str(device), num_images_per_prompt,
                        bool(do_classifier_free_guidance))
            memo = self.__dict__.setdefault("_prompt_memo", {})
            if memo_key in memo:
                return memo[memo_key][0]
            src_keep = (prompt_embeds, negative_prompt_embeds)     # (an identity only holds while the tensors live)
        if prompt_embeds is None:
            if text_encoder is None or getattr(self, "tokenizer", None) is None:
                raise ValueError("no text encoder / tokenizer registered: pass prompt_embeds (and negative_prompt_embeds)")
            eA = self._text_embeds(text_encoder, promptA, device)
            eB = self._text_embeds(text_encoder, promptB, device)
            prompt_embeds = eA * t + (1 - t) * eB                                          # :423
        bs = prompt_embeds.shape[0]
        prompt_embeds = prompt_embeds.to(device=device)
        _, seq, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, seq, -1)
        if do_classifier_free_guidance:
            if negative_prompt_embeds is None:
                if text_encoder is None:
                    raise ValueError("classifier-free guidance needs negative_prompt_embeds when no text encoder is set")
                # pipeline_PowerPaint.py:441-460: `negative_prompt` IS negative_promptA -- when it is None BOTH
                # unconditional prompts are "" (a lone negative_promptB is ignored), a str is used as given, a list must
                # match the prompt's type and batch size
                if negative_promptA is None:
                    nA = nB = [""] * bs
                elif promptA is not None and type(promptA) is not type(negative_promptA):
                    raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got "
                                    f"{type(negative_promptA)} != {type(promptA)}.")
                elif isinstance(negative_promptA, str):
                    nA, nB = [negative_promptA], [negative_promptB]
                elif bs != len(negative_promptA):
                    raise ValueError(f"`negative_prompt`: {negative_promptA} has batch size {len(negative_promptA)}, but "
                                     f"`prompt`: {promptA} has batch size {bs}. Please make sure that passed "
                                     "`negative_prompt` matches the batch size of `prompt`.")
                else:
                    nA, nB = negative_promptA, negative_promptB
                eA = self._text_embeds(text_encoder, nA, device)
                eB = self._text_embeds(text_encoder, nB, device)
                negative_prompt_embeds = eA * t_nag + (1 - t_nag) * eB                     # :499
            negative_prompt_embeds = negative_prompt_embeds.to(device=device)
            seq = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(
                bs * num_images_per_prompt, seq, -1)
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])             # :516  [uncond, cond]
        if memo_key is not None:
            memo = self.__dict__.setdefault("_prompt_memo", {})
            while len(memo) >= 4:                                  # (a few live entries: ppt-v2 encodes two prompt pairs per call)
                memo.pop(next(iter(memo)))
            memo[memo_key] = (prompt_embeds, src_keep)
        return prompt_embeds

    def _vae_encode(self, image, generator):
        vae = getattr(self, "vae", None)
        if vae is None:
            raise ValueError("no VAE registered: pass the latent-space inputs directly")
        if isinstance(generator, list):
            lat = torch.cat([vae.encode(image[i:i + 1]).latent_dist.sample(generator=generator[i])
                             for i in range(image.shape[0])], dim=0)
        else:
            lat = vae.encode(image).latent_dist.sample(generator=generator)
        return vae.config.scaling_factor * lat

    def _finish(self, latents, output_type, return_dict, prompt_dtype, generator=None):
        if output_type != "latent":
            vae = getattr(self, "vae", None)
            if vae is None:
                raise ValueError('no VAE registered: use output_type="latent"')
            image = vae.decode(latents.to(next(iter(vae.parameters())).dtype) / vae.config.scaling_factor,
                               return_dict=False)[0]
            ip = getattr(self, "image_processor", None)
            if ip is not None:
                image = ip.postprocess(image, output_type=output_type, do_denormalize=[True] * image.shape[0])
                loop = getattr(self, "_loop", None)
                if loop is not None and output_type in ("pil", "np"):     # (the images are on the host: the device is idle anyway)
                    loop.flush_faults()
        else:
            image = latents
        if not return_dict:
            return (image, None)
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None)


def randn_tensor(shape, generator=None, device=None, dtype=torch.float32):
    """diffusers.utils.torch_utils.randn_tensor: CPU generator -> sample on CPU then move."""
    gdev = generator.device.type if isinstance(generator, torch.Generator) else None
    if isinstance(generator, list):
        shape1 = (1,) + tuple(shape[1:])
        return torch.cat([randn_tensor(shape1, g, device, dtype) for g in generator], dim=0)
    if gdev == "cpu" or generator is None and (device is None or torch.device(device).type == "cpu"):
        return torch.randn(shape, generator=generator, dtype=dtype).to(device)
    return torch.randn(shape, generator=generator, device=device, dtype=dtype)
