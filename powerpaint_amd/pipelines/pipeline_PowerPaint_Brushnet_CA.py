"""ppt-v2 (BrushNet) pipeline on the MI355X HIP path: drop-in for
/root/reference/powerpaint/pipelines/pipeline_PowerPaint_Brushnet_CA.py:1026-1497
(`StableDiffusionPowerPaintBrushNetPipeline.__call__`).

Additive keyword extension for synthetic / VAE-free operation: `conditioning_latents=` ([B or 2B, 5, h, w]:
VAE latents of the masked image * scaling_factor concatenated with the latent-resolution mask, i.e. the tensor the
reference builds at :1338-1345) and `prompt_embedsU=` / `negative_prompt_embedsU=` for the UNet's plain prompt.
"""
import inspect
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ._base import PipelineBase, hip_mask_prep, randn_tensor
from ._loop import DenoiseLoop
from .image_processor import VaeImageProcessor


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, **kwargs):
    """pipeline_PowerPaint_Brushnet_CA.py:87-128: a custom `timesteps` list goes to schedulers whose `set_timesteps`
    takes one and is a ValueError for the others -- which, as in diffusers 0.27, is all four of this package
    (DDIM / DPM-Solver++ / PNDM / UniPC take a step count only)."""
    if timesteps is not None:
        if "timesteps" not in set(inspect.signature(scheduler.set_timesteps).parameters.keys()):
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" timestep schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
        return scheduler.timesteps, len(scheduler.timesteps)
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps


class StableDiffusionPowerPaintBrushNetPipeline(PipelineBase):
    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]     # pipeline_PowerPaint_Brushnet_CA.py:177

    def __init__(self, vae=None, text_encoder=None, text_encoder_brushnet=None, tokenizer=None, unet=None,
                 brushnet=None, scheduler=None, safety_checker=None, feature_extractor=None, image_encoder=None,
                 requires_safety_checker: bool = False):
        self.register_modules(vae=vae, text_encoder=text_encoder, text_encoder_brushnet=text_encoder_brushnet,
                              tokenizer=tokenizer, unet=unet, brushnet=brushnet, scheduler=scheduler,
                              safety_checker=safety_checker, feature_extractor=feature_extractor,
                              image_encoder=image_encoder)
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor) if vae is not None else None
        self._loop = None
        self.use_graph = True

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def clip_skip(self):
        """pipeline_PowerPaint_Brushnet_CA.py:1005-1007,1227: `__call__` records its `clip_skip` argument here and, like the
        reference, does NOT pass it on to `encode_prompt` (:1268-1277) -- only a direct `encode_prompt(clip_skip=k)` call
        selects an earlier hidden state."""
        return getattr(self, "_clip_skip", None)

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1 and self.unet.config.time_cond_proj_dim is None

    def encode_prompt(self, prompt, device, num_images_per_prompt, do_cfg, negative_prompt=None, prompt_embeds=None,
                      negative_prompt_embeds=None, lora_scale=None, clip_skip=None):
        """pipeline_PowerPaint_Brushnet_CA.py:442-629 (plain prompt for the UNet); returns cat([neg, pos])."""
        if prompt_embeds is None:
            if self.text_encoder is None:
                raise ValueError("no text encoder registered: pass prompt_embedsU / negative_prompt_embedsU")
            if clip_skip is None:
                prompt_embeds = self._text_embeds(self.text_encoder, prompt, device)
            else:
                # :537-552 -- the hidden state `clip_skip` layers before the last, then the tower's final LayerNorm.
                # (Only a direct `encode_prompt(clip_skip=...)` call gets here: the reference's `__call__` stores its
                # `clip_skip` argument in `self._clip_skip` (:1227) and never hands it to `encode_prompt` (:1268-1277),
                # and this `__call__` does the same.)
                tok = self.tokenizer
                ids = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True,
                          return_tensors="pt").input_ids
                out = self.text_encoder(ids.to(device), output_hidden_states=True)
                prompt_embeds = self.text_encoder.text_model.final_layer_norm(out[-1][-(clip_skip + 1)])
        bs, seq, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.to(device).repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, seq, -1)
        if do_cfg:
            if negative_prompt_embeds is None:
                if self.text_encoder is None:
                    raise ValueError("classifier-free guidance needs negative_prompt_embedsU")
                neg = negative_prompt if negative_prompt is not None else [""] * bs
                if isinstance(neg, str):
                    neg = [neg] * bs
                negative_prompt_embeds = self._text_embeds(self.text_encoder, neg, device)
            negative_prompt_embeds = negative_prompt_embeds.to(device).repeat(1, num_images_per_prompt, 1).view(
                bs * num_images_per_prompt, seq, -1)
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        return prompt_embeds

    @torch.no_grad()
    def __call__(self, promptA: Union[str, List[str]] = None, promptB: Union[str, List[str]] = None,
                 promptU: Union[str, List[str]] = None, tradoff: float = 1.0, tradoff_nag: float = 1.0, image=None,
                 mask=None, height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 timesteps: List[int] = None, guidance_scale: float = 7.5, negative_promptA=None,
                 negative_promptB=None, negative_promptU=None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.FloatTensor] = None,
                 prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None, ip_adapter_image=None,
                 ip_adapter_image_embeds=None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 brushnet_conditioning_scale: Union[float, List[float]] = 1.0, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0,
                 control_guidance_end: Union[float, List[float]] = 1.0, clip_skip: Optional[int] = None,
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"],
                 conditioning_latents: Optional[torch.FloatTensor] = None,
                 prompt_embedsU: Optional[torch.FloatTensor] = None,
                 negative_prompt_embedsU: Optional[torch.FloatTensor] = None, **kwargs):
        callback = kwargs.pop("callback", None)
        callback_steps = kwargs.pop("callback_steps", 1)
        if ip_adapter_image is not None or ip_adapter_image_embeds is not None:
            raise NotImplementedError("IP-Adapter is outside the PowerPaint hot path (never enabled by app.py)")
        if isinstance(control_guidance_start, list):
            control_guidance_start = control_guidance_start[0]
        if isinstance(control_guidance_end, list):
            control_guidance_end = control_guidance_end[0]
        if isinstance(brushnet_conditioning_scale, list):
            brushnet_conditioning_scale = brushnet_conditioning_scale[0]
        prompt = promptA
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        self._guidance_scale = guidance_scale
        self._clip_skip = clip_skip                      # (:1227; recorded, not applied -- see the property)
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self._execution_device
        do_cfg = self.do_classifier_free_guidance
        prompt_embeds = self._encode_prompt(promptA, promptB, tradoff, device, num_images_per_prompt, do_cfg,
                                            negative_promptA, negative_promptB, tradoff_nag,
                                            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                                            text_encoder=self.text_encoder_brushnet)
        prompt_embedsU = self.encode_prompt(promptU, device, num_images_per_prompt, do_cfg, negative_promptU,
                                            prompt_embeds=prompt_embedsU, negative_prompt_embeds=negative_prompt_embedsU)
        nb = batch_size * num_images_per_prompt
        # 4./6.1 conditioning latents: [VAE latents of the masked image | latent-resolution keep-mask]
        if conditioning_latents is None:
            ip = getattr(self, "image_processor", None)
            if ip is None or self.vae is None:
                raise ValueError("no VAE / image processor registered: pass conditioning_latents")
            img = ip.preprocess(image, height=height, width=width).to(device=device, dtype=torch.float32)
            msk = ip.preprocess(mask, height=height, width=width).to(device=device, dtype=torch.float32)
            img = img.repeat_interleave(nb // img.shape[0], dim=0)
            msk = msk.repeat_interleave(nb // msk.shape[0], dim=0)
            if do_cfg and not guess_mode:   # prepare_image (:949-950) duplicates BEFORE the VAE: the two CFG halves get
                img, msk = torch.cat([img] * 2), torch.cat([msk] * 2)   # their own posterior samples (the RNG advances 2x)
            B0, C0, H0, W0 = msk.shape
            original_mask = hip_mask_prep(3, msk, None, (B0, 1, H0, W0), B0, C0, H0, W0)   # (mask.sum(1) < 0)  :1312
            height, width = img.shape[-2:]
            # 6. latents are drawn BEFORE the conditioning posterior is sampled (:1323 precedes :1338)
            if latents is None:
                vs = getattr(self, "vae_scale_factor", 8)
                latents = randn_tensor((nb, self.unet.config.in_channels, height // vs, width // vs), generator=generator,
                                       device=device, dtype=self._noise_dtype(prompt_embeds))
            cl = self.vae.encode(img.to(next(iter(self.vae.parameters())).dtype)).latent_dist.sample() * \
                self.vae.config.scaling_factor                                               # :1338-1341
            hl, wl = cl.shape[-2:]
            ml = hip_mask_prep(2, original_mask, None, (B0, 1, hl, wl), B0, 1, H0, W0, hl, wl)  # nearest  :1342-1344
            conditioning_latents = torch.cat([cl.float(), ml], 1)                            # :1345
        conditioning_latents = conditioning_latents.to(device)
        # (an un-duplicated tensor stays so: the loop copies it to both CFG halves itself, without comparing them)
        if do_cfg and guess_mode and conditioning_latents.shape[0] != nb:
            raise ValueError("guess_mode runs BrushNet on the conditional half only: conditioning_latents must have "
                             f"batch {nb}, got {conditioning_latents.shape[0]}")
        h, w = conditioning_latents.shape[-2:]
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps)
        self._num_timesteps = len(timesteps)
        shape = (nb, self.unet.config.in_channels, h, w)
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=self._noise_dtype(prompt_embeds))
        latents = latents.to(device=device, dtype=torch.float32) * self.scheduler.init_noise_sigma
        n = len(timesteps)
        keep = [1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end) for i in range(n)]
        scales = [brushnet_conditioning_scale * k for k in keep]                              # :1370-1376,1405-1409
        if self._loop is None or self._loop.scheduler is not self.scheduler or self._loop.unet is not self.unet or \
                self._loop.side is not self.brushnet:        # (a replaced component must not keep driving the old one)
            self._loop = DenoiseLoop(self.unet, self.scheduler, side=self.brushnet, side_kind="brushnet")
        self._loop.bind(shape, do_cfg, guidance_scale, prompt_embedsU, prompt_embeds_side=prompt_embeds,
                        side_static_inputs=[(conditioning_latents, self.unet.config.in_channels)],
                        side_scale=scales[0], guess_mode=guess_mode, eta=eta, generator=generator,
                        noise_dtype=self._noise_dtype(prompt_embeds))
        cb = None
        bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
        if bad:                                                                              # check_inputs, :775-780
            raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but "
                             f"found {bad}")
        if callback is not None or callback_on_step_end is not None:
            state = {"prompt_embeds": prompt_embeds, "negative_prompt_embeds": negative_prompt_embeds}

            def cb(i, t, lat):
                if callback_on_step_end is not None:
                    # :1451-1459 -- the callback sees the tensors it asked for and may hand back replacements:
                    #   latents         the loop owns `lat`: new values are copied into it (the next replay reads it);
                    #   prompt_embeds   in the reference's loop that name is BrushNet's context (`encoder_hidden_states=
                    #                   prompt_embeds`, :1411-1419): the side network's hoisted cross-attention K / V^T are
                    #                   recomputed in place (same arena addresses -> the captured step graph stays valid);
                    #   negative_prompt_embeds   rebinds a local the reference never reads again: recorded only.
                    kwargs = {k: (lat if k == "latents" else state[k]) for k in callback_on_step_end_tensor_inputs}
                    ret = callback_on_step_end(self, i, t, kwargs)
                    if isinstance(ret, dict):
                        new = ret.pop("latents", lat)
                        if new is not lat:
                            lat.copy_(new.to(lat.device, lat.dtype))
                        pe = ret.pop("prompt_embeds", state["prompt_embeds"])
                        if pe is not state["prompt_embeds"]:
                            if tuple(pe.shape) != tuple(state["prompt_embeds"].shape):
                                raise ValueError(f"callback_on_step_end returned prompt_embeds of shape {tuple(pe.shape)}, "
                                                 f"the loop was built for {tuple(state['prompt_embeds'].shape)}")
                            state["prompt_embeds"] = pe
                            side_rt = self._loop.side_rt
                            ctx = pe.chunk(2)[1] if (guess_mode and do_cfg and pe.shape[0] == 2 * nb) else pe
                            side_rt.set_context(ctx.to(device), force=True)
                        state["negative_prompt_embeds"] = ret.pop("negative_prompt_embeds", state["negative_prompt_embeds"])
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, lat)
        out = self._loop.run(latents, n, use_graph=self.use_graph, callback=cb, timesteps=timesteps,
                             scale_schedule=scales)
        return self._finish(out.clone(), output_type, return_dict, prompt_embeds.dtype, generator)
