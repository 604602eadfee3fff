"""ppt-v1 pipeline on the MI355X HIP path: drop-in for
/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:723-1071 (`StableDiffusionInpaintPipeline.__call__`).

Same positional/keyword signature; two additive keyword extensions for synthetic / VAE-free operation:
`masked_image_latents=` and `mask_latents=` (latent-space inputs normally produced by `prepare_mask_latents`).
"""
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ._base import PipelineBase, hip_mask_prep, prepare_mask_and_masked_image
from ._loop import DenoiseLoop
from .image_processor import VaeImageProcessor


class StableDiffusionInpaintPipeline(PipelineBase):
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, safety_checker=None,
                 feature_extractor=None, requires_safety_checker: bool = False):
        self.register_modules(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler,
                              safety_checker=safety_checker, feature_extractor=feature_extractor)
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1) if vae is not None else 8
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor) if vae is not None else None
        self._loop = None
        self.use_graph = True

    def check_inputs(self, prompt, height, width, strength, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        """pipeline_PowerPaint.py:553-602."""
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None:
            if prompt_embeds.shape != negative_prompt_embeds.shape:
                raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape.")

    def prepare_mask_latents(self, mask, masked_image, batch_size, height, width, dtype, device, generator,
                             do_classifier_free_guidance, masked_image_latents=None):
        """pipeline_PowerPaint.py:671-710 (nearest mask downsample in the HIP kernel)."""
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        mb = mask.shape[0]
        mask = hip_mask_prep(2, mask.to(device), None, (mb, 1, h, w), mb, 1, mask.shape[-2], mask.shape[-1], h, w)
        if masked_image_latents is None:
            masked_image_latents = self._vae_encode(masked_image.to(device=device, dtype=dtype), generator)
        if mask.shape[0] < batch_size:
            if batch_size % mask.shape[0]:
                raise ValueError("The passed mask and the required batch size don't match.")
            mask = mask.repeat(batch_size // mask.shape[0], 1, 1, 1)
        if masked_image_latents.shape[0] < batch_size:
            if batch_size % masked_image_latents.shape[0]:
                raise ValueError("The passed images and the required batch size don't match.")
            masked_image_latents = masked_image_latents.repeat(batch_size // masked_image_latents.shape[0], 1, 1, 1)
        mask = torch.cat([mask] * 2) if do_classifier_free_guidance else mask
        masked_image_latents = torch.cat([masked_image_latents] * 2) if do_classifier_free_guidance else masked_image_latents
        return mask, masked_image_latents.to(device)

    @torch.no_grad()
    def __call__(self, promptA: Union[str, List[str]] = None, promptB: Union[str, List[str]] = None, image=None,
                 mask=None, height: Optional[int] = None, width: Optional[int] = None, strength: float = 1.0,
                 tradoff: float = 1.0, tradoff_nag: float = 1.0, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_promptA=None, negative_promptB=None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.FloatTensor] = None, prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: int = 1, cross_attention_kwargs: Optional[Dict[str, Any]] = None, task_class=None,
                 masked_image_latents: Optional[torch.FloatTensor] = None,
                 mask_latents: Optional[torch.FloatTensor] = None):
        if task_class is not None:
            raise NotImplementedError("task_class is not used by the released checkpoints")
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        prompt, negative_prompt = promptA, negative_promptA
        self.check_inputs(prompt, height, width, strength, callback_steps, negative_prompt, prompt_embeds,
                          negative_prompt_embeds)
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self._execution_device
        do_cfg = guidance_scale > 1.0
        prompt_embeds = self._encode_prompt(promptA, promptB, tradoff, device, num_images_per_prompt, do_cfg,
                                            negative_promptA, negative_promptB, tradoff_nag,
                                            prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, device)
        if num_inference_steps < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of "
                             f"pipelinesteps is {num_inference_steps} which is < 1 and not appropriate for this pipeline.")
        nb = batch_size * num_images_per_prompt
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        num_channels_unet = self.unet.config.in_channels
        if num_channels_unet not in (4, 9):
            raise ValueError(f"The unet {type(self.unet).__name__} should have either 4 or 9 input channels, not "
                             f"{num_channels_unet}.")                                     # pipeline_PowerPaint.py:976-979
        four = num_channels_unet == 4      # plain text-to-image UNet: known region re-imposed every step (:1025-1036)
        # 6. latents -- drawn BEFORE the masked-image posterior is sampled, in the prompt dtype, as the reference does
        #    (prepare_latents :930 precedes prepare_mask_latents :952): same seed / generator => same noise
        #    strength < 1 (:604-655, 713-720, 919): enter the schedule late, start from the noised init image
        shape = (nb, 4, h, w)
        pixels = not (mask_latents is not None and masked_image_latents is not None)
        mk = masked_image = init_image = None
        if pixels:
            mk, masked_image, init_image = prepare_mask_and_masked_image(image, mask, height, width, device,
                                                                          return_image=True)
        if four and not pixels:
            raise ValueError("a 4-channel UNet needs the init image (its latents are blended back every step): pass "
                             "`image` and `mask`, not latent-space inputs")
        latents, noise, image_latents = self._initial_latents(shape, strength, timesteps, latents, init_image, generator,
                                                              device, self._noise_dtype(prompt_embeds),
                                                              return_image_latents=four, return_all=True)
        # 5./7. mask + masked-image latents
        if not pixels:
            # (latent-space inputs stay un-duplicated: the loop copies a tensor with the un-duplicated batch to both CFG halves
            #  itself -- `torch.cat([m] * 2)` here would only make it compare the halves again, a device synchronisation)
            m = mask_latents.to(device=device, dtype=torch.float32)
            mil = masked_image_latents.to(device)
        else:
            m, mil = self.prepare_mask_latents(mk, masked_image, nb, height, width, prompt_embeds.dtype, device,
                                               generator, do_cfg, masked_image_latents)
        if not four and 4 + m.shape[1] + mil.shape[1] != num_channels_unet:
            raise ValueError("Incorrect configuration settings! mask / masked-image latents do not add up to "
                             f"unet.config.in_channels = {num_channels_unet}")
        # 10. fused denoising loop
        if self._loop is None or self._loop.scheduler is not self.scheduler or self._loop.unet is not self.unet:
            self._loop = DenoiseLoop(self.unet, self.scheduler)
        self._loop.bind(shape, do_cfg, guidance_scale, prompt_embeds, static_inputs=[] if four else [(m, 4), (mil, 5)],
                        eta=eta, generator=generator, noise_dtype=self._noise_dtype(prompt_embeds),
                        blend=(image_latents, m, noise) if four else None)
        cb = None
        if callback is not None:
            def cb(i, t, lat):
                if i % callback_steps == 0:
                    callback(i, t, lat)
        out = self._loop.run(latents, len(timesteps), use_graph=self.use_graph, callback=cb, timesteps=timesteps)
        return self._finish(out.clone(), output_type, return_dict, prompt_embeds.dtype)
