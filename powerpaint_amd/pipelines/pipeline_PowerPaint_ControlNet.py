"""ppt-v1 + ControlNet pipeline on the MI355X HIP path: drop-in for
/root/reference/powerpaint/pipelines/pipeline_PowerPaint_ControlNet.py:1349-1771
(`StableDiffusionControlNetInpaintPipeline.__call__`).  Additive keyword extensions as in the v1 pipeline
(`masked_image_latents=`, `mask_latents=`); `control_image` may be a ready [B,3,H,W] tensor in [0,1].
"""
from typing import Any, Callable, Dict, List, Optional, Union

import torch

from ._base import PipelineBase, prepare_mask_and_masked_image
from ._loop import DenoiseLoop
from .image_processor import VaeImageProcessor
from .pipeline_PowerPaint import StableDiffusionInpaintPipeline


class StableDiffusionControlNetInpaintPipeline(StableDiffusionInpaintPipeline):
    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, controlnet=None, scheduler=None,
                 safety_checker=None, feature_extractor=None, requires_safety_checker: bool = False):
        super().__init__(vae, text_encoder, tokenizer, unet, scheduler, safety_checker, feature_extractor)
        self.controlnet = controlnet
        self.control_image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_convert_rgb=True,
                                                         do_normalize=False)

    def prepare_control_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype,
                              do_classifier_free_guidance=False, guess_mode=False):
        """pipeline_PowerPaint_ControlNet.py:830-858 for tensor inputs (do_normalize=False -> values stay in [0,1])."""
        if not isinstance(image, torch.Tensor):
            ip = getattr(self, "control_image_processor", None)
            if ip is None:
                raise ValueError("control_image must be a [B,3,H,W] tensor in [0,1] (no image processor registered)")
            image = ip.preprocess(image, height=height, width=width)
        image = image.to(dtype=torch.float32)
        rep = batch_size if image.shape[0] == 1 else num_images_per_prompt
        image = image.repeat_interleave(rep, dim=0).to(device=device)
        if do_classifier_free_guidance and not guess_mode:
            image = torch.cat([image] * 2)
        return image

    @torch.no_grad()
    def __call__(self, promptA: Union[str, List[str]] = None, promptB: Union[str, List[str]] = None, image=None,
                 mask=None, control_image=None, height: Optional[int] = None, width: Optional[int] = None,
                 strength: float = 1.0, tradoff: float = 1.0, tradoff_nag: float = 1.0,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_promptA=None,
                 negative_promptB=None, num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.FloatTensor] = None, prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
                 callback_steps: int = 1, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 0.5, guess_mode: bool = False,
                 control_guidance_start: Union[float, List[float]] = 0.0,
                 control_guidance_end: Union[float, List[float]] = 1.0,
                 masked_image_latents: Optional[torch.FloatTensor] = None,
                 mask_latents: Optional[torch.FloatTensor] = None):
        if isinstance(control_guidance_start, list):
            control_guidance_start = control_guidance_start[0]
        if isinstance(control_guidance_end, list):
            control_guidance_end = control_guidance_end[0]
        if isinstance(controlnet_conditioning_scale, list):
            controlnet_conditioning_scale = controlnet_conditioning_scale[0]
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        prompt = promptA
        self.check_inputs(prompt, height, width, strength, callback_steps, negative_promptA, prompt_embeds,
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
        nb = batch_size * num_images_per_prompt
        control_image = self.prepare_control_image(control_image, width, height, nb, num_images_per_prompt, device,
                                                   torch.float32, do_cfg, guess_mode)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, device)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        # latents before the masked-image posterior sample (pipeline_PowerPaint_ControlNet.py:1614 precedes :1636)
        # strength < 1 (:1601-1625): enter the schedule late, start from the noised init image
        shape = (nb, 4, h, w)
        pixels = not (mask_latents is not None and masked_image_latents is not None)
        mk = masked_image = init_image = None
        if pixels:
            mk, masked_image, init_image = prepare_mask_and_masked_image(image, mask, height, width, device,
                                                                          return_image=True)
        num_channels_unet = self.unet.config.in_channels
        if num_channels_unet not in (4, 9):                                     # pipeline_PowerPaint_ControlNet.py:1251-1254
            raise ValueError(f"The unet {type(self.unet).__name__} should have either 4 or 9 input channels, not "
                             f"{num_channels_unet}.")
        four = num_channels_unet == 4      # known region re-imposed after every step (:1613, 1725-1736)
        if four and not pixels:
            raise ValueError("a 4-channel UNet needs the init image (its latents are blended back every step): pass "
                             "`image` and `mask`, not latent-space inputs")
        latents, noise, image_latents = self._initial_latents(shape, strength, timesteps, latents, init_image, generator,
                                                              device, self._noise_dtype(prompt_embeds),
                                                              return_image_latents=four, return_all=True)
        if not pixels:
            m = mask_latents.to(device=device, dtype=torch.float32)
            mil = masked_image_latents.to(device)
            # (left un-duplicated: the loop copies an un-duplicated tensor to both CFG halves itself, without comparing them)
        else:
            m, mil = self.prepare_mask_latents(mk, masked_image, nb, height, width, prompt_embeds.dtype, device,
                                               generator, do_cfg, masked_image_latents)
        n = len(timesteps)
        keep = [1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end) for i in range(n)]
        scales = [controlnet_conditioning_scale * k for k in keep]
        if self._loop is None or self._loop.scheduler is not self.scheduler or self._loop.unet is not self.unet or \
                self._loop.side is not self.controlnet:
            self._loop = DenoiseLoop(self.unet, self.scheduler, side=self.controlnet, side_kind="controlnet")
        self._loop.bind(shape, do_cfg, guidance_scale, prompt_embeds, prompt_embeds_side=prompt_embeds,
                        static_inputs=[] if four else [(m, 4), (mil, 5)], controlnet_cond=control_image,
                        side_scale=scales[0], guess_mode=guess_mode, eta=eta, generator=generator,
                        noise_dtype=self._noise_dtype(prompt_embeds), blend=(image_latents, m, noise) if four else None)
        cb = None
        if callback is not None:
            def cb(i, t, lat):
                if i % callback_steps == 0:
                    callback(i, t, lat)
        out = self._loop.run(latents, n, use_graph=self.use_graph, callback=cb, timesteps=timesteps,
                             scale_schedule=scales)
        return self._finish(out.clone(), output_type, return_dict, prompt_embeds.dtype)
