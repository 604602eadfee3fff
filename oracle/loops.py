"""Oracle (TEST INFRASTRUCTURE): the three denoising loop bodies and the bit-exact input prep.

Restates, with the same tensor order conventions:
  v1 loop         /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:986-1041
  v2 loop         /root/reference/powerpaint/pipelines/pipeline_PowerPaint_Brushnet_CA.py:1378-1466
  ControlNet loop /root/reference/powerpaint/pipelines/pipeline_PowerPaint_ControlNet.py:1660-1741
  mask prep       pipeline_PowerPaint.py:39-153 (tensor branch), 671-710
  BrushNet mask   pipeline_PowerPaint_Brushnet_CA.py:1312,1342-1345
  prompt blend    pipeline_PowerPaint.py:423,499,516
"""
from typing import Callable, List, Optional

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------- bit-exact prep (row a20)
def binarize_mask(mask: torch.Tensor) -> torch.Tensor:
    """pipeline_PowerPaint.py:143-144."""
    mask = mask.clone()
    mask[mask < 0.5] = 0
    mask[mask >= 0.5] = 1
    return mask


def masked_image(image: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """pipeline_PowerPaint.py:147 (mask already binarised)."""
    return image * (mask < 0.5)


def mask_to_latent(mask: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """pipeline_PowerPaint.py:677-679 -- F.interpolate default mode = nearest."""
    return F.interpolate(mask, size=(h, w))


def brushnet_original_mask(mask_rgb: torch.Tensor) -> torch.Tensor:
    """pipeline_PowerPaint_Brushnet_CA.py:1312 on a [-1,1] preprocessed RGB mask."""
    return (mask_rgb.sum(1)[:, None, :, :] < 0).to(mask_rgb.dtype)


def blend_prompt_embeds(embA, embB, t: float):
    """pipeline_PowerPaint.py:423."""
    return embA * t + (1 - t) * embB


# ---------------------------------------------------------------- loops
@torch.no_grad()
def loop_v1(unet, scheduler, latents, mask, masked_image_latents, prompt_embeds, num_inference_steps: int,
            guidance_scale: float = 7.5, controlnet=None, control_image=None, controlnet_conditioning_scale=0.5,
            eps_hook: Optional[Callable] = None, teacher_latents: Optional[List[torch.Tensor]] = None,
            t_start: int = 0, guess_mode: bool = False, eta: float = 0.0, generator=None, image_latents=None,
            noise=None):
    """ppt-v1 loop (optionally + ControlNet).  `mask`, `masked_image_latents`, `prompt_embeds`, `control_image`
    are already CFG-duplicated ([uncond, cond] order, pipeline_PowerPaint.py:516,703-706).
    `eps_hook(i, t, latents_in, noise_pred_2B)` lets tests record per-step tensors; `teacher_latents[i]`
    (if given) replaces the loop-carried latents at step i (teacher forcing).
    t_start > 0 (`strength < 1`, get_timesteps :713-720): the loop runs over `scheduler.timesteps[t_start:]` and
    `latents` is the already-noised init image (no init_noise_sigma scaling, :640-642).
    guess_mode (pipeline_PowerPaint_ControlNet.py:1669-1702): the ControlNet sees the conditional half only
    (`control_image` un-duplicated), the unconditional half of the UNet batch gets zero residuals.
    eta / generator: `prepare_extra_step_kwargs` (:536-551) -- handed to `scheduler.step` iff its signature names them.
    image_latents / noise: the `num_channels_unet == 4` branch (:1025-1036; ControlNet pipeline :1725-1736) -- after every
    step the unmasked region becomes the FIRST image's latents noised to the next timestep (clean on the last step),
    `mask[:1]` being the first row of the (CFG-duplicated) latent-resolution mask."""
    import inspect
    step_params = set(inspect.signature(scheduler.step).parameters)
    extra = {k: v for k, v in (("eta", eta), ("generator", generator)) if k in step_params}
    scheduler.set_timesteps(num_inference_steps)
    do_cfg = guidance_scale > 1.0
    if t_start == 0:
        latents = latents * scheduler.init_noise_sigma
    timesteps = scheduler.timesteps[t_start * scheduler.order:]
    for i, t in enumerate(timesteps):
        if teacher_latents is not None:
            latents = teacher_latents[i]
        x = torch.cat([latents] * 2) if do_cfg else latents
        x = scheduler.scale_model_input(x, t)
        kw = {}
        if controlnet is not None:
            half = guess_mode and do_cfg
            cx = scheduler.scale_model_input(latents, t) if half else x
            ce = prompt_embeds.chunk(2)[1] if half else prompt_embeds
            down, mid = controlnet(cx, t, encoder_hidden_states=ce, controlnet_cond=control_image,
                                   conditioning_scale=controlnet_conditioning_scale, guess_mode=guess_mode)
            if half:
                down = [torch.cat([torch.zeros_like(d), d]) for d in down]
                mid = torch.cat([torch.zeros_like(mid), mid])
            kw = dict(down_block_additional_residuals=down, mid_block_additional_residual=mid)
        if unet.config.in_channels == 9:
            x = torch.cat([x, mask, masked_image_latents], dim=1)
        noise_pred = unet(x, t, encoder_hidden_states=prompt_embeds, **kw)[0]
        if eps_hook is not None:
            eps_hook(i, t, latents, noise_pred)
        if do_cfg:
            u, c = noise_pred.chunk(2)
            noise_pred = u + guidance_scale * (c - u)
        latents = scheduler.step(noise_pred, t, latents, **extra)[0]
        if unet.config.in_channels == 4 and image_latents is not None:
            proper = image_latents[:1]
            if i < len(timesteps) - 1:
                proper = scheduler.add_noise(proper, noise, torch.tensor([timesteps[i + 1]]))
            latents = (1 - mask[:1]) * proper + mask[:1] * latents
    return latents


@torch.no_grad()
def loop_v2(unet, brushnet, scheduler, latents, conditioning_latents, prompt_embeds, prompt_embedsU,
            num_inference_steps: int, guidance_scale: float = 7.5, brushnet_conditioning_scale: float = 1.0,
            control_guidance_start: float = 0.0, control_guidance_end: float = 1.0,
            eps_hook: Optional[Callable] = None, teacher_latents: Optional[List[torch.Tensor]] = None,
            guess_mode: bool = False):
    """ppt-v2 (BrushNet) loop.  conditioning_latents is [2B,5,h,w] (already CFG-duplicated); in guess_mode
    (:1394-1425) [B,5,h,w]: BrushNet then sees the conditional half only and the unconditional half of the UNet batch
    gets zero residuals."""
    scheduler.set_timesteps(num_inference_steps)
    do_cfg = guidance_scale > 1.0
    latents = latents * scheduler.init_noise_sigma
    n = len(scheduler.timesteps)
    keep = [1.0 - float(i / n < control_guidance_start or (i + 1) / n > control_guidance_end) for i in range(n)]
    for i, t in enumerate(scheduler.timesteps):
        if teacher_latents is not None:
            latents = teacher_latents[i]
        x = torch.cat([latents] * 2) if do_cfg else latents
        x = scheduler.scale_model_input(x, t)
        half = guess_mode and do_cfg
        bx = scheduler.scale_model_input(latents, t) if half else x
        be = prompt_embeds.chunk(2)[1] if half else prompt_embeds
        down, mid, up = brushnet(bx, t, encoder_hidden_states=be, brushnet_cond=conditioning_latents,
                                 conditioning_scale=brushnet_conditioning_scale * keep[i], guess_mode=guess_mode)
        if half:
            down = [torch.cat([torch.zeros_like(d), d]) for d in down]
            mid = torch.cat([torch.zeros_like(mid), mid])
            up = [torch.cat([torch.zeros_like(d), d]) for d in up]
        noise_pred = unet(x, t, encoder_hidden_states=prompt_embedsU, down_block_add_samples=list(down),
                          mid_block_add_sample=mid, up_block_add_samples=list(up))[0]
        if eps_hook is not None:
            eps_hook(i, t, latents, noise_pred)
        if do_cfg:
            u, c = noise_pred.chunk(2)
            noise_pred = u + guidance_scale * (c - u)
        latents = scheduler.step(noise_pred, t, latents)[0]
    return latents
