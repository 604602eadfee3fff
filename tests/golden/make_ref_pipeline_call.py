#!/usr/bin/env python
"""Generate tests/golden/ref_pipeline_call.pt: final latents of the REFERENCE'S OWN
`StableDiffusionInpaintPipeline.__call__` (pipeline_PowerPaint.py:723-1071, run through oracle/ref_pipeline.py) for a
prompt-in / pixels-in call: promptA / promptB blend, image + mask tensors, VAE encode of the masked image, 9-channel
UNet, CFG 7.5, DDIM.  Components are seeded stand-ins shared with the tests (`components()`): the oracle's reduced SD-1.5
UNet and VAE (bf16-rounded matrix weights) and a one-layer transformers CLIP text encoder over the small vocabulary."""
import json
import os
import sys

import torch
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import schedulers as OS, sd_modules as OM, vae as OV  # noqa: E402

TINY = dict(block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))
VAE_CFG = dict(block_out_channels=(64, 128, 256, 256), layers_per_block=1)
CALL = dict(promptA="the cat and the dog", promptB="the empty scene", tradoff=0.4, tradoff_nag=0.6,
            negative_promptA="blur", negative_promptB="the scene", height=128, width=128, num_inference_steps=3,
            guidance_scale=7.5)


def bf16_(m):
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() >= 2:
                p.copy_(p.to(torch.bfloat16).float())
    return m


def components():
    with open(os.path.join(HERE, "ref_task_tokens.json")) as f:
        G = json.load(f)
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(G["vocab"])},
                                     merges=[tuple(m) for m in G["merges"]], model_max_length=77)
    n = len(tok)
    torch.manual_seed(31)
    enc = transformers.CLIPTextModel(transformers.CLIPTextConfig(
        vocab_size=n, hidden_size=768, intermediate_size=3072, num_hidden_layers=1, num_attention_heads=12,
        max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=n - 2, eos_token_id=n - 1,
        pad_token_id=n - 1)).eval()
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
    bf16_(enc)
    torch.manual_seed(32)
    unet = bf16_(OM.UNet2DConditionModel(in_channels=9, **TINY)).eval()
    torch.manual_seed(33)
    vae = bf16_(OV.AutoencoderKL(**VAE_CFG)).eval()
    return tok, enc, unet, vae


def inputs():
    g = torch.Generator().manual_seed(41)
    img = torch.rand(1, 3, 128, 128, generator=g) * 2 - 1
    mask = torch.zeros(1, 1, 128, 128)
    mask[:, :, 30:100, 20:90] = 1.0
    lat = torch.randn(1, 4, 16, 16, generator=g)
    return img, mask, lat


def main():
    from oracle import ref_pipeline
    Pipe, _ = ref_pipeline.load_reference_pipeline_class()
    tok, enc, unet, vae = components()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, scheduler=OS.DDIMScheduler(), safety_checker=None,
                feature_extractor=None, requires_safety_checker=False)
    img, mask, lat = inputs()
    seen = []
    with torch.no_grad():
        out = pipe(image=img, mask=mask, latents=lat.clone(), generator=torch.Generator().manual_seed(5),
                   output_type="latent", return_dict=False, callback=lambda i, t, l: seen.append((i, int(t), l.clone())),
                   **CALL)[0]
    torch.save(dict(latents=out, steps=seen), os.path.join(HERE, "ref_pipeline_call.pt"))
    print(out.shape, float(out.abs().max()), [s[:2] for s in seen])


CALL_V2 = dict(promptA="the cat", promptB="the scene", promptU="the dog", tradoff=0.4, tradoff_nag=0.6,
               negative_promptA="blur", negative_promptB="the cat", negative_promptU="blur", num_inference_steps=3,
               guidance_scale=7.5, brushnet_conditioning_scale=1.0, width=128, height=128)


def components_v2():
    tok, enc, _, vae = components()
    torch.manual_seed(34)
    unet = bf16_(OM.UNet2DConditionModel(in_channels=4, **TINY)).eval()
    torch.manual_seed(35)
    bn = OM.randomize_zero_convs(bf16_(OM.BrushNetModel(in_channels=4, conditioning_channels=5, **TINY))).eval()
    return tok, enc, unet, bn, vae


def inputs_v2():
    img, mask, lat = inputs()
    return img * (mask < 0.5), (mask * 2 - 1).repeat(1, 3, 1, 1), lat      # pre-masked image, RGB mask in [-1, 1]


def main_v2():
    """ref_pipeline_call_v2.pt: the same for `StableDiffusionPowerPaintBrushNetPipeline.__call__`
    (pipeline_PowerPaint_Brushnet_CA.py:1026-1497): BrushNet + 4-channel UNet, DPM-Solver++, two prompt encoders."""
    from oracle import ref_pipeline
    Pipe = ref_pipeline.load_reference_brushnet_pipeline_class(OM.BrushNetModel)
    tok, enc, unet, bn, vae = components_v2()
    pipe = Pipe(vae=vae, text_encoder=enc, text_encoder_brushnet=enc, tokenizer=tok, unet=unet, brushnet=bn,
                scheduler=OS.DPMSolverMultistepScheduler(), safety_checker=None, feature_extractor=None,
                requires_safety_checker=False)
    img, mask3, lat = inputs_v2()
    torch.manual_seed(9)                                   # the conditioning latents are sampled from the global RNG
    with torch.no_grad():
        out = pipe(image=img, mask=mask3, latents=lat.clone(), output_type="latent", return_dict=False, **CALL_V2)[0]
    torch.save(dict(latents=out), os.path.join(HERE, "ref_pipeline_call_v2.pt"))
    print("v2", out.shape, float(out.abs().max()))


CALL_CN = dict(promptA="the cat", promptB="the cat", tradoff=1.0, tradoff_nag=1.0, negative_promptA="blur",
               negative_promptB="blur", width=128, height=128, guidance_scale=7.5, controlnet_conditioning_scale=0.5,
               num_inference_steps=3)


def components_cn():
    tok, enc, unet, vae = components()
    torch.manual_seed(36)
    cn = OM.randomize_zero_convs(bf16_(OM.ControlNetModel(
        in_channels=4, **{k: v for k, v in TINY.items() if k != "up_block_types"}))).eval()
    return tok, enc, unet, cn, vae


def control_image():
    return torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(42))


def main_cn():
    """ref_pipeline_call_cn.pt: `StableDiffusionControlNetInpaintPipeline.__call__`
    (pipeline_PowerPaint_ControlNet.py:1349-1760): ControlNet residuals into the 9-channel UNet, DDIM."""
    from oracle import ref_pipeline
    Pipe = ref_pipeline.load_reference_controlnet_pipeline_class(OM.ControlNetModel)
    tok, enc, unet, cn, vae = components_cn()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, controlnet=cn, scheduler=OS.DDIMScheduler(),
                safety_checker=None, feature_extractor=None, requires_safety_checker=False)
    img, mask, lat = inputs()
    with torch.no_grad():
        out = pipe(image=img, mask=mask, control_image=control_image(), latents=lat.clone(),
                   generator=torch.Generator().manual_seed(5), output_type="latent", return_dict=False, **CALL_CN)[0]
    torch.save(dict(latents=out), os.path.join(HERE, "ref_pipeline_call_cn.pt"))
    print("controlnet", out.shape, float(out.abs().max()))


CALL_STRENGTH = dict(CALL, num_inference_steps=5, strength=0.6)
DPM_SD15 = dict(timestep_spacing="leading", steps_offset=1)     # DPMSolverMultistepScheduler.from_config(<SD-1.5 config>)


def main_strength():
    """ref_pipeline_call_strength.pt: the v1 `__call__` with `strength = 0.6` and no `latents`
    (pipeline_PowerPaint.py:604-655,713-720,906-944): the schedule is entered at entry 2 of 5, the initial latents are
    the VAE posterior sample of the init image noised to that timestep (posterior sample, then noise, then the masked
    image's posterior sample, all from one generator).  DPM-Solver++ so that the multistep warm-up restarts mid-schedule."""
    from oracle import ref_pipeline
    Pipe, _ = ref_pipeline.load_reference_pipeline_class()
    tok, enc, unet, vae = components()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, scheduler=OS.DPMSolverMultistepScheduler(**DPM_SD15),
                safety_checker=None, feature_extractor=None, requires_safety_checker=False)
    img, mask, _ = inputs()
    seen = []
    with torch.no_grad():
        out = pipe(image=img, mask=mask, generator=torch.Generator().manual_seed(5), output_type="latent",
                   return_dict=False, callback=lambda i, t, l: seen.append((i, int(t), l.clone())), **CALL_STRENGTH)[0]
    torch.save(dict(latents=out, steps=seen), os.path.join(HERE, "ref_pipeline_call_strength.pt"))
    print("strength", out.shape, float(out.abs().max()), [s[:2] for s in seen])


def main_guess():
    """ref_pipeline_call_v2_guess.pt / ref_pipeline_call_cn_guess.pt: the BrushNet and ControlNet `__call__`s with
    `guess_mode=True` under CFG (pipeline_PowerPaint_Brushnet_CA.py:1394-1425, pipeline_PowerPaint_ControlNet.py:
    1669-1702): side network on the conditional half only (image / control image not duplicated, :949), residual scales
    logspace(-1, 0, n), zeros for the unconditional half of the UNet batch."""
    from oracle import ref_pipeline
    Pipe = ref_pipeline.load_reference_brushnet_pipeline_class(OM.BrushNetModel)
    tok, enc, unet, bn, vae = components_v2()
    pipe = Pipe(vae=vae, text_encoder=enc, text_encoder_brushnet=enc, tokenizer=tok, unet=unet, brushnet=bn,
                scheduler=OS.DPMSolverMultistepScheduler(), safety_checker=None, feature_extractor=None,
                requires_safety_checker=False)
    img, mask3, lat = inputs_v2()
    torch.manual_seed(9)
    with torch.no_grad():
        out = pipe(image=img, mask=mask3, latents=lat.clone(), output_type="latent", return_dict=False, guess_mode=True,
                   **CALL_V2)[0]
    torch.save(dict(latents=out), os.path.join(HERE, "ref_pipeline_call_v2_guess.pt"))
    print("v2 guess", out.shape, float(out.abs().max()))
    Pipe = ref_pipeline.load_reference_controlnet_pipeline_class(OM.ControlNetModel)
    tok, enc, unet, cn, vae = components_cn()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, controlnet=cn, scheduler=OS.DDIMScheduler(),
                safety_checker=None, feature_extractor=None, requires_safety_checker=False)
    img, mask, lat = inputs()
    with torch.no_grad():
        out = pipe(image=img, mask=mask, control_image=control_image(), latents=lat.clone(),
                   generator=torch.Generator().manual_seed(5), output_type="latent", return_dict=False, guess_mode=True,
                   **CALL_CN)[0]
    torch.save(dict(latents=out), os.path.join(HERE, "ref_pipeline_call_cn_guess.pt"))
    print("controlnet guess", out.shape, float(out.abs().max()))


CALL_ETA = dict(CALL, eta=0.7)


def main_eta():
    """ref_pipeline_call_eta.pt: the v1 `__call__` with `eta = 0.7` on DDIM (pipeline_PowerPaint.py:536-551,1023): every
    step adds std_dev_t * randn_tensor(noise_pred.shape, generator) -- the generator that has already produced the
    masked image's posterior sample."""
    from oracle import ref_pipeline
    Pipe, _ = ref_pipeline.load_reference_pipeline_class()
    tok, enc, unet, vae = components()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, scheduler=OS.DDIMScheduler(), safety_checker=None,
                feature_extractor=None, requires_safety_checker=False)
    img, mask, lat = inputs()
    with torch.no_grad():
        out = pipe(image=img, mask=mask, latents=lat.clone(), generator=torch.Generator().manual_seed(5),
                   output_type="latent", return_dict=False, **CALL_ETA)[0]
    torch.save(dict(latents=out), os.path.join(HERE, "ref_pipeline_call_eta.pt"))
    print("eta", out.shape, float(out.abs().max()))


CALL_4CH = dict(CALL, num_inference_steps=4)
CALL_4CH_CN = dict(CALL_CN, num_inference_steps=4, strength=0.8)


def components_4ch():
    tok, enc, _, vae = components()
    torch.manual_seed(37)
    unet = bf16_(OM.UNet2DConditionModel(in_channels=4, **TINY)).eval()
    return tok, enc, unet, vae


def main_4ch():
    """ref_pipeline_call_4ch.pt: the `num_channels_unet == 4` branch of the v1 and the ControlNet `__call__`s
    (pipeline_PowerPaint.py:927-928,965-979,1025-1036; pipeline_PowerPaint_ControlNet.py:1612-1613,1725-1736): a plain
    4-channel UNet sees only the latents; after every scheduler step the unmasked region is replaced by the init
    image's latents noised to the next timestep.  No `latents` argument: init-image posterior sample, noise, masked-image
    posterior sample come from one generator, in that order.  Case "v1": DDIM, strength 1; case "cn": ControlNet,
    DPM-Solver++, strength 0.8 (schedule entered at entry 1 of 4, initial latents = noised init image)."""
    from oracle import ref_pipeline
    Pipe, _ = ref_pipeline.load_reference_pipeline_class()
    tok, enc, unet, vae = components_4ch()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, scheduler=OS.DDIMScheduler(), safety_checker=None,
                feature_extractor=None, requires_safety_checker=False)
    img, mask, _ = inputs()
    seen = []
    with torch.no_grad():
        out = pipe(image=img, mask=mask, generator=torch.Generator().manual_seed(5), output_type="latent",
                   return_dict=False, callback=lambda i, t, l: seen.append((i, int(t), l.clone())), **CALL_4CH)[0]
    Pipe = ref_pipeline.load_reference_controlnet_pipeline_class(OM.ControlNetModel)
    _, _, _, cn, _ = components_cn()
    pipe = Pipe(vae=vae, text_encoder=enc, tokenizer=tok, unet=unet, controlnet=cn,
                scheduler=OS.DPMSolverMultistepScheduler(**DPM_SD15), safety_checker=None, feature_extractor=None,
                requires_safety_checker=False)
    seen_cn = []
    with torch.no_grad():
        out_cn = pipe(image=img, mask=mask, control_image=control_image(), generator=torch.Generator().manual_seed(6),
                      output_type="latent", return_dict=False,
                      callback=lambda i, t, l: seen_cn.append((i, int(t), l.clone())), **CALL_4CH_CN)[0]
    torch.save(dict(latents=out, steps=seen, latents_cn=out_cn, steps_cn=seen_cn),
               os.path.join(HERE, "ref_pipeline_call_4ch.pt"))
    print("4ch", out.shape, float(out.abs().max()), [s[:2] for s in seen], "cn", float(out_cn.abs().max()),
          [s[:2] for s in seen_cn])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("strength", "guess", "eta", "4ch"):
        {"strength": main_strength, "guess": main_guess, "eta": main_eta, "4ch": main_4ch}[sys.argv[1]]()
        sys.exit(0)
    main()
    main_v2()
    main_cn()
    main_strength()
    main_guess()
    main_eta()
    main_4ch()
