"""SURVEY.md §8f-4 -- checkpoint directories (`from_pretrained`, `load_model`) and the image processor at the two ends
of the pipelines.  Host logic only: everything here runs on CPU (parameter packing works on any device; launches do not).
"""
import json
import os

import numpy as np
import pytest
import torch

from powerpaint_amd import _lib as L
from powerpaint_amd import loaders
from powerpaint_amd import models as PM
from powerpaint_amd.pipelines.image_processor import VaeImageProcessor

safetensors = pytest.importorskip("safetensors.torch")
PIL = pytest.importorskip("PIL.Image")

TINY = dict(block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"))


def write_dir(d, config, sd, weights="diffusion_pytorch_model.safetensors", class_name="X"):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(dict(config, _class_name=class_name, _diffusers_version="0.27.0"), f)
    sd = {k: v.contiguous() for k, v in sd.items()}
    if weights.endswith(".safetensors"):
        safetensors.save_file(sd, os.path.join(d, weights))
    else:
        torch.save(sd, os.path.join(d, weights))


def test_unet_and_controlnet_from_pretrained(tmp_path):
    ref = PM.UNet2DConditionModel(in_channels=9, device="cpu", dtype=torch.float16, **TINY)
    sd = {k: v.half() for k, v in ref.net.synthetic_state_dict(seed=3).items()}
    root = str(tmp_path / "sd-inpainting")
    write_dir(os.path.join(root, "unet"), dict(TINY, in_channels=9, out_channels=4, sample_size=64, act_fn="silu"), sd)
    m = PM.UNet2DConditionModel.from_pretrained(root, subfolder="unet", torch_dtype=torch.float16, device="cpu",
                                                local_files_only=True)
    assert m.config.in_channels == 9 and m.config.block_out_channels == (320, 640)
    assert m.dtype == torch.float16                  # torch_dtype=float16 (the reference's default) computes in fp16 ...
    ref.load_state_dict(sd)
    assert torch.equal(m.param_buffer(), ref.param_buffer())                 # same packed bytes
    w = m.net.params.tensor("mid_block.resnets.0.conv1.weight")              # ... and an fp16 checkpoint is not re-rounded
    assert w.dtype == torch.float16 and torch.equal(
        w, sd["mid_block.resnets.0.conv1.weight"].permute(0, 2, 3, 1).reshape(w.shape))
    mb = PM.UNet2DConditionModel.from_pretrained(root, subfolder="unet", device="cpu")      # default: bf16
    assert mb.dtype == torch.bfloat16 and mb.param_buffer().numel() == m.param_buffer().numel()
    assert m._sd is not None                                                 # kept for BrushNetModel.from_unet
    # load_model: the safetensors.torch.load_model stand-in (strict by default, (missing, unexpected) returned)
    f = os.path.join(root, "unet", "diffusion_pytorch_model.safetensors")
    fresh = PM.UNet2DConditionModel(in_channels=9, device="cpu", dtype=torch.float16, **TINY)
    assert loaders.load_model(fresh, f) == ([], [])
    assert torch.equal(fresh.param_buffer(), ref.param_buffer())
    extra = dict(sd, **{"not.a.key": torch.zeros(1)})
    safetensors.save_file(extra, str(tmp_path / "extra.safetensors"))
    with pytest.raises(RuntimeError):
        loaders.load_model(fresh, str(tmp_path / "extra.safetensors"))
    assert loaders.load_model(fresh, str(tmp_path / "extra.safetensors"), strict=False) == ([], ["not.a.key"])
    short = {k: v for k, v in sd.items() if k != "conv_out.bias"}
    torch.save(short, str(tmp_path / "short.bin"))
    with pytest.raises(RuntimeError):
        loaders.load_model(fresh, str(tmp_path / "short.bin"), strict=False)
    # errors: no such folder, not in the (offline) hub cache, no weight file
    with pytest.raises(L.PPError):
        PM.UNet2DConditionModel.from_pretrained(root, subfolder="vae", device="cpu")
    with pytest.raises(L.PPError):
        PM.UNet2DConditionModel.from_pretrained("nobody/not-a-cached-repo", subfolder="unet", device="cpu")
    os.remove(f)
    with pytest.raises(L.PPError):
        PM.UNet2DConditionModel.from_pretrained(root, subfolder="unet", device="cpu")


def test_vae_and_text_encoder_from_pretrained(tmp_path):
    transformers = pytest.importorskip("transformers")
    cfg = dict(block_out_channels=(64, 64, 64, 64), layers_per_block=1, in_channels=3, out_channels=3,
               latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, act_fn="silu",
               down_block_types=["DownEncoderBlock2D"] * 4, up_block_types=["UpDecoderBlock2D"] * 4, sample_size=512)
    ref = PM.AutoencoderKL(device="cpu", **cfg)
    sd = ref.net.synthetic_state_dict(seed=5)
    write_dir(str(tmp_path / "vae"), cfg, sd, weights="diffusion_pytorch_model.bin")
    vae = PM.AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae", device="cpu")
    ref.load_state_dict(sd)
    assert torch.equal(vae.param_buffer(), ref.param_buffer()) and vae.config.scaling_factor == 0.18215
    # CLIP: a transformers checkpoint directory as save_pretrained writes it
    hf_cfg = transformers.CLIPTextConfig(vocab_size=300, hidden_size=768, intermediate_size=3072, num_hidden_layers=1,
                                         num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                                         bos_token_id=298, eos_token_id=299, pad_token_id=299)
    hf = transformers.CLIPTextModel(hf_cfg)
    hf.save_pretrained(str(tmp_path / "text_encoder"))
    enc = PM.CLIPTextModel.from_pretrained(str(tmp_path), subfolder="text_encoder", device="cpu")
    assert enc.config.vocab_size == 300 and enc.config.num_hidden_layers == 1 and enc.config.eos_token_id == 299
    assert torch.equal(enc.text_model.embeddings.token_embedding.weight, hf.get_input_embeddings().weight)
    got = dict(enc.named_parameters())["text_model.encoder.layers.0.mlp.fc1.weight"]
    want = {k.replace("text_model.", ""): v for k, v in hf.state_dict().items()}["encoder.layers.0.mlp.fc1.weight"]
    assert torch.equal(got, want)
    # load_model on the nn.Module goes through its own load_state_dict (key mapping included)
    safetensors.save_file({k: v.contiguous() for k, v in hf.state_dict().items()}, str(tmp_path / "te.safetensors"))
    enc2 = PM.CLIPTextModel(device="cpu", vocab_size=300, num_hidden_layers=1)
    assert loaders.load_model(enc2, str(tmp_path / "te.safetensors")) == ([], [])


def test_image_processor_roundtrip():
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, size=(70, 90, 3), dtype=np.uint8)
    img = PIL.fromarray(arr)
    ip = VaeImageProcessor(vae_scale_factor=8)
    t = ip.preprocess(img)                                     # default size: rounded down to multiples of 8
    assert t.shape == (1, 3, 64, 88) and t.dtype == torch.float32 and -1.0 <= float(t.min()) and float(t.max()) <= 1.0
    same = PIL.fromarray(arr[:64, :88])
    t2 = ip.preprocess(same, height=64, width=88)
    assert torch.equal(t2[0], torch.from_numpy(arr[:64, :88].astype(np.float32) / 255.0).permute(2, 0, 1) * 2 - 1)
    back = ip.postprocess(t2, output_type="pil")
    assert isinstance(back, list) and np.array_equal(np.array(back[0]), arr[:64, :88])
    assert np.allclose(ip.postprocess(t2, output_type="np")[0], arr[:64, :88] / 255.0, atol=1e-6)
    pt = ip.postprocess(t2 * 3.0, output_type="pt")
    assert float(pt.min()) >= 0.0 and float(pt.max()) <= 1.0   # denormalise clamps
    assert torch.equal(ip.postprocess(t2, output_type="pt", do_denormalize=[False]), t2)
    assert ip.postprocess(t2, output_type="latent") is t2
    lst = ip.preprocess([img, img], height=32, width=48)
    assert lst.shape == (2, 3, 32, 48)
    # control images: RGB conversion, values stay in [0, 1]
    cp = VaeImageProcessor(vae_scale_factor=8, do_convert_rgb=True, do_normalize=False)
    gray = PIL.fromarray(arr[..., 0], mode="L")
    c = cp.preprocess(gray, height=64, width=88)
    assert c.shape == (1, 3, 64, 88) and float(c.min()) >= 0.0 and float(c.max()) <= 1.0
    # tensors pass through (already [-1, 1]: not normalised twice); numpy in [0, 1] is normalised
    x = torch.rand(2, 3, 16, 16) * 2 - 1
    assert torch.equal(ip.preprocess(x), x)
    a = rng.random((1, 16, 16, 3)).astype(np.float32)
    assert torch.allclose(ip.preprocess(a), torch.from_numpy(a).permute(0, 3, 1, 2) * 2 - 1)
    with pytest.raises(ValueError):
        ip.preprocess([])
    with pytest.raises(ValueError):
        ip.postprocess(np.zeros((1, 3, 8, 8)))


def test_pipelines_register_processors():
    from powerpaint_amd import pipelines as PP
    vae = PM.AutoencoderKL(device="cpu", block_out_channels=(64, 64, 64, 64), layers_per_block=1)
    p1 = PP.StableDiffusionInpaintPipeline(vae=vae)
    assert isinstance(p1.image_processor, VaeImageProcessor) and p1.vae_scale_factor == 8
    p2 = PP.StableDiffusionControlNetInpaintPipeline(vae=vae)
    assert p2.control_image_processor.config.do_normalize is False and p2.control_image_processor.config.do_convert_rgb
    assert PP.StableDiffusionPowerPaintBrushNetPipeline(vae=None).image_processor is None
    # the DiffusionPipeline memory switches app.py / users call are accepted (nothing to offload on 288 GB)
    for name in ("enable_model_cpu_offload", "enable_vae_slicing", "enable_vae_tiling", "enable_attention_slicing",
                 "enable_xformers_memory_efficient_attention", "maybe_free_model_hooks"):
        assert getattr(p1, name)() is None
    assert p1.to("cuda") is p1


def test_pipeline_from_pretrained(tmp_path):
    """`Pipeline.from_pretrained(folder)` over a diffusers pipeline folder (model_index.json + component sub-folders),
    the first call of app.py:90; component overrides as app.py:168-176 passes them."""
    transformers = pytest.importorskip("transformers")
    from powerpaint_amd import pipelines as PP, schedulers as PS
    root = str(tmp_path / "sd-inpainting")
    unet = PM.UNet2DConditionModel(in_channels=9, device="cpu", **TINY)
    write_dir(os.path.join(root, "unet"), dict(TINY, in_channels=9, out_channels=4, sample_size=64),
              {k: v.half() for k, v in unet.net.synthetic_state_dict(seed=3).items()})
    vcfg = dict(block_out_channels=(64, 64, 64, 64), layers_per_block=1, in_channels=3, out_channels=3, latent_channels=4,
                norm_num_groups=32, scaling_factor=0.18215)
    vae = PM.AutoencoderKL(device="cpu", **vcfg)
    write_dir(os.path.join(root, "vae"), vcfg, vae.net.synthetic_state_dict(seed=5))
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_task_tokens.json")) as f:
        G = json.load(f)
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(G["vocab"])},
                                     merges=[tuple(m) for m in G["merges"]], model_max_length=77)
    tok.save_pretrained(os.path.join(root, "tokenizer"))
    n = len(tok)
    hf = transformers.CLIPTextModel(transformers.CLIPTextConfig(
        vocab_size=n, hidden_size=768, intermediate_size=3072, num_hidden_layers=1, num_attention_heads=12,
        max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=n - 2, eos_token_id=n - 1, pad_token_id=n - 1))
    hf.save_pretrained(os.path.join(root, "text_encoder"))
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(_class_name="PNDMScheduler", _diffusers_version="0.6.0", beta_end=0.012, beta_schedule="scaled_linear",
                       beta_start=0.00085, num_train_timesteps=1000, set_alpha_to_one=False, skip_prk_steps=True,
                       steps_offset=1, trained_betas=None, clip_sample=False), f)
    with open(os.path.join(root, "model_index.json"), "w") as f:
        json.dump({"_class_name": "StableDiffusionInpaintPipeline", "_diffusers_version": "0.6.0",
                   "feature_extractor": ["transformers", "CLIPImageProcessor"],
                   "safety_checker": ["stable_diffusion", "StableDiffusionSafetyChecker"],
                   "scheduler": ["diffusers", "PNDMScheduler"], "text_encoder": ["transformers", "CLIPTextModel"],
                   "tokenizer": ["transformers", "CLIPTokenizer"], "unet": ["diffusers", "UNet2DConditionModel"],
                   "vae": ["diffusers", "AutoencoderKL"]}, f)
    pipe = PP.StableDiffusionInpaintPipeline.from_pretrained(root, torch_dtype=torch.float16, device="cpu",
                                                             local_files_only=True)
    assert isinstance(pipe.unet, PM.UNet2DConditionModel) and pipe.unet.config.in_channels == 9
    assert isinstance(pipe.vae, PM.AutoencoderKL) and pipe.vae_scale_factor == 8 and pipe.image_processor is not None
    assert isinstance(pipe.text_encoder, PM.CLIPTextModel) and pipe.text_encoder.config.vocab_size == n
    assert isinstance(pipe.scheduler, PS.PNDMScheduler) and pipe.scheduler.config.steps_offset == 1
    assert pipe.tokenizer("the cat").input_ids == tok("the cat").input_ids
    assert pipe.safety_checker is None and pipe.feature_extractor is None
    # the scheduler swap of app.py:197, and a component handed in instead of loaded
    pipe.scheduler = PS.UniPCMultistepScheduler.from_config(pipe.scheduler.config)
    assert pipe.scheduler.config.timestep_spacing == "leading" and pipe.scheduler.config.steps_offset == 1
    assert isinstance(PS.DDIMScheduler.from_config(pipe.scheduler.config), PS.DDIMScheduler)
    mine = PS.DPMSolverMultistepScheduler()
    p2 = PP.StableDiffusionPowerPaintBrushNetPipeline.from_pretrained(root, device="cpu", scheduler=mine, brushnet="B",
                                                                      text_encoder_brushnet=pipe.text_encoder,
                                                                      unet=pipe.unet, vae=pipe.vae)
    assert p2.scheduler is mine and p2.brushnet == "B" and p2.unet is pipe.unet and p2.text_encoder_brushnet is pipe.text_encoder
    with pytest.raises(TypeError):
        PP.StableDiffusionInpaintPipeline.from_pretrained(root, device="cpu", brushnet="B")
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(_class_name="EulerDiscreteScheduler"), f)
    with pytest.raises(L.PPError):
        PP.StableDiffusionInpaintPipeline.from_pretrained(root, device="cpu", unet=pipe.unet, vae=pipe.vae,
                                                          text_encoder=pipe.text_encoder)


def test_model_constructors_refuse_config_values_they_do_not_implement():
    """a diffusers config.json key the compiled networks fix at one value (use_linear_projection,
    class_embed_type, upcast_attention, act_fn ...) is an error when it carries another one, not a silently dropped
    keyword; the SD-1.5 values, bookkeeping keys and unknown (newer) keys pass."""
    from powerpaint_amd import _lib as L
    from powerpaint_amd import models as PM
    tiny = dict(block_out_channels=(320, 640), layers_per_block=1, down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), device="cpu")
    ok = dict(_class_name="UNet2DConditionModel", _diffusers_version="0.27.0", act_fn="silu", use_linear_projection=False,
              class_embed_type=None, upcast_attention=False, mid_block_type="UNetMidBlock2DCrossAttn", dropout=0.0,
              some_future_option=1)
    PM.UNet2DConditionModel(in_channels=9, **tiny, **ok)
    for bad in (dict(use_linear_projection=True), dict(class_embed_type="timestep"), dict(act_fn="gelu"),
                dict(upcast_attention=True), dict(resnet_time_scale_shift="scale_shift"), dict(transformer_layers_per_block=2)):
        for cls, kw in ((PM.UNet2DConditionModel, dict(in_channels=9)), (PM.BrushNetModel, {}), (PM.ControlNetModel, {})):
            t = {k: v for k, v in tiny.items() if not (cls is PM.ControlNetModel and k == "up_block_types")}
            with pytest.raises(L.PPError):
                cls(**kw, **t, **bad)
