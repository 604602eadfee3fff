"""SURVEY.md §8f-4 -- controller pre / post-processing (powerpaint_amd/controller.py) against the reference's own
`PowerPaintController.predict / infer` (tests/golden/ref_controller.json, made by tests/golden/make_ref_controller.py):
every keyword handed to the pipeline and every returned image must match -- strings and numbers verbatim, images by
size, mode and SHA-256.  CPU only (PIL / NumPy host logic)."""
import hashlib
import json
import os
import types

import numpy as np
import pytest
import torch

PILImage = pytest.importorskip("PIL.Image")
from powerpaint_amd.controller import PowerPaintController, fit_short_side, outpaint_canvas, snap_to_eight  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def digest(img):
    return dict(size=list(img.size), mode=img.mode, sha=hashlib.sha256(np.array(img).tobytes()).hexdigest())


def make_inputs(w, h, seed):
    rng = np.random.default_rng(seed)
    img = PILImage.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
    m = np.zeros((h, w), dtype=np.uint8)
    m[h // 4: 3 * h // 4, w // 3: 2 * w // 3] = 255
    return {"image": img, "mask": PILImage.fromarray(m).convert("RGB")}


class RecordingPipe:
    def __init__(self):
        self.calls = []

    def __call__(self, **kw):
        rec = {}
        for k, v in kw.items():
            if isinstance(v, PILImage.Image):
                rec[k] = digest(v)
            elif isinstance(v, torch.Generator):
                rec[k] = dict(generator_seed=int(v.initial_seed()))
            else:
                rec[k] = v
        self.calls.append(rec)
        return types.SimpleNamespace(images=[PILImage.fromarray(255 - np.array(kw["image"].convert("RGB")))])


@pytest.fixture(scope="module")
def G():
    with open(os.path.join(HERE, "golden", "ref_controller.json")) as f:
        return json.load(f)


def test_predict_matches_reference(G):
    seeds = []
    for c in G["cases"]:
        pl = c["plan"]
        pipe = RecordingPipe()
        ctl = PowerPaintController(pipe, version=pl["version"], seed_fn=seeds.append, generator_device="cpu")
        inp = make_inputs(*c["size"], seed=c["input_seed"])
        out, res = ctl.predict(inp, "a red cat", 0.7, 12, 6.5, c["seed"], "blurry", pl["task"], pl["v"], pl["h"])
        assert pipe.calls[0] == c["call"], (pl, {k: (pipe.calls[0].get(k), c["call"].get(k)) for k in c["call"]
                                                  if pipe.calls[0].get(k) != c["call"].get(k)})
        assert [digest(o) for o in out] == c["out"] and [digest(r) for r in res] == c["res"], pl
        assert digest(inp["image"]) == c["final_inputs"]["image"] and digest(inp["mask"]) == c["final_inputs"]["mask"]
    assert seeds == [c["seed"] for c in G["cases"]]                     # set_seed(seed) right before the pipeline call


def test_infer_dispatch_matches_reference(G):
    for c in G["infer"]:
        pipe = RecordingPipe()
        ctl = PowerPaintController(pipe, version=c["version"], seed_fn=lambda s: None, generator_device="cpu")
        inp = make_inputs(240, 180, seed=c["input_seed"])
        ctl.infer(inp, "tg", "tg-neg", "sg", "sg-neg", 0.5, 5, 7.0, 9, c["task"], 1.2, 1.4, "op", "op-neg", "rm", "rm-neg")
        assert pipe.calls[0] == c["call"], c["task"]


def test_helpers_and_controlnet_path():
    img = PILImage.fromarray(np.zeros((200, 301, 3), dtype=np.uint8))
    assert fit_short_side(img, False).size == (int(301 / 200 * 640), 640) and fit_short_side(img, True).size[1] == 512
    tall = PILImage.fromarray(np.zeros((301, 200, 3), dtype=np.uint8))
    assert fit_short_side(tall, False).size == (640, int(301 / 200 * 640))
    canvas, mask = outpaint_canvas(PILImage.fromarray(np.full((40, 60, 3), 9, dtype=np.uint8)), 1.5, 1)
    assert canvas.size == (60, 60) and np.array(canvas)[0, 0, 0] == 127 and np.array(canvas)[10, 0, 0] == 9
    m = np.array(mask)[:, :, 0]
    assert m[:20].all() and (m[20:40] == 0).all() and m[40:].all()           # 10-pixel gap inside the 40 original rows
    _, _, wh = snap_to_eight(PILImage.fromarray(np.zeros((37, 45, 3), dtype=np.uint8)),
                             PILImage.fromarray(np.zeros((37, 45, 3), dtype=np.uint8)))
    assert wh == (40, 32)
    # ControlNet path: control image supplied by the caller, prompts as app.py:398-401, paste-back returned
    pipe = RecordingPipe()
    ctl = PowerPaintController(None, control_pipe=pipe, seed_fn=lambda s: None)
    inp = make_inputs(300, 200, seed=3)
    ctrl = PILImage.fromarray(np.zeros((50, 50, 3), dtype=np.uint8))
    out, res = ctl.predict_controlnet(inp, ctrl, "a dog", 7, 5.0, 1, "bad", 0.8)
    call = pipe.calls[0]
    assert call["promptA"] == call["promptB"] == "a dog P_obj" and call["negative_promptA"] == "bad"
    assert call["control_image"]["size"] == call["image"]["size"] == [960, 640]
    assert call["controlnet_conditioning_scale"] == 0.8 and call["tradoff"] == 1.0
    assert out[1].size == (960, 640) and res[0].size == (960, 640)
    with pytest.raises(ValueError):
        PowerPaintController(pipe).predict_controlnet(inp, ctrl, "a", 1, 1.0, 1, "", 1.0)


def test_predict_controlnet_matches_reference(G):
    """`predict_controlnet` (app.py:389-475) with the annotator output handed in: pipeline keywords, the paste-back
    (Gaussian radius 4) and the red-tint preview equal the reference's, by SHA-256."""
    for c in G["controlnet"]:
        pipe = RecordingPipe()
        seeds = []
        ctl = PowerPaintController(None, control_pipe=pipe, seed_fn=seeds.append)
        inp = make_inputs(*c["size"], seed=c["input_seed"])
        ctrl = make_inputs(90, 70, seed=c["ctrl_seed"])["image"]
        out, res = ctl.predict_controlnet(inp, ctrl, "a dog", 7, 5.0, c["seed"], "bad", 0.8)
        assert pipe.calls[0] == c["call"], {k: (pipe.calls[0].get(k), c["call"].get(k)) for k in c["call"]
                                             if pipe.calls[0].get(k) != c["call"].get(k)}
        assert [digest(o) for o in out] == c["out"] and [digest(r) for r in res] == c["res"]
        assert seeds == [c["seed"]]


@pytest.mark.gpu
def test_checkpoint_folder_to_controller_to_image_on_the_hip_path(tmp_path):
    """SURVEY.md section 8f-4 end to end on the GPU (app.py:84-200 + 245-387): a diffusers-layout pipeline folder on disk
    -> `StableDiffusionInpaintPipeline.from_pretrained` -> TokenizerWrapper + add_tokens (app.py:93-108) -> controller
    `predict` (text-guided, then an outpainting canvas) -> PIL, with UNet, VAE, CLIP tower and the fused loop all on the
    HIP kernels.  Checked against the same run driven by hand (the controller's documented pre-processing + a direct
    pipeline call under the same seed): identical pixels."""
    transformers = pytest.importorskip("transformers")
    from test_loaders import TINY, write_dir
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    from powerpaint_amd.controller import _set_seed, add_task
    from powerpaint_amd.utils import TokenizerWrapper, add_tokens
    root = str(tmp_path / "ppt-v1")
    unet = PM.UNet2DConditionModel(in_channels=9, device="cpu", **TINY)
    write_dir(os.path.join(root, "unet"), dict(TINY, in_channels=9, out_channels=4, sample_size=64),
              {k: v.half() for k, v in unet.net.synthetic_state_dict(seed=3).items()})
    vcfg = dict(block_out_channels=(64, 128, 256, 256), layers_per_block=1, in_channels=3, out_channels=3,
                latent_channels=4, norm_num_groups=32, scaling_factor=0.18215)
    vae = PM.AutoencoderKL(device="cpu", **vcfg)
    write_dir(os.path.join(root, "vae"), vcfg, vae.net.synthetic_state_dict(seed=5))
    with open(os.path.join(HERE, "golden", "ref_task_tokens.json")) as f:
        T = json.load(f)
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(T["vocab"])},
                                     merges=[tuple(m) for m in T["merges"]], model_max_length=77)
    tok.save_pretrained(os.path.join(root, "tokenizer"))
    n = len(tok)
    torch.manual_seed(0)
    hf = transformers.CLIPTextModel(transformers.CLIPTextConfig(
        vocab_size=n, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12,
        max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=n - 2, eos_token_id=n - 1, pad_token_id=n - 1))
    hf.save_pretrained(os.path.join(root, "text_encoder"))
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(dict(_class_name="PNDMScheduler", beta_end=0.012, beta_schedule="scaled_linear", beta_start=0.00085,
                       num_train_timesteps=1000, set_alpha_to_one=False, skip_prk_steps=True, steps_offset=1,
                       trained_betas=None, clip_sample=False), f)
    with open(os.path.join(root, "model_index.json"), "w") as f:
        json.dump({"_class_name": "StableDiffusionInpaintPipeline", "scheduler": ["diffusers", "PNDMScheduler"],
                   "text_encoder": ["transformers", "CLIPTextModel"], "tokenizer": ["transformers", "CLIPTokenizer"],
                   "unet": ["diffusers", "UNet2DConditionModel"], "vae": ["diffusers", "AutoencoderKL"],
                   "feature_extractor": ["transformers", "CLIPImageProcessor"],
                   "safety_checker": ["stable_diffusion", "StableDiffusionSafetyChecker"]}, f)

    pipe = PP.StableDiffusionInpaintPipeline.from_pretrained(root, torch_dtype=torch.bfloat16, device="cuda",
                                                             local_files_only=True)
    assert isinstance(pipe.scheduler, PS.PNDMScheduler) and pipe.unet.device.type == "cuda"
    pipe.tokenizer = TokenizerWrapper(from_pretrained=root, subfolder="tokenizer", revision=None)      # app.py:93-97
    add_tokens(tokenizer=pipe.tokenizer, text_encoder=pipe.text_encoder,
               placeholder_tokens=["P_ctxt", "P_shape", "P_obj"], initialize_tokens=["a", "a", "a"],
               num_vectors_per_token=10)                                                                # app.py:102-108
    with torch.no_grad():
        for e in pipe.text_encoder.text_model.embeddings.token_embedding.external_embeddings:
            e["embedding"].copy_(torch.randn_like(e["embedding"]) * 0.05)
    ctl = PowerPaintController(pipe, version="ppt-v1")
    inp = make_inputs(320, 320, seed=7)
    keep = {"image": inp["image"].copy(), "mask": inp["mask"].copy()}
    out, res = ctl.predict(inp, "a red cat", 0.7, 3, 6.5, 123, "blurry", "text-guided")
    assert len(out) == 1 and out[0].size == (640, 640) and out[0].mode == "RGB" and len(res) == 2
    assert np.array(out[0]).std() > 1.0
    # the same request by hand: the controller's pre-processing, then the pipeline under the same seed
    image = fit_short_side(keep["image"], False)
    image, mask, (w, h) = snap_to_eight(image, keep["mask"])
    pA, pB, nA, nB = add_task("a red cat", "blurry", "text-guided", "ppt-v1")
    _set_seed(123)
    again = pipe(promptA=pA, promptB=pB, tradoff=0.7, tradoff_nag=0.7, negative_promptA=nA, negative_promptB=nB,
                 image=image.convert("RGB"), mask=mask.convert("RGB"), width=w, height=h, guidance_scale=6.5,
                 num_inference_steps=3).images[0]
    assert np.array_equal(np.array(out[0]), np.array(again))
    other, _ = ctl.predict({"image": keep["image"].copy(), "mask": keep["mask"].copy()}, "a red cat", 0.7, 3, 6.5, 124,
                           "blurry", "text-guided")
    assert not np.array_equal(np.array(other[0]), np.array(out[0]))                  # the seed reaches the noise
    # outpainting canvas (app.py:262-304): 512-short-side image on a 1.5x taller grey canvas
    o2, r2 = ctl.predict({"image": keep["image"].copy(), "mask": keep["mask"].copy()}, "", 1.0, 2, 7.5, 5, "",
                         "image-outpainting", 1.5, 1.0)
    assert o2[0].size == (512, 768) and r2[0].size == (512, 768) and np.isfinite(np.array(o2[0], dtype=np.float32)).all()
