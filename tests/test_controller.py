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
