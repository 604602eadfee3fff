#!/usr/bin/env python
"""Generate tests/golden/ref_controller.json from the REFERENCE'S OWN controller code (build container only).

/root/reference/app.py cannot be imported (gradio, cv2, controlnet_aux, ...), so `PowerPaintController.predict / infer`,
`add_task` and `set_seed` are lifted out by AST and executed unmodified over a recording stand-in for the pipeline
(returns the inverted input image, so the post-processing sees image-dependent data).  `torch.Generator("cuda")` of the
v2 branch is redirected to the CPU (there is no GPU here); nothing else is touched.

Stored per case: the inputs' recipe (seeded), every keyword the reference hands to the pipeline (strings / numbers
verbatim, images as size + mode + SHA-256) and the SHA-256 of the images it returns.
"""
import ast
import hashlib
import json
import os
import random
import types

import numpy as np
import torch
from PIL import Image, ImageFilter

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/app.py"


def digest(img):
    return dict(size=list(img.size), mode=img.mode, sha=hashlib.sha256(np.array(img).tobytes()).hexdigest())


def make_inputs(w, h, seed):
    rng = np.random.default_rng(seed)
    img = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
    m = np.zeros((h, w), dtype=np.uint8)
    m[h // 4: 3 * h // 4, w // 3: 2 * w // 3] = 255
    return {"image": img, "mask": Image.fromarray(m).convert("RGB")}


class RecordingPipe:
    def __init__(self):
        self.calls = []

    def __call__(self, **kw):
        rec = {}
        for k, v in kw.items():
            if isinstance(v, Image.Image):
                rec[k] = digest(v)
            elif isinstance(v, torch.Generator):
                rec[k] = dict(generator_seed=int(v.initial_seed()))
            else:
                rec[k] = v
        self.calls.append(rec)
        out = Image.fromarray(255 - np.array(kw["image"].convert("RGB")))
        return types.SimpleNamespace(images=[out])


def load_reference():
    tree = ast.parse(open(REF).read())
    fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("add_task", "set_seed")}
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PowerPaintController"][0]
    meths = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("predict", "infer",
                                                                                   "predict_controlnet")]
    stub = ast.ClassDef(name="Ref", bases=[], keywords=[], body=meths, decorator_list=[])
    mod = ast.Module(body=[fns["set_seed"], fns["add_task"], stub], type_ignores=[])
    ast.fix_missing_locations(mod)
    fake_torch = types.SimpleNamespace(Generator=lambda dev=None: torch.Generator("cpu"), manual_seed=torch.manual_seed,
                                       cuda=types.SimpleNamespace(manual_seed=lambda s: None,
                                                                  manual_seed_all=lambda s: None))
    ns = dict(np=np, Image=Image, ImageFilter=ImageFilter, torch=fake_torch, random=random, print=lambda *a, **k: None)
    exec(compile(mod, "app.py", "exec"), ns)
    return ns["Ref"]


def main():
    Ref = load_reference()
    cases = []
    sizes = [(300, 200), (200, 300), (257, 257)]
    plans = []
    for version in ("ppt-v1", "ppt-v2"):
        for task in ("text-guided", "shape-guided", "object-removal"):
            plans.append(dict(version=version, task=task, v=None, h=None))
        for v, h in ((1, 1.5), (1.5, 1), (1.3, 1.7), (1, 1)):
            plans.append(dict(version=version, task="image-outpainting", v=v, h=h))
    for i, pl in enumerate(plans):
        w, hh = sizes[i % len(sizes)]
        ctl = Ref()
        ctl.version, ctl.pipe = pl["version"], RecordingPipe()
        inp = make_inputs(w, hh, seed=i)
        out, res = ctl.predict(inp, "a red cat", 0.7, 12, 6.5, 40 + i, "blurry", pl["task"], pl["v"], pl["h"])
        cases.append(dict(plan=pl, size=[w, hh], input_seed=i, seed=40 + i, call=ctl.pipe.calls[0],
                          out=[digest(o) for o in out], res=[digest(r) for r in res],
                          final_inputs=dict(image=digest(inp["image"]), mask=digest(inp["mask"]))))
    infer_cases = []
    for i, (task, version) in enumerate((("shape-guided", "ppt-v1"), ("image-outpainting", "ppt-v2"),
                                         ("something-else", "ppt-v1"), ("object-removal", "ppt-v2"))):
        ctl = Ref()
        ctl.version, ctl.pipe = version, RecordingPipe()
        inp = make_inputs(240, 180, seed=100 + i)
        ctl.infer(inp, "tg", "tg-neg", "sg", "sg-neg", 0.5, 5, 7.0, 9, task, 1.2, 1.4, "op", "op-neg", "rm", "rm-neg")
        infer_cases.append(dict(task=task, version=version, input_seed=100 + i, call=ctl.pipe.calls[0]))
    # ControlNet path: the annotator is a stand-in (HED slot, returns its input) -- the real ones are third-party models
    cn_cases = []
    for i, (w, hh) in enumerate(((300, 200), (200, 300))):
        ctl = Ref()
        ctl.version, ctl.control_pipe, ctl.current_control = "ppt-v1", RecordingPipe(), "hed"
        ctl.hed = lambda im: im
        inp = make_inputs(w, hh, seed=200 + i)
        ctrl = make_inputs(90, 70, seed=300 + i)["image"]
        out, res = ctl.predict_controlnet(inp, ctrl, "hed", "a dog", 7, 5.0, 3 + i, "bad", 0.8)
        cn_cases.append(dict(size=[w, hh], input_seed=200 + i, ctrl_seed=300 + i, seed=3 + i,
                             call=ctl.control_pipe.calls[0], out=[digest(o) for o in out],
                             res=[digest(r) for r in res]))
    with open(os.path.join(HERE, "ref_controller.json"), "w") as f:
        json.dump(dict(cases=cases, infer=infer_cases, controlnet=cn_cases), f)
    print(len(cases), "predict cases,", len(infer_cases), "infer cases,", len(cn_cases), "controlnet cases")


if __name__ == "__main__":
    main()
