#!/usr/bin/env python
"""Generate tests/golden/ref_mask_prep.json from the REFERENCE'S OWN `prepare_mask_and_masked_image`
(/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:39-153), lifted out by AST (the module itself imports
diffusers) and run unmodified on seeded PIL / ndarray / tensor inputs.  Stored: the input recipe and the SHA-256 + shape
of the returned mask, masked image and image (row a20: integer mask ops are bit-exact)."""
import ast
import hashlib
import json
import os

import numpy as np
import PIL.Image
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py"


def sha(t):
    a = np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
    return dict(shape=list(a.shape), sha=hashlib.sha256(a.tobytes()).hexdigest())


def inputs(kind, seed, h, w, batch=1):
    rng = np.random.default_rng(seed)
    if kind == "pil":
        img = [PIL.Image.fromarray(rng.integers(0, 256, size=(h + 5, w + 3, 3), dtype=np.uint8)) for _ in range(batch)]
        msk = [PIL.Image.fromarray(rng.integers(0, 256, size=(h + 5, w + 3), dtype=np.uint8)) for _ in range(batch)]
        return (img[0], msk[0]) if batch == 1 else (img, msk)
    if kind == "numpy":
        img = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for _ in range(batch)]
        msk = [rng.random((h, w)).astype(np.float32) for _ in range(batch)]
        return (img[0], msk[0]) if batch == 1 else (img, msk)
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, h, w, generator=g) * 2 - 1
    msk = torch.rand(batch, 1, h, w, generator=g)
    if kind == "tensor3":
        return img[0], msk[0]
    if kind == "tensor2":
        return img[0], msk[0, 0]
    return img, msk


def main():
    tree = ast.parse(open(REF).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "prepare_mask_and_masked_image"][0]
    ns = dict(torch=torch, np=np, PIL=PIL)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "pipeline_PowerPaint.py", "exec"), ns)
    ref = ns["prepare_mask_and_masked_image"]
    cases = []
    for i, (kind, h, w, batch) in enumerate([("pil", 64, 96, 1), ("pil", 40, 40, 2), ("numpy", 32, 48, 1),
                                            ("numpy", 16, 16, 3), ("tensor", 24, 40, 2), ("tensor3", 24, 24, 1),
                                            ("tensor2", 8, 72, 1), ("pil", 128, 128, 1)]):
        img, msk = inputs(kind, 10 + i, h, w, batch)
        m, mi, im = ref(img, msk, h, w, return_image=True)
        cases.append(dict(kind=kind, seed=10 + i, h=h, w=w, batch=batch, mask=sha(m), masked=sha(mi), image=sha(im),
                          ones=int(m.sum())))
    with open(os.path.join(HERE, "ref_mask_prep.json"), "w") as f:
        json.dump(dict(cases=cases), f)
    print(len(cases), "cases")


if __name__ == "__main__":
    main()
