#!/usr/bin/env python
"""Generate tests/golden/ref_check_inputs.json: the verdict ("ok" or the exception type) of the REFERENCE'S OWN
`StableDiffusionInpaintPipeline.check_inputs` (pipeline_PowerPaint.py:553-602, lifted by AST) on 1728 combinations."""
import ast
import itertools
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py"


def main():
    tree = ast.parse(open(REF).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "StableDiffusionInpaintPipeline"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "check_inputs"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    ref = ns["check_inputs"]
    emb = {"E": torch.zeros(1, 77, 8), "E2": torch.zeros(2, 77, 8), None: None}
    rows = []
    for p, (h, w), s, cb, ng, pe, ne in itertools.product([None, "a", ["a", "b"], 5],
                                                          [(512, 512), (500, 512), (512, 33)], [1.0, -0.1, 1.5],
                                                          [1, 0, None, 2.5], [None, "n"], [None, "E"], [None, "E", "E2"]):
        try:
            ref(None, p, h, w, s, cb, ng, emb[pe], emb[ne])
            r = "ok"
        except Exception as e:
            r = type(e).__name__
        rows.append([p, h, w, s, cb, ng, pe, ne, r])
    json.dump(rows, open(os.path.join(HERE, "ref_check_inputs.json"), "w"))
    print(len(rows), sum(r[-1] == "ok" for r in rows))


if __name__ == "__main__":
    main()
