#!/usr/bin/env python
"""Generate tests/golden/ref_encode_prompt.pt from the REFERENCE'S OWN `StableDiffusionInpaintPipeline._encode_prompt`
(/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:317-518), lifted out by AST (the module imports diffusers)
and run unmodified on a stand-in `self` holding a real transformers CLIP tokenizer (the small vocabulary of
ref_task_tokens.json) and a small randomly initialised `transformers.CLIPTextModel`.  Stored: the encoder's config and
state dict (so the test rebuilds the identical encoder), the call arguments and the returned prompt_embeds.
"""
import ast
import json
import os
import types
from typing import List, Optional

import torch
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py"
CFG = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77,
           hidden_act="quick_gelu")

CASES = [
    dict(promptA="the cat and the dog", promptB="the empty scene", t=0.3, n=1, cfg=True, nA="blur", nB="the dog", tn=0.8),
    dict(promptA="the cat", promptB="the cat", t=1.0, n=2, cfg=True, nA=None, nB=None, tn=1.0),
    dict(promptA=["the cat", "the dog and the scene"], promptB=["blur", "the empty scene"], t=0.5, n=1, cfg=True,
         nA=["blur", "the dog"], nB=["the cat", "blur"], tn=0.25),
    dict(promptA="the scene", promptB="the dog", t=0.0, n=1, cfg=False, nA=None, nB=None, tn=1.0),
    dict(promptA=["the cat", "the dog"], promptB=["the dog", "the cat"], t=0.7, n=3, cfg=True, nA=None, nB=None, tn=0.5),
]


def tokenizer_and_encoder():
    with open(os.path.join(HERE, "ref_task_tokens.json")) as f:
        G = json.load(f)
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(G["vocab"])},
                                     merges=[tuple(m) for m in G["merges"]], model_max_length=77)
    n = len(tok)
    cfg = transformers.CLIPTextConfig(vocab_size=n, bos_token_id=n - 2, eos_token_id=n - 1, pad_token_id=n - 1, **CFG)
    torch.manual_seed(0)
    enc = transformers.CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() >= 2:
                p.mul_(3.0)
    return tok, enc, n


def main():
    tree = ast.parse(open(REF).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "StableDiffusionInpaintPipeline"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_encode_prompt"][0]
    ns = dict(torch=torch, Optional=Optional, List=List, LoraLoaderMixin=type("L", (), {}),
              TextualInversionLoaderMixin=type("T", (), {}),
              logger=types.SimpleNamespace(warning=lambda *a, **k: None, info=lambda *a, **k: None))
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "pipeline_PowerPaint.py", "exec"), ns)
    ref = ns["_encode_prompt"]
    tok, enc, n = tokenizer_and_encoder()
    me = types.SimpleNamespace(tokenizer=tok, text_encoder=enc, unet=None)
    outs = []
    with torch.no_grad():
        for c in CASES:
            outs.append(ref(me, c["promptA"], c["promptB"], c["t"], torch.device("cpu"), c["n"], c["cfg"], c["nA"],
                            c["nB"], c["tn"]).clone())
        # pre-computed embeds pass straight through (repeat + CFG concat only)
        pe, ne = torch.randn(2, 77, 32, generator=torch.Generator().manual_seed(1)), \
            torch.randn(2, 77, 32, generator=torch.Generator().manual_seed(2))
        outs.append(ref(me, None, None, 0.5, torch.device("cpu"), 2, True, None, None, 0.5, prompt_embeds=pe,
                        negative_prompt_embeds=ne).clone())
    # the plain (promptU) encoder of the BrushNet pipeline: pipeline_PowerPaint_Brushnet_CA.py:442-629
    tree2 = ast.parse(open(REF.replace("pipeline_PowerPaint.py", "pipeline_PowerPaint_Brushnet_CA.py")).read())
    cls2 = [n for n in tree2.body if isinstance(n, ast.ClassDef) and n.name == "StableDiffusionPowerPaintBrushNetPipeline"][0]
    fn2 = [n for n in cls2.body if isinstance(n, ast.FunctionDef) and n.name == "encode_prompt"][0]
    ns2 = dict(ns, USE_PEFT_BACKEND=False, adjust_lora_scale_text_encoder=lambda *a, **k: None,
               scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None)
    exec(compile(ast.Module(body=[fn2], type_ignores=[]), "pipeline_PowerPaint_Brushnet_CA.py", "exec"), ns2)
    ref2 = ns2["encode_prompt"]
    U_CASES = [dict(prompt="the cat and the dog", n=1, neg="blur"), dict(prompt="the empty scene", n=2, neg=None),
               dict(prompt=["the cat", "the dog"], n=1, neg=["blur", "the scene"]),
               dict(prompt=["the cat", "blur"], n=3, neg=None)]
    outs_u = []
    with torch.no_grad():
        for c in U_CASES:
            outs_u.append(ref2(me, c["prompt"], torch.device("cpu"), c["n"], True, c["neg"]).clone())
        outs_u.append(ref2(me, None, torch.device("cpu"), 2, True, None, prompt_embeds=pe, negative_prompt_embeds=ne).clone())
    torch.save(dict(cfg=CFG, vocab_size=n, state_dict=enc.state_dict(), cases=CASES, outs=outs, pe=pe, ne=ne,
                    u_cases=U_CASES, outs_u=outs_u), os.path.join(HERE, "ref_encode_prompt.pt"))
    print([tuple(o.shape) for o in outs], [tuple(o.shape) for o in outs_u])


if __name__ == "__main__":
    main()
