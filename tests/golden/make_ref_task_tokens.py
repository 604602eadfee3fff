#!/usr/bin/env python
"""Generate tests/golden/ref_task_tokens.json from the REFERENCE'S OWN source (run in the build container only).

/root/reference/powerpaint/utils/utils.py is imported unmodified (its one missing import, `mmengine.print_log`, is
replaced by a no-op) and `add_task` is lifted out of /root/reference/app.py by AST (the file itself needs gradio).
There is no CLIP checkpoint here, so the tokenizer is a real `transformers.CLIPTokenizer` over a small hand-made BPE
vocabulary (stored in the fixture so the tests rebuild the identical tokenizer anywhere) and the text encoder is a stub
exposing `.text_model.embeddings.token_embedding`.

Stored:
  * vocabulary / merges, the placeholder registration and the resulting token_map and id ranges;
  * for a list of prompts: the expanded text, the padded ids, and column 0 of the reference embedding layer's output
    -- or the error type where the reference raises.  Column 0 of every table row holds that row's identity, and the
    generator asserts that each full output row equals the table row its column 0 names, so column 0 carries the
    whole (bit-exact) result;
  * 300 seeded synthetic id sequences (runs, adjacent runs, orphans, truncated runs), same encoding;
  * add_task outputs for every task / version; the state-dict keys of the patched encoder.
"""
import ast
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
DIM = 16
SEQ = 40


def load_reference():
    mm = types.ModuleType("mmengine")
    mm.print_log = lambda *a, **k: None
    sys.modules["mmengine"] = mm
    spec = importlib.util.spec_from_file_location("ref_pp_utils", os.path.join(REF, "powerpaint/utils/utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tree = ast.parse(open(os.path.join(REF, "app.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "add_task"][0]
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "app.py", "exec"), ns)
    return mod, ns["add_task"]


def tiny_vocab():
    from tokenizers import pre_tokenizers
    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    merges = [("a", "n"), ("an", "d</w>"), ("o", "f</w>"), ("t", "h"), ("th", "e</w>"), ("s", "c"), ("e", "n"),
              ("sc", "en"), ("scen", "e</w>"), ("e", "m"), ("p", "t"), ("em", "pt"), ("empt", "y</w>"), ("b", "l"),
              ("u", "r</w>"), ("bl", "ur</w>"), ("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>")]
    vocab = alphabet + [c + "</w>" for c in alphabet] + [a + b for a, b in merges] + ["<|startoftext|>",
                                                                                     "<|endoftext|>"]
    return vocab, [list(m) for m in merges]


class _Emb(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.token_embedding = nn.Embedding(n, DIM)


class _TextModel(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.embeddings = _Emb(n)


class StubEncoder(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.text_model = _TextModel(n)


def identity_weights(n_base, layer):
    """Column 0 of every row names the row: base row r -> r, external block k row j -> 100000 + 100 k + j."""
    g = torch.Generator().manual_seed(1234)
    w = torch.randn(n_base, DIM, generator=g)
    w[:, 0] = torch.arange(n_base, dtype=torch.float32)
    with torch.no_grad():
        layer.wrapped.weight.copy_(w)
        for k, e in enumerate(layer.external_embeddings):
            blk = torch.randn(e["end"] - e["start"], DIM, generator=g)
            blk[:, 0] = 100000 + 100 * k + torch.arange(blk.shape[0], dtype=torch.float32)
            e["embedding"].copy_(blk)


def col0(layer, out):
    """Identity column of a reference output [batch, length, dim], after checking every row is an exact table row."""
    c = out[..., 0].to(torch.int64)
    rows = torch.empty_like(out)
    for b in range(out.shape[0]):
        for i in range(out.shape[1]):
            v = int(c[b, i])
            if v < 100000:
                rows[b, i] = layer.wrapped.weight[v]
            else:
                k, j = divmod(v - 100000, 100)
                rows[b, i] = layer.external_embeddings[k]["embedding"][j]
    assert torch.equal(rows, out)
    return c.tolist()


def run(layer, ids):
    try:
        with torch.no_grad():
            return {"out": col0(layer, layer(torch.tensor(ids, dtype=torch.long)))}
    except AssertionError:
        return {"error": "AssertionError"}
    except IndexError:
        return {"error": "IndexError"}


def synthetic_ids(rng, n_base, spans):
    """One sequence mixing ordinary ids, complete runs, adjacent runs, orphan pieces and (sometimes) a cut run."""
    seq = []
    while len(seq) < SEQ:
        r = rng.random()
        sp = spans[rng.integers(len(spans))]
        full = list(range(sp["start"], sp["end"]))
        if r < 0.45:
            piece = [int(rng.integers(n_base))]
        elif r < 0.70:
            piece = full
        elif r < 0.80:
            piece = full + full                                  # adjacent runs: the second is not spliced
        elif r < 0.90:
            piece = [int(rng.integers(sp["start"] + 1, sp["end"])) if sp["end"] - sp["start"] > 1
                     else int(rng.integers(n_base))]             # orphan piece
        elif r < 0.92:
            piece = full[: max(1, len(full) // 2)] + [int(rng.integers(n_base))]   # incomplete run -> assert
        else:
            piece = [sp["end"] + 50]                             # beyond every range -> row 0
        if len(seq) + len(piece) > SEQ and rng.random() < 0.9:
            piece = [int(rng.integers(n_base))]                  # mostly: do not let the sequence end cut a run
        seq += piece
    return seq[:SEQ]


def main():
    ref, add_task = load_reference()
    import transformers
    vocab, merges = tiny_vocab()
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(vocab)}, merges=[tuple(m) for m in merges],
                                     model_max_length=77)
    d = tempfile.mkdtemp()
    tok.save_pretrained(d)
    wrapper = ref.TokenizerWrapper(from_pretrained=d)
    n_base = len(wrapper.wrapped)
    enc = StubEncoder(n_base)
    placeholders = ["P_ctxt", "P_shape", "P_obj"]
    ref.add_tokens(tokenizer=wrapper, text_encoder=enc, placeholder_tokens=placeholders,
                   initialize_tokens=["a", "a", "a"], num_vectors_per_token=10)
    layer = enc.text_model.embeddings.token_embedding
    init_equal = [bool(torch.equal(e["embedding"].detach(),
                                   layer.weight[wrapper("a").input_ids[1]].detach()[None].repeat(10, 1)))
                  for e in layer.external_embeddings]
    # a single-vector placeholder registered directly, not trainable
    wrapper.add_placeholder_token("P_one", num_vec_per_token=1)
    info = wrapper.get_token_info("P_one")
    info["embedding"] = torch.zeros(1, DIM)
    layer.add_embeddings(info)
    identity_weights(n_base, layer)

    out = dict(vocab=vocab, merges=merges, dim=DIM, n_base=n_base, placeholders=placeholders, num_vec=10,
               token_map=wrapper.token_map, init_equal=init_equal,
               spans=[dict(name=e["name"], start=int(e["start"]), end=int(e["end"])) for e in
                      layer.external_embeddings],
               state_dict_keys=sorted(enc.state_dict().keys()),
               weight_seed=1234)

    # ---- task prompts
    tasks = []
    for task in ["text-guided", "object-removal", "image-outpainting", "shape-guided", "context-aware"]:
        for version in ["ppt-v1", "ppt-v2"]:
            tasks.append(dict(task=task, version=version, prompt="a cat", negative="blur dog",
                              out=list(add_task("a cat", "blur dog", task, version))))
    out["add_task"] = tasks

    # ---- prompts through tokenizer + embedding layer
    prompts = [t for tk in tasks for t in tk["out"]]
    prompts = sorted(set(prompts)) + [
        "the empty scene", "P_obj", "P_obj P_obj", "P_objP_ctxt", "P_obj P_ctxt P_shape P_one",
        "a P_obj_3 of the cat", "P_one P_one", "P_oneP_one", "the scene P_one and P_obj_0 P_obj_1",
        "the empty scene and the cat and the dog " * 8 + "P_obj",                   # run cut by max_length -> assert
        "the empty scene and the cat " * 9 + "P_shape",
        ["a cat P_ctxt", "P_shape the dog"],
    ]
    cases = []
    for p in prompts:
        text = wrapper.replace_placeholder_tokens_in_text(p)
        ids = wrapper(p, padding="max_length", max_length=77, truncation=True).input_ids
        ids2 = ids if isinstance(p, list) else [ids]
        r = run(layer, ids2)
        c = dict(prompt=p, text=text, ids=ids, encode_ids=wrapper.encode(p).input_ids)
        if "out" in r:
            c["col0"] = r["out"]
        else:
            c["error"] = r["error"]
        cases.append(c)
    out["cases"] = cases
    out["prop_half"] = wrapper.replace_placeholder_tokens_in_text("a P_obj", prop_tokens_to_load=0.5)
    out["prop_half_list"] = wrapper.replace_placeholder_tokens_in_text(["a P_obj"], prop_tokens_to_load=0.5)
    ids = wrapper("a cat P_obj").input_ids
    out["decode"] = dict(ids=ids, text=wrapper.decode(ids), raw=wrapper.decode(ids, return_raw=True))
    out["token_info"] = {p: wrapper.get_token_info(p) for p in placeholders + ["P_one"]}

    # ---- synthetic id sequences straight into the embedding layer
    rng = np.random.default_rng(2024)
    synth = []
    for _ in range(300):
        seq = synthetic_ids(rng, n_base, out["spans"])
        r = run(layer, [seq])
        synth.append(dict(ids=seq, col0=r["out"][0]) if "out" in r else dict(ids=seq, error=r["error"]))
    out["synthetic"] = synth
    # per-call external embedding (forward's second argument) and the 1-D id form
    extra = dict(name="extra", start=n_base + 200, end=n_base + 203,
                 embedding=torch.full((3, DIM), 7.0))
    with torch.no_grad():
        seq = [5, n_base + 200, n_base + 201, n_base + 202, 9, out["spans"][0]["start"]] + \
              list(range(out["spans"][0]["start"] + 1, out["spans"][0]["end"]))
        o = layer(torch.tensor(seq), external_embeddings=extra)
    assert torch.equal(o[0, 1:4], extra["embedding"])
    o[0, 1:4, 0] = torch.tensor([200000.0, 200001.0, 200002.0])
    out["extra_case"] = dict(ids=seq, extra=dict(name="extra", start=extra["start"], end=extra["end"], value=7.0),
                             col0=[int(v) for v in o[0, :, 0].tolist()])
    n_err = sum("error" in s for s in synth)
    print(f"cases {len(cases)} (errors {sum('error' in c for c in cases)}), synthetic 300 (errors {n_err})")
    with open(os.path.join(HERE, "ref_task_tokens.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
