"""SURVEY.md §8 row a21 -- task-prompt tokens (TokenizerWrapper / EmbeddingLayerWithFixes / add_tokens / add_task).

tests/golden/ref_task_tokens.json holds outputs of the reference's own classes (tests/golden/make_ref_task_tokens.py).
CPU: the oracle restatement and the product's host logic (text expansion, id ranges, the splice plan, error
behaviour) against that fixture.  GPU: the embedding layer itself, bit-exact, through pp_embed_splice.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import task_tokens as OT
from powerpaint_amd.utils import EmbeddingLayerWithFixes, TokenizerWrapper, add_task, add_tokens, splice_plan

HERE = os.path.dirname(os.path.abspath(__file__))
transformers = pytest.importorskip("transformers")


@pytest.fixture(scope="module")
def G():
    with open(os.path.join(HERE, "golden", "ref_task_tokens.json")) as f:
        return json.load(f)


class _Emb(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.token_embedding = nn.Embedding(n, dim)


class _TextModel(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.embeddings = _Emb(n, dim)


class StubEncoder(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.text_model = _TextModel(n, dim)


def make_tokenizer(G):
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(G["vocab"])},
                                     merges=[tuple(m) for m in G["merges"]], model_max_length=77)
    return TokenizerWrapper(tokenizer=tok)


def tables(G):
    """The identity-column tables of the generator (column 0 of a row names the row)."""
    g = torch.Generator().manual_seed(G["weight_seed"])
    w = torch.randn(G["n_base"], G["dim"], generator=g)
    w[:, 0] = torch.arange(G["n_base"], dtype=torch.float32)
    blocks = []
    for k, sp in enumerate(G["spans"]):
        blk = torch.randn(sp["end"] - sp["start"], G["dim"], generator=g)
        blk[:, 0] = 100000 + 100 * k + torch.arange(blk.shape[0], dtype=torch.float32)
        blocks.append(blk)
    return w, blocks


def build(G, device="cpu", dtype=torch.float32):
    """Same registration sequence as the generator ran on the reference, on the product's classes."""
    wrapper = make_tokenizer(G)
    enc = StubEncoder(G["n_base"], G["dim"])
    add_tokens(tokenizer=wrapper, text_encoder=enc, placeholder_tokens=G["placeholders"],
               initialize_tokens=["a"] * len(G["placeholders"]), num_vectors_per_token=G["num_vec"])
    layer = enc.text_model.embeddings.token_embedding
    init_equal = [bool(torch.equal(e["embedding"].detach(),
                                   layer.weight[wrapper("a").input_ids[1]].detach()[None].repeat(G["num_vec"], 1)))
                  for e in layer.external_embeddings]
    wrapper.add_placeholder_token("P_one", num_vec_per_token=1)
    info = wrapper.get_token_info("P_one")
    info["embedding"] = torch.zeros(1, G["dim"])
    layer.add_embeddings(info)
    w, blocks = tables(G)
    with torch.no_grad():
        layer.wrapped.weight.copy_(w)
        for e, b in zip(layer.external_embeddings, blocks):
            e["embedding"].copy_(b)
    enc.to(device=device, dtype=dtype)
    info["embedding"] = info["embedding"].to(device=device, dtype=dtype)    # a plain tensor: Module.to() skips it
    return wrapper, enc, layer, init_equal


def plan_to_col0(plan, spans_with_row0):
    out = plan.astype(np.int64).copy()
    for k, sp in enumerate(spans_with_row0):
        w = sp["end"] - sp["start"]
        for j in range(w):
            out[plan == -(sp["row0"] + j) - 1] = 100000 + 100 * k + j
    return out


def with_row0(spans):
    out, r = [], 0
    for sp in spans:
        out.append(dict(sp, row0=r))
        r += sp["end"] - sp["start"]
    return out


# ------------------------------------------------------------------------------------------------ CPU: oracle pinned
def test_oracle_embedding_matches_reference(G):
    w, blocks = tables(G)
    ext = [dict(sp, embedding=b.numpy()) for sp, b in zip(G["spans"], blocks)]
    n_ok = n_err = 0
    for c in G["synthetic"] + [dict(ids=c["ids"], **({"col0": c["col0"]} if "col0" in c else {"error": c["error"]}))
                              for c in G["cases"]]:
        ids = np.asarray(c["ids"])
        if "error" in c:
            with pytest.raises(AssertionError):
                OT.embedding_with_fixes(ids, w.numpy(), ext)
            n_err += 1
        else:
            out = OT.embedding_with_fixes(ids, w.numpy(), ext)
            ref = np.asarray(c["col0"]).reshape(out.shape[:2])
            assert np.array_equal(out[..., 0].astype(np.int64), ref)
            n_ok += 1
    assert n_ok > 200 and n_err > 50


def test_oracle_text_and_tasks_match_reference(G):
    for c in G["cases"]:
        assert OT.expand_placeholders(c["prompt"], G["token_map"]) == c["text"]
    assert OT.expand_placeholders("a P_obj", G["token_map"], 0.5) == G["prop_half"]
    for t in G["add_task"]:
        assert list(OT.task_prompts(t["prompt"], t["negative"], t["task"], t["version"])) == t["out"]
    for p in G["placeholders"]:
        assert OT.placeholder_names(p, G["num_vec"]) == G["token_map"][p]


# ------------------------------------------------------------------------------------------------ CPU: host logic
def test_tokenizer_wrapper_matches_reference(G):
    wrapper, enc, layer, init_equal = build(G)
    assert wrapper.token_map == G["token_map"]
    assert init_equal == G["init_equal"] == [True] * 3
    assert [dict(name=e["name"], start=int(e["start"]), end=int(e["end"])) for e in layer.external_embeddings] \
        == G["spans"]
    assert sorted(enc.state_dict().keys()) == G["state_dict_keys"]          # reference checkpoints load unchanged
    for p, info in G["token_info"].items():
        assert wrapper.get_token_info(p) == info
    for c in G["cases"]:
        assert wrapper.replace_placeholder_tokens_in_text(c["prompt"]) == c["text"]
        assert wrapper(c["prompt"], padding="max_length", max_length=77, truncation=True).input_ids == c["ids"]
        assert wrapper.encode(c["prompt"]).input_ids == c["encode_ids"]
    assert wrapper.replace_placeholder_tokens_in_text("a P_obj", prop_tokens_to_load=0.5) == G["prop_half"]
    assert wrapper.replace_placeholder_tokens_in_text(["a P_obj"], prop_tokens_to_load=0.5) == G["prop_half_list"]
    d = G["decode"]
    assert wrapper.decode(d["ids"]) == d["text"] and wrapper.decode(d["ids"], return_raw=True) == d["raw"]
    # forwarded attributes and the error paths
    assert wrapper.model_max_length == 77 and wrapper.eos_token_id == wrapper.wrapped.eos_token_id
    with pytest.raises(AttributeError):
        wrapper.no_such_attribute
    with pytest.raises(AssertionError):
        wrapper.add_placeholder_token("P_obj", num_vec_per_token=10)        # pieces already in the vocabulary
    with pytest.raises(ValueError):
        wrapper.add_placeholder_token("my_P_obj_x", num_vec_per_token=2)    # contains a known placeholder
    with pytest.raises(AssertionError):
        TokenizerWrapper(from_pretrained="a", from_config="b")
    shuffled = wrapper.replace_placeholder_tokens_in_text("P_obj", vector_shuffle=True).split(" ")
    assert sorted(shuffled) == sorted(G["token_map"]["P_obj"])


def test_add_task_matches_reference(G):
    for t in G["add_task"]:
        assert list(add_task(t["prompt"], t["negative"], t["task"], t["version"])) == t["out"]


def test_splice_plan_matches_reference(G):
    spans = with_row0(G["spans"])
    n_ok = n_err = 0
    cases = G["synthetic"] + [c for c in G["cases"]]
    for c in cases:
        ids = np.asarray(c["ids"])
        ids = ids[None] if ids.ndim == 1 else ids
        if "error" in c:
            with pytest.raises(AssertionError):
                splice_plan(ids, G["n_base"], spans)
            n_err += 1
            continue
        plan = splice_plan(ids, G["n_base"], spans)
        assert plan.dtype == np.int32
        assert np.array_equal(plan_to_col0(plan, spans), np.asarray(c["col0"]).reshape(plan.shape))
        n_ok += 1
    assert n_ok > 200 and n_err > 50


def test_splice_plan_matches_oracle_random():
    """Independent derivations (source map vs concatenation) agree on seeded random id soup, errors included."""
    rng = np.random.default_rng(5)
    n_base, dim = 50, 4
    spans = with_row0([dict(name="a", start=50, end=53), dict(name="b", start=53, end=54),
                       dict(name="c", start=60, end=64)])
    w = np.arange(n_base, dtype=np.float32)[:, None].repeat(dim, 1)
    ext = [dict(sp, embedding=(1000 + sp["row0"] + np.arange(sp["end"] - sp["start"], dtype=np.float32))[:, None]
                .repeat(dim, 1)) for sp in spans]
    n_err = 0
    for _ in range(3000):
        n = int(rng.integers(1, 14))
        # small alphabet around the ranges so that runs, near-runs and adjacent runs all occur by chance
        ids = rng.choice([1, 2, 50, 51, 52, 53, 54, 60, 61, 62, 63, 70], size=(2, n))
        if rng.random() < 0.5:
            p = int(rng.integers(0, n))
            run = np.arange(50, 53) if rng.random() < 0.5 else np.arange(60, 64)
            ids[0, p:p + len(run)] = run[: n - p]
        try:
            want = OT.embedding_with_fixes(ids, w, ext)
        except AssertionError:
            n_err += 1
            with pytest.raises(AssertionError):
                splice_plan(ids, n_base, spans)
            continue
        plan = splice_plan(ids, n_base, spans)
        got = np.where(plan >= 0, plan, 1000 + (-plan - 1)).astype(np.float32)
        assert np.array_equal(got, want[..., 0])
    assert 100 < n_err < 2900


def test_embedding_layer_host_checks(G):
    layer = EmbeddingLayerWithFixes(nn.Embedding(10, 4))
    e1 = dict(name="x", start=10, end=12, embedding=torch.zeros(2, 4))
    layer.add_embeddings(e1)
    with pytest.raises(AssertionError):
        layer.add_embeddings(dict(name="x", start=20, end=22, embedding=torch.zeros(2, 4)))      # duplicate name
    layer = EmbeddingLayerWithFixes(nn.Embedding(10, 4), [e1])
    with pytest.raises(AssertionError):
        layer.add_embeddings(dict(name="y", start=11, end=13, embedding=torch.zeros(2, 4)))      # overlapping ids
    assert torch.equal(layer.replace_input_ids(torch.tensor([[1, 10, 11, 9]])), torch.tensor([[1, 0, 0, 9]]))
    assert layer.weight is layer.wrapped.weight
    from powerpaint_amd._lib import PPError
    with pytest.raises(PPError):                       # no CPU execution path: the table has to be on the GPU
        layer(torch.tensor([[1, 2, 3]]))


# ------------------------------------------------------------------------------------------------ GPU
def _check_layer(G, layer, dev, dtype):
    w, blocks = tables(G)
    cases = G["synthetic"] + G["cases"]
    ok = [c for c in cases if "col0" in c]
    for c in ok:
        ids = torch.tensor(c["ids"], device=dev)
        out = layer(ids)
        ids2 = ids if ids.ndim == 2 else ids[None]
        assert out.shape == (ids2.shape[0], ids2.shape[1], G["dim"]) and out.dtype == dtype and out.is_cuda
        col0 = torch.tensor(c["col0"]).reshape(out.shape[:2])
        want = torch.empty(out.shape, dtype=torch.float32)
        for b in range(out.shape[0]):
            for i in range(out.shape[1]):
                v = int(col0[b, i])
                want[b, i] = w[v] if v < 100000 else blocks[(v - 100000) // 100][(v - 100000) % 100]
        assert torch.equal(out.cpu(), want.to(dtype))                                       # bit-exact
    for c in [c for c in cases if "error" in c][:20]:
        with pytest.raises(AssertionError):
            layer(torch.tensor(c["ids"], device=dev))
    return len(ok)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_embedding_layer_matches_reference_gpu(G, dtype):
    dev = torch.device("cuda:0")
    wrapper, enc, layer, _ = build(G, dev, dtype)
    assert _check_layer(G, layer, dev, dtype) > 200
    # 1-D ids and a per-call external embedding (forward's second argument)
    ec = G["extra_case"]
    extra = dict(name=ec["extra"]["name"], start=ec["extra"]["start"], end=ec["extra"]["end"],
                 embedding=torch.full((3, G["dim"]), ec["extra"]["value"], device=dev, dtype=dtype))
    out = layer(torch.tensor(ec["ids"], device=dev), external_embeddings=extra)
    assert out.shape == (1, len(ec["ids"]), G["dim"])
    col0 = out[0, :, 0].float().cpu()
    want = torch.tensor([float(v) if v < 200000 else ec["extra"]["value"] for v in ec["col0"]]).to(dtype).float()
    assert torch.equal(col0, want)
    # the per-call embedding must not stick
    assert torch.equal(layer(torch.tensor([[1, 2, 3]], device=dev))[0].cpu(), layer.weight[[1, 2, 3]].cpu())


@pytest.mark.gpu
def test_embedding_layer_edge_cases_gpu(G):
    dev = torch.device("cuda:0")
    base = nn.Embedding(32, 24).to(dev)
    layer = EmbeddingLayerWithFixes(base)
    ids = torch.randint(0, 32, (3, 11), device=dev)
    assert torch.equal(layer(ids), base(ids))                                               # no externals: plain lookup
    with pytest.raises(IndexError):
        layer(torch.tensor([[1, 40]], device=dev))
    # mixed dtypes promote like torch.cat does (fp16 table + fp32 block -> fp32)
    half = nn.Embedding(32, 24).to(dev).half()
    layer = EmbeddingLayerWithFixes(half, dict(name="p", start=32, end=34,
                                               embedding=torch.randn(2, 24, device=dev)))
    out = layer(torch.tensor([[3, 32, 33, 5, 33]], device=dev))
    assert out.dtype == torch.float32
    want = torch.cat([half.weight[[3]].float(), layer.external_embeddings[0]["embedding"], half.weight[[5]].float(),
                      half.weight[[0]].float()])
    assert torch.equal(out[0], want)
    # odd row width (3 fp16 = 6 bytes per row: the byte-granular copy path) and a checkpoint-style in-place update
    odd = nn.Embedding(9, 3).to(dev).half()
    layer = EmbeddingLayerWithFixes(odd, dict(name="p", start=9, end=10, trainable=True,
                                              embedding=torch.ones(1, 3, device=dev).half()))
    assert torch.equal(layer(torch.tensor([2, 9], device=dev))[0], torch.cat([odd.weight[[2]], torch.ones(1, 3).to(odd.weight)]))
    with torch.no_grad():
        layer.trainable_embeddings["p"].fill_(5.0)
    assert torch.equal(layer(torch.tensor([9], device=dev))[0, 0], torch.full((3,), 5.0, device=dev).half())
    from powerpaint_amd._lib import PPError
    with pytest.raises(PPError):
        EmbeddingLayerWithFixes(base, dict(name="bad", start=32, end=35, embedding=torch.zeros(2, 24)))(ids)


@pytest.mark.gpu
def test_prompt_blend_through_clip_text_model_gpu(G):
    """promptA / promptB through a (randomly initialised, tiny) HF CLIPTextModel whose token embedding is the
    product's layer, blended as pipeline_PowerPaint.py:423 -- against the same model with the oracle's embedding
    output injected, fp32."""
    from powerpaint_amd.pipelines._base import PipelineBase
    dev = torch.device("cuda:0")
    wrapper = make_tokenizer(G)
    cfg = transformers.CLIPTextConfig(vocab_size=G["n_base"], hidden_size=32, intermediate_size=64,
                                      num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77,
                                      bos_token_id=G["n_base"] - 2, eos_token_id=G["n_base"] - 1)
    torch.manual_seed(0)
    enc = transformers.CLIPTextModel(cfg).eval()
    add_tokens(tokenizer=wrapper, text_encoder=enc, placeholder_tokens=G["placeholders"],
               initialize_tokens=["a"] * 3, num_vectors_per_token=G["num_vec"])
    emb_holder = getattr(enc, "text_model", enc).embeddings              # transformers 4.x vs 5.x layout
    layer = emb_holder.token_embedding
    assert isinstance(layer, EmbeddingLayerWithFixes)
    with torch.no_grad():
        for e in layer.external_embeddings:
            e["embedding"].copy_(torch.randn_like(e["embedding"]))
    enc.to(dev)
    pipe = PipelineBase()
    pipe.register_modules(tokenizer=wrapper, text_encoder=enc)
    pA, pB, nA, nB = add_task("a cat", "blur dog", "shape-guided")
    with torch.no_grad():
        got = pipe._encode_prompt(pA, pB, 0.3, dev, 1, True, negative_promptA=nA, negative_promptB=nB, t_nag=0.3)
    assert got.shape == (2, 77, 32)

    # the same, with the embedding computed by the oracle and fed through a plain nn.Embedding-free path
    w = layer.weight.detach().cpu().numpy()
    ext = [dict(name=e["name"], start=e["start"], end=e["end"], embedding=e["embedding"].detach().cpu().numpy())
           for e in layer.external_embeddings]

    class _Inject(nn.Module):
        def forward(self, ids):
            return torch.from_numpy(OT.embedding_with_fixes(ids.cpu().numpy(), w, ext)).to(ids.device)

    emb_holder.token_embedding = _Inject()

    def embed(p):
        ids = wrapper(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        with torch.no_grad():
            return enc(ids.to(dev))[0]

    want = torch.cat([embed(nA) * 0.3 + 0.7 * embed(nB), embed(pA) * 0.3 + 0.7 * embed(pB)])
    assert torch.allclose(got, want, atol=1e-6, rtol=0)
