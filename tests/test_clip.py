"""SURVEY.md §8f-2 -- the CLIP text tower (`text_encoder`) on the HIP path.

The reference's text encoder IS `transformers.CLIPTextModel` (pipeline_PowerPaint.py:23,189), and transformers is
installed in this image, so the pin is the real class: same weights -> same last_hidden_state.  CPU: checkpoint key
compatibility (4.x / 5.x / PowerPaint-wrapped), add_tokens on the HIP module, plan shape, loud failure without a GPU.
GPU: the small causal attention kernel, the GEMM SiLU epilogue quick_gelu rides on, the tower against transformers,
and prompt -> embeds through the pipeline helper with task tokens.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from powerpaint_amd import _lib as L
from powerpaint_amd.engine import Arena, Builder
from powerpaint_amd.models import CLIPTextModel
from powerpaint_amd.utils import EmbeddingLayerWithFixes, TokenizerWrapper, add_task, add_tokens

transformers = pytest.importorskip("transformers")
HERE = os.path.dirname(os.path.abspath(__file__))


def hf_model(vocab=49408, layers=12, seed=0):
    cfg = transformers.CLIPTextConfig(vocab_size=vocab, hidden_size=768, intermediate_size=3072,
                                      num_hidden_layers=layers, num_attention_heads=12, max_position_embeddings=77,
                                      hidden_act="quick_gelu", bos_token_id=vocab - 2, eos_token_id=vocab - 1,
                                      pad_token_id=vocab - 1)
    torch.manual_seed(seed)
    m = transformers.CLIPTextModel(cfg).eval()
    with torch.no_grad():                       # the default init is tiny (std 0.02): give biases / norms real values
        for n, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.05)
            elif "embedding" not in n:
                p.mul_(2.0)
            p.copy_(p.to(torch.bfloat16).float() if p.dim() >= 2 else p)
    return m


def tiny_tokenizer():
    with open(os.path.join(HERE, "golden", "ref_task_tokens.json")) as f:
        G = json.load(f)
    tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(G["vocab"])},
                                     merges=[tuple(m) for m in G["merges"]], model_max_length=77)
    return TokenizerWrapper(tokenizer=tok), G


# ------------------------------------------------------------------------------------------------ CPU
def test_checkpoint_key_compatibility():
    hf = hf_model(vocab=600, layers=2)
    m = CLIPTextModel(device="cpu", vocab_size=600, num_hidden_layers=2)
    ours = set(m.state_dict())
    want = {("text_model." + k if not k.startswith("text_model.") else k) for k in hf.state_dict()
            if not k.endswith("position_ids")}
    want = {k.replace("token_embedding.weight", "token_embedding.wrapped.weight") for k in want}
    assert ours == want                                                     # transformers 4.x names, wrapped table
    r = m.load_state_dict(hf.state_dict())                                  # as transformers (5.x here) names them
    assert not r.missing_keys and not r.unexpected_keys
    assert torch.equal(m.text_model.embeddings.token_embedding.weight, hf.get_input_embeddings().weight)
    r = m.load_state_dict({"text_model." + k.replace("text_model.", ""): v for k, v in hf.state_dict().items()})
    assert not r.missing_keys and not r.unexpected_keys
    # the PowerPaint text_encoder checkpoint layout: add_tokens first, then wrapped + trainable_embeddings keys
    wrapper, G = tiny_tokenizer()
    m2 = CLIPTextModel(device="cpu", vocab_size=G["n_base"], num_hidden_layers=1)
    layer = m2.text_model.embeddings.token_embedding
    add_tokens(tokenizer=wrapper, text_encoder=m2, placeholder_tokens=G["placeholders"],
               initialize_tokens=["a"] * 3, num_vectors_per_token=G["num_vec"])
    assert m2.text_model.embeddings.token_embedding is layer and isinstance(layer, EmbeddingLayerWithFixes)
    keys = set(m2.state_dict())
    for p in G["placeholders"]:
        assert f"text_model.embeddings.token_embedding.trainable_embeddings.{p}" in keys
    assert "text_model.embeddings.token_embedding.wrapped.weight" in keys
    sd = {k: torch.randn_like(v) for k, v in m2.state_dict().items()}
    m2.load_state_dict(sd)
    assert torch.equal(layer.external_embeddings[2]["embedding"], sd[
        "text_model.embeddings.token_embedding.trainable_embeddings.P_obj"])   # the dict entry IS the parameter


def test_plan_shape_and_loud_failure_on_cpu():
    m = CLIPTextModel(device="cpu", vocab_size=100, num_hidden_layers=3)
    sd = {n[len("text_model."):]: p for n, p in m.named_parameters() if ".token_embedding." not in n}
    m.net.pack(sd, "cpu")
    dry = Arena()
    pb = Builder(dry)
    m.net.build(pb, dry.alloc(2 * 77 * 768 * 2), 2, 77)
    names = [c[2] for c in pb.plan.calls]
    assert names.count("linear") == 12 and names.count("attention_small") == 3 and names.count("layernorm") == 7
    assert names.count("add") == 2
    with pytest.raises(L.PPError):
        m(torch.zeros(1, 77, dtype=torch.long))
    with pytest.raises(L.PPError):
        CLIPTextModel(device="cpu", hidden_act="gelu")
    with pytest.raises(L.PPError):
        CLIPTextModel(device="cpu", hidden_size=1024, num_attention_heads=8)      # head_dim 128


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("B,H,n,causal", [(4, 12, 77, True), (2, 3, 128, False), (1, 2, 5, True), (3, 12, 77, False)])
def test_attention_small_gpu(B, H, n, causal):
    from powerpaint_amd import ops
    torch.manual_seed(B * n)
    qkv = (torch.randn(B * n, 3 * H * 64, device="cuda") * 1.5).bfloat16()
    q, k, v = qkv[:, :H * 64], qkv[:, H * 64:2 * H * 64], qkv[:, 2 * H * 64:]
    o = ops.attention_small(q, k, v, B, H, n, n, causal=causal)

    def heads(t):
        return t.float().view(B, n, H, 64).transpose(1, 2)

    want = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), is_causal=causal)
    want = want.transpose(1, 2).reshape(B * n, H * 64)
    assert torch.allclose(o.float(), want, atol=2e-2, rtol=2e-2)
    with pytest.raises(L.PPError):
        L.check(L.lib().pp_attention_small(q.data_ptr(), 1, k.data_ptr(), 2, v.data_ptr(), 2, o.data_ptr(), 1, 1, 1, 200,
                                           200, 64, 0.125, 0, L.PP_DT_BF16, torch.cuda.current_stream().cuda_stream),
                "too long")


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,splitk", [(308, 3072, 768, 0), (77, 3072, 768, 0), (308, 768, 3072, 0), (308, 768, 3072, 2)])
def test_gemm_silu_epilogue_gpu(M, N, K, splitk):
    """quick_gelu(u) = silu(1.702 u) / 1.702 rides on the GEMM's SiLU epilogue (powerpaint_amd/clip.py)."""
    from powerpaint_amd import ops
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    b = torch.randn(N, device="cuda")
    y = ops.gemm(x, w, b, act=L.PP_ACT_SILU, splitk=splitk)
    want = F.silu(x.float() @ w.float().t() + b)
    assert torch.allclose(y.float(), want, atol=2e-2, rtol=1e-2)
    qg = ops.gemm(x, (w.float() * 1.702).bfloat16(), b * 1.702, act=L.PP_ACT_SILU, splitk=splitk).float() / 1.702
    u = x.float() @ w.float().t() + b
    assert torch.allclose(qg, u * torch.sigmoid(1.702 * u), atol=3e-2, rtol=2e-2)


@pytest.mark.gpu
def test_clip_text_model_matches_transformers_gpu():
    hf = hf_model()
    m = CLIPTextModel(device="cuda")
    m.load_state_dict(hf.state_dict())
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 49406, (4, 77), generator=g)
    ids[:, 0] = 49406
    ids[:, 40:] = 49407
    with torch.no_grad():
        want = hf(ids)[0]
    out = m(ids.cuda())
    got = out[0]
    assert got.shape == (4, 77, 768) and got.dtype == torch.float32 and out.last_hidden_state is got
    cos = F.cosine_similarity(got.cpu().flatten(), want.flatten(), dim=0).item()
    err = (got.cpu() - want).abs().max().item()
    assert cos >= 0.9995 and err <= 0.03 * max(1.0, want.abs().max().item()), (cos, err)
    # causal: a token's state does not depend on later tokens
    ids2 = ids.clone()
    ids2[:, 30:] = 5
    got2 = m(ids2.cuda(), return_dict=False)[0]
    assert torch.equal(got2[:, :30], got[:, :30]) and not torch.equal(got2[:, 30:], got[:, 30:])
    # a weight update is picked up (repack on version change); shorter sequences build their own plan
    with torch.no_grad():
        m.text_model.final_layer_norm.bias.add_(1.0)
    assert torch.allclose(m(ids.cuda())[0], got + 1.0, atol=2e-2)      # the output is rounded to bf16
    assert m(ids[:1, :16].cuda())[0].shape == (1, 16, 768)
    assert out.pooler_output.shape == (4, 768)


@pytest.mark.gpu
def test_clip_skip_hidden_states_match_transformers_gpu():
    """`encode_prompt(clip_skip=k)` (pipeline_PowerPaint_Brushnet_CA.py:537-552): `output_hidden_states=True` returns the
    embeddings + every layer's pre-final-LayerNorm state as transformers does, and the pipeline helper takes
    [-(k + 1)] through `text_model.final_layer_norm` -- against the same recipe on transformers.CLIPTextModel."""
    from powerpaint_amd.pipelines import StableDiffusionPowerPaintBrushNetPipeline
    wrapper, G = tiny_tokenizer()
    hf = hf_model(vocab=G["n_base"], layers=4, seed=5)
    m = CLIPTextModel(device="cuda", vocab_size=G["n_base"], num_hidden_layers=4, eos_token_id=G["n_base"] - 1)
    m.load_state_dict(hf.state_dict())
    ids = wrapper(["a cat on a mat", "dog"], padding="max_length", max_length=77, truncation=True,
                  return_tensors="pt").input_ids
    with torch.no_grad():
        ref = hf(ids, output_hidden_states=True)
    out = m(ids.cuda(), output_hidden_states=True)
    assert len(out.hidden_states) == len(ref.hidden_states) == 5
    assert m(ids.cuda(), output_hidden_states=True, return_dict=False)[-1][2].shape == (2, 77, 768)
    for i, (a, b) in enumerate(zip(out.hidden_states, ref.hidden_states)):
        cos = F.cosine_similarity(a.float().cpu().flatten(), b.flatten(), dim=0).item()
        assert cos >= 0.9995 and (a.float().cpu() - b).abs().max().item() <= 0.03 * max(1.0, b.abs().max().item()), (i, cos)
    pipe = StableDiffusionPowerPaintBrushNetPipeline(text_encoder=m, tokenizer=wrapper)
    hf_tm = getattr(hf, "text_model", hf)
    for k in (1, 2):
        got = pipe.encode_prompt(["a cat on a mat", "dog"], torch.device("cuda"), 1, False, clip_skip=k)
        with torch.no_grad():
            want = hf_tm.final_layer_norm(ref.hidden_states[-(k + 1)])
        cos = F.cosine_similarity(got.float().cpu().flatten(), want.flatten(), dim=0).item()
        assert got.shape == (2, 77, 768) and cos >= 0.9995, (k, cos)
        assert (got.float().cpu() - want).abs().max().item() <= 0.03 * max(1.0, want.abs().max().item())
    plain = pipe.encode_prompt(["a cat on a mat", "dog"], torch.device("cuda"), 1, False)
    assert not torch.allclose(plain.float(), got.float(), atol=1e-2)         # clip_skip changes the embedding


@pytest.mark.gpu
def test_task_prompts_to_embeds_through_pipeline_helper_gpu():
    """add_task -> TokenizerWrapper -> HIP CLIPTextModel (spliced task tokens) -> blended prompt_embeds, against
    transformers' tower fed with the reference-rule embedding (oracle/task_tokens.py)."""
    from oracle import task_tokens as OT
    from powerpaint_amd.pipelines._base import PipelineBase
    wrapper, G = tiny_tokenizer()
    hf = hf_model(vocab=G["n_base"], layers=2, seed=3)
    m = CLIPTextModel(device="cuda", vocab_size=G["n_base"], num_hidden_layers=2, eos_token_id=G["n_base"] - 1)
    m.load_state_dict(hf.state_dict())
    add_tokens(tokenizer=wrapper, text_encoder=m, placeholder_tokens=G["placeholders"], initialize_tokens=["a"] * 3,
               num_vectors_per_token=G["num_vec"])
    layer = m.text_model.embeddings.token_embedding
    with torch.no_grad():
        for e in layer.external_embeddings:
            e["embedding"].copy_(torch.randn_like(e["embedding"]) * 0.05)
    pipe = PipelineBase()
    pipe.register_modules(tokenizer=wrapper, text_encoder=m)
    pA, pB, nA, nB = add_task("a cat", "blur dog", "shape-guided")
    got = pipe._encode_prompt(pA, pB, 0.3, torch.device("cuda"), 1, True, negative_promptA=nA, negative_promptB=nB,
                              t_nag=0.3)
    assert got.shape == (2, 77, 768)
    w = layer.weight.detach().cpu().numpy()
    ext = [dict(name=e["name"], start=e["start"], end=e["end"], embedding=e["embedding"].detach().cpu().numpy())
           for e in layer.external_embeddings]

    def embed(p):
        ids = wrapper(p, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        emb = torch.from_numpy(OT.embedding_with_fixes(ids.numpy(), w, ext))
        with torch.no_grad():
            return hf(inputs_embeds=emb)[0] if _accepts_embeds(hf) else _hf_from_embeds(hf, emb)

    want = torch.cat([embed(nA) * 0.3 + 0.7 * embed(nB), embed(pA) * 0.3 + 0.7 * embed(pB)])
    cos = F.cosine_similarity(got.cpu().flatten(), want.flatten(), dim=0).item()
    assert cos >= 0.9995 and (got.cpu() - want).abs().max().item() <= 0.03 * max(1.0, want.abs().max().item())


def _accepts_embeds(hf):
    import inspect
    return "inputs_embeds" in inspect.signature(hf.forward).parameters


def _hf_from_embeds(hf, emb):
    """transformers versions whose CLIPTextModel.forward has no inputs_embeds: swap the token embedding for one call."""
    holder = getattr(hf, "text_model", hf).embeddings
    old = holder.token_embedding

    class _Inject(torch.nn.Module):
        def forward(self, ids):
            return emb

    holder.token_embedding = _Inject()
    try:
        return hf(torch.zeros(emb.shape[:2], dtype=torch.long))[0]
    finally:
        holder.token_embedding = old
