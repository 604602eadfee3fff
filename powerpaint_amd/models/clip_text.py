"""CLIPTextModel on the MI355X HIP path.

Drop-in for `transformers.CLIPTextModel` as the pipelines and app.py use it
(/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:378-423: `self.text_encoder(ids.to(device))[0]`;
app.py:94-117: `add_tokens(tokenizer, text_encoder, ...)` then `load_model(text_encoder, ".../text_encoder/model.safetensors")`).

It IS an nn.Module whose parameter tree has transformers' names (4.x layout: `text_model.embeddings.*`,
`text_model.encoder.layers.{i}.{self_attn.{q,k,v,out}_proj, layer_norm1, mlp.fc1, mlp.fc2, layer_norm2}`,
`text_model.final_layer_norm`), so `state_dict()`, `load_state_dict()`, `safetensors.torch.load_model`, `.to()` and
`powerpaint_amd.utils.add_tokens` work on it unchanged; the sub-modules are parameter containers only.  `forward`
repacks the parameters into the kernels' layout when they changed and replays `powerpaint_amd.clip.CLIPTextNet`'s
launch plan.  The token embedding is always an `EmbeddingLayerWithFixes` (key `...token_embedding.wrapped.weight`, the
name the PowerPaint checkpoints use; the stock `...token_embedding.weight` is mapped on load).
"""
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib as L
from ..clip import CLIPRuntime, CLIPTextNet
from ..utils.utils import EmbeddingLayerWithFixes
from ._base import Output
from ..loaders import PretrainedMixin


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(c, c) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, c, f):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(c, f), nn.Linear(f, c)


class _Layer(nn.Module):
    def __init__(self, c, f, eps):
        super().__init__()
        self.self_attn = _Attn(c)
        self.layer_norm1 = nn.LayerNorm(c, eps=eps)
        self.mlp = _MLP(c, f)
        self.layer_norm2 = nn.LayerNorm(c, eps=eps)


class _Encoder(nn.Module):
    def __init__(self, n, c, f, eps):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c, f, eps) for _ in range(n)])


class _Embeddings(nn.Module):
    def __init__(self, vocab, npos, c):
        super().__init__()
        self.token_embedding = EmbeddingLayerWithFixes(nn.Embedding(vocab, c))
        self.position_embedding = nn.Embedding(npos, c)


class _FinalLayerNorm(nn.LayerNorm):
    """`text_encoder.text_model.final_layer_norm(hidden)` is called by `encode_prompt(clip_skip=...)`
    (/root/reference/powerpaint/pipelines/pipeline_PowerPaint_Brushnet_CA.py:547-552): same parameters and state-dict keys
    as nn.LayerNorm, the arithmetic is the HIP `pp_layernorm` kernel for 16-bit device tensors."""

    def forward(self, x):
        if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16):
            from .. import ops
            x2 = x.reshape(-1, x.shape[-1]).contiguous()
            return ops.layernorm(x2, self.weight.float(), self.bias.float(), self.eps).view(x.shape)
        return super().forward(x)


class _TextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg.vocab_size, cfg.max_position_embeddings, cfg.hidden_size)
        self.encoder = _Encoder(cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps)
        self.final_layer_norm = _FinalLayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPTextModel(nn.Module, PretrainedMixin):
    def __init__(self, config=None, device="cuda", **kw):
        super().__init__()
        d = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                 pad_token_id=1, bos_token_id=49406, eos_token_id=49407, projection_dim=768)
        if config is not None:
            d.update({k: getattr(config, k) for k in d if hasattr(config, k)})
        d.update(kw)
        self.config = SimpleNamespace(**d)
        if self.config.hidden_act != "quick_gelu":
            raise L.PPError("CLIPTextModel: only hidden_act='quick_gelu' (the SD-1.5 text encoder) is built")
        self.text_model = _TextTransformer(self.config)
        self.net = CLIPTextNet(self.config.hidden_size, self.config.intermediate_size, self.config.num_hidden_layers,
                               self.config.num_attention_heads, self.config.max_position_embeddings,
                               self.config.layer_norm_eps)
        self._rt: Optional[CLIPRuntime] = None
        self._stamp = None
        self.requires_grad_(False)
        self.to(device)

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    # ---- checkpoints: transformers 4.x names, 5.x names (no `text_model.` prefix), stock or wrapped token embedding
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("position_ids"):
                continue
            if not k.startswith("text_model."):
                k = "text_model." + k
            if k == "text_model.embeddings.token_embedding.weight":
                k = "text_model.embeddings.token_embedding.wrapped.weight"
            sd[k] = v
        return super().load_state_dict(sd, strict=strict, assign=assign)

    def _params_stamp(self):
        return tuple((p.data_ptr(), p._version) for n, p in self.named_parameters() if ".token_embedding." not in n)

    def _ensure_packed(self):
        dev = self.device
        if dev.type != "cuda":
            raise L.PPError("CLIPTextModel: parameters must live on the GPU (no CPU path)")
        stamp = self._params_stamp()
        if stamp != self._stamp or self._rt is None or self._rt.device != dev:
            sd = {n[len("text_model."):]: p for n, p in self.named_parameters() if ".token_embedding." not in n}
            self.net.pack(sd, dev)
            self._rt = CLIPRuntime(self.net, dev)
            self._stamp = stamp

    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask=None, position_ids=None,
                output_attentions=None, output_hidden_states=None, return_dict: Optional[bool] = True, **unused):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is not None and not bool(torch.all(attention_mask == 1)):
            raise NotImplementedError("attention_mask with padding holes is outside the PowerPaint path (the pipelines "
                                      "pass none: pipeline_PowerPaint.py:400-409)")
        if position_ids is not None or output_attentions:
            raise NotImplementedError("position_ids / output_attentions are not used by the pipelines")
        ids = input_ids.reshape(-1, input_ids.shape[-1])
        if ids.shape[1] > self.config.max_position_embeddings:
            raise ValueError(f"Sequence length must be less than max_position_embeddings (got `sequence length`: "
                             f"{ids.shape[1]} and max_position_embeddings: {self.config.max_position_embeddings}")
        self._ensure_packed()
        tok = self.text_model.embeddings.token_embedding(ids.to(self.device))          # [B, n, C], splice included
        last = self._rt.run(tok).to(self.dtype)
        # pooled output as transformers computes it (unused by the pipelines): the EOS position
        if self.config.eos_token_id == 2:
            pos = ids.to(torch.int).argmax(dim=-1)
        else:
            pos = (ids.to(torch.int) == self.config.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(last.shape[0], device=last.device), pos.to(last.device)]
        if output_hidden_states:
            # (embeddings, layer 1, ..., layer N) before final_layer_norm, as transformers returns them:
            # `encode_prompt(clip_skip=k)` takes [-(k + 1)] and applies text_model.final_layer_norm itself
            hs = tuple(h.to(self.dtype) for h in self._rt.hidden_states())
            if not return_dict:
                return (last, pooled, hs)
            return Output(last_hidden_state=last, pooler_output=pooled, hidden_states=hs)
        if not return_dict:
            return (last, pooled)
        return Output(last_hidden_state=last, pooler_output=pooled)
