"""CLIP text tower (transformers.CLIPTextModel, the `text_encoder` of the pipelines) compiled to a HIP launch plan --
SURVEY.md §8f-2.  Call sites: /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:378-423 (`self.text_encoder(ids)[0]`
for promptA / promptB and their negatives), app.py:94-117 (from_pretrained + add_tokens + load_model).

Per layer (pre-LN transformer, causal mask, quick_gelu):
    h  = x + out_proj(attn(LN1(x)))            pp_layernorm, pp_gemm_bf16 (fused q|k|v, N = 3C), pp_attention_small,
                                               pp_gemm_bf16 (+ residual in the epilogue)
    x' = h + fc2(quick_gelu(fc1(LN2(h))))      pp_layernorm, pp_gemm_bf16 (SiLU epilogue), pp_gemm_bf16 (+ residual)
quick_gelu(u) = u * sigmoid(1.702 u) = silu(1.702 u) / 1.702, so fc1 is packed as 1.702 * (W1, b1), runs with the GEMM's
SiLU epilogue, and the 1 / 1.702 is folded into fc2's weights -- no extra activation pass.
The token embedding (incl. the task-prompt splice) is `powerpaint_amd.utils.EmbeddingLayerWithFixes`.
"""
from typing import Dict, Optional

import torch

from . import _lib as L
from .engine import Arena, Builder, ParamPack, Plan, _align

QG = 1.702


class CLIPTextNet:
    def __init__(self, hidden_size: int = 768, intermediate_size: int = 3072, num_hidden_layers: int = 12,
                 num_attention_heads: int = 12, max_position_embeddings: int = 77, layer_norm_eps: float = 1e-5):
        self.C, self.F, self.n_layers, self.heads = hidden_size, intermediate_size, num_hidden_layers, num_attention_heads
        self.max_pos, self.eps = max_position_embeddings, layer_norm_eps
        if hidden_size % 64 or intermediate_size % 64 or hidden_size // num_attention_heads != 64 or \
                hidden_size % num_attention_heads:
            raise L.PPError("CLIPTextModel: hidden / intermediate sizes must be multiples of 64 and head_dim must be 64")
        if max_position_embeddings > 128:
            raise L.PPError("CLIPTextModel: at most 128 positions (pp_attention_small)")
        self.params: Optional[ParamPack] = None
        self.P: Dict[str, int] = {}

    def pack(self, sd: Dict[str, torch.Tensor], device):
        """sd: `encoder.layers.{i}.*`, `final_layer_norm.*`, `embeddings.position_embedding.weight` (no prefix)."""
        pk = ParamPack()
        bf, f32 = torch.bfloat16, torch.float32

        def W(k):
            return sd[k].detach().float()

        pk.add("pos", W("embeddings.position_embedding.weight"), bf)
        for i in range(self.n_layers):
            p = f"encoder.layers.{i}"
            for n in ("layer_norm1", "layer_norm2"):
                pk.add(f"{p}.{n}.weight", W(f"{p}.{n}.weight"), f32)
                pk.add(f"{p}.{n}.bias", W(f"{p}.{n}.bias"), f32)
            pk.add(f"{p}.qkv.weight", torch.cat([W(f"{p}.self_attn.{n}_proj.weight") for n in "qkv"], 0), bf)
            pk.add(f"{p}.qkv.bias", torch.cat([W(f"{p}.self_attn.{n}_proj.bias") for n in "qkv"], 0), f32)
            pk.add(f"{p}.out.weight", W(f"{p}.self_attn.out_proj.weight"), bf)
            pk.add(f"{p}.out.bias", W(f"{p}.self_attn.out_proj.bias"), f32)
            pk.add(f"{p}.fc1.weight", W(f"{p}.mlp.fc1.weight") * QG, bf)
            pk.add(f"{p}.fc1.bias", W(f"{p}.mlp.fc1.bias") * QG, f32)
            pk.add(f"{p}.fc2.weight", W(f"{p}.mlp.fc2.weight") / QG, bf)
            pk.add(f"{p}.fc2.bias", W(f"{p}.mlp.fc2.bias"), f32)
        pk.add("final_layer_norm.weight", W("final_layer_norm.weight"), f32)
        pk.add("final_layer_norm.bias", W("final_layer_norm.bias"), f32)
        pk.to_device(device)
        self.params, self.P = pk, pk.ptr
        return self

    def build(self, pb: Builder, x_in: int, B: int, n: int) -> int:
        """x_in: bf16 token embeddings [B*n][C]; returns the pointer of last_hidden_state [B*n][C] (bf16)."""
        P, Cc, rows = self.P, self.C, B * n
        x = pb.alloc(rows * Cc * 2)
        for b in range(B):                                   # + position embedding (rows 0..n-1 of the table)
            pb.add(x_in + b * n * Cc * 2, P["pos"], x + b * n * Cc * 2, n * Cc)
        # transformers' `hidden_states`: the embeddings and every layer's output BEFORE final_layer_norm; all of them
        # are persistent arena tensors, so `output_hidden_states=True` (clip_skip) costs nothing extra
        self.hidden = [x]
        for i in range(self.n_layers):
            p = f"encoder.layers.{i}"
            nxt = pb.alloc(rows * Cc * 2)
            mid = pb.alloc(rows * Cc * 2)
            m = pb.mark()
            h = pb.layernorm(x, rows, Cc, P[f"{p}.layer_norm1.weight"], P[f"{p}.layer_norm1.bias"], self.eps)
            qkv = pb.linear(h, rows, Cc, P[f"{p}.qkv.weight"], 3 * Cc, P[f"{p}.qkv.bias"], name="linear")
            a = pb.alloc(rows * Cc * 2)
            pb.plan.add("attention_small", pb.lib.pp_attention_small, qkv, 3 * Cc, qkv + 2 * Cc, 3 * Cc,
                        qkv + 4 * Cc, 3 * Cc, a, Cc, B, self.heads, n, n, 64, 0.125, 1, pb.dt)
            pb.plan.count("attention_small", 4.0 * B * self.heads * n * n * 64)
            pb.linear(a, rows, Cc, P[f"{p}.out.weight"], Cc, P[f"{p}.out.bias"], res1=x, out=mid, name="linear")
            h = pb.layernorm(mid, rows, Cc, P[f"{p}.layer_norm2.weight"], P[f"{p}.layer_norm2.bias"], self.eps)
            f = pb.linear(h, rows, Cc, P[f"{p}.fc1.weight"], self.F, P[f"{p}.fc1.bias"], act=L.PP_ACT_SILU, name="linear")
            pb.linear(f, rows, self.F, P[f"{p}.fc2.weight"], Cc, P[f"{p}.fc2.bias"], res1=mid, out=nxt, name="linear")
            pb.release(m)
            x = nxt
            self.hidden.append(x)
        return pb.layernorm(x, rows, Cc, P["final_layer_norm.weight"], P["final_layer_norm.bias"], self.eps)


class CLIPRuntime:
    def __init__(self, net: CLIPTextNet, device):
        self.net, self.device = net, torch.device(device)
        self.key = None
        self.plan: Optional[Plan] = None
        self.arena: Optional[Arena] = None

    def _build(self, arena: Arena, B: int, n: int):
        pb = Builder(arena)
        x_in = arena.alloc(B * n * self.net.C * 2)
        out = self.net.build(pb, x_in, B, n)
        return x_in, out, pb.plan

    def ensure(self, B: int, n: int):
        if (B, n) == self.key:
            return
        dry = Arena()
        self._build(dry, B, n)
        self.arena = Arena(_align(dry.peak, 4096), self.device)
        self.x_in, self.out, self.plan = self._build(self.arena, B, n)
        self.hidden = list(self.net.hidden)
        self.key = (B, n)

    def hidden_states(self):
        """transformers' `hidden_states` of the last `run` (embeddings + one tensor per layer, pre final_layer_norm), as
        copies (the arena is overwritten by the next call)."""
        B, n = self.key
        return tuple(self.arena.view(p, (B, n, self.net.C), torch.bfloat16).clone() for p in self.hidden)

    def run(self, tok_emb: torch.Tensor) -> torch.Tensor:
        """tok_emb [B, n, C] (any float dtype, on the device) -> last_hidden_state [B, n, C] bf16 (arena view)."""
        B, n, Cc = tok_emb.shape
        self.ensure(B, n)
        self.arena.view(self.x_in, (B, n, Cc), torch.bfloat16).copy_(tok_emb)
        self.plan.run(torch.cuda.current_stream().cuda_stream)
        return self.arena.view(self.out, (B, n, Cc), torch.bfloat16)
