"""CPU, fp64: the algebra behind the two "folded" operators of the C ABI, stated once in plain torch and checked against the
unfused computation they replace (the GPU tests check the kernels against the same references numerically).

* pp_xattn_fold / pp_xattn_block (include/pp_hip.h): norm2 -> to_q -> softmax(q K^T / sqrt d) V -> to_out + residual with
  K and V contracted into the projections, the folded-LayerNorm terms carried as a logit column sum / bias, padded keys
  masked by a -inf logit bias, and H stored with the contraction index permuted inside every group of 32.
"""
import math

import torch
import torch.nn.functional as F

F64 = torch.float64


def gen(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)


def test_cross_attention_fold_is_the_attention_block():
    B, hw, C, heads, nctx, KP = 2, 24, 320, 8, 77, 80
    d = C // heads
    h = gen(B, hw, C, seed=1) * 1.5 + 0.3
    g2, b2 = gen(C, seed=2) * 0.3 + 1, gen(C, seed=3) * 0.2
    wq, wo, bo = gen(C, C, seed=4) / math.sqrt(C), gen(C, C, seed=5) / math.sqrt(C), gen(C, seed=6) * 0.1
    K, V = gen(B, nctx, C, seed=7), gen(B, nctx, C, seed=8)
    # the block as the reference computes it
    q = F.layer_norm(h, (C,), g2, b2, 1e-5) @ wq.t()
    qh, kh, vh = (t.reshape(B, -1, heads, d).transpose(1, 2) for t in (q, K, V))
    o = (torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), -1) @ vh).transpose(1, 2).reshape(B, hw, C)
    ref = o @ wo.t() + bo + h
    # folded form.  LayerNorm fold of to_q (include/pp_hip.h, PPGemmArgs.ln_*): W' = gamma . W, colsum, bias W beta
    wqf, cs, tq = wq * g2[None, :], (wq * g2[None, :]).sum(1), wq @ b2
    scale = d ** -0.5 * math.log2(math.e)                                      # softmax in the exp2 domain
    Kp = torch.zeros(B, KP, heads, d, dtype=F64)
    Kp[:, :nctx] = K.reshape(B, nctx, heads, d)
    Vp = torch.zeros(B, KP, heads, d, dtype=F64)
    Vp[:, :nctx] = V.reshape(B, nctx, heads, d)
    Gt = scale * torch.einsum("bkhd,hdc->bhkc", Kp, wqf.reshape(heads, d, C)).reshape(B, heads * KP, C)
    gcs = scale * torch.einsum("bkhd,hd->bhk", Kp, cs.reshape(heads, d)).reshape(B, heads * KP)
    gb = scale * torch.einsum("bkhd,hd->bhk", Kp, tq.reshape(heads, d))
    gb[:, :, nctx:] = -math.inf                                                # padded keys
    gb = gb.reshape(B, heads * KP)
    Ht = torch.einsum("nhd,bkhd->bnhk", wo.reshape(C, heads, d), Vp).reshape(B, C, heads * KP)
    # storage order of H's contraction index: position 8 kg + j of a 32-group holds index 16 (j >> 2) + 4 kg + (j & 3)
    kp = torch.arange(heads * KP)
    kk = 32 * (kp // 32) + 16 * ((kp % 8) // 4) + 4 * ((kp // 8) % 4) + (kp % 4)
    assert sorted(kk.tolist()) == list(range(heads * KP))                      # a permutation
    Ht_stored = Ht[:, :, kk]
    mean = h.mean(-1, keepdim=True)
    rstd = torch.rsqrt(h.var(-1, unbiased=False, keepdim=True) + 1e-5)
    logits = rstd * (h @ Gt.transpose(1, 2) - mean * gcs[:, None, :]) + gb[:, None, :]
    p = torch.exp2(logits.reshape(B, hw, heads, KP) - logits.reshape(B, hw, heads, KP).amax(-1, keepdim=True))
    p = (p / p.sum(-1, keepdim=True)).reshape(B, hw, heads * KP)
    assert torch.all(p.reshape(B, hw, heads, KP)[..., nctx:] == 0)
    # the kernel's second GEMM contracts P (in accumulator-register order = the same permutation) with H as stored
    out = p[:, :, kk] @ Ht_stored.transpose(1, 2) + bo + h
    assert torch.allclose(out, ref, atol=1e-10, rtol=1e-10), (out - ref).abs().max()
