#!/usr/bin/env python
"""Fused cross-attention block (pp_xattn_block) against the three-launch chain it replaces, at the UNet's 64x64 level
(B = 8 -> M = 32768) and config 5's 128x128 level (B = 4 -> M = 65536): us per sub-block, hot and behind a 1 GiB memset."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402


def timeit(fn, cold=False, reps=20):
    big = torch.empty(1 << 28, device="cuda") if cold else None
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        if cold:
            big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    C, heads, nctx, d = 320, 8, 77, 40
    for B, hw in [(8, 4096), (4, 16384)]:
        M = B * hw
        dt = torch.bfloat16
        h = torch.randn(M, C, device="cuda").to(dt)
        ctx = torch.randn(B * nctx, 768, device="cuda").to(dt)
        wkv = (torch.randn(2 * C, 768, device="cuda") * 768 ** -0.5).to(dt)
        wq = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt)
        wo = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt)
        bo, cs, tq = torch.randn(C, device="cuda"), wq.float().sum(1).contiguous(), torch.randn(C, device="cuda")
        hf = h.float()
        st = torch.stack([hf.reshape(M, 2, 160).sum(-1), (hf * hf).reshape(M, 2, 160).sum(-1)], -1).contiguous()
        k, vt = ops.gemm(ctx, wkv, vt_col0=C, rows_per_batch=nctx)
        vtp = torch.zeros(B, C, 80, dtype=dt, device="cuda")
        vtp[:, :, :nctx] = vt
        folded = ops.xattn_fold(k, vtp, B, nctx, heads, wq, wo, q_colsum=cs, q_bias=tq)

        def chain():
            q = ops.gemm(h, wq, ln_stats=st, ln_colsum=cs, ln_dim=C, bias=tq)
            ao = ops.attention(q, k, vtp, B, heads, hw, nctx, d)
            return ops.gemm(ao, wo, bias=bo, res1=h, row_stats=True)

        def fused():
            return ops.xattn_block(h, folded, bias_o=bo, res=h, ln_stats=st, rows_per_batch=hw, row_stats=True)

        def fold():
            return ops.xattn_fold(k, vtp, B, nctx, heads, wq, wo, q_colsum=cs, q_bias=tq)

        for cold in (False, True):
            print(f"M={M:6d} {'cold' if cold else 'hot '}: chain {timeit(chain, cold):7.1f} us   fused {timeit(fused, cold):7.1f} us"
                  f"   (fold, once per prompt: {timeit(fold, cold):6.1f} us)", flush=True)


if __name__ == "__main__":
    main()
