#!/usr/bin/env python
"""Fused feed-forward (pp_ff_fused) against the two launches it replaces, hot, at the UNet's 64x64 level (M = 32768 rows) and
at config 5's 128x128 level.  usage: python tools/ff_one.py [M ...]   (GPU box; lab switches of the library apply)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L, ops
from powerpaint_amd.engine import _geglu_interleave, _kperm_geglu

C = 320
dev = "cuda"


def main():
    only = None
    if "--only" in sys.argv:                     # (rocprofv3 --pmc passes: run ONE of fused / fused4 / chain)
        i = sys.argv.index("--only")
        only = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    Ms = [int(a) for a in sys.argv[1:]] or [32768, 131072]
    for M in Ms:
        g = torch.Generator("cpu").manual_seed(0)
        r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
        hs = r(M, C).to(torch.bfloat16)
        w1 = _geglu_interleave(r(8 * C, C, sc=C ** -0.5)).to(torch.bfloat16).contiguous()
        b1 = r(8 * C, sc=0.1)
        cs1 = w1.float().sum(1).contiguous()
        hf = hs.float()
        st = torch.stack([hf.reshape(M, 2, 160).sum(-1), (hf * hf).reshape(M, 2, 160).sum(-1)], -1).contiguous()
        w2 = r(C, 5 * C, sc=(5 * C) ** -0.5).to(torch.bfloat16).contiguous()
        w2kp = torch.cat([_kperm_geglu(w2[:, :4 * C]), w2[:, 4 * C:]], 1).contiguous()
        b2 = r(C, sc=0.1)
        res = r(M, C).to(torch.bfloat16)
        acc = torch.zeros(M // 4096 if M >= 4096 else 1, 32, 2, dtype=torch.int64, device=dev)
        rpb = 4096 if M >= 4096 else M
        gn = [(acc, 10, 0, 32)]

        def fused():      # the 8-wave kernel (W2' natural order)
            return ops.ff_fused(hs, w1, b1, w2, b2, cs1=cs1, ln_stats=st, res1=res, rows_per_batch=rpb, gn=gn, w2_kperm=False)

        def fused4():     # the 4-wave kernel (hidden index permuted)
            return ops.ff_fused(hs, w1, b1, w2kp, b2, cs1=cs1, ln_stats=st, res1=res, rows_per_batch=rpb, gn=gn, w2_kperm=True)

        def chain():
            gg = ops.gemm(hs, w1, bias=b1, act=L.PP_ACT_GEGLU, ln_stats=st, ln_colsum=cs1, ln_dim=C)
            return ops.gemm(gg, w2, bias=b2, x2=hs, res1=res, rows_per_batch=rpb, gn=gn)

        if only:
            fn = {"fused": fused, "fused4": fused4, "chain": chain}[only]
            for _ in range(6):
                fn()
            torch.cuda.synchronize()
            continue
        a, b = fused(), chain()
        d = (a.float() - b.float()).abs()
        print(f"M={M}: max |fused - chain| {float(d.max()):.4g}, differing {float((d > 0).float().mean()):.4f}")
        d4 = (fused4().float() - b.float()).abs()
        print(f"M={M}: max |fused4 - chain| {float(d4.max()):.4g}, differing {float((d4 > 0).float().mean()):.4f}")
        for name, fn in (("fused", fused), ("fused4", fused4), ("chain", chain), ("fused", fused), ("fused4", fused4), ("chain", chain)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            fl = 2.0 * M * C * (8 * C + 5 * C)
            print(f"  {name}: {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
