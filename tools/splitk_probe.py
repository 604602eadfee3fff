#!/usr/bin/env python
"""Cost of the split-K combine: the same GEMM shape with a tiny K (the MFMA part is negligible) timed with
splitk = 1 / 2 / 4 / 8 -- the difference is partial-slab write + combine kernel.    python tools/splitk_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402
from epi_probe import time_launch  # noqa: E402


def main():
    lib = L.lib()
    dev = "cuda"
    for M, N in ((512, 1280), (2048, 1280), (8192, 640)):
        for K in (1024, 11520):
            x = torch.randn(M, K, device=dev).bfloat16()
            w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
            bias = torch.randn(N, device=dev)
            res = torch.randn(M, N, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            ws = torch.empty(8 * M * N, device=dev)
            row = []
            for sk in (1, 2, 4, 8):
                a = L.PPGemmArgs()
                a.M, a.N, a.K, a.x_mode = M, N, K, L.PP_X_PLAIN
                a.x1, a.c1, a.ldx1 = x.data_ptr(), K, K
                a.w, a.bias = w.data_ptr(), bias.data_ptr()
                a.res1, a.ldres1, a.ldres2 = res.data_ptr(), N, N
                a.scale, a.act = 1.0, 0
                a.out, a.ldo = out.data_ptr(), N
                a.tile, a.splitk = 21, sk
                a.workspace = ws.data_ptr()
                t = time_launch(lib, a)
                row.append(f"{t:7.1f}" if t else "    err")
            print(f"M={M:5d} N={N:5d} K={K:6d} tile 21   splitk 1/2/4/8 us: {' '.join(row)}", flush=True)


if __name__ == "__main__":
    main()
