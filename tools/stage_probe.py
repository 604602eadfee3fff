#!/usr/bin/env python
"""Pipeline depth probe for the small-M plain GEMMs (64x160 tiles, one 4-wave workgroup per CU): tile codes 32 / 42 / 62 =
3 / 4 / 5 LDS-DMA stages, hot (graph of 10 back-to-back launches) and behind a 256 MB memset (cold L2 / MALL).
    python tools/stage_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402

SHAPES = [(2048, 1280, 1280, 0), (2048, 1280, 5120, 1280), (8192, 640, 640, 0), (8192, 640, 2560, 640), (512, 1280, 1280, 0),
          (2048, 3840, 1280, 0)]


def hot(fn, rep=10, iters=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep):
            fn()
    best = 1e9
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
    return best


BIG = None


def cold(fn, reps=7):
    global BIG
    if BIG is None:
        BIG = torch.empty(1 << 26, device="cuda")
    fn()
    ts = []
    for _ in range(reps):
        BIG.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev, dt = "cuda", torch.bfloat16
    for (M, N, K, K2) in SHAPES:
        x = torch.randn(M, K, device=dev).to(dt)
        x2 = torch.randn(M, K2, device=dev).to(dt) if K2 else None
        w = (torch.randn(N, K + K2, device=dev) * (K + K2) ** -0.5).to(dt)
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev).to(dt)
        line = f"M={M:5d} N={N:5d} K={K + K2:5d}:"
        for tile in (0, 32, 42, 62, 31, 54):
            try:
                f = lambda: ops.gemm(x, w, b, x2=x2, res1=res, tile=tile)  # noqa: E731
                line += f"  t{tile} hot {hot(f):6.1f} cold {cold(f):6.1f}"
            except Exception as e:
                line += f"  t{tile} ERR {str(e)[:30]}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
