#!/usr/bin/env python
"""Tile / split-K probe of the transformer GEMMs of the 16x16 and 32x32 levels (round 5): every (tile, split-K) form
against the automatic choice, HOT (a hipGraph of 10 back-to-back launches) and COLD IN A GRAPH (10 x [256 MB memset +
launch] minus 10 x [memset]: cold L2 / MALL as inside the step, no eager launch gaps).
    python tools/tile_probe.py [shape-name ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L, ops  # noqa: E402

#          name        M     N      K1    K2    act               res   row_stats
SHAPES = [("ff2_16", 2048, 1280, 5120, 1280, L.PP_ACT_NONE, True, False),
          ("ff2_32", 8192, 640, 2560, 640, L.PP_ACT_NONE, True, False),
          ("geglu_16", 2048, 10240, 1280, 0, L.PP_ACT_GEGLU, False, False),
          ("geglu_32", 8192, 5120, 640, 0, L.PP_ACT_GEGLU, False, False),
          ("qkv_16", 2048, 3840, 1280, 0, L.PP_ACT_NONE, False, False),
          ("qkv_32", 8192, 1920, 640, 0, L.PP_ACT_NONE, False, False),
          ("out_16", 2048, 1280, 1280, 0, L.PP_ACT_NONE, True, True),
          ("out_32", 8192, 640, 640, 0, L.PP_ACT_NONE, True, True),
          ("out_8", 512, 1280, 1280, 0, L.PP_ACT_NONE, True, True),
          ("qkv_8", 512, 3840, 1280, 0, L.PP_ACT_NONE, False, False),
          ("q_16", 2048, 1280, 1280, 0, L.PP_ACT_NONE, False, False),
          ("w8_11520", 512, 1280, 11520, 0, L.PP_ACT_NONE, True, False),      # the 8x8-level convs as plain GEMMs
          ("w8_23040", 512, 1280, 23040, 0, L.PP_ACT_NONE, True, False)]
TILES = tuple(int(t) for t in os.environ.get("PP_TILE_PROBE_TILES", "0,32,42,31,54,21,24,33,53").split(","))
SPLITS = tuple(int(t) for t in os.environ.get("PP_TILE_PROBE_SPLITS", "1,2,4").split(","))
BIG = None


def graph_us(fns, rep=10, iters=5):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep):
            for f in fns:
                f()
    best = 1e9
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / rep)
    return best


def main():
    global BIG
    dev, dt = "cuda", torch.bfloat16
    BIG = torch.empty(1 << 26, device=dev)
    flush = lambda: BIG.zero_()  # noqa: E731
    t_flush = graph_us([flush])
    want = set(sys.argv[1:])
    print(f"# flush alone {t_flush:.1f} us; columns: tile code x split-K -> hot | cold-in-graph us (incl. the combine launch)")
    for (name, M, N, K, K2, act, res, rs) in SHAPES:
        if want and name not in want:
            continue
        x = torch.randn(M, K, device=dev).to(dt)
        x2 = torch.randn(M, K2, device=dev).to(dt) if K2 else None
        w = (torch.randn(N, K + K2, device=dev) * (K + K2) ** -0.5).to(dt)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).to(dt) if res else None
        print(f"{name}: M={M} N={N} K={K + K2}", flush=True)
        for tile in TILES:
            line = f"   tile {tile:2d}:"
            for sk in ((0,) if tile == 0 else SPLITS):
                if rs and sk > 1:
                    continue
                try:
                    f = lambda: ops.gemm(x, w, b, x2=x2, res1=r, act=act, tile=tile, splitk=sk, row_stats=rs)  # noqa: E731
                    h = graph_us([f])
                    c = graph_us([flush, f]) - t_flush
                    line += f"   sk{sk} {h:6.1f} |{c:6.1f}"
                except Exception as e:
                    line += f"   sk{sk} ERR({str(e)[-24:]})"
            print(line, flush=True)


if __name__ == "__main__":
    main()
