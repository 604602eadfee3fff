#!/usr/bin/env python
"""Phase ablation of the split-K weight-stream launches of the 8x8 level (M = 512, N = 1280, K = 11520 / 23040 as plain GEMMs on
the ping-pong 128x160 tile x 8 splits; lab build: PPGemmArgs.dbg 1 no refill DMA, 2 no MFMA, 4 no epilogue), hot (weights in
the Infinity Cache) and cold (behind a 256 MiB memset).  What bounds the main loop: memory latency, DMA issue or the matrix pipe?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_ablate import run  # noqa: E402

names = {0: "full", 4: "no-epi", 1: "no-refill", 2: "no-mfma", 3: "no-refill,no-mfma", 5: "no-refill,no-epi", 7: "loop only"}
for (M, N, K) in [(512, 1280, 11520), (512, 1280, 23040), (2048, 1280, 11520)]:
    for tile, sk in ((54, 8), (54, 4), (53, 4)):
        if M == 2048 and sk == 8:
            continue
        line = f"M={M} N={N} K={K} tile={tile} x {sk} splits (hot, incl. the separate combine): "
        for d in names:
            ts = sorted(run(M, N, K, tile, sk, d, iters=10) for _ in range(3))
            line += f"  {names[d]} {ts[1]:.1f}"
        print(line, flush=True)
