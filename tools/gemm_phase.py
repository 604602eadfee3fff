#!/usr/bin/env python
"""Phase decomposition of the v2 GEMM main loop on long-K shapes (PPGemmArgs.dbg switches: 1 no DMA refill,
2 no fragment reads / MFMA, 4 no epilogue).    python tools/gemm_phase.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_ablate import run  # noqa: E402

names = {0: "full", 4: "no-epi", 5: "no-epi,no-dma", 6: "no-epi,no-lds/mfma", 7: "loop only"}
for (M, N, K, tile) in [(8192, 1280, 11520, 33), (32768, 320, 2880, 33), (8192, 640, 5760, 31), (32768, 320, 2880, 24)]:
    line = f"M={M:6d} N={N:5d} K={K:6d} tile={tile:3d}: "
    fl = 2.0 * M * N * K
    for dbg in (0, 4, 5, 6, 7):
        t = run(M, N, K, tile, 1, dbg, iters=10)
        line += f"{names[dbg]}={t:7.1f}us ({fl / t / 1e6:5.0f} TF)  " if dbg in (0, 4, 5) else f"{names[dbg]}={t:7.1f}us  "
    print(line, flush=True)
