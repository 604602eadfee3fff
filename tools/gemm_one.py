#!/usr/bin/env python
"""One GEMM shape, a few launches (PMC / rocprof target).   python tools/gemm_one.py M N K tile [dbg]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_ablate import run  # noqa: E402

M, N, K, tile = (int(v) for v in sys.argv[1:5])
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
t = run(M, N, K, tile, 1, dbg, iters=5)
print(f"M={M} N={N} K={K} tile={tile} dbg={dbg}: {t:.1f} us  {2.0 * M * N * K / t / 1e6:.0f} TF")
