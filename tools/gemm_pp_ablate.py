#!/usr/bin/env python
"""Phase ablation of the ping-pong GEMM (debug switches in PPGemmArgs.dbg: 1 no refill DMA, 2 no MFMA,
4 no epilogue, 8 no s_setprio) on long-K shapes; interleaved rounds, median."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_ablate import run  # noqa: E402

names = {0: "full", 8: "no-setprio", 1: "no-refill", 2: "no-mfma", 3: "no-refill,no-mfma", 9: "no-refill,no-setprio"}
for (M, N, K) in [(8192, 1280, 11520), (32768, 320, 5760)]:
    for tile in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "53,33,54").split(",")]:
        res = {d: [] for d in names}
        for _ in range(3):
            for d in names:
                res[d].append(run(M, N, K, tile, 1, d, iters=10))
        line = f"M={M} N={N} K={K} tile={tile}: " + "  ".join(f"{names[d]} {sorted(v)[1]:.1f}" for d, v in res.items())
        print(line, flush=True)
