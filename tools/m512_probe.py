#!/usr/bin/env python
"""Tile / split-K probe for the 8x8-level 3x3 convs (M = 512, N = 1280): conv + combine, hot (hipGraph of 10) and behind
a 256 MB memset.   python tools/m512_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from powerpaint_amd import ops  # noqa: E402
from stage_probe import hot, cold  # noqa: E402


def main():
    dev, dt = "cuda", torch.bfloat16
    for (C1, C2) in ((1280, 0), (1280, 1280)):
        x = torch.randn(8, 8, 8, C1, device=dev).to(dt)
        x2 = torch.randn(8, 8, 8, C2, device=dev).to(dt) if C2 else None
        K = 9 * (C1 + C2)
        w = (torch.randn(1280, K, device=dev) * K ** -0.5).to(dt)
        b = torch.randn(1280, device=dev)
        print(f"M=512 N=1280 K={K}")
        for tile in (0, 54, 53, 44, 32, 42, 31, 24):
            line = f"  tile {tile:2d}:"
            for sk in (0, 2, 4, 8):
                if tile == 0 and sk:
                    continue
                try:
                    f = lambda: ops.conv3x3(x, w, b, x2=x2, tile=tile, splitk=sk)  # noqa: E731
                    line += f"  sk{sk} {hot(f):5.1f}|{cold(f):5.1f}"
                except Exception as e:
                    line += f"  sk{sk} ERR"
            print(line, flush=True)


if __name__ == "__main__":
    main()
