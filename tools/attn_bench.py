#!/usr/bin/env python
"""Attention kernel micro-benchmark on the SD-1.5 shapes (B=8, heads=8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402

B, H = 8, 8
for d, nq, nk in [(40, 4096, 4096), (80, 1024, 1024), (160, 256, 256), (160, 64, 64), (40, 4096, 77), (80, 1024, 77),
                  (160, 256, 77)]:
    C = H * d
    q = torch.randn(B * nq, 2 * C, device="cuda").to(torch.bfloat16)
    k = torch.randn(B * nk, 2 * C, device="cuda").to(torch.bfloat16)
    v = torch.randn(B * nk, C, device="cuda").to(torch.bfloat16)
    vt = ops.transpose_v(v, B, nk)
    for _ in range(3):
        ops.attention(q[:, :C], k[:, C:], vt, B, H, nq, nk, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.attention(q[:, :C], k[:, C:], vt, B, H, nq, nk, d)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    fl = 4.0 * B * H * nq * nk * d
    print(f"attn d={d:3d} nq={nq:5d} nk={nk:5d}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s")
