#!/usr/bin/env python
"""Run ONE attention shape a few times (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402

d, nq, nk = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (40, 4096, 4096)))
B, H = 8, 8
C = H * d
q = torch.randn(B * nq, 2 * C, device="cuda").to(torch.bfloat16)
k = torch.randn(B * nk, 2 * C, device="cuda").to(torch.bfloat16)
v = torch.randn(B * nk, C, device="cuda").to(torch.bfloat16)
vt = ops.transpose_v(v, B, nk)
for _ in range(4):
    ops.attention(q[:, :C], k[:, C:], vt, B, H, nq, nk, d)
torch.cuda.synchronize()
