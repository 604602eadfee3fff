#!/usr/bin/env python
"""One `GroupNorm -> SiLU -> conv3x3` shape through the fused kernel, a few launches (PMC / rocprof target).
    python tools/conv_gn_one.py H C1 C2 Cout [batch]        (lab: PP_CONV_GN_NMODE=0|1|2|3)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from powerpaint_amd import ops  # noqa: E402
from conv_gn_shapes import gn_acc  # noqa: E402

H, C1, C2, Cout = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev, dt = "cuda", torch.bfloat16
g = torch.Generator("cpu").manual_seed(1)
x1 = torch.randn(B, H, H, C1, generator=g).to(dev).to(dt)
x2 = torch.randn(B, H, H, C2, generator=g).to(dev).to(dt) if C2 else None
K = 9 * (C1 + C2)
w = (torch.randn(Cout, K, generator=g) * K ** -0.5).to(dev).to(dt)
acc = gn_acc(torch.cat([x1, x2], -1) if C2 else x1)
gb = ops.gn_gamma_beta(torch.ones(C1 + C2, device=dev), torch.zeros(C1 + C2, device=dev))
for _ in range(3):
    ops.conv3x3(x1, w, None, x2=x2, gn_in=(acc, gb, 32, 1e-5))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv3x3(x1, w, None, x2=x2, gn_in=(acc, gb, 32, 1e-5))
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e3
print(f"conv_gn B={B} {H}x{H} C={C1}+{C2} -> {Cout}: {t:.1f} us (eager, incl. launch gaps)  {2.0 * B * H * H * Cout * K / t / 1e6:.0f} TF")
