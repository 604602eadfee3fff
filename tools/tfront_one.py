#!/usr/bin/env python
"""pp_tfront at the headline shape (batch 8, 64x64, C = 320), a few launches (PMC / rocprof target).  python tools/tfront_one.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402
from powerpaint_amd.engine import _kperm  # noqa: E402

B, hw, C, dev, dt = 8, 4096, 320, "cuda", torch.bfloat16
g = torch.Generator("cpu").manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
x = r(B * hw, C).to(dt)
xf = x.float().reshape(B, hw, 32, 10)
acc = torch.stack([(xf.sum((1, 3)).double() * 2 ** 24).round().long(), ((xf * xf).sum((1, 3)).double() * 2 ** 20).round().long()], -1).contiguous()
w1, w2 = (r(C, C) * C ** -0.5).to(dt), (r(3 * C, C) * C ** -0.5).to(dt)
args = (x, acc, torch.ones(C, device=dev), torch.zeros(C, device=dev), w1, r(C) * 0.1, _kperm(w2).contiguous(), w2.float().sum(1).contiguous(),
        r(3 * C) * 0.1, hw)
for _ in range(3):
    ops.tfront(*args)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    for _ in range(10):
        ops.tfront(*args)
best = 1e9
for _ in range(5):
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record(); gr.replay(); a1.record(); torch.cuda.synchronize()
    best = min(best, a0.elapsed_time(a1) * 100.0)
print(f"tfront M={B * hw}: {best:.1f} us hot (hipGraph of 10)  PP_TF_QD={os.environ.get('PP_TF_QD')}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.tfront(*args)
e1.record()
torch.cuda.synchronize()
print(f"tfront M={B * hw}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us (eager, incl. launch gaps)")
