#!/usr/bin/env python
"""conv_norm_out + SiLU + conv_out: the fused launch (pp_gn_conv3x3_smallcout) against groupnorm_apply_acc + conv3x3_smallcout,
at the UNet's output shape (B = 8, 64 x 64 x 320) and config 5's (B = 4, 128 x 128): us, hot and behind a 1 GiB memset."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402
from tools.xattn_ab import timeit  # noqa: E402

for B, H in [(8, 64), (4, 128)]:
    C = 320
    x = torch.randn(B, H, H, C, device="cuda").bfloat16()
    xf = x.float().reshape(B, H * H, 32, C // 32)
    acc = torch.stack([(xf.sum((1, 3)).double() * 2 ** 24).round().long(),
                       ((xf * xf).sum((1, 3)).double() * 2 ** 20).round().long()], -1).contiguous()
    g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    w = (torch.randn(4, 9 * C, device="cuda") * (9 * C) ** -0.5).bfloat16()
    bias = torch.randn(4, device="cuda")
    fused = lambda: ops.gn_conv3x3_smallcout(x, acc, g, b, 1e-5, w, bias)
    ap = lambda: ops.groupnorm_apply_acc(x, acc, g, b, 1e-5, True)
    y = ap()
    cv = lambda: ops.conv3x3_smallcout(y, w, bias)
    for cold in (False, True):
        print(f"B={B} {H}x{H} {'cold' if cold else 'hot '}: apply {timeit(ap, cold):6.1f} + conv {timeit(cv, cold):6.1f} us   fused {timeit(fused, cold):6.1f} us",
              flush=True)
