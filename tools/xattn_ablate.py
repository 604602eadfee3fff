#!/usr/bin/env python
"""Lab build: time pp_xattn_block with parts removed (PP_XA_DBG), M = 32768.  One process per setting."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops
from tools.xattn_ab import timeit
B, hw, C, heads, nctx = 8, 4096, 320, 8, 77
M = B * hw
dt = torch.bfloat16
h = torch.randn(M, C, device="cuda").to(dt)
k = torch.randn(B * nctx, C, device="cuda").to(dt)
vtp = torch.zeros(B, C, 80, dtype=dt, device="cuda"); vtp[:, :, :nctx].normal_()
wq = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt); wo = (torch.randn(C, C, device="cuda") * C ** -0.5).to(dt)
bo = torch.randn(C, device="cuda")
hf = h.float()
st = torch.stack([hf.reshape(M, 2, 160).sum(-1), (hf * hf).reshape(M, 2, 160).sum(-1)], -1).contiguous()
folded = ops.xattn_fold(k, vtp, B, nctx, heads, wq, wo)
f = lambda: ops.xattn_block(h, folded, bias_o=bo, res=h, ln_stats=st, rows_per_batch=hw, row_stats=True)
print(f"PP_XA_DBG={os.environ.get('PP_XA_DBG', '0'):>3}: {timeit(f, False, 30):7.1f} us", flush=True)
