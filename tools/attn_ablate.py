#!/usr/bin/env python
"""Ablation of attn_pipe_kernel<40> at N = 4096 (B = 8, 8 heads) through PP_ATTN_DBG (read once per process, so one
subprocess per variant): 1 no MFMA, 2 no exp, 4 no tile DMA / barrier, 8 no LDS fragment reads, sums thereof."""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "one":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from powerpaint_amd import ops
    B, H, d, n = 8, 8, 40, 4096
    C = H * d
    q = torch.randn(B * n, 2 * C, device="cuda").to(torch.bfloat16)
    v = torch.randn(B * n, C, device="cuda").to(torch.bfloat16)
    vt = ops.transpose_v(v, B, n)
    for _ in range(3):
        ops.attention(q[:, :C], q[:, C:], vt, B, H, n, n, d)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(q[:, :C], q[:, C:], vt, B, H, n, n, d)
    e1.record()
    torch.cuda.synchronize()
    print(f"{e0.elapsed_time(e1) / 20 * 1e3:.1f}")
    sys.exit(0)

names = {0: "full", 1: "no MFMA", 2: "no exp", 4: "no DMA/barrier", 8: "no LDS reads", 12: "no DMA, no LDS reads",
         13: "VALU only", 15: "loop only"}
for rnd in range(2):
    for dbg, nm in names.items():
        env = dict(os.environ, PP_ATTN_DBG=str(dbg))
        out = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True).stdout.strip()
        print(f"round {rnd}  dbg {dbg:2d} {nm:22s} {out} us", flush=True)
