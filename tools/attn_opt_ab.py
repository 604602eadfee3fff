#!/usr/bin/env python
"""A/B of the round-3 loop changes of attn_pipe_kernel<40, QB=2> (template parameter OPT, lab build: PP_LAB=1
PP_LIB=powerpaint_amd/libpp_hip_lab.so), N = 4096 and 16384, B = 8 / 4, hot, interleaved rounds; outputs must be bit-identical."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402

OPTS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,5,13,21,29".split(","))]
for B, n in ((8, 4096), (4, 16384)):
    H, d = 8, 40
    C = H * d
    g = torch.Generator("cpu").manual_seed(3)
    q = torch.randn(B * n, C, generator=g).to("cuda", torch.bfloat16)
    k = torch.randn(B * n, C, generator=g).to("cuda", torch.bfloat16)
    v = torch.randn(B * n, C, generator=g).to("cuda", torch.bfloat16)
    vt = ops.transpose_v(v, B, n)
    ref = None
    ts = {o: [] for o in OPTS}
    for rnd in range(3):
        for o in OPTS:
            os.environ["PP_ATTN_OPT"] = str(o)
            out = ops.attention(q, k, vt, B, H, n, n, d)
            if ref is None:
                ref = out.clone()
            if not torch.equal(out, ref):
                print(f"!! OPT {o} changes the result: max diff {(out.float() - ref.float()).abs().max().item():.3g}", flush=True)
            it = 10 if n == 4096 else 3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                ops.attention(q, k, vt, B, H, n, n, d)
            e1.record()
            torch.cuda.synchronize()
            ts[o].append(e0.elapsed_time(e1) / it * 1e3)
    fl = 4.0 * B * H * n * n * d
    for o in OPTS:
        m = sorted(ts[o])[1]
        print(f"N={n:5d} B={B}  OPT {o}: {m:8.1f} us  {fl / m / 1e6:6.1f} TF/s   (rounds: {' '.join(f'{t:.1f}' for t in ts[o])})", flush=True)
