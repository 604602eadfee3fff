#!/usr/bin/env python
"""Ablation of the v2 GEMM kernel phases on a few shapes (debug switches in PPGemmArgs.dbg)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402

lib = L.lib()
dev = "cuda"


def run(M, N, K, tile, sk, dbg, res=True, iters=20):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(max(1, sk) * M * N + (1 << 20), dtype=torch.float32, device=dev)      # slabs + combine scratch
    a = L.PPGemmArgs()
    a.M, a.N, a.K, a.x_mode = M, N, K, 0
    a.dtype = L.PP_DT_BF16
    a.x1, a.c1, a.ldx1 = x.data_ptr(), K, K
    a.w, a.bias = w.data_ptr(), b.data_ptr()
    if res:
        a.res1, a.ldres1 = r.data_ptr(), N
    a.scale, a.out, a.ldo = 1.0, out.data_ptr(), N
    a.tile, a.splitk, a.workspace = tile, sk, ws.data_ptr()
    a.dbg = dbg
    st = torch.cuda.current_stream()
    for _ in range(3):
        L.check(lib.pp_gemm_bf16(C.byref(a), st.cuda_stream), "gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        lib.pp_gemm_bf16(C.byref(a), st.cuda_stream)
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if __name__ == "__main__":
    names = {0: "full", 4: "no-epi", 5: "no-epi,no-refill", 6: "no-epi,no-mfma", 7: "nothing(loop only)", 1: "no-refill", 2: "no-mfma"}
    for (M, N, K) in [(2048, 1280, 1280), (512, 1280, 1280), (32768, 320, 320), (32768, 2560, 320), (8192, 640, 640),
                      (2048, 1280, 11520), (32768, 320, 2880)]:
        for tile in (21, 22, 42, 23):
            line = f"M={M:6d} N={N:5d} K={K:6d} tile={tile:3d}: "
            for dbg in (0, 4, 5, 6, 7):
                t = run(M, N, K, tile, 1, dbg)
                line += f"{names[dbg]}={t:7.1f}  "
            print(line, flush=True)
