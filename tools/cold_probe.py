#!/usr/bin/env python
"""Cold-cache cost of a launch: the same split-K GEMM (+ combine kernel) timed (a) back to back (hot I-cache / L2 / MALL)
and (b) with a 1 GB memset + an unrelated kernel between repetitions (what it sees inside the real launch sequence).
    python tools/cold_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L, ops  # noqa: E402


def main():
    lib = L.lib()
    dev = "cuda"
    M, N, K = 512, 1280, 11520
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(8 * M * N, device=dev)
    big = torch.empty(1 << 28, device=dev)          # 1 GiB
    gx = torch.randn(8, 32, 32, 640, device=dev).bfloat16()
    gg, gb = torch.randn(640, device=dev), torch.randn(640, device=dev)
    st = torch.cuda.current_stream()
    for sk in (1, 2, 8):
        a = L.PPGemmArgs()
        a.M, a.N, a.K, a.x_mode = M, N, K, L.PP_X_PLAIN
        a.x1, a.c1, a.ldx1 = x.data_ptr(), K, K
        a.w, a.bias = w.data_ptr(), bias.data_ptr()
        a.res1, a.ldres1, a.ldres2 = res.data_ptr(), N, N
        a.scale, a.act, a.out, a.ldo = 1.0, 0, out.data_ptr(), N
        a.tile, a.splitk, a.workspace = 21, sk, ws.data_ptr()
        for mode in ("hot", "cold"):
            ts = []
            for it in range(12):
                if mode == "cold":
                    big.zero_()
                    ops.groupnorm(gx, gg, gb, 1e-5, True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                lib.pp_gemm_bf16(C.byref(a), st.cuda_stream)
                e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            print(f"splitk={sk} {mode:4s}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}", flush=True)


if __name__ == "__main__":
    main()
