#!/usr/bin/env python
"""What makes a launch slow inside the real sequence?  One small GEMM timed (hot) back to back, (data) behind a 1 GiB
memset = operands evicted from L2 / MALL, (code) behind eight other GEMM template instances on small operands =
instruction cache evicted, (both).    python tools/cold_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402


def mk(lib, M, N, K, tile, sk, dev, ws=None):
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    a = L.PPGemmArgs()
    a.M, a.N, a.K, a.x_mode = M, N, K, L.PP_X_PLAIN
    a.x1, a.c1, a.ldx1 = x.data_ptr(), K, K
    a.w, a.bias = w.data_ptr(), bias.data_ptr()
    a.ldres1 = a.ldres2 = N
    a.scale, a.act, a.out, a.ldo = 1.0, 0, out.data_ptr(), N
    a.tile, a.splitk = tile, sk
    if ws is not None:
        a.workspace = ws.data_ptr()
    a._keep = (x, w, bias, out)
    return a


def main():
    lib = L.lib()
    dev = "cuda"
    big = torch.empty(1 << 28, device=dev)
    ws = torch.empty(8 * 2048 * 1280, device=dev)
    others = [mk(lib, 256, 320, 320, t, 1, dev) for t in (21, 31, 22, 32, 42, 23, 33, 24)]
    st = torch.cuda.current_stream()
    for name, a in (("lin 2048x1280x640  t32", mk(lib, 2048, 1280, 640, 32, 1, dev)),
                    ("lin 32768x320x320  t24", mk(lib, 32768, 320, 320, 24, 1, dev)),
                    ("lin 512x1280x2560  t32 splitk2", mk(lib, 512, 1280, 2560, 32, 2, dev, ws))):
        for mode in ("hot", "data", "code", "both"):
            ts = []
            for it in range(14):
                if mode in ("data", "both"):
                    big.zero_()
                if mode in ("code", "both"):
                    for o in others:
                        lib.pp_gemm_bf16(C.byref(o), st.cuda_stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                lib.pp_gemm_bf16(C.byref(a), st.cuda_stream)
                e1.record(st)
                st.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts = sorted(ts[2:])
            print(f"{name:32s} {mode:5s}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f}", flush=True)


if __name__ == "__main__":
    main()
