#!/usr/bin/env python
"""Per-shape timing of the FeedForward GEGLU GEMM (hot back-to-back and cold behind a 1 GiB memset).  Run once per
setting with the lab library:  PP_LAB=1 PP_LIB=powerpaint_amd/libpp_hip_lab.so PP_GEGLU_WS=0|1 python tools/geglu_ws_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402
from tools.gemm_sweep import time_launch, time_launch_cold  # noqa: E402


def main():
    lib = L.lib()
    tag = os.environ.get("PP_GEGLU_WS", "default")
    for M, Cc in [(32768, 320), (8192, 640), (2048, 1280), (65536, 320), (16384, 640), (4096, 1280)]:
        N = 8 * Cc
        x = torch.randn(M, Cc, device="cuda").bfloat16()
        w = (torch.randn(N, Cc, device="cuda") * Cc ** -0.5).bfloat16()
        b = torch.randn(N, device="cuda")
        st = torch.randn(M, Cc // 160, 2, device="cuda").abs() + 1
        cs = torch.randn(N, device="cuda")
        out = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)
        a = L.PPGemmArgs()
        a.M, a.N, a.K, a.x_mode = M, N, Cc, L.PP_X_PLAIN
        a.x1, a.c1, a.ldx1 = x.data_ptr(), Cc, Cc
        a.w, a.bias = w.data_ptr(), b.data_ptr()
        a.scale, a.act = 1.0, L.PP_ACT_GEGLU
        a.dtype = L.PP_DT_BF16
        a.out, a.ldo = out.data_ptr(), N // 2
        a.ldres1 = a.ldres2 = N
        a.ln_stats, a.ln_colsum, a.ln_tiles, a.ln_dim, a.ln_eps = st.data_ptr(), cs.data_ptr(), Cc // 160, Cc, 1e-5
        hot = min(time_launch(lib, a, 50) for _ in range(3))
        cold = time_launch_cold(lib, a, 9)
        fl = 2.0 * M * N * Cc
        print(f"WS={tag} M={M:6d} C={Cc:5d}: hot {hot:8.1f} us {fl / hot / 1e6:7.1f} TF/s   cold {cold:8.1f} us {fl / cold / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
