#!/usr/bin/env python
"""Epilogue cost probe for the small-K transformer GEMMs: FF1 (GEGLU) / QKV with and without the folded-LayerNorm
correction, against the same launch with the epilogue disabled (debug flag 4) -- shows how much of a K=320..1280 GEMM
is epilogue latency / VALU rather than MFMA.    python tools/epi_probe.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402


def time_launch(lib, a, iters=30):
    st = torch.cuda.current_stream()
    for _ in range(3):
        rc = lib.pp_gemm_bf16(C.byref(a), st.cuda_stream)
        if rc != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        lib.pp_gemm_bf16(C.byref(a), st.cuda_stream)
    e1.record(st)
    st.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    lib = L.lib()
    dev = "cuda"
    for M, Cc in ((32768, 320), (8192, 640), (2048, 1280)):
        x = torch.randn(M, Cc, device=dev).bfloat16()
        tiles = (Cc + 159) // 160
        st = torch.randn(M, tiles, 2, device=dev).abs().contiguous()
        for name, N, act in (("ff1_geglu", 8 * Cc, L.PP_ACT_GEGLU), ("ff1_noact", 8 * Cc, 0), ("qkv", 3 * Cc, 0),
                             ("to_q", Cc, 0)):
            w = (torch.randn(N, Cc, device=dev) * Cc ** -0.5).bfloat16()
            bias = torch.randn(N, device=dev)
            cs = torch.randn(N, device=dev)
            n_out = N // 2 if act == L.PP_ACT_GEGLU else N
            out = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
            stats_out = torch.empty(M, (N + 159) // 160, 2, device=dev)
            row = []
            for variant in ("plain", "ln", "rowstats", "noepi"):
                if variant == "rowstats" and act:
                    row.append("     -")
                    continue
                a = L.PPGemmArgs()
                a.M, a.N, a.K, a.x_mode = M, N, Cc, L.PP_X_PLAIN
                a.x1, a.c1, a.ldx1 = x.data_ptr(), Cc, Cc
                a.w, a.bias = w.data_ptr(), bias.data_ptr()
                a.ldres1 = a.ldres2 = N
                a.scale, a.act = 1.0, act
                a.out, a.ldo = out.data_ptr(), n_out
                if variant == "ln":
                    a.ln_stats, a.ln_colsum, a.ln_tiles, a.ln_dim, a.ln_eps = st.data_ptr(), cs.data_ptr(), tiles, Cc, 1e-5
                if variant == "rowstats":
                    a.row_stats_out = stats_out.data_ptr()
                if variant == "noepi":
                    a.dbg = 4
                t = time_launch(lib, a)
                row.append(f"{t:6.1f}" if t else "   err")
            tf = 2.0 * M * N * Cc / 1e6
            print(f"M={M:6d} C={Cc:5d} {name:10s} N={N:6d}  plain/ln/rowstats/noepi us: {' '.join(row)}   "
                  f"(plain {tf / float(row[0]):.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
