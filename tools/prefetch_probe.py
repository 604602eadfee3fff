#!/usr/bin/env python
"""What would a weight prefetch buy?  Per GEMM launch of the UNet plan: time it hot (back to back), cold (behind a
1 GiB memset), and cold with W (and W + X) touched by a streaming read first (lands in the Infinity Cache / L2).
    python tools/prefetch_probe.py [--min-gflop 1]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd.engine import SDNet  # noqa: E402
from powerpaint_amd.runtime import NetRuntime  # noqa: E402
from tools.gemm_sweep import FIELDS, signature, time_launch  # noqa: E402

BIG = None


def timed(lib, a, pre=()):
    global BIG
    if BIG is None:
        BIG = torch.empty(1 << 28, device="cuda")
    st = torch.cuda.current_stream()
    ts = []
    for _ in range(5):
        BIG.zero_()
        BIG[:1 << 20].add_(1.0)
        for t in pre:
            t.sum()                      # streaming read of the operand
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        lib.pp_gemm_bf16(C.byref(a), st.cuda_stream)
        e1.record(st)
        st.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-gflop", type=float, default=0.0)
    args = ap.parse_args()
    dev = "cuda"
    lib = L.lib()
    net = SDNet("unet", 9)
    net.load_state_dict(net.synthetic_state_dict(device=dev, seed=0), dev)
    rt = NetRuntime(net, dev)
    rt.ensure(8, 64, 64, 77, 9, ("plain",))
    rt.arena.buf.view(torch.bfloat16).normal_(0, 1)
    groups = {}
    for a in rt.step_plan.keep:
        groups.setdefault(signature(a), []).append(a)
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    tot = dict(hot=0.0, cold=0.0, pw=0.0, pwx=0.0)
    for sig, lst in groups.items():
        a = L.PPGemmArgs.from_buffer_copy(lst[0])
        a.workspace = ws.data_ptr()
        if 2.0 * a.M * a.N * a.K < args.min_gflop * 1e9:
            continue
        wbytes = a.N * a.K * 2
        xbytes = (a.batch * a.hin * a.win * (a.c1 + a.c2) if a.x_mode else a.M * a.ldx1) * 2
        pbuf = net.params.buf
        woff = a.w - pbuf.data_ptr()
        wt = pbuf[woff:woff + wbytes].view(torch.int32)
        abuf = rt.arena.buf
        xoff = a.x1 - abuf.data_ptr()
        xt = abuf[xoff:xoff + min(xbytes, abuf.numel() - xoff)].view(torch.int16)[: (min(xbytes, abuf.numel() - xoff) // 4) * 2].view(torch.int32)
        hot = time_launch(lib, a, iters=10)
        cold = timed(lib, a)
        pw = timed(lib, a, (wt,))
        pwx = timed(lib, a, (wt, xt))
        n = len(lst)
        for k, v in (("hot", hot), ("cold", cold), ("pw", pw), ("pwx", pwx)):
            tot[k] += v * n
        s = dict(zip(FIELDS, sig))
        print(f"{'conv' if s['x_mode'] else 'lin':4} M{s['M']:6d} N{s['N']:6d} K{s['K']:6d} x{n:2d}  W {wbytes / 1e6:6.1f} MB  "
              f"hot {hot:6.1f}  cold {cold:6.1f}  cold+W {pw:6.1f}  cold+W+X {pwx:6.1f}", flush=True)
    print("per step (us): " + "  ".join(f"{k} {v:8.1f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
