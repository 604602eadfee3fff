#!/usr/bin/env python
"""Per-shape GEMM / implicit-GEMM sweep over the launches of the real SD-1.5 UNet plan (B=8, 64x64 latents).

For every distinct launch signature in the plan: time the auto configuration and every (tile, split-K) alternative on
synthetic buffers with HIP events, print a table sorted by time contribution per denoising step.
    python tools/gemm_sweep.py [--quick] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import _lib as L  # noqa: E402
from powerpaint_amd.engine import SDNet  # noqa: E402
from powerpaint_amd.runtime import NetRuntime  # noqa: E402

FIELDS = ["x_mode", "M", "N", "K", "c1", "c2", "batch", "hin", "win", "hout", "wout", "stride", "up", "act"]


def signature(a):
    return tuple(getattr(a, f) for f in FIELDS) + (bool(a.res1), bool(a.res2), bool(a.rowvec), bool(a.out_vt))


def time_launch(lib, a, iters=20):
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
    return e0.elapsed_time(e1) / iters * 1e3   # us


_BIG = None


def time_launch_cold(lib, a, reps=5):
    """Median over `reps` launches, each preceded by a 1 GiB memset (+ an unrelated elementwise kernel): L2 / MALL /
    instruction cache as cold as they are inside the real launch sequence, where ~350 other launches and > 1 GB of
    activations and weights pass between two uses of the same kernel + operands.  Hot back-to-back timing makes
    split-K look much cheaper than it is in the pipeline (the combine kernel alone: ~8 us hot, ~31 us in the graph)."""
    global _BIG
    if _BIG is None:
        _BIG = torch.empty(1 << 28, device="cuda")
    st = torch.cuda.current_stream()
    if lib.pp_gemm_bf16(C.byref(a), st.cuda_stream) != 0:
        return None
    ts = []
    for _ in range(reps):
        _BIG.zero_()
        _BIG[:1 << 20].add_(1.0)
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
    ap.add_argument("--cold", action="store_true", help="cold-cache timing (see time_launch_cold)")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    args = ap.parse_args()
    dev = "cuda"
    lib = L.lib()
    net = SDNet("unet", 9)
    net.load_state_dict(net.synthetic_state_dict(device=dev, seed=0), dev)
    rt = NetRuntime(net, dev)
    rt.ensure(args.batch, args.latent, args.latent, 77, 9, ("plain",))
    # fill the arena with finite random bf16 so DVFS / data effects are realistic
    rt.arena.buf.view(torch.bfloat16).normal_(0, 1)
    groups = {}
    for a in rt.step_plan.keep:
        groups.setdefault(signature(a), []).append(a)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    rows = []
    for sig, lst in groups.items():
        a0 = lst[0]
        a = L.PPGemmArgs.from_buffer_copy(a0)
        a.workspace = ws.data_ptr()
        flops = 2.0 * a.M * a.N * a.K
        res = {}
        a.tile, a.splitk = 0, 0
        timer = time_launch_cold if args.cold else time_launch
        res["auto"] = timer(lib, a)
        if not args.quick:
            kt = a.K // 64
            for tile in (21, 31, 22, 32, 42, 23, 33, 24):
                for sk in (1, 2, 3, 4, 6, 8):
                    if sk > 1 and (kt // sk < 8 or a.M * a.N * 4 * sk > ws.numel()):
                        continue
                    bm = {1: 128, 2: 64, 3: 256, 4: 128}[tile % 10]
                    nblk = -(-a.M // bm) * -(-a.N // 160) * sk
                    if nblk > 4096 and sk > 1:
                        continue
                    a.tile, a.splitk = tile, sk
                    t = timer(lib, a, 5) if args.cold else time_launch(lib, a, iters=10)
                    if t is not None:
                        res[f"t{tile}s{sk}"] = t
        best = min((v, k) for k, v in res.items() if v is not None)
        rows.append(dict(sig=dict(zip(FIELDS + ["res1", "res2", "rowvec", "vt"], sig)), count=len(lst), flops=flops,
                         auto_us=res["auto"], best_us=best[0], best=best[1], all=res))
    rows.sort(key=lambda r: -r["auto_us"] * r["count"])
    tot_auto = sum(r["auto_us"] * r["count"] for r in rows)
    tot_best = sum(r["best_us"] * r["count"] for r in rows)
    print(f"{'mode':5} {'M':>6} {'N':>6} {'K':>6} {'cnt':>3} {'auto_us':>8} {'TF':>6} {'best_us':>8} {'TF':>6} {'cfg':>6}  extra")
    for r in rows:
        s = r["sig"]
        extra = ("s2 " if s["stride"] == 2 else "") + ("up " if s["up"] else "") + ("cat " if s["c2"] else "") + \
            ("geglu " if s["act"] == 1 else "") + ("vt " if s["vt"] else "")
        print(f"{'conv' if s['x_mode'] else 'lin':5} {s['M']:6d} {s['N']:6d} {s['K']:6d} {r['count']:3d} "
              f"{r['auto_us']:8.1f} {r['flops'] / r['auto_us'] / 1e6:6.0f} {r['best_us']:8.1f} "
              f"{r['flops'] / r['best_us'] / 1e6:6.0f} {r['best']:>6}  {extra}")
    print(f"total per step: auto {tot_auto / 1e3:.2f} ms, best-of-sweep {tot_best / 1e3:.2f} ms")
    if args.json:
        json.dump(rows, open(args.json, "w"))


if __name__ == "__main__":
    main()
