#!/usr/bin/env python
"""A/B of GEMM tile configurations on the launch signatures of the real SD-1.5 UNet plan (B=8, 64x64 latents):
every distinct launch is timed hot (back to back) and cold (behind a 1 GiB memset) for the auto choice and for each
tile in --tiles (split-K as given), interleaved over --rounds rounds; prints median us and TFLOP/s per (shape, tile).
    python tools/gemm_pp_bench.py --tiles 33,53,31,54 --splitk 1,2,4 [--min-gflop 10]
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
from tools.gemm_sweep import FIELDS, signature, time_launch, time_launch_cold  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="33,53,31,54")
    ap.add_argument("--splitk", default="1")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--min-gflop", type=float, default=0.0)
    ap.add_argument("--conv-only", action="store_true")
    ap.add_argument("--lin-only", action="store_true")
    ap.add_argument("--max-m", type=int, default=1 << 30)
    ap.add_argument("--min-m", type=int, default=0)
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    args = ap.parse_args()
    tiles = [int(t) for t in args.tiles.split(",")]
    sks = [int(t) for t in args.splitk.split(",")]
    dev = "cuda"
    lib = L.lib()
    net = SDNet("unet", 9)
    net.load_state_dict(net.synthetic_state_dict(device=dev, seed=0), dev)
    rt = NetRuntime(net, dev)
    rt.ensure(args.batch, args.latent, args.latent, 77, 9, ("plain",))
    rt.arena.buf.view(torch.bfloat16).normal_(0, 1)
    groups = {}
    for a in rt.step_plan.keep:
        groups.setdefault(signature(a), []).append(a)
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    rows = []
    for sig, lst in groups.items():
        a = L.PPGemmArgs.from_buffer_copy(lst[0])
        a.workspace = ws.data_ptr()
        flops = 2.0 * a.M * a.N * a.K
        if flops < args.min_gflop * 1e9 or (args.conv_only and not a.x_mode) or (args.lin_only and a.x_mode):
            continue
        if a.M > args.max_m or a.M < args.min_m:
            continue
        cfgs = [(0, 0)]
        kt = a.K // 64
        for t in tiles:
            for sk in sks:
                if sk > 1 and (kt // sk < 6 or a.M * a.N * 4 * sk > ws.numel()):
                    continue
                if t % 10 == 3 and a.M < 256:
                    continue
                cfgs.append((t, sk))
        hot = {c: [] for c in cfgs}
        cold = {c: [] for c in cfgs}
        for _ in range(args.rounds):
            for c in cfgs:
                a.tile, a.splitk = c
                t = time_launch(lib, a, iters=10)
                if t is None:
                    continue
                hot[c].append(t)
                if not args.no_cold:
                    cold[c].append(time_launch_cold(lib, a, 3))
        med = lambda v: sorted(v)[len(v) // 2] if v else None
        s = dict(zip(FIELDS + ["res1", "res2", "rowvec", "vt"], sig))
        extra = ("s2 " if s["stride"] == 2 else "") + ("up " if s["up"] else "") + ("cat " if s["c2"] else "") + \
            ("geglu " if s["act"] == 1 else "") + ("vt " if s["vt"] else "")
        head = f"{'conv' if s['x_mode'] else 'lin':4} M{s['M']:6d} N{s['N']:6d} K{s['K']:6d} x{len(lst)} {extra}"
        line = []
        for c in cfgs:
            h, cd = med(hot[c]), med(cold[c])
            if h is None:
                continue
            line.append(f"t{c[0]}s{c[1]}: {h:6.1f}us {flops / h / 1e6:5.0f}TF" + (f" (cold {cd:6.1f})" if cd else ""))
            rows.append(dict(sig=s, count=len(lst), flops=flops, tile=c[0], splitk=c[1], hot_us=h, cold_us=cd))
        print(head + "\n    " + "\n    ".join(line), flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"))


if __name__ == "__main__":
    main()
