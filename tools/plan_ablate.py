#!/usr/bin/env python
"""Marginal cost of each launch family INSIDE the hipGraph: capture the UNet step plan with one family removed at a
time and time graph replays.  (Per-launch HIP-event timings of an eager run include ~5 us of launch overhead each and
do not show what a kernel costs once the graph has removed the gaps.)  Outputs are garbage when launches are dropped --
this is a timing tool only.     python tools/plan_ablate.py [--batch 8] [--latent 64]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd.engine import SDNet  # noqa: E402
from powerpaint_amd.runtime import NetRuntime  # noqa: E402


def time_graph(rt, iters=20):
    rt.graph = None
    rt.capture()
    for _ in range(3):
        rt.graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        rt.graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    args = ap.parse_args()
    dev = "cuda"
    net = SDNet("unet", 9)
    net.load_state_dict(net.synthetic_state_dict(device=dev, seed=0), dev)
    rt = NetRuntime(net, dev)
    rt.ensure(args.batch, args.latent, args.latent, 77, 9, ("plain",))
    rt.set_timestep(500)
    full = list(rt.step_plan.calls)
    names = []
    for c in full:
        if c[2] not in names:
            names.append(c[2])
    base = time_graph(rt)
    print(f"full plan: {len(full)} launches, {base:.3f} ms / forward (graph replay)")
    for n in names:
        rt.step_plan.calls = [c for c in full if c[2] != n]
        k = len(full) - len(rt.step_plan.calls)
        t = time_graph(rt)
        print(f"  without {n:20s} ({k:3d} launches): {t:7.3f} ms  -> marginal {base - t:6.3f} ms "
              f"({(base - t) / k * 1e3:6.1f} us / launch)", flush=True)
    rt.step_plan.calls = full


if __name__ == "__main__":
    main()
