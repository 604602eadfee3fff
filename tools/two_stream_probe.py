#!/usr/bin/env python
"""Does splitting the batch over two concurrent streams pay?  4 images as ONE batch-8 (CFG) step graph against two
half batches (2 images each, same weights, own arenas and step graphs) replayed on two streams at once.
    python tools/two_stream_probe.py [--steps 50]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from powerpaint_amd import pipelines as PP, schedulers as PS  # noqa: E402
from powerpaint_amd.runtime import NetRuntime  # noqa: E402


def clone_with_own_runtime(m):
    c = object.__new__(type(m))
    c.__dict__.update(m.__dict__)
    c.rt = NetRuntime(m.net, m.device)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    pipe, nets, _ = bench.build_pipeline("v1", dev, 0, 1)
    kw = bench.synthetic_inputs("v1", dev, 0, 4, 64, args.steps)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps

    t_one = timed(lambda: pipe(**kw))
    print(f"one stream, batch 4 images: {t_one * 1e3:.1f} ms  ({4 / t_one:.2f} images/s)", flush=True)

    u2 = clone_with_own_runtime(nets[0])
    u3 = clone_with_own_runtime(nets[0])
    pa = PP.StableDiffusionInpaintPipeline(unet=u2, scheduler=PS.DDIMScheduler())
    pb = PP.StableDiffusionInpaintPipeline(unet=u3, scheduler=PS.DDIMScheduler())
    half = lambda lo: {k: (v[lo:lo + 2] if torch.is_tensor(v) and v.shape[0] == 4 else v) for k, v in kw.items()}  # noqa: E731
    ka, kb = half(0), half(2)
    t_half = timed(lambda: pa(**ka))
    print(f"one stream, batch 2 images: {t_half * 1e3:.1f} ms  ({2 / t_half:.2f} images/s)", flush=True)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    # prime both loops (bind + capture) on their streams
    with torch.cuda.stream(sa):
        pa(**ka)
    with torch.cuda.stream(sb):
        pb(**kb)
    torch.cuda.synchronize()
    la, lb = pa._loop, pb._loop

    def both():
        # interleave the per-step replays of the two loops so both streams stay fed
        for lp, k in ((la, ka), (lb, kb)):
            lp.scheduler.reset()
            lp.latents.copy_(k["latents"].to(torch.float32))
        for _ in range(args.steps):
            with torch.cuda.stream(sa):
                la.graph.replay()
            with torch.cuda.stream(sb):
                lb.graph.replay()

    t_two = timed(both)
    print(f"two streams, 2 + 2 images : {t_two * 1e3:.1f} ms  ({4 / t_two:.2f} images/s)", flush=True)


if __name__ == "__main__":
    main()
