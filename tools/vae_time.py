#!/usr/bin/env python
"""Wall time of the stages either side of the loop on the HIP path (synthetic weights): AutoencoderKL decode / encode
and the CLIP text tower.    python tools/vae_time.py [--batch 4] [--latent 64]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd.models import AutoencoderKL, CLIPTextModel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    vae = AutoencoderKL(device="cuda")
    vae.load_state_dict(vae.net.synthetic_state_dict(seed=0))
    z = torch.randn(args.batch, 4, args.latent, args.latent, device="cuda")
    img = torch.rand(args.batch, 3, 8 * args.latent, 8 * args.latent, device="cuda") * 2 - 1
    for name, fn, rt in (("decode", lambda: vae.decode(z), vae._dec), ("encode", lambda: vae.encode(img), vae._enc)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        fl = rt.plan.flops
        print(f"{name}: {ms:8.2f} ms / {args.batch} images of {8 * args.latent}^2   {fl / 1e12:6.2f} TFLOP  "
              f"{fl / ms / 1e9:7.1f} TFLOP/s   {len(rt.plan.calls)} launches   arena {rt.arena.size / 2 ** 30:.2f} GiB",
              flush=True)
        per = rt.plan.run_timed(torch.cuda.current_stream())
        top = sorted(per.items(), key=lambda kv: -kv[1])[:6]
        print("   per kernel family (ms, eager): " + ", ".join(f"{k} {v:.2f}" for k, v in top), flush=True)
    enc = CLIPTextModel(device="cuda")
    ids = torch.randint(0, 49406, (args.batch, 77), device="cuda")
    enc(ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        enc(ids)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.iters * 1e3
    print(f"clip text tower: {ms:8.2f} ms / {args.batch} prompts of 77 tokens   {len(enc._rt.plan.calls)} launches "
          f"(eager, incl. the host-side splice plan)", flush=True)


if __name__ == "__main__":
    main()
