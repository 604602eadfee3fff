#!/usr/bin/env python
"""Per-shape timing of ResnetBlock2D's `GroupNorm -> SiLU -> conv3x3` at the headline step's shapes (batch 8): the fused
launch (csrc/conv_gn.hip) against the two launches it replaces (pp_groupnorm_apply_acc + pp_gemm_bf16(PP_X_CONV3X3)).
Each variant is captured into a hipGraph of REP back-to-back launches and replayed (no host gaps), timed with events.

  python tools/conv_gn_shapes.py [--rep 10] [--out file.json]
Lab switches (PP_LAB=1 PP_LIB=.../libpp_hip_lab.so): PP_CONV_GN_NMODE=0|1|2|3, PP_CONV_GN_PP=0.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from powerpaint_amd import ops  # noqa: E402

# (H = W, C1, C2, Cout, C3, C4, count per UNet forward) -- SD-1.5 at 64x64 latents
SHAPES = [
    (64, 320, 0, 320, 0, 0, 4), (64, 640, 320, 320, 0, 0, 1), (64, 320, 320, 320, 0, 0, 2), (64, 320, 0, 320, 640, 320, 1),
    (64, 320, 0, 320, 320, 320, 2),
    (32, 320, 0, 640, 0, 0, 1), (32, 640, 0, 640, 0, 0, 2), (32, 640, 0, 640, 320, 0, 1), (32, 1280, 640, 640, 0, 0, 1),
    (32, 640, 640, 640, 0, 0, 1), (32, 640, 320, 640, 0, 0, 1), (32, 640, 0, 640, 1280, 640, 1), (32, 640, 0, 640, 640, 640, 1),
    (32, 640, 0, 640, 640, 320, 1),
    (16, 640, 0, 1280, 0, 0, 1), (16, 1280, 0, 1280, 0, 0, 2), (16, 1280, 0, 1280, 640, 0, 1), (16, 1280, 1280, 1280, 0, 0, 2),
    (16, 1280, 640, 1280, 0, 0, 1), (16, 1280, 0, 1280, 1280, 1280, 2), (16, 1280, 0, 1280, 1280, 640, 1),
    (8, 1280, 0, 1280, 0, 0, 8), (8, 1280, 1280, 1280, 0, 0, 3), (8, 1280, 0, 1280, 1280, 1280, 3),
]


def gn_acc(x, groups=32):
    B, H, W, C = x.shape
    xf = x.double().reshape(B, H * W, groups, C // groups)
    return torch.stack([(xf.sum((1, 3)) * 2 ** 24).round().long(), ((xf * xf).sum((1, 3)) * 2 ** 20).round().long()],
                       -1).contiguous()


def timed(fn, rep, iters=5):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(rep):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000.0 / rep)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rep", type=int, default=10)
    ap.add_argument("--out", default="")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--only-h", type=int, default=0, help="only the shapes of this latent size")
    args = ap.parse_args()
    dev, dt, B = "cuda", torch.bfloat16, args.batch
    rows, tot = [], {"fused": 0.0, "conv": 0.0, "apply": 0.0}
    for (H, C1, C2, Cout, C3, C4, cnt) in SHAPES:
        if args.only_h and H != args.only_h:
            continue
        g = torch.Generator("cpu").manual_seed(H + C1 + C2)
        mk = lambda *s: torch.randn(*s, generator=g).to(dev).to(dt)  # noqa: E731
        x1 = mk(B, H, H, C1)
        x2 = mk(B, H, H, C2) if C2 else None
        x3 = mk(B, H, H, C3) if C3 else None
        x4 = mk(B, H, H, C4) if C4 else None
        Ct = C1 + C2
        K = 9 * Ct + C3 + C4
        w = (torch.randn(Cout, K, generator=g) * K ** -0.5).to(dev).to(dt)
        bias = torch.randn(Cout, generator=g).to(dev)
        gam, bet = torch.ones(Ct, device=dev), torch.zeros(Ct, device=dev)
        acc = gn_acc(torch.cat([x1, x2], -1) if C2 else x1)
        gb = ops.gn_gamma_beta(gam, bet)
        y = ops.groupnorm_apply_acc(x1, acc, gam, bet, 1e-5, True, x2=x2)
        t_f = timed(lambda: ops.conv3x3(x1, w, bias, x2=x2, x3=x3, x4=x4, gn_in=(acc, gb, 32, 1e-5), tile=args.tile, splitk=args.splitk), args.rep) \
            if ops.conv_gn_supported(x1, Cout, x2=x2, x3=x3, x4=x4) else float("nan")
        t_c = timed(lambda: ops.conv3x3(y, w, bias, x3=x3, x4=x4), args.rep)
        t_a = timed(lambda: ops.groupnorm_apply_acc(x1, acc, gam, bet, 1e-5, True, x2=x2), args.rep)
        M = B * H * H
        fl = 2.0 * M * Cout * K
        rows.append(dict(M=M, N=Cout, K=K, cat=bool(C2), tail=C3 + C4, count=cnt, fused_us=t_f, conv_us=t_c, apply_us=t_a,
                         fused_tflops=fl / t_f * 1e-6, conv_tflops=fl / t_c * 1e-6))
        tot["fused"] += cnt * t_f
        tot["conv"] += cnt * t_c
        tot["apply"] += cnt * t_a
        print(f"M={M:5d} N={Cout:4d} K={K:5d} {'cat ' if C2 else '    '}{'tail' if C3 else '    '} x{cnt}: fused {t_f:6.1f} us "
              f"({fl / t_f * 1e-6:5.0f} TF)  conv {t_c:6.1f} ({fl / t_c * 1e-6:5.0f} TF) + apply {t_a:5.1f}  "
              f"-> fused - (conv + apply) = {t_f - t_c - t_a:+6.1f}", flush=True)
    print(f"per UNet forward (hot, back to back): fused {tot['fused']:.0f} us; conv {tot['conv']:.0f} + apply {tot['apply']:.0f} "
          f"= {tot['conv'] + tot['apply']:.0f} us")
    if args.out:
        json.dump(dict(env={k: v for k, v in os.environ.items() if k.startswith("PP_")}, rows=rows, total=tot),
                  open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
