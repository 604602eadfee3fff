#!/usr/bin/env python
"""Headline benchmark of the denoising hot path (BASELINE.json config 2).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over one batch: a full 50-step DDIM, CFG-7.5 denoise of 4 synthetic 512x512
images (latents 4x64x64, 9-channel UNet input) per GPU, `output_type="latent"` (VAE / CLIP excluded, as the metric
says).  Inputs are resident in HBM before the timed region.  Weak scaling: every rank denoises its own 4 images; the
only collective is the start-up RCCL broadcast of the packed UNet parameters from rank 0.

Prints ONE JSON line on rank 0 (fields: see the driver contract; plus `roofline` and `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from powerpaint_amd import dist as ppdist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0        # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md (2495 TF measured)
# algorithmic GFLOP per sample per forward at 64x64 latents (2*MAC of conv/linear/attention)
GFLOP_PER_SAMPLE = {"unet9": 803.4, "unet4": 803.3, "brushnet": 826.2, "controlnet": 283.3}


def build_pipeline(cfg, device, rank, world, net_kw=None):
    """Networks + pipeline of one rank.  Rank 0 creates the (random-init) weights; every other rank only lays out its
    packed parameter buffer (`meta=True` state dict, `materialize=False`) and receives the bytes in the one start-up
    broadcast.  `net_kw` overrides the architecture (tests run this function with a reduced network)."""
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    net_kw = dict(net_kw or {})
    unet = PM.UNet2DConditionModel(in_channels=9 if cfg != "v2" else 4, device=device, **net_kw)
    side = None
    nets = [unet]
    if cfg == "v2":
        side = PM.BrushNetModel(in_channels=4, conditioning_channels=5, device=device, **net_kw)
        nets.append(side)
    elif cfg == "controlnet":
        side = PM.ControlNetModel(in_channels=4, device=device,
                                  **{k: v for k, v in net_kw.items() if k != "up_block_types"})
        nets.append(side)
    for i, m in enumerate(nets):
        if rank == 0:
            sd = m.net.synthetic_state_dict(device=device, seed=i)       # random init directly in HBM
            m.load_state_dict(sd)
            del sd
        else:
            m.load_state_dict(m.net.synthetic_state_dict(meta=True), materialize=False)
    on_gpu = torch.device(device).type == "cuda"
    if on_gpu:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    ppdist.broadcast_params([m.param_buffer() for m in nets], src=0)     # the ONE collective (RCCL over xGMI)
    if on_gpu:
        torch.cuda.synchronize()
    bcast_s = time.perf_counter() - t0
    if cfg == "v1":
        pipe = PP.StableDiffusionInpaintPipeline(unet=unet, scheduler=PS.DDIMScheduler())
    elif cfg == "v2":
        pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(unet=unet, brushnet=side,
                                                            scheduler=PS.DPMSolverMultistepScheduler())
    else:
        pipe = PP.StableDiffusionControlNetInpaintPipeline(unet=unet, controlnet=side, scheduler=PS.DDIMScheduler())
    return pipe, nets, bcast_s


def synthetic_inputs(cfg, device, rank, per_gpu, lat_hw):
    """SURVEY.md section 8d seeds: generator keyed by the GLOBAL image index, generated on CPU in fp32."""
    h = w = lat_hw
    lat, mil, pos, neg, posU, negU, ctrl = [], [], [], [], [], [], []
    for j in range(per_gpu):
        g = ppdist.image_generator(rank * per_gpu + j)
        lat.append(torch.randn(1, 4, h, w, generator=g))
        mil.append(torch.randn(1, 4, h, w, generator=g) * 0.5)
        neg.append(torch.randn(1, 77, 768, generator=g))
        pos.append(torch.randn(1, 77, 768, generator=g))
        negU.append(torch.randn(1, 77, 768, generator=g))
        posU.append(torch.randn(1, 77, 768, generator=g))
        if cfg == "controlnet":
            ctrl.append(torch.rand(1, 3, h * 8, w * 8, generator=g))
    mask = torch.zeros(per_gpu, 1, h, w)
    mask[:, :, h // 4: 3 * h // 4, w // 4: 3 * w // 4] = 1.0
    cat = lambda l: torch.cat(l).to(device)  # noqa: E731
    kw = dict(prompt_embeds=cat(pos), negative_prompt_embeds=cat(neg), latents=cat(lat), guidance_scale=7.5,
              num_inference_steps=50, output_type="latent", return_dict=False)
    if cfg in ("v1", "controlnet"):
        kw.update(mask_latents=mask.to(device), masked_image_latents=cat(mil), height=h * 8, width=w * 8)
        if cfg == "controlnet":
            kw.update(control_image=cat(ctrl), controlnet_conditioning_scale=0.5)
    else:
        kw.update(conditioning_latents=torch.cat([cat(mil), mask.to(device)], 1), prompt_embedsU=cat(posU),
                  negative_prompt_embedsU=cat(negU))
    return kw


def roofline_pass(pipe):
    """Instrumented eager replay of one denoising step: HIP events around every launch on the launch stream."""
    loop = pipe._loop
    loop.scheduler.reset()
    st = torch.cuda.current_stream()
    loop.program.run(st.cuda_stream)               # warm
    loop.scheduler.reset()
    per = {}
    reps = 3
    for _ in range(reps):
        loop.scheduler.reset()
        for k, v in loop.program.run_timed(st).items():
            per[k] = per.get(k, 0.0) + v / reps
    flops = {}
    for plan in ([loop.side_rt.step_plan] if loop.side_rt is not None else []) + [loop.rt.step_plan]:
        for k, v in plan.flops_by_kind.items():
            flops[k] = flops.get(k, 0.0) + v
    counts = {}
    for _, _, name in loop.program.calls:
        counts[name] = counts.get(name, 0) + 1
    # algorithmic HBM bytes of the 3x3 implicit-GEMM launches: input pixels once, weights once, output once,
    # residual operands once (bf16 = 2 B, temb row vector fp32)
    from powerpaint_amd import _lib as L
    alg_bytes = 0.0
    for fn, args, name in loop.program.calls:
        if name != "conv3x3":
            continue
        a = args[0]._obj
        if isinstance(a, L.PPGemmArgs):
            alg_bytes += 2.0 * a.batch * a.hin * a.win * (a.c1 + a.c2) + 2.0 * a.N * a.K + 2.0 * a.M * a.N
            alg_bytes += (2.0 * a.M * a.N if a.res1 else 0.0) + (2.0 * a.M * a.N if a.res2 else 0.0)
    return per, flops, counts, alg_bytes


def measured_traffic():
    """HBM bytes per 3x3 implicit-GEMM launch from the committed PMC passes (tools/hbm_traffic.sh -> profiles/; two
    rocprofv3 --pmc runs of this benchmark cannot happen inside this process).  None if the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        fam = json.load(open(path))["families"]["conv3x3 implicit GEMM"]
        return fam["bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline():
    """The CPU oracle (kind 'port': plain-PyTorch fp32 restatement of the reference's diffusers path) timed on this
    box's host cores on a BOUNDED sample of the same workload: ONE UNet forward of one image with CFG (batch 2) at
    32x32 latents (256x256 px, BASELINE config 1's shape; 0.360 TFLOP), scaled to the 64x64-latent forward by the
    algorithmic FLOP ratio (803.4 / 180.1 GFLOP per sample) and to 50 steps."""
    from oracle import sd_modules as OM
    cores = min(os.cpu_count() or 1, 64)        # more threads only add contention for these conv sizes
    torch.set_num_threads(cores)
    with torch.device("meta"):
        o = OM.UNet2DConditionModel(in_channels=9)
    o = o.to_empty(device="cpu")
    g = torch.Generator("cpu").manual_seed(0)
    with torch.no_grad():
        for n, p in o.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0) if ("norm" in n and n.endswith("weight")) else p.zero_()
            else:
                p.normal_(0, (1.0 / p[0].numel()) ** 0.5, generator=g)
        x = torch.randn(2, 9, 32, 32, generator=g)
        e = torch.randn(2, 77, 768, generator=g)
        o(x[:, :, :8, :8], 500, e)                    # touch weights / warm the thread pool
        t0 = time.perf_counter()
        o(x, 500, e)
        dt = time.perf_counter() - t0
    scale = 803.4 / 180.1
    return {"value": 1.0 / (dt * scale * 50), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 UNet forward, 1 image with CFG (batch 2), 32x32 latents, fp32 torch CPU oracle: {dt:.2f} s; "
                      f"scaled x{scale:.2f} (FLOP ratio to 64x64 latents) x50 steps"}


def roofline_report(pipe, dump_launches=None) -> dict:
    """`roofline` (dominant kernel = the implicit-GEMM 3x3 convolution launches of pp_gemm_kernel) and the per-kernel
    tables, from one eager replay of the step program with a HIP event pair around every launch."""
    per, flops, counts, alg_bytes = roofline_pass(pipe)
    k = "conv3x3"
    ach = flops[k] / 1e12 / (per[k] * 1e-3)
    out = {"roofline": {"bound": "mfma", "kernel": "pp_gemm_kernel<...,CONV3X3> (implicit-GEMM 3x3 conv)",
                        "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / MFMA_PEAK_TFLOPS, "traffic": measured_traffic(),
                        "traffic_unit": "bytes / launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_hbm_traffic.json)",
                        "launches_per_step": counts[k], "avg_launch_ms": per[k] / counts[k],
                        "alg_flop_per_launch": flops[k] / counts[k],
                        "alg_bytes_per_launch": alg_bytes / counts[k]},
           "per_kernel_ms_per_denoise_step": {n: round(v, 4) for n, v in sorted(per.items(), key=lambda kv: -kv[1])},
           "per_kernel_tflops": {n: round(flops[n] / 1e12 / (per[n] * 1e-3), 1) for n in flops if per.get(n)}}
    if dump_launches:
        prog = pipe._loop.program
        json.dump([{"i": i, "what": prog.describe(i), "ms": prog.last_launch_ms[i]}
                   for i in range(len(prog.calls))], open(dump_launches, "w"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="v1", choices=["v1", "v2", "controlnet"])
    ap.add_argument("--per-gpu", type=int, default=4)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write per-launch HIP-event timings of one step (JSON)")
    args = ap.parse_args()

    rank, world, local = ppdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    pipe, nets, bcast_s = build_pipeline(args.config, device, rank, world)
    pipe.use_graph = not args.no_graph
    kw = synthetic_inputs(args.config, device, rank, args.per_gpu, args.latent)

    for _ in range(args.warmup):
        pipe(**kw)
    ppdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe(**kw)[0]
    torch.cuda.synchronize()
    ppdist.barrier()
    dt = ppdist.max_over_ranks(time.perf_counter() - t0, device)
    assert torch.isfinite(out).all()

    if rank == 0:
        images = args.per_gpu * world * args.steps
        ms_per_step = dt / args.steps * 1e3
        unet_steps = 50
        gf = {"v1": GFLOP_PER_SAMPLE["unet9"], "v2": GFLOP_PER_SAMPLE["unet4"] + GFLOP_PER_SAMPLE["brushnet"],
              "controlnet": GFLOP_PER_SAMPLE["unet9"] + GFLOP_PER_SAMPLE["controlnet"]}[args.config]
        scale_hw = (args.latent / 64.0) ** 2 if args.latent != 64 else 1.0
        tflop_step = gf * 2 * args.per_gpu * scale_hw / 1e3         # CFG doubles the batch
        ms_denoise_step = ms_per_step / unet_steps
        px = args.latent * 8
        res = {
            "metric": ("inpainted images/sec @512x512, 50-step DDIM CFG, batch4/GPU" if (px, args.per_gpu, args.config) == (512, 4, "v1")
                       else f"inpainted images/sec @{px}x{px}, 50-step, batch{args.per_gpu}/GPU ({args.config})"),
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": {"v1": f"ppt-v1 SD1.5-inpaint UNet (9-ch in), {px}x{px}, 50-step DDIM, CFG=7.5, batch={args.per_gpu}/GPU",
                                    "v2": f"ppt-v2 BrushNet + SD1.5 UNet, {px}x{px}, 50-step DPMSolver++, CFG=7.5, batch={args.per_gpu}/GPU",
                                    "controlnet": f"ppt-v1 + ControlNet, {px}x{px}, 50-step DDIM, CFG=7.5, batch={args.per_gpu}/GPU"}[args.config],
                       "global_batch": args.per_gpu * world, "latent": [4, args.latent, args.latent],
                       "denoise_steps": unet_steps, "parallelism": f"dp{world} (image shards, no step collectives)",
                       "hipgraph": not args.no_graph, "weights": "random init (no checkpoints offline)",
                       "weight_broadcast_s": round(bcast_s, 4)},
            "ms_per_denoise_step": ms_denoise_step,
            "unet_step_mfma_util": tflop_step / (ms_denoise_step * 1e-3) / MFMA_PEAK_TFLOPS,
        }
        if not args.no_roofline:
            try:
                res.update(roofline_report(pipe, args.dump_launches))
            except Exception as e:      # the headline number must still be reported: keep the one JSON line
                res["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(res), flush=True)
    ppdist.barrier()


if __name__ == "__main__":
    main()
