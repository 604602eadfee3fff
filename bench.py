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
# ... at 32x32 / 128x128 latents (attention is quadratic in the token count, so not a plain area ratio)
GFLOP_PER_SAMPLE_BY_LATENT = {32: {"unet9": 180.1, "unet4": 180.1, "brushnet": 185.8, "controlnet": 62.7},
                              64: GFLOP_PER_SAMPLE,
                              128: {"unet9": 4674.5, "unet4": 4674.0, "brushnet": 4765.5, "controlnet": 1717.1}}


def build_pipeline(cfg, device, rank, world, net_kw=None, dtype=torch.bfloat16, scheduler=None, bcast_timeout=0.0):
    """Networks + pipeline of one rank.  Rank 0 creates the (random-init) weights; every other rank only lays out its
    packed parameter buffer (`meta=True` state dict, `materialize=False`) and receives the bytes in the one start-up
    broadcast.  `net_kw` overrides the architecture (tests run this function with a reduced network)."""
    from powerpaint_amd import models as PM, pipelines as PP, schedulers as PS
    net_kw = dict(net_kw or {}, dtype=dtype)
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
    # (the watchdog covers the collective only: a slow but healthy local build -- random init, packing -- must not be
    #  killed and blamed on the broadcast)
    with ppdist.Watchdog(bcast_timeout if world > 1 else 0.0, "weight broadcast"):
        ppdist.broadcast_params([m.param_buffer() for m in nets], src=0)     # the ONE collective (RCCL over xGMI)
        if on_gpu:
            torch.cuda.synchronize()
    for m in nets:
        m.params_changed()
    bcast_s = time.perf_counter() - t0
    sched = {"ddim": PS.DDIMScheduler, "dpm": PS.DPMSolverMultistepScheduler, "pndm": PS.PNDMScheduler,
             "unipc": PS.UniPCMultistepScheduler}[scheduler or ("dpm" if cfg == "v2" else "ddim")]()
    if cfg == "v1":
        pipe = PP.StableDiffusionInpaintPipeline(unet=unet, scheduler=sched)
    elif cfg == "v2":
        pipe = PP.StableDiffusionPowerPaintBrushNetPipeline(unet=unet, brushnet=side, scheduler=sched)
    else:
        pipe = PP.StableDiffusionControlNetInpaintPipeline(unet=unet, controlnet=side, scheduler=sched)
    return pipe, nets, bcast_s


def synthetic_inputs(cfg, device, rank, per_gpu, lat_hw, denoise_steps=50):
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
              num_inference_steps=denoise_steps, output_type="latent", return_dict=False)
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
    hbm_bytes = {}          # algorithmic bytes (elements read + written) of the HBM-bound kernels
    for fn, args, name in loop.program.calls:
        if name == "groupnorm_apply":        # (x1, c1, x2, c2, batch, hw, ...): read + write of [batch][hw][c1 + c2] 16-bit
            hbm_bytes[name] = hbm_bytes.get(name, 0.0) + 2.0 * 2.0 * args[4] * args[5] * (args[1] + args[3])
        elif name == "cfg_sched_step":       # eps (2 x n) + latents r/w (+ scheduler state) in fp32: 20 B / latent element
            hbm_bytes[name] = hbm_bytes.get(name, 0.0) + 20.0 * args[5]
        if name != "conv3x3":
            continue
        a = args[0]._obj
        if isinstance(a, L.PPGemmArgs):
            alg_bytes += 2.0 * a.batch * a.hin * a.win * (a.c1 + a.c2) + 2.0 * a.N * a.K + 2.0 * a.M * a.N
            alg_bytes += (2.0 * a.M * a.N if a.res1 else 0.0) + (2.0 * a.M * a.N if a.res2 else 0.0)
    return per, flops, counts, alg_bytes, hbm_bytes


TRAFFIC_PROFILE = "profiles/r06_hbm_traffic.json"         # tools/hbm_traffic.sh (two rocprofv3 --pmc passes of this bench)
KERNEL_STATS_PROFILE = "profiles/r06_kernel_stats.txt"     # rocprofv3 --kernel-trace --stats of this bench (tools/gpu_round.sh)


def lib_sha16() -> str:
    """Identity of the libpp_hip.so this process runs (pp_build_id: a digest of the sources, headers and flags it was built
    from -- stable across rebuilds, unlike a hash of the binary): a committed profile names the build it was taken from."""
    from powerpaint_amd import _lib as L
    try:
        return L.build_id()
    except Exception:
        return ""


def family_replay_us(pipe, names, reps: int = 20):
    """LIVE kernel time of one launch family: the family's launches of the step program (C-ABI calls, i.e. a split-K
    GEMM together with its combine launch) are captured into a hipGraph of their own and that graph is replayed `reps`
    times between ONE HIP event pair on the launch stream.  No event sits between kernels, nothing is subtracted:
    (elapsed / reps) is the wall time the device spends on the family, launch boundaries included, with each launch
    reading the arena addresses and weights it reads in the real step.  Returns (us per replay, number of launches)."""
    from powerpaint_amd.engine import Plan
    loop = pipe._loop
    sub = Plan()
    sub.calls = [c for c in loop.program.calls if c[2] in names]
    if not sub.calls:
        return None, 0
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sub.run(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sub.run(torch.cuda.current_stream().cuda_stream)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, len(sub.calls)


def measured_traffic():
    """HBM bytes per 3x3 implicit-GEMM launch from the committed PMC passes (two rocprofv3 --pmc runs of this benchmark
    cannot happen inside this process).  None if the profile is absent."""
    for rel in (TRAFFIC_PROFILE, "profiles/r04_hbm_traffic.json"):
        try:
            fam = json.load(open(os.path.join(ROOT, rel)))["families"]["conv3x3 implicit GEMM"]
            return fam["bytes_per_launch"], rel
        except Exception:
            continue
    return None, None


def profiled_conv_launch_us():
    """Average duration of the 3x3 convolution kernels -- pp_conv_gn_kernel (halo-tile implicit GEMM; round 6: on input that
    is normalised already, NMODE = 2) and pp_gemm_kernel_v2 with template argument XMODE = 1 (tap-major implicit
    GEMM: 8x8 level, strided / upsampling convs) -- in the committed rocprofv3 --stats summary of this benchmark, and the
    build that summary was taken from (`# lib_sha16:` header line).  -> (us per conv launch, lib sha) or (None, None)."""
    import re
    for rel in (KERNEL_STATS_PROFILE, "profiles/r04_kernel_stats.txt"):
        try:
            calls, total_ms, sha = 0, 0.0, None
            for line in open(os.path.join(ROOT, rel)):
                m = re.match(r"#\s*lib_sha16:\s*(\w+)", line)
                if m:
                    sha = m.group(1)
                m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+.*(pp_gemm_kernel_v2<\d+, 160, \d, 2, 1,|pp_conv_gn_kernel<)", line)
                if m:
                    calls += int(m.group(1))
                    total_ms += float(m.group(2))
            if calls:
                return total_ms / calls * 1e3, sha, rel
        except Exception:
            continue
    return None, None, None


def cpu_baseline(budget_s: float = 40.0):
    """SURVEY.md section 8(d) "CPU reference timing": the CPU oracle (kind "port": the plain-PyTorch fp32 restatement of
    the reference's diffusers path -- diffusers itself cannot be installed) on this box's host cores, BASELINE config 1
    END TO END: ppt-v1 loop, 256x256 (latents 32x32), 10-step DDIM, CFG 7.5, batch 1 -- 20 UNet sample-forwards,
    3.6 TFLOP, ~8 s on 16 threads (bounded: stopped and extrapolated after `budget_s`), THEN the headline shape the way
    section 8(d) prescribes for configs 2-5: two warm denoising steps at 64x64 latents (one image + CFG twin, 1.6 TFLOP
    per step, ~4 s each), extrapolated x 50 -> `value` in the metric's unit.  `config1_images_per_s` is the end-to-end
    config-1 figure, `value_by_flop_ratio_from_config1` the old FLOP-ratio scaling of it (kept for comparison)."""
    from oracle import loops as OL, schedulers as OS, sd_modules as OM
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(ncpu, 8))
    with torch.device("meta"):
        o = OM.UNet2DConditionModel(in_channels=9)
    o = o.to_empty(device="cpu").eval()
    g = torch.Generator("cpu").manual_seed(0)

    class Stop(Exception):
        pass

    done = []
    with torch.no_grad():
        for n, p in o.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0) if ("norm" in n and n.endswith("weight")) else p.zero_()
            else:
                p.normal_(0, (1.0 / p[0].numel()) ** 0.5, generator=g)
        lat = torch.randn(1, 4, 32, 32, generator=g)
        mil = torch.randn(1, 4, 32, 32, generator=g) * 0.5
        mask = torch.zeros(1, 1, 32, 32)
        mask[:, :, 8:24, 8:24] = 1.0
        pe = torch.randn(2, 77, 768, generator=g)
        o(torch.randn(2, 9, 8, 8, generator=g), 500, pe)                  # touch weights / warm the thread pool
        # thread count: these conv sizes stop scaling early and many-core hosts get SLOWER with every core in use
        # (measured: 2.1 s per forward on 64 threads of the GPU box against 0.5 s on 8 threads of the build box) --
        # time one forward at 8 / 16 / 32 threads and keep the fastest
        x9 = torch.randn(2, 9, 32, 32, generator=g)
        best = None
        for nt in sorted({min(ncpu, 8), min(ncpu, 16), min(ncpu, 32)}):
            torch.set_num_threads(nt)
            tt = time.perf_counter()
            o(x9, 500, pe)
            tt = time.perf_counter() - tt
            if best is None or tt < best[0]:
                best = (tt, nt)
            if tt > 8.0:
                break
        cores = best[1]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()

        def hook(i, t, latents, eps):
            done.append(time.perf_counter() - t0)
            if done[-1] > budget_s and len(done) >= 2:
                raise Stop()

        try:
            OL.loop_v1(o, OS.DDIMScheduler(), lat, torch.cat([mask] * 2), torch.cat([mil] * 2), pe, 10, 7.5, eps_hook=hook)
            dt, how = time.perf_counter() - t0, "end to end"
        except Stop:
            dt = done[-1] / len(done) * 10
            how = f"first {len(done)} of 10 steps timed ({done[-1]:.1f} s), the rest extrapolated"
        # "two warm steps, extrapolated x N" -- the headline shape itself (latents
        # 64x64, one image + its CFG twin = 1.6 TFLOP per step), one untimed step first, then two timed ones
        lat64 = torch.randn(1, 4, 64, 64, generator=g)
        mil64 = torch.randn(1, 4, 64, 64, generator=g) * 0.5
        mask64 = torch.zeros(1, 1, 64, 64)
        mask64[:, :, 16:48, 16:48] = 1.0
        marks = []

        def hook64(i, t, latents, eps):
            marks.append(time.perf_counter())
            if len(marks) == 3:
                raise Stop()

        try:
            OL.loop_v1(o, OS.DDIMScheduler(), lat64, torch.cat([mask64] * 2), torch.cat([mil64] * 2), pe, 50, 7.5, eps_hook=hook64)
        except Stop:
            pass
        step64 = (marks[2] - marks[0]) / 2 if len(marks) == 3 else None
    scale = (803.4 / 180.1) * (50 / 10)
    by_flops = 1.0 / (dt * scale)
    value = 1.0 / (step64 * 50) if step64 else by_flops
    return {"value": value, "unit": "images/s", "cores": cores, "kind": "port",
            "config1_images_per_s": 1.0 / dt, "value_by_flop_ratio_from_config1": by_flops,
            "headline_step_s": step64,
            "sample": (f"headline shape (ppt-v1, 512x512, CFG 7.5, one image): two warm denoising steps of the fp32 torch CPU "
                       f"oracle on {cores} threads, {step64:.2f} s per step, extrapolated x50 steps -> `value`; "
                       if step64 else "") +
                      f"BASELINE config 1 (256x256, 10-step DDIM, batch 1; 3.6 TFLOP) {how}: {dt:.1f} s "
                      f"(`config1_images_per_s`; scaled by the algorithmic FLOP ratio x{scale:.1f}: `value_by_flop_ratio_from_config1`)"}


HBM_PEAK_GBS = 6290.0            # measured float4 copy, /opt/skills/guides/MI355X_MICROARCH.md ("6.29 TB/s measured")
HBM_SPEC_GBS = 8000.0            # the part's HBM3E specification (fractions are quoted against both)


def roofline_report(pipe, dump_launches=None, peak=MFMA_PEAK_TFLOPS, profile_matches=True) -> dict:
    """`roofline` of the dominant kernel family (the 3x3 convolution launches: pp_conv_gn_kernel -- halo-tile implicit GEMM,
    since round 6 on input normalised by a separate apply -- and pp_gemm_kernel_v2<XMODE = 1>, with their split-K combines;
    the FLOPs counted are the convolution's MACs only).  `achieved` / `frac` are a LIVE measurement of this run: the family's launches replayed as
    their own hipGraph between one HIP event pair (`family_replay_us`).  Next to it: `frac_event` (an event pair around
    every launch of an eager step: includes the eager launch gap) and `frac_profile` (kernel-only average of the
    committed rocprofv3 --stats summary, valid only for the build named in that file).  `hbm_roofline`: the plain
    linear / 1x1 launches (K <= 1280, below the MFMA ridge) and the GroupNorm apply against the measured 6.29 TB/s."""
    from powerpaint_amd import _lib as L
    per, flops, counts, alg_bytes, hbm_bytes = roofline_pass(pipe)
    k = "conv3x3"
    ach_event = flops[k] / 1e12 / (per[k] * 1e-3)
    live_us, n_live = family_replay_us(pipe, ("conv3x3",))
    ach_live = flops[k] / 1e12 / (live_us * 1e-6)
    prof_us, prof_sha, prof_src = profiled_conv_launch_us() if profile_matches else (None, None, None)
    ach_prof = (flops[k] / counts[k]) / (prof_us * 1e-6) / 1e12 if prof_us else None
    traffic, traffic_src = measured_traffic() if profile_matches else (None, None)
    sha = lib_sha16()
    out = {"roofline": {"bound": "mfma", "kernel": "pp_conv_gn_kernel<...,NMODE=2,...> (halo-tile implicit-GEMM 3x3 conv on normalised input) / "
                                                            "pp_gemm_kernel_v2<...,XMODE=1,...> (tap-major implicit GEMM) + split-K combines",
                        "achieved": ach_live, "peak": peak, "unit": "TFLOP/s", "frac": ach_live / peak,
                        "note": "FLOPs = the convolutions' MACs only.  Round 6: the GroupNorm + SiLU of the resnet convs' inputs is "
                                "NOT in these launches any more (rounds 4-5 fused 30 of 44 into the conv loaders): it is the "
                                "`groupnorm_apply` family of hbm_roofline (or rides in the producers' split-K combines); "
                                "the combines of launches with <= 4 splits run inside the conv kernels",
                        "time_base": "live: the family's launches replayed as their own hipGraph between one HIP event "
                                     "pair in this run (launch boundaries and split-K combines included)",
                        "avg_launch_us": live_us / n_live,
                        "achieved_event": ach_event, "frac_event": ach_event / peak,
                        "avg_launch_us_event": per[k] / counts[k] * 1e3,
                        "frac_profile": (ach_prof / peak) if ach_prof else None, "avg_launch_us_profile": prof_us,
                        "profile_source": prof_src, "profile_lib_sha16": prof_sha, "lib_sha16": sha,
                        "profile_is_this_build": bool(prof_sha) and prof_sha == sha,
                        "traffic": traffic, "traffic_unit": "bytes / launch (2 x FETCH_SIZE + WRITE_SIZE)",
                        "traffic_source": traffic_src,
                        "launches_per_step": counts[k], "alg_flop_per_launch": flops[k] / counts[k],
                        "alg_bytes_per_launch": alg_bytes / counts[k]}}
    # ---- HBM-side families, live
    hbm = {}
    lin_bytes, lin_n = 0.0, 0
    for fn, args, name in pipe._loop.program.calls:
        a = getattr(args[0], "_obj", None) if args else None
        if name in ("linear", "conv1x1") and isinstance(a, L.PPGemmArgs):
            # algorithmic bytes: X once, W once, output once, residual operands once (16-bit), row moments ignored
            out_cols = a.N
            nw = a.M // a.rows_per_batch if a.w_batch_stride > 0 else 1      # (one weight matrix per batch item)
            lin_bytes += 2.0 * a.M * a.K + 2.0 * a.N * a.K * nw + 2.0 * a.M * out_cols
            lin_bytes += (2.0 * a.M * a.N if a.res1 else 0.0) + (2.0 * a.M * a.N if a.res2 else 0.0)
            lin_n += 1
    for fam, names, nbytes in (("linear + conv1x1 (plain GEMMs)", ("linear", "conv1x1"), lin_bytes),
                               ("groupnorm_apply", ("groupnorm_apply",), hbm_bytes.get("groupnorm_apply", 0.0))):
        us, n = family_replay_us(pipe, names)
        if us:
            gbs = nbytes / (us * 1e-6) / 1e9
            hbm[fam] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "peak_spec": HBM_SPEC_GBS,
                        "frac_of_spec": round(gbs / HBM_SPEC_GBS, 4), "launches": n, "avg_launch_us": round(us / n, 2),
                        "alg_MB_per_step": round(nbytes / 1e6, 1), "time_base": "live hipGraph replay of the family"}
            if fam.startswith("linear"):
                fl = flops.get("linear", 0.0) + flops.get("conv1x1", 0.0)
                hbm[fam]["tflops"] = round(fl / 1e12 / (us * 1e-6), 1)
    out["hbm_roofline"] = hbm
    fam_us = {}
    for fam in ("attention", "linear_geglu", "ff_fused", "xattn_block", "tfront"):
        us, n = family_replay_us(pipe, (fam,))
        if us:
            fam_us[fam] = {"us_per_step": round(us, 1), "launches": n,
                           "tflops": round(flops.get(fam, 0.0) / 1e12 / (us * 1e-6), 1)}
    out["mfma_families_live"] = fam_us
    out["per_kernel_ms_per_denoise_step"] = {n: round(v, 4) for n, v in sorted(per.items(), key=lambda kv: -kv[1])}
    out["per_kernel_tflops"] = {n: round(flops[n] / 1e12 / (per[n] * 1e-3), 1) for n in flops if per.get(n)}
    out["launches_per_denoise_step"] = len(pipe._loop.program.calls)
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
    ap.add_argument("--denoise-steps", type=int, default=50, help="scheduler steps per image (config 5: 30)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"], help="16-bit compute format (config 5: fp16)")
    ap.add_argument("--scheduler", default=None, choices=["ddim", "dpm", "pndm", "unipc"],
                    help="default: ddim (v1, controlnet), dpm = DPM-Solver++(2M) (v2)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write per-launch HIP-event timings of one step (JSON)")
    ap.add_argument("--dist-timeout", type=float, default=180.0,
                    help="N > 1: seconds the RCCL rendezvous / first all-reduce / weight broadcast may take before the "
                         "rank exits with a message naming the stuck stage (0 = wait forever)")
    args = ap.parse_args()

    rank, world, local = ppdist.init_from_env(timeout_s=args.dist_timeout)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    pipe, nets, bcast_s = build_pipeline(args.config, device, rank, world, dtype=dtype, scheduler=args.scheduler,
                                         bcast_timeout=args.dist_timeout)
    rank_log = ppdist.gather_strings(f"rank {rank}: {ppdist.device_identity(local)}; weight_broadcast_s {bcast_s:.3f}")
    if rank == 0 and world > 1:
        for line in rank_log:
            print(line, file=sys.stderr, flush=True)
    pipe.use_graph = not args.no_graph
    kw = synthetic_inputs(args.config, device, rank, args.per_gpu, args.latent, args.denoise_steps)
    sched_name = {"ddim": "DDIM", "dpm": "DPMSolver++", "pndm": "PNDM", "unipc": "UniPC"}[
        args.scheduler or ("dpm" if args.config == "v2" else "ddim")]

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
    pipe._loop.flush_faults()          # (in-kernel split-K combines: no share left uncombined in any of the timed calls)

    if rank == 0:
        images = args.per_gpu * world * args.steps
        ms_per_step = dt / args.steps * 1e3
        unet_steps = len(pipe.scheduler.timesteps)          # network evaluations per image (PNDM: denoise_steps + 1)
        tab = GFLOP_PER_SAMPLE_BY_LATENT.get(args.latent)
        if tab is not None:
            gf = {"v1": tab["unet9"], "v2": tab["unet4"] + tab["brushnet"],
                  "controlnet": tab["unet9"] + tab["controlnet"]}[args.config]
            tflop_step = gf * 2 * args.per_gpu / 1e3                # CFG doubles the batch
        else:                                                       # other sizes: the launch plan's own count (2 * MAC)
            tflop_step = pipe._loop.program.flops / 1e12
        ms_denoise_step = ms_per_step / unet_steps
        # FLOPs the launch plan EXECUTES per step: below the algorithmic figure where the loop runs the CFG-identical prefix
        # of the networks (conv_in .. first self-attention) on one half of the pair (engine.SDNet.build_step, twin).  Every
        # utilisation figure of this line is computed from executed FLOPs; the algorithmic one is reported beside it.
        prog = getattr(getattr(pipe, "_loop", None), "program", None)
        tflop_exec = prog.flops / 1e12 if prog is not None else tflop_step
        px = args.latent * 8
        res = {
            "metric": ("inpainted images/sec @512x512, 50-step DDIM CFG, batch4/GPU"
                       if (px, args.per_gpu, args.config, args.denoise_steps, sched_name) == (512, 4, "v1", 50, "DDIM")
                       else f"inpainted images/sec @{px}x{px}, {args.denoise_steps}-step {sched_name}, "
                            f"batch{args.per_gpu}/GPU ({args.config})"),
            "value": images / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": {"v1": f"ppt-v1 SD1.5-inpaint UNet (9-ch in), {px}x{px}, {args.denoise_steps}-step {sched_name}, CFG=7.5, {args.dtype}, batch={args.per_gpu}/GPU",
                                    "v2": f"ppt-v2 BrushNet + SD1.5 UNet, {px}x{px}, {args.denoise_steps}-step {sched_name}, CFG=7.5, {args.dtype}, batch={args.per_gpu}/GPU",
                                    "controlnet": f"ppt-v1 + ControlNet, {px}x{px}, {args.denoise_steps}-step {sched_name}, CFG=7.5, {args.dtype}, batch={args.per_gpu}/GPU"}[args.config],
                       "global_batch": args.per_gpu * world, "latent": [4, args.latent, args.latent],
                       "denoise_steps": args.denoise_steps, "network_evaluations": unet_steps, "parallelism": f"dp{world} (image shards, no step collectives)",
                       "hipgraph": not args.no_graph, "weights": "random init (no checkpoints offline)",
                       "weight_broadcast_s": round(bcast_s, 4), "ranks": rank_log},
            "ms_per_denoise_step": ms_denoise_step,
            "algorithmic_tflop_per_step": tflop_step, "executed_tflop_per_step": tflop_exec,
            "unet_step_mfma_util": tflop_exec / (ms_denoise_step * 1e-3) / MFMA_PEAK_TFLOPS,
            "unet_step_mfma_util_algorithmic": tflop_step / (ms_denoise_step * 1e-3) / MFMA_PEAK_TFLOPS,
            "cfg_twin_prefix": bool(getattr(getattr(getattr(pipe, "_loop", None), "rt", None), "twin", False)),
            "launches_per_denoise_step": len(pipe._loop.program.calls) if getattr(pipe, "_loop", None) is not None
            and getattr(pipe._loop, "program", None) is not None else None,
        }
        if not args.no_roofline:
            try:
                headline = (args.config, args.latent, args.per_gpu, args.dtype) == ("v1", 64, 4, "bf16")
                res.update(roofline_report(pipe, args.dump_launches, profile_matches=headline))
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
