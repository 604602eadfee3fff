"""The fused denoising loop: one launch program per step, replayed eagerly or as a hipGraph.

Per step (reference: /root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:988-1041,
pipeline_PowerPaint_Brushnet_CA.py:1384-1466, pipeline_PowerPaint_ControlNet.py:1663-1741):
    (round 5: the next two lines and the zeroing of the GroupNorm accumulators are ONE launch, pp_step_head)
    temb_all     <- temb_table[step]                                  (one row copy, pp_embed_splice indexed by the device
                                                                       step counter: the sinusoid -> time_embedding MLP ->
                                                                       every resnet's time_emb_proj chain depends on t
                                                                       alone, unet_2d_condition.py:1155-1156, so it is run
                                                                       once per schedule row when the timesteps change,
                                                                       not four launches per network and step)
    x_in[:, 0:4] <- cat([latents] * 2)                                (pp_nchw_to_nhwc with batch wrap, no copy of the cat)
    [BrushNet | ControlNet forward]                                   (own launch plan, residuals stay in HBM as NHWC)
    eps          <- UNet(x_in, t, ctx, residuals)
    latents      <- scheduler.step(eps_u + g (eps_c - eps_u), t, latents)   (pp_cfg_sched_step, fp32)
    [latents     <- (1 - m) add_noise(x0, noise, t_next) + m latents]       (pp_latent_blend; ppt-v1 with a 4-channel UNet)
    step         <- step + 1                                          (pp_step_advance; round 5: by the last block of
                                                                       pp_cfg_sched_step when nothing behind it reads the counter)
No host->device traffic and no host synchronisation inside the loop.

A scheduler that is not one of powerpaint_amd.schedulers (any object with the diffusers protocol the reference
duck-types: `.timesteps`, `.scale_model_input`, `.step(...)`, e.g. the UniPC scheduler app.py:197 installs) is driven
the way the reference drives it: the network part of the step is the same captured launch program, the guidance
combine and the scheduler's own `.step` run as that object's tensor code on the device, once per step.
"""
import os
from typing import Callable, List, Optional

import torch

from .. import _lib as L
from ..engine import Plan
from ..schedulers import _SchedulerBase, variance_noise


def _temb_table_enabled() -> bool:
    """(lab) PP_LAB=1 PP_TEMB_TABLE=0: every step runs the time-embedding chain again."""
    return not (os.environ.get("PP_LAB") == "1" and os.environ.get("PP_TEMB_TABLE", "1") == "0")


class DenoiseLoop:
    def __init__(self, unet, scheduler, side=None, side_kind: Optional[str] = None):
        self.foreign = not isinstance(scheduler, _SchedulerBase)     # duck-typed scheduler: its own .step per step
        if self.foreign and not (hasattr(scheduler, "step") and hasattr(scheduler, "timesteps")):
            raise TypeError(f"{type(scheduler).__name__} does not follow the scheduler protocol (.timesteps, .step)")
        self.unet, self.scheduler, self.side, self.side_kind = unet, scheduler, side, side_kind
        self.lib = L.lib()
        self.program: Optional[Plan] = None
        self.graph = None
        self._key = None
        self._graph_scale = None      # side-branch conditioning scale baked into the captured graph (by-value kernel arg)
        self.latents: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------
    def bind(self, latents_shape, do_cfg: bool, guidance_scale: float, prompt_embeds, prompt_embeds_side=None,
             static_inputs=(), side_static_inputs=(), controlnet_cond=None, side_scale: float = 1.0,
             guess_mode: bool = False, eta: float = 0.0, generator=None, noise_dtype=torch.float32, blend=None):
        """Compile the per-step program.  latents_shape = (B, 4, h, w) of the *un-duplicated* latents.
        blend = (image_latents [1,4,h,w], mask [1,1,h,w], noise [B,4,h,w]): the `num_channels_unet == 4` branch of the
        ppt-v1 loop body (pipeline_PowerPaint.py:1025-1039) -- after every scheduler step the unmasked region is
        replaced by the init image's latents noised to the NEXT timestep.
        eta > 0 with this package's DDIM (the only scheduler whose `step` takes it, pipeline_PowerPaint.py:536-551):
        the step program gains `latents += std_dev_t * z`; `run` draws z from `generator` before every step, in
        `noise_dtype`, exactly where the reference's `scheduler.step` calls `randn_tensor`.
        guess_mode (pipeline_PowerPaint_Brushnet_CA.py:1394-1425, pipeline_PowerPaint_ControlNet.py:1669-1702): the
        side network sees only the conditional half of a CFG pair (its inputs -- `prompt_embeds_side`, conditioning
        latents / control image -- arrive un-duplicated) with its residual scales log-spaced from 0.1 to 1; the
        unconditional half of the UNet batch gets zeros."""
        dev = self.unet.device
        B, Cl, h, w = latents_shape
        Be = 2 * B if do_cfg else B
        sch = self.scheduler
        if self.latents is None or tuple(self.latents.shape) != tuple(latents_shape):
            self.latents = torch.zeros(latents_shape, dtype=torch.float32, device=dev)
        lat = self.latents
        prog = Plan()
        lib = self.lib
        cin = self.unet.config.in_channels
        side_rt = None
        wiring_kw = {}
        self._eta = float(eta) if (not self.foreign and sch.kind == 0) else 0.0
        self._gen, self._noise_dtype = generator, noise_dtype
        self._extra_step_kwargs = {}
        if self.foreign:                                     # prepare_extra_step_kwargs for a duck-typed scheduler
            import inspect
            names = set(inspect.signature(sch.step).parameters)
            self._extra_step_kwargs = {k: v for k, v in (("eta", eta), ("generator", generator)) if k in names}
        else:
            sch.set_eta(self._eta)
        self._blend = None
        if blend is not None:
            # loop-owned fp32 copies at stable addresses (a captured graph reads them); first image / first mask only
            x0, mk, nz = blend
            want = ((1, Cl, h, w), (1, 1, h, w), tuple(latents_shape))
            bufs = getattr(self, "_blend_bufs", None)
            if bufs is None or tuple(tuple(b.shape) for b in bufs) != want:
                bufs = self._blend_bufs = tuple(torch.zeros(sh, dtype=torch.float32, device=dev) for sh in want)
            for b_, v in zip(bufs, (x0[:1], mk[:1], nz)):
                b_.copy_(v.to(device=dev, dtype=torch.float32).reshape(b_.shape))
            self._blend = bufs
        half = bool(guess_mode and do_cfg)                   # side network on the conditional half only
        # guess mode scales the n residuals by logspace(-1, 0, n) * conditioning_scale (BrushNet_CA.py:905-928): `run`'s
        # per-step scalar schedule is expanded the same way before it is patched into the zero-conv launches
        self._guess_ramp = None
        if guess_mode and self.side is not None and not self.side.config.global_pool_conditions:
            self._guess_ramp = [float(v) for v in torch.logspace(-1, 0, len(self.side.net._zero_conv_specs()))]
        Bs = B if half else Be

        # The CFG pair is built here, as `cat([latents] * 2)` (pipeline_PowerPaint.py:990): where every other network input
        # is CFG-duplicated as well, the two halves of a forward pass are identical until the prompt enters and the
        # networks run that prefix once (`twin`, SDNet.build_step).  A static input counts as duplicated when it arrives
        # with the un-duplicated batch (load_input copies it to both halves) or when its halves compare equal.
        def halves_equal(t):
            if t is None or not torch.is_tensor(t):
                return False
            if t.shape[0] == B:
                return True
            return t.shape[0] == 2 * B and bool(torch.equal(t[:B], t[B:]))

        twin_u = bool(do_cfg) and all(halves_equal(t) for t, _ in static_inputs)
        # (the control image goes through set_cond, which accepts the un-duplicated batch as well and copies it to both halves)
        twin_s = bool(do_cfg) and not half and all(halves_equal(t) for t, _ in side_static_inputs) and \
            (self.side_kind == "brushnet" or halves_equal(controlnet_cond))
        if self.side is not None and self.side.dtype != self.unet.dtype:
            # the fused loop hands the side network's residuals to the UNet as raw NHWC arena pointers, every step: both
            # must store activations in the same 16-bit format (the reference raises a dtype error in its first conv)
            raise L.PPError(f"{type(self.side).__name__} computes in {self.side.dtype}, the UNet in {self.unet.dtype}: "
                            f"load both with the same torch_dtype")
        if self.side is not None:
            if half and prompt_embeds_side.shape[0] == Be:
                prompt_embeds_side = prompt_embeds_side.chunk(2)[1]
            if self.side_kind == "brushnet":
                side_rt = self.side.prepare((Bs, Cl, h, w), prompt_embeds_side, side_scale, bool(guess_mode), half,
                                            twin=twin_s)
                d, m, u = self.side.outputs()
                wiring_kw = dict(down_block_add_samples=d, mid_block_add_sample=m, up_block_add_samples=u)
            else:
                side_rt = self.side.prepare((Bs, Cl, h, w), prompt_embeds_side, controlnet_cond, side_scale,
                                            bool(guess_mode), half, twin=twin_s)
                d, m = self.side.outputs()
                wiring_kw = dict(down_block_additional_residuals=d, mid_block_additional_residual=m)
        rt = self.unet.prepare((Be, cin, h, w), prompt_embeds, twin=twin_u, **wiring_kw)
        # one-time (per call) static channels of the UNet / side inputs
        rt.load_input(list(static_inputs))
        if side_rt is not None and side_static_inputs:
            side_rt.load_input(list(side_static_inputs))

        if self.foreign:
            # network input (scaled by the scheduler per step), timestep table and step counter owned by the loop
            tsv = torch.as_tensor(sch.timesteps).detach().to(dev, torch.float32).contiguous()
            if getattr(self, "_f_ts", None) is None or self._f_ts.shape != tsv.shape:
                self._f_ts = tsv
                self._f_step = torch.zeros(1, dtype=torch.int32, device=dev)
                self._f_x = torch.zeros(latents_shape, dtype=torch.float32, device=dev)
            else:
                self._f_ts.copy_(tsv)
            if tuple(self._f_x.shape) != tuple(latents_shape):
                self._f_x = torch.zeros(latents_shape, dtype=torch.float32, device=dev)
            ts, step, mp, kind, src = self._f_ts, self._f_step, None, -1, self._f_x
            self._do_cfg, self._g = bool(do_cfg), float(guidance_scale)
        else:
            ts, step = sch.timesteps_f32(), sch.step_counter()
            mp = sch.m_prev(lat) if sch.kind >= 1 else None  # scheduler state: DPM m_{i-1}; PNDM history + saved sample
            kind, src = sch.kind, lat
        key = (tuple(latents_shape), bool(do_cfg), bool(guess_mode), self._eta > 0, float(guidance_scale), id(rt.step_plan),
               id(side_rt.step_plan) if side_rt is not None else None, kind, ts.data_ptr(), step.data_ptr(),
               0 if self.foreign else sch.coef_table().data_ptr(), mp.data_ptr() if mp is not None else 0, src.data_ptr(),
               _temb_table_enabled(), int(ts.numel()), tuple(getattr(r.net.params, "version", 0) for r in (rt, side_rt) if r is not None),
               tuple(b.data_ptr() for b in self._blend) if self._blend is not None else None,
               sch.renoise_table().data_ptr() if (self._blend is not None and not self.foreign) else 0)
        if key == self._key and self.program is not None:
            # The conditioning scale is a by-value argument of the zero-conv launches: `prepare` has patched the launch
            # records (eager runs see it), but a graph captured with another value still carries the old one.
            if self.graph is not None and self._scale_now() != self._graph_scale:
                self.graph = None
            return self
        self._key = key
        hw = h * w
        mod = B if do_cfg else 0
        self._temb = {}
        head_skip = {}
        for r in ([side_rt] if side_rt is not None else []) + [rt]:
            info = self._temb_split(r, ts) if _temb_table_enabled() else None
            x = r.lay["x_in"]
            nb_r = Bs if r is side_rt else Be                    # (guess mode: the side network takes `latents` as is)
            calls = r.step_plan.calls
            zero = calls[0] if (calls and calls[0][2] == "zero_u64") else None
            if info is not None and zero is not None:
                # time-embedding row + network input + accumulator zeroing: ONE launch at the head of the step (pp_step_head)
                self._temb[id(r)] = info
                head_skip[id(r)] = {0}
                prog.add("step_head", lib.pp_step_head, info["table"].data_ptr(), step.data_ptr(), info["out"], info["total"],
                         src.data_ptr(), nb_r, Cl, hw, mod if nb_r != B else 0, x.ptr, x.C, 0, L.dtype_code(r.net.dtype),
                         zero[1][0], zero[1][1])
                continue
            if info is not None:
                self._temb[id(r)] = info
                prog.add("temb_row", lib.pp_embed_splice, info["table"].data_ptr(), None, step.data_ptr(), info["out"], 1,
                         info["total"] * 4)
            else:
                prog.add("step_select_t", lib.pp_step_select_t, ts.data_ptr(), step.data_ptr(), r.lay["t_dev"])
            prog.add("nchw_to_nhwc", lib.pp_nchw_to_nhwc, src.data_ptr(), 0, nb_r, Cl, hw, mod if nb_r != B else 0,
                     x.ptr, x.C, 0, L.dtype_code(r.net.dtype))
        for r in ([side_rt] if side_rt is not None else []) + [rt]:
            skip = set(self._temb[id(r)]["idx"]) if id(r) in self._temb else set()
            skip |= head_skip.get(id(r), set())
            prog.calls += [c for i, c in enumerate(r.step_plan.calls) if i not in skip]
            prog.flops += r.step_plan.flops
        # (round 5) where pp_cfg_sched_step is the step's last reader of the counter, its last block moves the counter on:
        # no pp_step_advance launch behind it
        fold_advance = (not self.foreign) and not (self._eta > 0) and self._blend is None and \
            not (os.environ.get("PP_LAB") == "1" and os.environ.get("PP_FOLD_ADVANCE", "1") == "0")     # (lab: the two launches)
        if not self.foreign:
            if fold_advance and (getattr(self, "_ticket", None) is None or self._ticket.device != lat.device):
                self._ticket = torch.zeros(1, dtype=torch.int32, device=lat.device)
            prog.add("cfg_sched_step", lib.pp_cfg_sched_step, rt.outputs["eps"], int(do_cfg), float(guidance_scale),
                     lat.data_ptr(), mp.data_ptr() if mp is not None else None, lat.numel(), sch.kind,
                     sch.coef_table().data_ptr(), step.data_ptr(), self._ticket.data_ptr() if fold_advance else None)
            if self._eta > 0:
                if getattr(self, "_var_noise", None) is None or tuple(self._var_noise.shape) != tuple(latents_shape):
                    self._var_noise = torch.zeros(latents_shape, dtype=torch.float32, device=dev)
                prog.add("ddim_variance_noise", lib.pp_ddim_variance_noise, lat.data_ptr(), self._var_noise.data_ptr(),
                         lat.numel(), sch.coef_table().data_ptr(), step.data_ptr())
            if self._blend is not None:
                x0, mk, nz = self._blend
                prog.add("latent_blend", lib.pp_latent_blend, lat.data_ptr(), x0.data_ptr(), mk.data_ptr(), nz.data_ptr(),
                         sch.renoise_table().data_ptr(), step.data_ptr(), B, Cl, hw)
        if not fold_advance:
            prog.add("step_advance", lib.pp_step_advance, step.data_ptr())
        self.program = prog
        self.rt, self.side_rt = rt, side_rt
        self.graph = None
        self._keep = (ts, step, mp, lat)
        return self

    # ---- time-embedding table
    def _temb_split(self, r, ts):
        """The four leading time-embedding launches of a network's step plan (`timestep_embedding` + three
        `linear_skinny`: sinusoid, time_embedding.linear_1/2, all time_emb_proj rows at once) and where their result
        lands; plus a [rows of the schedule][temb_total] fp32 table at a stable address.  None = unexpected layout."""
        calls = r.step_plan.calls
        idx = [i for i, c in enumerate(calls) if c[2] in ("timestep_embedding", "linear_skinny")]
        if len(idx) != 4 or idx != list(range(idx[0], idx[0] + 4)) or calls[idx[0]][2] != "timestep_embedding":
            return None
        a = calls[idx[-1]][1]
        out, total = int(a[6]), int(a[5])
        tabs = self.__dict__.setdefault("_temb_tables", {})
        k = (id(r), int(ts.numel()), total)
        if k not in tabs:
            tabs[k] = dict(table=torch.zeros(int(ts.numel()), total, dtype=torch.float32, device=ts.device), ts=None,
                           plan=None)
        ent = tabs[k]
        pver = getattr(r.net.params, "version", 0)
        if ent["plan"] is not r.step_plan or ent.get("pver") != pver:     # new plan / weights rewritten in place: stale rows
            ent["plan"], ent["ts"], ent["pver"], ent["hkey"] = r.step_plan, None, pver, None
        return dict(idx=idx, out=out, total=total, table=ent["table"], ent=ent, rt=r)

    def _fill_temb_tables(self):
        """(Re)compute the table rows when the timestep table changed since they were made: per row the network's own
        four launches, eagerly, with the step counter pointing at that row."""
        ts, step = self._keep[0], self._keep[1]
        for info in getattr(self, "_temb", {}).values():
            ent, r = info["ent"], info["rt"]
            # (this package's schedulers keep the timesteps on the host: compare there -- a device compare synchronises)
            host = getattr(self.scheduler, "_ts_host", None) if not self.foreign else None
            hkey = tuple(host.tolist()) if host is not None else None
            if hkey is not None and ent.get("hkey") == hkey:
                continue
            if hkey is None and ent["ts"] is not None and ent["ts"].shape == ts.shape and torch.equal(ent["ts"], ts):
                continue
            stream = torch.cuda.current_stream().cuda_stream
            saved = step.clone()
            view = r.arena.view(info["out"], (info["total"],), torch.float32)
            calls = [r.step_plan.calls[i] for i in info["idx"]]
            for i in range(int(ts.numel())):
                step.fill_(i)
                L.check(self.lib.pp_step_select_t(ts.data_ptr(), step.data_ptr(), r.lay["t_dev"], stream), "pp_step_select_t")
                for fn, args, name in calls:
                    L.check(fn(*args, stream), name)
                info["table"][i].copy_(view)
            step.copy_(saved)
            ent["ts"] = ts.clone()
            ent["hkey"] = hkey

    def _side_scale(self, v):
        """One entry of `run`'s scale schedule as the side runtime stores it (guess mode: the per-residual ramp)."""
        ramp = getattr(self, "_guess_ramp", None)
        return tuple(r * v for r in ramp) if ramp is not None else v

    def _scale_now(self):
        sc = getattr(self.side_rt, "_scale", None) if getattr(self, "side_rt", None) is not None else None
        return tuple(sc) if isinstance(sc, (list, tuple)) else sc

    def capture(self):
        """Capture one step into a hipGraph (torch.cuda.CUDAGraph).  The captured launches read the step counter from
        device memory, so the same graph serves every step."""
        self._graph_scale = self._scale_now()
        torch.cuda.synchronize()
        step = self._keep[1]
        saved_step = step.clone()
        saved_lat = self.latents.clone()
        saved_m = self._keep[2].clone() if self._keep[2] is not None else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.program.run(s.cuda_stream)        # warm-up outside capture (func attributes, lazy module load)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.program.run(torch.cuda.current_stream().cuda_stream)
        # undo the warm-up's side effects
        step.copy_(saved_step)
        self.latents.copy_(saved_lat)
        if saved_m is not None:
            self._keep[2].copy_(saved_m)
        torch.cuda.synchronize()
        self.graph = g

    def run(self, latents: torch.Tensor, num_steps: int, use_graph: bool = True,
            callback: Optional[Callable] = None, timesteps=None, scale_schedule: Optional[List[float]] = None):
        """Runs `num_steps` steps starting from step counter 0.  Returns the fp32 latents tensor (owned by the loop)."""
        if self.foreign:
            return self._run_foreign(latents, num_steps, use_graph, callback, timesteps, scale_schedule)
        self.scheduler.reset()
        if getattr(self, "_ticket", None) is not None:
            self._ticket.zero_()       # (an aborted launch must not leave the step counter's ticket mid-count)
        if self.scheduler.kind >= 1:
            self._keep[2].zero_()
        self._fill_temb_tables()
        self.latents.copy_(latents.to(self.latents.device, torch.float32))
        varying = scale_schedule is not None and len(set(scale_schedule)) > 1
        if not varying and scale_schedule and self.side_rt is not None and \
                self._scale_now() != self._side_scale(scale_schedule[0]):
            self.side_rt._patch_scale(self._side_scale(scale_schedule[0]))   # (a previous call may have left a windowed value)
        if use_graph and not varying and (self.graph is None or self._graph_scale != self._scale_now()):
            self.capture()
        stream = torch.cuda.current_stream().cuda_stream
        for i in range(num_steps):
            if varying and self.side_rt is not None:
                self.side_rt._patch_scale(self._side_scale(scale_schedule[i]))
            if self._eta > 0:         # this step's variance noise (stream-ordered before the step that consumes it)
                self._var_noise.copy_(variance_noise(self._var_noise.shape, self._gen, self._var_noise.device,
                                                     self._noise_dtype))
            if use_graph and not varying:
                self.graph.replay()
            else:
                self.program.run(stream)
            if callback is not None:
                callback(i, timesteps[i] if timesteps is not None else None, self.latents)
        self._check_faults()
        return self.latents

    def _check_faults(self, blocking: bool = False):
        """Where a plan combines split-K in-kernel (launches of 2 / 4 co-resident splits: pp_gemm_combine_ctr_bytes()): the
        plan's fault word is examined WITHOUT a synchronisation -- copied to pinned memory behind this call's work, read by a
        later call (NetRuntime.check_faults).  `flush_faults()` is the blocking form for a caller that synchronises anyway."""
        for r in (getattr(self, "rt", None), getattr(self, "side_rt", None)):
            if r is not None and r.combines_in_kernel():
                r.check_faults(blocking=blocking)

    def flush_faults(self):
        """Synchronise and raise if any in-kernel split-K combine since the last check reported a fault."""
        self._check_faults(blocking=True)

    def _run_foreign(self, latents, num_steps, use_graph, callback, timesteps, scale_schedule):
        """Duck-typed scheduler: network part = the captured program (eps lands in the UNet runtime's fp32 output), then
        the guidance combine and `scheduler.step` exactly as the reference's loop body does
        (pipeline_PowerPaint.py:1018-1023)."""
        sch = self.scheduler
        tl = timesteps if timesteps is not None else sch.timesteps
        # The timestep table the captured program indexes holds exactly the timesteps this run steps through: with
        # `strength < 1` the pipelines hand over `scheduler.timesteps[t_start:]` (get_timesteps,
        # pipeline_PowerPaint.py:713-720), and the networks must see those -- not the head of the full schedule that
        # `bind` uploaded -- while `scheduler.step` receives the same values below.
        tsv = torch.as_tensor(tl).detach().to(self._f_ts.device, torch.float32).reshape(-1)
        if tsv.numel() > self._f_ts.numel():
            raise L.PPError(f"{tsv.numel()} timesteps for a loop bound to a schedule of {self._f_ts.numel()}")
        if num_steps > tsv.numel():
            raise L.PPError(f"{num_steps} steps requested, {tsv.numel()} timesteps given")
        self._f_ts[:tsv.numel()].copy_(tsv)
        self._fill_temb_tables()
        self._f_step.zero_()
        lat = latents.to(self.latents.device, torch.float32).clone()
        varying = scale_schedule is not None and len(set(scale_schedule)) > 1
        stream = torch.cuda.current_stream().cuda_stream
        for i in range(num_steps):
            t = tl[i]
            x = sch.scale_model_input(lat, t) if hasattr(sch, "scale_model_input") else lat
            self._f_x.copy_(x)
            if varying and self.side_rt is not None:
                self.side_rt._patch_scale(self._side_scale(scale_schedule[i]))
            if use_graph and not varying:
                if self.graph is None or self._graph_scale != self._scale_now():
                    self.capture()
                    self._f_step.fill_(i)
                self.graph.replay()
            else:
                self.program.run(stream)
            eps = self.rt.eps_tensor()
            if self._do_cfg:
                eu, ec = eps.chunk(2)
                eps = eu + self._g * (ec - eu)
            lat = sch.step(eps, t, lat, **self._extra_step_kwargs, return_dict=False)[0].to(torch.float32)
            if self._blend is not None:                       # pipeline_PowerPaint.py:1025-1039, with the scheduler's own add_noise
                x0, mk, nz = self._blend
                proper = x0
                if i < len(tl) - 1:
                    proper = sch.add_noise(x0, nz, torch.as_tensor(tl[i + 1]).reshape(1).to(x0.device)).to(torch.float32)
                lat = (1 - mk) * proper + mk * lat
            if callback is not None:
                callback(i, t, lat)
        self.latents.copy_(lat)
        self._check_faults()
        return self.latents
