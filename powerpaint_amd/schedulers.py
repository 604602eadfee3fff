"""Schedulers of the hot path with the diffusers duck-type the reference pipelines rely on
(/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:536-551,906,993,1023,642;
 pipeline_PowerPaint_Brushnet_CA.py:87-128,1391,1449,969): `.set_timesteps`, `.timesteps`, `.order`,
`.init_noise_sigma`, `.scale_model_input`, `.step(..., return_dict=False)[0]`, `.add_noise`, `.config.steps_offset`.

The arithmetic restates diffusers==0.27.0 `DDIMScheduler` / `DPMSolverMultistepScheduler` (pinned at
/root/reference/requirements/requirements.txt:3, not vendored).  Per-step coefficients are computed on the host in
fp32 torch exactly in the library's operation order and uploaded as a device table `coef[step][8]`; the tensor math
(CFG combine + step) runs in the fused HIP kernel `pp_cfg_sched_step` on fp32 latents.  `.step()` itself launches
that kernel, so a foreign loop calling `scheduler.step` still runs on the HIP path.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib as L


def _betas(T, beta_start, beta_end):
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2


# Options of the diffusers schedulers that change the arithmetic and that the fused step kernel does not implement: a
# config carrying another value is REFUSED (a silently dropped key would run and give wrong images).
_ONLY = {"prediction_type": ("epsilon",), "beta_schedule": ("scaled_linear",), "clip_sample": (False,),
         "thresholding": (False,), "rescale_betas_zero_snr": (False,), "use_karras_sigmas": (False,),
         "use_lu_lambdas": (False,), "euler_at_final": (False,), "variance_type": (None,),
         "trained_betas": (None,), "algorithm_type": ("dpmsolver++",), "solver_type": ("midpoint", "bh2"),
         "final_sigmas_type": ("zero",), "lower_order_final": (True,), "lambda_min_clipped": (-float("inf"),),
         "timestep_spacing": ("leading", "linspace", "trailing")}


def variance_noise(shape, generator, device, dtype) -> torch.Tensor:
    """diffusers' `randn_tensor` for the per-step DDIM variance noise: drawn on the generator's device (a CPU generator
    gives the same numbers whatever the compute device), returned as contiguous fp32 on `device`."""
    gdev = generator.device if isinstance(generator, torch.Generator) else torch.device(device)
    if isinstance(generator, (list, tuple)):
        z = torch.cat([torch.randn((1,) + tuple(shape[1:]), generator=g, device=g.device, dtype=dtype).to(device)
                       for g in generator])
    else:
        z = torch.randn(tuple(shape), generator=generator, device=gdev, dtype=dtype).to(device)
    return z.to(torch.float32).contiguous()


def _check_config(cls_name, cfg, keys=None):
    """`keys`: restrict the check to the options the target class's diffusers constructor names (a donor config's other
    keys never reach it in `from_config`)."""
    for k, ok in _ONLY.items():
        if keys is not None and k not in keys:
            continue
        if k in cfg and cfg[k] not in ok:
            raise L.PPError(f"{cls_name}: {k}={cfg[k]!r} is not implemented on the HIP path (supported: {ok})")


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0
    kind = -1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, **cfg):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, **cfg)
        self.betas = _betas(num_train_timesteps, beta_start, beta_end)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.timesteps = None
        self.num_inference_steps = None
        self._coef_dev = None
        self._ts_dev = None
        self._step_dev = None
        self._m_prev = None
        self._device = None
        self._begin = 0

    def set_begin_index(self, begin_index: int = 0):
        """diffusers' `set_begin_index`: the loop enters the schedule at row `begin_index` (`strength < 1`: the pipelines'
        `get_timesteps` hands `scheduler.timesteps[t_start:]` to the loop, pipeline_PowerPaint.py:713-720).  The
        multistep warm-up restarts there -- first-order first step of DPM-Solver++, PLMS start-up, UniPC order ramp and
        no corrector on the first step -- exactly what the library's step-index / counter logic does when the first
        `step()` call carries a later timestep.  `set_timesteps` resets it to 0."""
        if self.timesteps is None:
            raise L.PPError("set_begin_index before set_timesteps")
        if not 0 <= int(begin_index) < len(self._ts_host):
            raise ValueError(f"begin_index {begin_index} outside the schedule of {len(self._ts_host)} entries")
        self._begin = int(begin_index)
        self._fill_table()
        self.timesteps = self._ts_host.clone()
        self._upload(self._device)

    @property
    def begin_index(self):
        return self._begin

    @classmethod
    def from_config(cls, config, **kw):
        """`Scheduler.from_config(other.config)` / a `scheduler_config.json` dict: the keys this class's constructor
        names are taken over, the rest (other schedulers' options, `_class_name`, ...) ignored."""
        import inspect
        src = dict(config) if isinstance(config, dict) else dict(vars(config))
        src.update(kw)
        _check_config(cls.__name__, src)
        names = set(inspect.signature(cls.__init__).parameters) - {"self", "kw", "cfg"}
        return cls(**{k: v for k, v in src.items() if k in names})

    # -- device state for the fused kernel
    def scale_model_input(self, sample, timestep=None):
        return sample

    def _upload(self, device):
        self._device = torch.device(device) if device is not None else None
        if self._device is not None and self._device.type == "cuda":
            same = (self._coef_dev is not None and self._coef_dev.shape == self._coef.shape
                    and self._coef_dev.device.type == "cuda"
                    and (self._device.index is None or self._coef_dev.device.index == self._device.index))
            if same:   # keep device addresses stable across calls so captured graphs stay valid
                self._coef_dev.copy_(self._coef)
                self._ts_dev.copy_(self.timesteps.to(torch.float32))
                self._step_dev.fill_(self._begin)
            else:
                self._coef_dev = self._coef.to(self._device).contiguous()
                self._ts_dev = self.timesteps.to(self._device, torch.float32).contiguous()
                self._step_dev = torch.full((1,), self._begin, dtype=torch.int32, device=self._device)
            self.timesteps = self.timesteps.to(self._device)
        if self._m_prev is not None:
            self._m_prev.zero_()

    def coef_table(self) -> torch.Tensor:
        return self._coef_dev

    def timesteps_f32(self) -> torch.Tensor:
        return self._ts_dev

    def step_counter(self) -> torch.Tensor:
        return self._step_dev

    def renoise_table(self) -> torch.Tensor:
        """[rows][2] fp32 on the scheduler's device, indexed by the step counter like the coefficient table: what
        `add_noise(x0, noise, timesteps[i + 1])` multiplies x0 and the noise by AFTER step i -- (sqrt(abar), sqrt(1-abar))
        of the next timestep, (1, 0) after the last one.  The ppt-v1 loop with a 4-channel UNet re-noises the known
        region with it every step (pipeline_PowerPaint.py:1025-1039; pp_latent_blend)."""
        ts = self._ts_host.long()
        a = self.alphas_cumprod[ts[1:]].to(torch.float32)
        tab = torch.stack([torch.cat([a ** 0.5, torch.ones(1)]), torch.cat([(1 - a) ** 0.5, torch.zeros(1)])], 1)
        dev = self._device if self._device is not None else "cpu"
        cur = getattr(self, "_renoise_dev", None)
        want = torch.device(dev)
        if cur is not None and cur.shape == tab.shape and cur.device.type == want.type and \
                (want.index is None or cur.device.index == want.index):
            cur.copy_(tab)            # (stable address: a captured step graph reads it)
        else:
            self._renoise_dev = tab.to(dev).contiguous()
        return self._renoise_dev

    state_slots = 1          # fp32 copies of the latents the step kernel keeps between steps (DPM: 1, PNDM: 5)

    def m_prev(self, like: torch.Tensor) -> torch.Tensor:
        shape = (self.state_slots,) + tuple(like.shape) if self.state_slots > 1 else tuple(like.shape)
        if self._m_prev is None or tuple(self._m_prev.shape) != shape or self._m_prev.device != like.device:
            self._m_prev = torch.zeros(shape, dtype=torch.float32, device=like.device)
        return self._m_prev

    def reset(self):
        if self._step_dev is not None:
            self._step_dev.fill_(self._begin)
        if self._m_prev is not None:
            self._m_prev.zero_()

    def _index_of(self, timestep) -> int:
        t = int(timestep)
        idx = (self._ts_host == t).nonzero()
        if len(idx) == 0:
            raise ValueError(f"timestep {t} is not in the schedule")
        return int(idx[0])

    def set_eta(self, eta: float = 0.0):
        """`eta` of the pipelines (pipeline_PowerPaint.py:736-745): only DDIM's step takes it, the other schedulers
        ignore it -- as `prepare_extra_step_kwargs` never hands it to them."""
        return self

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict: bool = True, **kw):
        """x_t -> x_{t-1} on the HIP kernel (fp32 math).  Returns a NEW tensor in sample's dtype.  `eta` / `generator`:
        stochastic DDIM (kind 0 only), the variance noise drawn like diffusers' `randn_tensor(model_output.shape, ...)`."""
        if not sample.is_cuda:
            raise L.PPError("scheduler.step needs CUDA tensors: the step runs in the HIP kernel, no CPU fallback")
        if self.kind == 0 and float(eta) != self.eta:
            self.set_eta(eta)
        i = self._index_of(timestep)
        x = sample.detach().to(torch.float32).contiguous().clone()
        e = model_output.detach().to(torch.float32).contiguous()
        step = torch.full((1,), i, dtype=torch.int32, device=x.device)
        mp = self.m_prev(x) if self.kind >= 1 else None
        L.check(L.lib().pp_cfg_sched_step(e.data_ptr(), 0, 0.0, x.data_ptr(), mp.data_ptr() if mp is not None else None,
                                           x.numel(), self.kind, self._coef_dev.data_ptr(), step.data_ptr(), None,
                                           torch.cuda.current_stream().cuda_stream), "pp_cfg_sched_step")
        if self.kind == 0 and self.eta > 0:
            z = variance_noise(model_output.shape, generator, x.device, model_output.dtype)
            L.check(L.lib().pp_ddim_variance_noise(x.data_ptr(), z.data_ptr(), x.numel(), self._coef_dev.data_ptr(),
                                                    step.data_ptr(), torch.cuda.current_stream().cuda_stream),
                    "pp_ddim_variance_noise")
        out = x.to(sample.dtype)
        if not return_dict:
            return (out,)
        return SimpleNamespace(prev_sample=out)

    def add_noise(self, original_samples, noise, timesteps):
        a = self.alphas_cumprod.to(original_samples.device)[timesteps.to(original_samples.device).long()]
        sa = (a ** 0.5).flatten().to(original_samples.dtype)
        s1 = ((1 - a) ** 0.5).flatten().to(original_samples.dtype)
        while sa.dim() < original_samples.dim():
            sa, s1 = sa.unsqueeze(-1), s1.unsqueeze(-1)
        return sa * original_samples + s1 * noise


class DDIMScheduler(_SchedulerBase):
    """epsilon prediction, `leading` spacing, steps_offset = 1, set_alpha_to_one = False, no clipping; eta in [0, 1]."""
    kind = 0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, timestep_spacing="leading", **kw):
        _check_config("DDIMScheduler", kw)
        if timestep_spacing != "leading":
            raise L.PPError(f"DDIMScheduler: timestep_spacing={timestep_spacing!r} is not implemented (leading only)")
        super().__init__(num_train_timesteps, beta_start, beta_end, steps_offset=steps_offset,
                         set_alpha_to_one=set_alpha_to_one, timestep_spacing="leading", prediction_type="epsilon")
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.eta = 0.0

    def set_eta(self, eta: float = 0.0):
        """Stochastic DDIM: std_dev_t = eta * sqrt(variance_t) enters the table (columns 3 and 4); the device table is
        rewritten in place, so a captured step graph picks the new values up."""
        if not 0.0 <= float(eta) <= 1.0:
            raise ValueError(f"eta must be in [0, 1], got {eta}")
        if float(eta) == self.eta and self.timesteps is not None:
            return self               # (the table set_timesteps / set_begin_index filled already carries this eta)
        self.eta = float(eta)
        if self.timesteps is not None:
            self._fill_table()
            self.timesteps = self._ts_host.clone()
            begin = self._begin
            self._upload(self._device)
            self._begin = begin
        return self

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self._ts_host = torch.from_numpy(ts)
        self.timesteps = self._ts_host.clone()
        self._begin = 0
        self._fill_table()
        self._upload(device)

    def _fill_table(self):
        """(rows do not depend on where the loop enters: DDIM keeps no history)"""
        ts, n = self._ts_host.numpy(), self.num_inference_steps
        ratio = self.config.num_train_timesteps // n
        coef = torch.zeros(n, 8, dtype=torch.float32)
        for i, t in enumerate(ts.tolist()):
            prev = t - ratio
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
            coef[i, 0] = (1 - a_t) ** 0.5
            coef[i, 1] = a_t ** 0.5
            coef[i, 2] = a_p ** 0.5
            # DDIMScheduler.step: variance = (1-a_prev)/(1-a_t) * (1-a_t/a_prev); std_dev_t = eta * sqrt(variance);
            # direction coefficient sqrt(1-a_prev-std_dev_t^2); prev_sample += std_dev_t * noise (pp_ddim_variance_noise)
            sd = self.eta * (((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)) ** 0.5
            coef[i, 3] = (1 - a_p - sd ** 2) ** 0.5
            coef[i, 4] = sd
        self._coef = coef


def dpm_timesteps(T, n, spacing="linspace", steps_offset=0):
    """DPMSolverMultistepScheduler.set_timesteps of diffusers 0.27 (lambda_min_clipped = -inf => last_timestep = T):
    what `DPMSolverMultistepScheduler.from_config(pipe.scheduler.config)` yields on an SD-1.5 checkpoint is the
    `leading` form with steps_offset 1, the class default is `linspace`."""
    if spacing == "linspace":
        return np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
    if spacing == "leading":
        return (np.arange(0, n + 1) * (T // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + steps_offset
    if spacing == "trailing":
        return np.arange(T, 0, -T / n).round().copy().astype(np.int64) - 1
    raise ValueError(f"unknown timestep_spacing {spacing}")


class DPMSolverMultistepScheduler(_SchedulerBase):
    """dpmsolver++ (2M), midpoint, `linspace` spacing, final_sigmas_type = "zero", lower_order_final."""
    kind = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2,
                 timestep_spacing="linspace", steps_offset=0, **kw):
        _check_config("DPMSolverMultistepScheduler", kw)
        super().__init__(num_train_timesteps, beta_start, beta_end, solver_order=solver_order, steps_offset=steps_offset,
                         algorithm_type="dpmsolver++", solver_type="midpoint", final_sigmas_type="zero",
                         timestep_spacing=timestep_spacing, prediction_type="epsilon")
        if solver_order != 2:
            raise NotImplementedError("only the 2M solver is on the hot path")
        if timestep_spacing not in ("linspace", "leading", "trailing"):
            raise ValueError(f"unknown timestep_spacing {timestep_spacing}")

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ts = dpm_timesteps(T, num_inference_steps, self.config.timestep_spacing, self.config.steps_offset)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self._ts_host = torch.from_numpy(ts)
        self.timesteps = self._ts_host.clone()
        self._begin = 0
        self._fill_table()
        self._upload(device)

    def _fill_table(self):
        n = self.num_inference_steps
        coef = torch.zeros(n, 8, dtype=torch.float32)
        for i in range(self._begin, n):
            a_cur, s_cur = self._alpha_sigma(self.sigmas[i])
            a_t, sg_t = self._alpha_sigma(self.sigmas[i + 1])
            lam_t = torch.log(a_t) - torch.log(sg_t)
            lam_s0 = torch.log(a_cur) - torch.log(s_cur)
            h = lam_t - lam_s0
            c3 = a_t * (torch.exp(-h) - 1.0)
            coef[i, 0], coef[i, 1] = s_cur, a_cur
            coef[i, 2] = sg_t / s_cur
            coef[i, 3] = c3
            first_order = (i == self._begin) or (i == n - 1)   # lower_order_nums < 1, lower_order_final (sigma_last = 0)
            if not first_order:
                a_s1, sg_s1 = self._alpha_sigma(self.sigmas[i - 1])
                lam_s1 = torch.log(a_s1) - torch.log(sg_s1)
                r0 = (lam_s0 - lam_s1) / h
                coef[i, 4] = 0.5 * c3
                coef[i, 5] = 1.0 / r0
        self._coef = coef


class PNDMScheduler(_SchedulerBase):
    """`PNDMScheduler(skip_prk_steps=True)` = PLMS with the SD-1.5 checkpoint config (`leading` spacing, steps_offset = 1,
    set_alpha_to_one = False, epsilon prediction): what the reference's v1 app runs when no scheduler is chosen.
    N inference steps are N + 1 entries in `.timesteps` (the second one repeats): the pipelines loop over
    `scheduler.timesteps`, so nothing else changes.  Host side: one table row per evaluation (linear-multistep weights,
    transfer coefficients, history ring slots); the tensor math runs in `pp_cfg_sched_step` (kind 2)."""
    kind = 2
    state_slots = 5

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False, skip_prk_steps=True, timestep_spacing="leading", **kw):
        _check_config("PNDMScheduler", kw)
        if timestep_spacing != "leading":
            raise L.PPError(f"PNDMScheduler: timestep_spacing={timestep_spacing!r} is not implemented (leading only)")
        if not skip_prk_steps:
            raise NotImplementedError("only the PLMS form (skip_prk_steps=True, the SD-1.5 config) is on the hot path")
        super().__init__(num_train_timesteps, beta_start, beta_end, steps_offset=steps_offset,
                         set_alpha_to_one=set_alpha_to_one, skip_prk_steps=True, timestep_spacing="leading",
                         prediction_type="epsilon")
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def _transfer(self, t, prev_t):
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return float(sample_coeff), float(-(a_p - a_t) / denom)

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + self.config.steps_offset
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        self._ts_host = torch.from_numpy(plms)
        self.timesteps = self._ts_host.clone()
        self._begin = 0
        self._fill_table()
        self._upload(device)

    def _fill_table(self):
        """One row per evaluation, keyed on the evaluation COUNT since the loop entered the schedule (the library's
        `counter`): entered late (`set_begin_index`), the second call is still treated as the repeat evaluation."""
        plms = self._ts_host.numpy()
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        rows = len(plms)
        coef = torch.zeros(rows, 16, dtype=torch.float32)
        n_hist, head = 0, 0                       # stored predictions so far, ring slot of the next push
        for k, t in enumerate(plms.tolist()[self._begin:]):
            r = self._begin + k
            slot = lambda back: float((head - back) % 4)          # noqa: E731  (slot of the prediction `back` pushes ago)
            if k == 1:
                # second evaluation at the repeated timestep: redo the first transfer from the saved sample with the
                # average of the two predictions; this prediction is not stored
                a, b = self._transfer(t + ratio, t)
                coef[r, :6] = torch.tensor([0.5, 0.5, 0.0, 0.0, a, b])
                coef[r, 6:9] = torch.tensor([slot(1), slot(1), slot(1)])
                coef[r, 9], coef[r, 10], coef[r, 11] = -1.0, 1.0, 0.0
                continue
            n_hist = min(n_hist + 1, 4)
            w = {1: (1.0, 0.0, 0.0, 0.0), 2: (1.5, -0.5, 0.0, 0.0), 3: (23 / 12, -16 / 12, 5 / 12, 0.0),
                 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[n_hist]
            a, b = self._transfer(t, t - ratio)
            coef[r, :6] = torch.tensor([w[0], w[1], w[2], w[3], a, b])
            coef[r, 6:9] = torch.tensor([slot(1), slot(2), slot(3)])      # h1, h2, h3 = previous pushes
            coef[r, 9] = float(head)                                      # this prediction goes to ring slot `head`
            coef[r, 10], coef[r, 11] = 0.0, (1.0 if k == 0 else 0.0)      # the first evaluation saves its input sample
            head = (head + 1) % 4
        self._coef = coef

    def _index_of(self, timestep) -> int:
        """The repeated timestep is disambiguated by call order (a foreign loop calls step() once per entry)."""
        t = int(timestep)
        idx = (self._ts_host == t).nonzero().flatten().tolist()
        if not idx:
            raise ValueError(f"timestep {t} is not in the schedule")
        if len(idx) == 1:
            return idx[0]
        self._dup_calls = getattr(self, "_dup_calls", 0)
        i = idx[min(self._dup_calls, len(idx) - 1)]
        self._dup_calls += 1
        return i

    def reset(self):
        super().reset()
        self._dup_calls = 0


class UniPCMultistepScheduler(_SchedulerBase):
    """UniPC (arXiv:2302.04867) as diffusers' `UniPCMultistepScheduler` runs it -- the scheduler app.py:197 installs on
    the ppt-v2 pipeline: predict_x0, solver_type "bh2", solver_order 2 (3 supported), lower_order_final, corrector on
    every step after the first (minus `disable_corrector`).

    Every step is linear in (sample, eps, last_sample, m1, m2, m3): the host solves the small UniPC systems per step
    in float64 and uploads one 16-float row; `pp_cfg_sched_step` (kind 3) evaluates
        x0 = (x - c0 eps) / c1
        xc = use_corr ? c3 last + c4 m1 + c5 m2 + c6 m3 + c7 x0 : x          (UniC on the previous transition)
        x' = c8 xc + c9 x0 + c10 m1 + c11 m2                                  (UniP to the next time step)
        (last, m1, m2, m3) <- (xc, x0, m1, m2)
    on the state [4][n] kept between steps."""
    kind = 3
    state_slots = 4

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2,
                 solver_type="bh2", lower_order_final=True, disable_corrector=(), timestep_spacing="linspace",
                 steps_offset=0, predict_x0=True, prediction_type="epsilon", **kw):
        super().__init__(num_train_timesteps, beta_start, beta_end, solver_order=solver_order, solver_type=solver_type,
                         lower_order_final=lower_order_final, disable_corrector=list(disable_corrector),
                         timestep_spacing=timestep_spacing, steps_offset=steps_offset, predict_x0=predict_x0,
                         prediction_type=prediction_type)
        _check_config("UniPCMultistepScheduler", dict(kw, prediction_type=prediction_type,
                                                      lower_order_final=lower_order_final), self._CHECKED)
        if solver_order not in (1, 2, 3) or solver_type not in ("bh1", "bh2") or not predict_x0 or \
                prediction_type != "epsilon":
            raise NotImplementedError("UniPC: solver_order 1-3, bh1 / bh2, predict_x0, epsilon prediction")

    # arithmetic options of diffusers-0.27 `UniPCMultistepScheduler.__init__` that the fused step does not implement (a
    # donor config's keys outside that constructor -- algorithm_type, euler_at_final, clip_sample ... -- never reach it;
    # `solver_type` has its own rule below)
    _CHECKED = ("prediction_type", "beta_schedule", "thresholding", "use_karras_sigmas", "trained_betas",
                "final_sigmas_type", "lower_order_final")

    @classmethod
    def from_config(cls, config, **kw):
        """`UniPCMultistepScheduler.from_config(pipe.scheduler.config)` (app.py:197): the keys this class shares with
        the donor scheduler's config are taken over (betas, timestep_spacing, steps_offset), the rest keep defaults."""
        src = dict(config) if isinstance(config, dict) else dict(vars(config))
        # refuse what would change the arithmetic (beta_schedule, thresholding, Karras sigmas, trained_betas,
        # final_sigmas_type ...) instead of dropping it: the donor's config is what app.py:197 hands over
        _check_config(cls.__name__, {**src, **kw}, cls._CHECKED)
        take = ("num_train_timesteps", "beta_start", "beta_end", "timestep_spacing", "steps_offset", "prediction_type",
                "lower_order_final")
        args = {k: src[k] for k in take if k in src}
        if src.get("solver_order") in (1, 2, 3):
            args["solver_order"] = src["solver_order"]
        # a donor's solver_type outside bh1 / bh2 (DPM-Solver's midpoint / heun, logrho) becomes bh2, as in diffusers
        if src.get("solver_type") in ("bh1", "bh2"):
            args["solver_type"] = src["solver_type"]
        args.update(kw)
        if args.get("solver_type") in ("midpoint", "heun", "logrho"):
            args["solver_type"] = "bh2"
        return cls(**args)

    def _grid(self, N):
        T, sp = self.config.num_train_timesteps, self.config.timestep_spacing
        if sp == "linspace":
            return np.linspace(0, T - 1, N + 1).round()[::-1][:-1].copy().astype(np.int64)
        if sp == "leading":
            return (np.arange(0, N + 1) * (T // (N + 1))).round()[::-1][:-1].copy().astype(np.int64) + \
                self.config.steps_offset
        if sp == "trailing":
            return (np.arange(T, 0, -T / N).round() - 1).astype(np.int64)
        raise ValueError(f"unknown timestep_spacing {sp}")

    def _system(self, order, rks, hh):
        """b / R of the UniPC conditions and h*phi_1, B(h); float64."""
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1.0
        B_h = hh if self.config.solver_type == "bh1" else np.expm1(hh)
        R, b, fact = [], [], 1.0
        for i in range(1, order + 1):
            R.append(rks ** (i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1.0 / fact
        return np.stack(R), np.array(b), h_phi_1, B_h

    def set_timesteps(self, num_inference_steps: int, device=None):
        N = num_inference_steps
        self.num_inference_steps = N
        ts = self._grid(N)
        sig_all = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()      # fp32, as the library does
        sig = np.concatenate([np.interp(ts, np.arange(0, len(sig_all)), sig_all), [sig_all[0]]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        sg = sig.astype(np.float64)                     # the library keeps the grid in fp32; coefficients from it in f64
        alpha = 1.0 / np.sqrt(sg ** 2 + 1.0)
        sigma = sg * alpha
        lam = np.log(alpha) - np.log(sigma)
        self._ts_host = torch.from_numpy(ts)
        self.timesteps = self._ts_host.clone()
        self._grid_f64 = (alpha, sigma, lam)
        self._begin = 0
        self._fill_table()
        self._upload(device)

    def _fill_table(self):
        alpha, sigma, lam = self._grid_f64
        N, b0 = self.num_inference_steps, self._begin
        K = self.config.solver_order
        coef = np.zeros((N, 16), dtype=np.float64)
        lower, prev_order = 0, 1
        for i in range(b0, N):
            c = coef[i]
            c[0], c[1] = sigma[i], alpha[i]
            # ---- UniC: re-estimate x_i from x_{i-1} (order = the order the previous predictor ran at); not on the step
            # the loop enters at (the library's `last_sample is None`).  `disable_corrector` lists step indices of the
            # full schedule, as the library's `step_index` does
            if i > b0 and (i - 1) not in self.config.disable_corrector:
                order = prev_order
                h = lam[i] - lam[i - 1]
                rks = np.array([(lam[i - (k + 1)] - lam[i - 1]) / h for k in range(1, order)] + [1.0])
                R, b, h_phi_1, B_h = self._system(order, rks, -h)
                rhos = np.array([0.5]) if order == 1 else np.linalg.solve(R, b)
                c[2] = 1.0
                c[3] = sigma[i] / sigma[i - 1]                                   # last_sample
                m0 = -alpha[i] * h_phi_1 + alpha[i] * B_h * rhos[-1]
                for k in range(1, order):                                         # D1s_k = (m_{-k} - m0) / rk_k
                    w = -alpha[i] * B_h * rhos[k - 1] / rks[k - 1]
                    c[4 + k] = w                                                  # m2 (k = 1), m3 (k = 2)
                    m0 -= w
                c[4] = m0                                                         # m1 = x0_{i-1}
                c[7] = -alpha[i] * B_h * rhos[-1]                                 # x0_i  (D1_t = x0_i - m0)
            # ---- UniP: x_{i+1} from the corrected x_i
            order = min(K, N - i) if self.config.lower_order_final else K
            order = min(order, lower + 1)
            h = lam[i + 1] - lam[i]
            rks = np.array([(lam[i - k] - lam[i]) / h for k in range(1, order)] + [1.0])
            R, b, h_phi_1, B_h = self._system(order, rks, -h)
            c[8] = sigma[i + 1] / sigma[i]
            m0 = -alpha[i + 1] * h_phi_1
            if order > 1:
                rhos = np.array([0.5]) if order == 2 else np.linalg.solve(R[:-1, :-1], b[:-1])
                for k in range(1, order):
                    w = -alpha[i + 1] * B_h * rhos[k - 1] / rks[k - 1]
                    c[9 + k] = w                                                  # m1 (k = 1), m2 (k = 2): older x0's
                    m0 -= w
            c[9] = m0                                                             # x0_i
            prev_order = order
            if lower < K:
                lower += 1
        self._coef = torch.from_numpy(coef.astype(np.float32))


SCHEDULERS = {"DDIMScheduler": DDIMScheduler, "DPMSolverMultistepScheduler": DPMSolverMultistepScheduler,
              "PNDMScheduler": PNDMScheduler, "UniPCMultistepScheduler": UniPCMultistepScheduler}
