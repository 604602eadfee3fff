"""Oracle (TEST INFRASTRUCTURE): DDIM, DPM-Solver++(2M), PNDM (PLMS) and UniPC restated from diffusers==0.27.0.

The reference pipelines are scheduler-agnostic and only duck-type the scheduler
(/root/reference/powerpaint/pipelines/pipeline_PowerPaint.py:906,993,1023,642;
 pipeline_PowerPaint_Brushnet_CA.py:87-128,1391,1449,969).  The arithmetic lives in the
un-vendored dependency diffusers==0.27.0 (requirements/requirements.txt:3):
`DDIMScheduler`, `DPMSolverMultistepScheduler`.  SURVEY.md section 8a row a19 / Appendix B give
the closed forms.  Torch classes below follow the diffusers call protocol in fp32; the
`*_f64` functions are independent float64 NumPy re-derivations used to pin them.
"""
import math
from typing import List, Optional

import numpy as np
import torch


def sd_betas(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012) -> torch.Tensor:
    """`scaled_linear` schedule, float32 exactly as diffusers builds it."""
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2


class DDIMScheduler:
    """epsilon prediction, `leading` spacing, steps_offset=1, set_alpha_to_one=False, no clipping
    (the `runwayml/stable-diffusion-inpainting` scheduler config applied to DDIM)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False):
        self.config = type("C", (), dict(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                         beta_start=beta_start, beta_end=beta_end))()
        self.betas = sd_betas(num_train_timesteps, beta_start, beta_end)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict=False):
        """DDIMScheduler.step of diffusers 0.27 (epsilon prediction, no clipping): eta > 0 adds
        std_dev_t * randn_tensor(model_output.shape, generator) with std_dev_t = eta * sqrt(_get_variance(t, prev_t))."""
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5
        variance = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
        std_dev_t = eta * variance ** 0.5
        direction = (1 - a_prev - std_dev_t ** 2) ** 0.5 * model_output
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            gdev = generator.device if generator is not None else model_output.device
            noise = torch.randn(model_output.shape, generator=generator, device=gdev, dtype=model_output.dtype)
            prev = prev + std_dev_t * noise.to(model_output.device)
        return (prev,)

    def add_noise(self, x0, noise, timesteps):
        a = self.alphas_cumprod[timesteps.cpu()].to(device=x0.device, dtype=x0.dtype)   # (diffusers moves its tables too)
        sa = (a ** 0.5).flatten()
        s1 = ((1 - a) ** 0.5).flatten()
        while sa.dim() < x0.dim():
            sa, s1 = sa.unsqueeze(-1), s1.unsqueeze(-1)
        return sa * x0 + s1 * noise


class PNDMScheduler:
    """`PNDMScheduler(skip_prk_steps=True)` = PLMS, the scheduler the SD-1.5 (inpainting) checkpoint config names and
    therefore what the reference's v1 app runs by default: epsilon prediction, `leading`
    spacing, steps_offset=1, set_alpha_to_one=False.  N inference steps are N+1 UNet evaluations: the second timestep
    of the schedule appears twice and the first transfer is redone with the average of the two first model outputs
    (diffusers scheduling_pndm.py: set_timesteps / step_plms / _get_prev_sample)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False):
        self.config = type("C", (), dict(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset,
                                         beta_start=beta_start, beta_end=beta_end, skip_prk_steps=True))()
        self.betas = sd_betas(num_train_timesteps, beta_start, beta_end)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round().astype(np.int64) + self.config.steps_offset
        plms = np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms)
        self.ets: List[torch.Tensor] = []
        self.counter = 0
        self.cur_sample = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return sample_coeff * sample - (a_p - a_t) * model_output / denom

    def step(self, model_output, timestep, sample, generator=None, return_dict=False):
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = t
            t = t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev = self._get_prev_sample(sample, t, prev_t, model_output)
        self.counter += 1
        return (prev,)

    add_noise = DDIMScheduler.add_noise


class DPMSolverMultistepScheduler:
    """dpmsolver++, order 2, midpoint, `linspace` spacing, final_sigmas_type='zero', lower_order_final."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2,
                 timestep_spacing="linspace", steps_offset=0):
        self.config = type("C", (), dict(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                                         steps_offset=steps_offset, timestep_spacing=timestep_spacing))()
        self.betas = sd_betas(num_train_timesteps, beta_start, beta_end)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        T, N = self.config.num_train_timesteps, num_inference_steps
        # diffusers 0.27 DPMSolverMultistepScheduler.set_timesteps, lambda_min_clipped = -inf (last_timestep = T)
        if self.config.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, N + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.config.timestep_spacing == "leading":
            ts = (np.arange(0, N + 1) * (T // (N + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.config.steps_offset
        elif self.config.timestep_spacing == "trailing":
            ts = np.arange(T, 0, -T / N).round().copy().astype(np.int64) - 1
        else:
            raise ValueError(self.config.timestep_spacing)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        sig = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = num_inference_steps
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * self.config.solver_order
        self.lower_order_nums = 0
        self._step_index = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def step(self, model_output, timestep, sample, generator=None, return_dict=False):
        if self._step_index is None:
            idx = (self.timesteps == int(timestep)).nonzero()
            self._step_index = int(idx[0])
        i = self._step_index
        n = len(self.timesteps)
        lower_order_final = (i == n - 1)  # final_sigmas_type == "zero"
        # convert_model_output: epsilon -> x0
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        x0 = (sample - sigma_t * model_output) / alpha_t
        for k in range(self.config.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0

        s_t, s_s0 = self.sigmas[i + 1], self.sigmas[i]
        a_t, sg_t = self._alpha_sigma(s_t)
        a_s0, sg_s0 = self._alpha_sigma(s_s0)
        lam_t = torch.log(a_t) - torch.log(sg_t)
        lam_s0 = torch.log(a_s0) - torch.log(sg_s0)
        h = lam_t - lam_s0
        if self.config.solver_order == 1 or self.lower_order_nums < 1 or lower_order_final:
            prev = (sg_t / sg_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * x0
        else:
            s_s1 = self.sigmas[i - 1]
            a_s1, sg_s1 = self._alpha_sigma(s_s1)
            lam_s1 = torch.log(a_s1) - torch.log(sg_s1)
            m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
            h_0 = lam_s0 - lam_s1
            r0 = h_0 / h
            D0, D1 = m0, (1.0 / r0) * (m0 - m1)
            prev = ((sg_t / sg_s0) * sample - (a_t * (torch.exp(-h) - 1.0)) * D0
                    - 0.5 * (a_t * (torch.exp(-h) - 1.0)) * D1)
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        self._step_index += 1
        return (prev,)


    def add_noise(self, x0, noise, timesteps):
        """diffusers 0.27 DPMSolverMultistepScheduler.add_noise: sigma of the schedule entry holding each timestep,
        (alpha_t, sigma_t) from it."""
        idx = [int((self.timesteps == int(t)).nonzero()[0]) for t in timesteps]
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[idx].flatten().to(device=x0.device, dtype=x0.dtype))
        while alpha_t.dim() < x0.dim():
            alpha_t, sigma_t = alpha_t.unsqueeze(-1), sigma_t.unsqueeze(-1)
        return alpha_t * x0 + sigma_t * noise


# ---------------------------------------------------------------------------------------
# independent float64 closed forms (pins for the classes above and for the HIP step kernel)
# ---------------------------------------------------------------------------------------
class UniPCMultistepScheduler:
    """UniPC (Zhao et al. 2023, arXiv:2302.04867) as diffusers==0.27.0 `UniPCMultistepScheduler` runs it -- the scheduler
    app.py:197 puts on the ppt-v2 pipeline (`UniPCMultistepScheduler.from_config(pipe.scheduler.config)`): data
    prediction (predict_x0), B(h) = expm1 ("bh2"), solver_order 2 (3 supported), lower_order_final, corrector on every
    step after the first.  Restated in the library's own shape -- history lists, a `multistep_uni_p_bh_update`
    predictor and a `multistep_uni_c_bh_update` corrector -- in fp32; **parity unpinned** (diffusers is not
    installable here); tests pin it to the paper's structure instead (UniP-1 == DDIM, order of convergence on a
    closed-form ODE).  `timestep_spacing` / `steps_offset` come from the pipeline's scheduler config ("leading", 1 for
    the SD-1.5 family); `linspace` is the class default."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, solver_order=2,
                 solver_type="bh2", lower_order_final=True, disable_corrector=(), timestep_spacing="linspace",
                 steps_offset=0):
        self.T = num_train_timesteps
        self.alphas_cumprod = torch.cumprod(1.0 - sd_betas(num_train_timesteps, beta_start, beta_end), dim=0)
        self.solver_order, self.solver_type, self.lower_order_final = solver_order, solver_type, lower_order_final
        self.disable_corrector = list(disable_corrector)
        self.timestep_spacing, self.steps_offset = timestep_spacing, steps_offset
        self.timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        N, T = num_inference_steps, self.T
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, N + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, N + 1) * (T // (N + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        elif self.timestep_spacing == "trailing":
            ts = (np.arange(T, 0, -T / N).round() - 1).astype(np.int64)
        else:
            raise ValueError(self.timestep_spacing)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigma_last = ((1 - self.alphas_cumprod[0]) / self.alphas_cumprod[0]) ** 0.5
        sig = np.concatenate([np.interp(ts, np.arange(0, len(sig)), sig), [float(sigma_last)]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = torch.from_numpy(ts)
        self.num_inference_steps = N
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = None          # found from the first timestep `step` sees (`_init_step_index`): img2img-style
        self.this_order = 1             # loops enter the schedule late (get_timesteps slices scheduler.timesteps)

    def scale_model_input(self, sample, timestep=None):
        return sample

    @staticmethod
    def _alpha_sigma(sigma):
        alpha_t = 1 / ((sigma ** 2 + 1) ** 0.5)
        return alpha_t, sigma * alpha_t

    def _lambda(self, i):
        a, s = self._alpha_sigma(self.sigmas[i])
        return torch.log(a) - torch.log(s)

    def _bh(self, order, rks, hh):
        """R, b of the UniPC linear system and the scalars h*phi_1(h), B(h)."""
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return torch.stack(R), torch.stack(b), h_phi_1, B_h

    def _predict(self, sample, order):
        """multistep_uni_p_bh_update: x_{i+1} from the (corrected) x_i and the x0 history."""
        i = self.step_index
        m0, x = self.model_outputs[-1], sample
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i + 1])
        _, sigma_s0 = self._alpha_sigma(self.sigmas[i])
        h = self._lambda(i + 1) - self._lambda(i)
        rks, D1s = [], []
        for k in range(1, order):
            rk = (self._lambda(i - k) - self._lambda(i)) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack(rks)
        hh = -h
        R, b, h_phi_1, B_h = self._bh(order, rks, hh)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        pred_res = 0
        if D1s:
            rhos_p = torch.tensor([0.5]) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
            pred_res = sum(r * d for r, d in zip(rhos_p, D1s))
        return x_t_ - alpha_t * B_h * pred_res

    def _correct(self, this_model_output, last_sample, order):
        """multistep_uni_c_bh_update: x_i re-estimated from x_{i-1} with the x0 prediction at x_i included."""
        i = self.step_index
        m0, x = self.model_outputs[-1], last_sample
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        _, sigma_s0 = self._alpha_sigma(self.sigmas[i - 1])
        h = self._lambda(i) - self._lambda(i - 1)
        rks, D1s = [], []
        for k in range(1, order):
            rk = (self._lambda(i - (k + 1)) - self._lambda(i - 1)) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(k + 1)] - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack(rks)
        hh = -h
        R, b, h_phi_1, B_h = self._bh(order, rks, hh)
        rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        corr_res = sum(r * d for r, d in zip(rhos_c[:-1], D1s)) if D1s else 0
        D1_t = this_model_output - m0
        return x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * D1_t)

    def step(self, model_output, timestep, sample, generator=None, return_dict=False):
        if self.step_index is None:
            self.step_index = int((self.timesteps == int(timestep)).nonzero()[0])
        i = self.step_index
        use_corrector = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        alpha_t, sigma_t = self._alpha_sigma(self.sigmas[i])
        x0 = (sample - sigma_t * model_output) / alpha_t                       # convert_model_output (epsilon, predict_x0)
        if use_corrector:
            sample = self._correct(x0, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [x0]
        this_order = min(self.solver_order, len(self.timesteps) - i) if self.lower_order_final else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)           # warm-up
        self.last_sample = sample
        prev = self._predict(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return (prev,)

    def add_noise(self, x0, noise, timesteps):
        a = self.alphas_cumprod[timesteps.long().cpu()].to(device=x0.device, dtype=x0.dtype)
        sa, s1 = (a ** 0.5).flatten(), ((1 - a) ** 0.5).flatten()
        while sa.dim() < x0.dim():
            sa, s1 = sa.unsqueeze(-1), s1.unsqueeze(-1)
        return sa * x0 + s1 * noise


def alphas_cumprod_f64(T=1000, beta_start=0.00085, beta_end=0.012) -> np.ndarray:
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def ddim_timesteps(N: int, T: int = 1000, offset: int = 1) -> np.ndarray:
    return np.array([(N - 1 - i) * (T // N) + offset for i in range(N)], dtype=np.int64)


def ddim_step_f64(x, eps, t: int, N: int, T: int = 1000):
    ac = alphas_cumprod_f64(T)
    prev_t = t - T // N
    a_t = ac[t]
    a_prev = ac[prev_t] if prev_t >= 0 else ac[0]
    x0 = (x - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
    return math.sqrt(a_prev) * x0 + math.sqrt(1 - a_prev) * eps


def dpm_timesteps(N: int, T: int = 1000) -> np.ndarray:
    return np.linspace(0, T - 1, N + 1).round()[::-1][:-1].astype(np.int64)


def dpm_sigmas_f64(N: int, T: int = 1000) -> np.ndarray:
    ac = alphas_cumprod_f64(T)
    sig = np.sqrt((1 - ac) / ac)
    ts = dpm_timesteps(N, T)
    return np.concatenate([np.interp(ts, np.arange(T), sig), [0.0]])


def dpm_run_f64(x, eps_list, N: int):
    """Run all N DPM-Solver++(2M) steps in float64 given the eps fed at every step."""
    sig = dpm_sigmas_f64(N)
    x = np.asarray(x, dtype=np.float64)
    m1 = None
    for i in range(N):
        s0, st = sig[i], sig[i + 1]
        a0, at = 1 / math.sqrt(s0 * s0 + 1), 1 / math.sqrt(st * st + 1)
        g0, gt = s0 * a0, st * at
        m0 = (x - g0 * np.asarray(eps_list[i], dtype=np.float64)) / a0
        if st == 0.0:
            x = m0            # sigma_t/sigma_s -> 0, alpha_t (e^{-h}-1) -> -1
        else:
            lam0, lamt = math.log(a0) - math.log(g0), math.log(at) - math.log(gt)
            h = lamt - lam0
            c = at * (math.exp(-h) - 1.0)
            if i == 0:
                x = (gt / g0) * x - c * m0
            else:
                sp = sig[i - 1]
                ap = 1 / math.sqrt(sp * sp + 1)
                lamp = math.log(ap) - math.log(sp * ap)
                r0 = (lam0 - lamp) / h
                x = (gt / g0) * x - c * m0 - 0.5 * c * (m0 - m1) / r0
        m1 = m0
    return x


def pndm_timesteps(N: int, T: int = 1000, offset: int = 1) -> np.ndarray:
    ts = np.array([i * (T // N) + offset for i in range(N)], dtype=np.int64)
    return np.concatenate([ts[:-1], ts[-2:-1], ts[-1:]])[::-1].copy()


def pndm_run_f64(x, eps_list, N: int, T: int = 1000):
    """All N+1 PLMS evaluations in float64 given the eps fed at every evaluation (independent re-derivation: the linear
    multistep combination of the noise predictions, then the PNDM transfer formula (eq. 9 of the PNDM paper))."""
    ac = alphas_cumprod_f64(T)
    ts = pndm_timesteps(N, T)
    ratio = T // N

    def transfer(xx, t, tp, e):
        a_t, a_p = ac[t], (ac[tp] if tp >= 0 else ac[0])
        return math.sqrt(a_p / a_t) * xx - (a_p - a_t) * e / (a_t * math.sqrt(1 - a_p) + math.sqrt(a_t * (1 - a_t) * a_p))

    x = np.asarray(x, dtype=np.float64)
    hist: list = []
    x_start = None
    for k, t in enumerate(ts.tolist()):
        e = np.asarray(eps_list[k], dtype=np.float64)
        if k == 0:
            hist = [e]
            x_start = x
            x = transfer(x, t, t - ratio, e)
        elif k == 1:                        # same transfer again, from the saved sample, with the averaged prediction
            x = transfer(x_start, t + ratio, t, 0.5 * (e + hist[-1]))
        else:
            hist = (hist + [e])[-4:]
            if len(hist) == 2:
                m = (3 * hist[-1] - hist[-2]) / 2
            elif len(hist) == 3:
                m = (23 * hist[-1] - 16 * hist[-2] + 5 * hist[-3]) / 12
            else:
                m = (55 * hist[-1] - 59 * hist[-2] + 37 * hist[-3] - 9 * hist[-4]) / 24
            x = transfer(x, t, t - ratio, m)
    return x
