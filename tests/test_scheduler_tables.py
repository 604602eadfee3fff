"""CPU: the per-step coefficient tables of the product schedulers, driven through a NumPy restatement of the step
kernel's linear forms (`cfg_sched_step_kernel`, csrc/small.hip: kinds 0-3), against the oracle's diffusers-protocol
classes -- full schedules, schedules entered late (`set_begin_index`, strength < 1) and stochastic DDIM (`set_eta`).
The GPU suite checks the same things through the HIP kernel (tests/test_ops_gpu.py); this file pins the HOST half
(table construction) where no GPU is available."""
import numpy as np
import pytest
import torch

from oracle import schedulers as OS
from powerpaint_amd import schedulers as PS


class KernelEmu:
    """State and arithmetic of pp_cfg_sched_step (+ pp_ddim_variance_noise) for one scheduler, in float64."""

    def __init__(self, sch):
        self.kind, self.coef = sch.kind, sch._coef.double().numpy()
        self.state = None

    def step(self, i, x, e, noise=None):
        c = self.coef[i]
        if self.kind == 0:
            x0 = (x - c[0] * e) / c[1]
            out = c[2] * x0 + c[3] * e
            return out + c[4] * noise if noise is not None else out
        if self.kind == 1:
            if self.state is None:
                self.state = np.zeros_like(x)
            x0 = (x - c[0] * e) / c[1]
            d1 = c[5] * (x0 - self.state)
            self.state = x0
            return c[2] * x - c[3] * x0 - c[4] * d1
        if self.kind == 2:
            if self.state is None:
                self.state = np.zeros((5,) + x.shape)
            h1, h2, h3 = (self.state[int(c[k])].copy() for k in (6, 7, 8))
            mo = c[0] * e + c[1] * h1 + c[2] * h2 + c[3] * h3
            src = self.state[4].copy() if c[10] != 0 else x
            if c[11] != 0:
                self.state[4] = x
            if int(c[9]) >= 0:
                self.state[int(c[9])] = e
            return c[4] * src + c[5] * mo
        if self.state is None:
            self.state = np.zeros((4,) + x.shape)
        last, m1, m2, m3 = (self.state[k].copy() for k in range(4))
        x0 = (x - c[0] * e) / c[1]
        xc = c[3] * last + c[4] * m1 + c[5] * m2 + c[6] * m3 + c[7] * x0 if c[2] != 0 else x
        out = c[8] * xc + c[9] * x0 + c[10] * m1 + c[11] * m2
        self.state[0], self.state[1], self.state[2], self.state[3] = xc, x0, m1, m2
        return out


CASES = [("DDIMScheduler", {}), ("DPMSolverMultistepScheduler", {}),
         ("DPMSolverMultistepScheduler", dict(timestep_spacing="leading", steps_offset=1)), ("PNDMScheduler", {}),
         ("UniPCMultistepScheduler", dict(solver_order=2)), ("UniPCMultistepScheduler", dict(solver_order=3))]


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("N,begin", [(10, 0), (10, 4), (6, 5), (7, 1)])
def test_tables_reproduce_the_oracle_schedulers(name, kw, N, begin):
    o, h = getattr(OS, name)(**kw), getattr(PS, name)(**kw)
    o.set_timesteps(N)
    h.set_timesteps(N)                                   # (device=None: tables stay on the host)
    if begin:
        h.set_begin_index(begin)
    assert h.timesteps.tolist() == o.timesteps.tolist() and h.begin_index == begin
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 6, 6, generator=g, dtype=torch.float64)
    emu, xo, xe = KernelEmu(h), x.clone().float(), x.numpy().copy()
    for k, t in enumerate(o.timesteps[begin:]):
        e = torch.randn(2, 4, 6, 6, generator=g)
        xo = o.step(e, t, xo)[0]
        xe = emu.step(begin + k, xe, e.double().numpy())
        ref = xo.double().numpy()
        assert np.abs(xe - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), (name, N, begin, k)
    h.set_timesteps(N)
    assert h.begin_index == 0


@pytest.mark.parametrize("eta", [0.3, 1.0])
def test_ddim_eta_table_reproduces_the_oracle(eta):
    o, h = OS.DDIMScheduler(), PS.DDIMScheduler()
    o.set_timesteps(8)
    h.set_timesteps(8)
    base = h._coef.clone()
    h.set_eta(eta)
    assert float(h._coef[:, 4].min()) > 0 and torch.equal(h._coef[:, :3], base[:, :3])
    go, ge = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 4, 4, generator=g)
    emu, xo, xe = KernelEmu(h), x.clone(), x.double().numpy()
    for k, t in enumerate(o.timesteps):
        e = torch.randn(1, 4, 4, 4, generator=g)
        xo = o.step(e, t, xo, eta=eta, generator=go)[0]
        z = PS.variance_noise(e.shape, ge, "cpu", e.dtype)
        xe = emu.step(k, xe, e.double().numpy(), z.double().numpy())
        assert np.abs(xe - xo.double().numpy()).max() <= 2e-4 * max(1.0, float(xo.abs().max())), k
    h.set_eta(0.0)
    assert torch.equal(h._coef, base)
    with pytest.raises(ValueError):
        h.set_eta(-0.1)


def test_begin_index_is_validated():
    h = PS.DPMSolverMultistepScheduler()
    with pytest.raises(Exception):
        h.set_begin_index(1)                             # before set_timesteps
    h.set_timesteps(5)
    with pytest.raises(ValueError):
        h.set_begin_index(5)
    h.set_begin_index(4)
    assert (h._coef[:4] == 0).all() and (h._coef[4] != 0).any()


def test_unipc_from_config_refuses_what_it_cannot_compute():
    """`UniPCMultistepScheduler.from_config(pipe.scheduler.config)` is the call app.py:197 makes: arithmetic options of
    the donor config that the fused step does not implement must be REFUSED, not dropped; the donor's
    non-UniPC keys (algorithm_type ...) never reach the class in diffusers and are ignored; a DPM-Solver solver_type
    becomes bh2 as in diffusers."""
    from powerpaint_amd import _lib as L
    donor = PS.DPMSolverMultistepScheduler()
    u = PS.UniPCMultistepScheduler.from_config(donor.config)
    assert (u.config.solver_type, u.config.solver_order) == ("bh2", 2)
    base = dict(vars(donor.config))
    for bad in (dict(beta_schedule="linear"), dict(thresholding=True), dict(use_karras_sigmas=True),
                dict(trained_betas=[0.1, 0.2]), dict(final_sigmas_type="sigma_min"), dict(prediction_type="v_prediction"),
                dict(lower_order_final=False)):
        with pytest.raises(L.PPError):
            PS.UniPCMultistepScheduler.from_config({**base, **bad})
        with pytest.raises(L.PPError):
            PS.UniPCMultistepScheduler.from_config(base, **bad)
        with pytest.raises(L.PPError):
            PS.UniPCMultistepScheduler(**bad)
    ok = PS.UniPCMultistepScheduler.from_config({**base, "solver_type": "heun", "algorithm_type": "sde-dpmsolver++",
                                                 "euler_at_final": True})
    assert ok.config.solver_type == "bh2"
    assert PS.UniPCMultistepScheduler(solver_type="bh1").config.solver_type == "bh1"
    assert PS.UniPCMultistepScheduler.from_config({**base, "solver_type": "bh1", "solver_order": 3}).config.solver_order == 3


@pytest.mark.parametrize("name,kw", [("DDIMScheduler", {}), ("DPMSolverMultistepScheduler", {}), ("PNDMScheduler", {}),
                                     ("UniPCMultistepScheduler", {})])
def test_renoise_table_is_add_noise_of_the_next_timestep(name, kw):
    """`renoise_table()[i]` = what `scheduler.add_noise(x0, noise, timesteps[i + 1])` multiplies x0 and the noise by
    (pipeline_PowerPaint.py:1029-1033), (1, 0) after the last step; rows follow the step counter (full schedule)."""
    from oracle import schedulers as OS
    from powerpaint_amd import schedulers as PS
    o, h = getattr(OS, name)(**kw), getattr(PS, name)(**kw)
    o.set_timesteps(7)
    h.set_timesteps(7)
    tab = h.renoise_table()
    ts = [int(t) for t in o.timesteps]
    assert tab.shape == (len(ts), 2)
    x0, nz = torch.full((1, 1), 1.0), torch.zeros(1, 1)
    for i in range(len(ts) - 1):
        a = float(o.add_noise(x0, nz, torch.tensor([ts[i + 1]])))
        b = float(o.add_noise(nz, x0, torch.tensor([ts[i + 1]])))
        assert abs(float(tab[i, 0]) - a) < 1e-6 and abs(float(tab[i, 1]) - b) < 1e-6, (i, tab[i], a, b)
    assert tab[-1].tolist() == [1.0, 0.0]
    h.set_timesteps(7)
    assert h.renoise_table().data_ptr() == tab.data_ptr()             # stable address for the captured step
