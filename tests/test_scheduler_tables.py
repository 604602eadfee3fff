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
