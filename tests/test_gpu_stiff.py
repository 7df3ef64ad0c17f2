"""SURVEY.md 8(d) "stiff C4": Fisher-KPP on 1024 points with the domain kept at X = 1 (D/dx^2 = 1.05e4): Tsit5 runs at its
stability limit (thousands of steps per unit time), the dense store and the adjoint walk as many steps.  A short horizon
keeps the oracle affordable; examples/fisher_kpp_stiff.py runs the full T = 5 (56 698 steps) on the device."""
import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_parity import REL_GRAD_SUM, assert_bitwise, check_per_trajectory

pytestmark = pytest.mark.gpu


def test_stiff_fisher_kpp_1024_forward_and_adjoint_match_oracle():
    nx, D, r = 1024, 0.01, 1.0
    dx = 1.0 / (nx - 1)
    x = np.arange(nx) * dx
    rho0 = 0.5 * (np.tanh((x - 0.3) / 0.2) - np.tanh((x - 0.7) / 0.2))
    rng = np.random.default_rng(0)
    u0 = rho0[None, :] * (1 + 0.05 * rng.uniform(-1, 1, (2, 1)))
    T = 0.04
    t = np.linspace(0.0, T, 5)
    # the true model: step counts of a stability-limited solve, bit for bit
    sol = U.solve(U.EnsembleProblem(U.ODEProblem(models.rc_ode(nx, D, r, dx), u0[0], (0.0, T), []), u0), U.Tsit5(), saveat=t)
    out, st, rc = O.solve_ensemble(O.kpp_true(nx, D, r, dx), O.opts(O.TSIT5), u0, [0.0, T], [], t)
    assert (rc == 0).all() and st[0, 1] > 400                      # ~12 000 steps per unit time
    assert_bitwise(sol.stats[:, :4], st[:, :4], "true model counts")
    assert_bitwise(sol.u, out, "true model states")
    # the UDE with D0 near D/dx^2: loss + interpolating-adjoint gradient
    th = models.kpp_theta(models.kpp_chain(), rng)
    f = models.nn_ode(nx)
    th[f.d0_offset] = 0.95 * D / dx ** 2
    th[f.stencil_offset:f.stencil_offset + 3] = [1.01, -2.0, 0.99]
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, T), th), u0)
    r_ = U.loss_and_gradient(ens, U.Tsit5(), out, saveat=t, ensemblealg=U.EnsembleMI355(0, 4096))
    ref = O.loss_grad_ensemble(O.kpp_ude(nx), O.opts(O.TSIT5), u0, [0.0, T], th, t, out, nthreads=2)
    assert (r_.retcode == 0).all() and r_.stats[0, 1] > 400 and r_.stats[0, 5] > 400
    check_per_trajectory(r_, ref)
    assert np.linalg.norm(r_.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
