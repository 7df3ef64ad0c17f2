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


def _same(a, b, what):
    for k in ("loss", "grad_theta", "grad_u0", "stats", "u", "loss_per_traj", "retcode"):
        assert_bitwise(np.asarray(getattr(a, k)), np.asarray(getattr(b, k)), "%s: %s" % (what, k))


def test_checkpointed_adjoint_equals_dense_store_bit_for_bit():
    """InterpolatingAdjoint(checkpointing = true) as store-u-only + recompute: the forward store keeps (t, dt, u) per
    accepted step, the adjoint kernel re-runs the step's stages when it enters the interval.  Same operations on the same inputs:
    loss, every gradient entry, dL/du0, forward and backward step counts identical to the dense-store mode -- at the reference's
    size (Fisher-KPP-CNN.jl:111-143: 26 points, the oracle as third party) and on the stiff 1024-point variant that needs it."""
    rng = np.random.default_rng(3)
    ck = U.InterpolatingAdjoint(checkpointing=True)
    # reference size
    nx = 26
    f = models.nn_ode(nx)
    th = models.kpp_theta(models.kpp_chain(), rng)
    rho = models.rho0(nx)[None, :] * (1 + 0.05 * rng.uniform(-1, 1, (3, 1)))
    t = np.arange(11) * 0.5
    truth, _, rc = O.solve_ensemble(O.kpp_true(nx), O.opts(O.TSIT5), rho, [0.0, 5.0], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(f, rho[0], (0.0, 5.0), th), rho)
    dense = U.loss_and_gradient(ens, U.Tsit5(), truth, saveat=t)
    rec = U.loss_and_gradient(ens, U.Tsit5(), truth, saveat=t, sensealg=ck)
    _same(rec, dense, "26 points")
    ref = O.loss_grad_ensemble(O.kpp_ude(nx), O.opts(O.TSIT5), rho, [0.0, 5.0], th, t, truth)
    check_per_trajectory(rec, ref)
    # the stiff 1024-point variant (short horizon)
    nx, D = 1024, 0.01
    dx = 1.0 / (nx - 1)
    x = np.arange(nx) * dx
    u0 = (0.5 * (np.tanh((x - 0.3) / 0.2) - np.tanh((x - 0.7) / 0.2)))[None, :] * (1 + 0.05 * rng.uniform(-1, 1, (2, 1)))
    T = 0.04
    t = np.linspace(0.0, T, 5)
    f = models.nn_ode(nx)
    th = models.kpp_theta(models.kpp_chain(), rng)
    th[f.d0_offset] = 0.95 * D / dx ** 2
    th[f.stencil_offset:f.stencil_offset + 3] = [1.01, -2.0, 0.99]
    data = np.repeat(u0[:, None, :], 5, axis=1)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, T), th), u0)
    dense = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, ensemblealg=U.EnsembleMI355(0, 4096))
    rec = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, sensealg=ck, ensemblealg=U.EnsembleMI355(0, 4096))
    assert (rec.retcode == 0).all() and rec.stats[0, 1] > 400 and rec.stats[0, 5] > 400
    _same(rec, dense, "stiff 1024 points")


def test_checkpointed_adjoint_is_refused_where_no_instance_exists(golden):
    g = golden("Scenario_1_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    prob = U.ODEProblem(models.ude_dynamics(), X[0], (t[0], t[-1]), th)
    with pytest.raises(Exception, match="checkpointed adjoint"):
        U.loss_and_gradient(prob, U.Vern7(), X[None], saveat=t, abstol=1e-6, reltol=1e-6, sensealg=U.InterpolatingAdjoint(checkpointing=True))
    f = models.nn_ode(26)
    thk = models.kpp_theta(models.kpp_chain(), np.random.default_rng(0))
    probk = U.ODEProblem(f, models.rho0(26), (0.0, 1.0), thk)
    with pytest.raises(Exception, match="checkpointed adjoint"):   # Vern7: its dense output needs six stages beyond the step's own
        U.loss_and_gradient(probk, U.Vern7(), np.zeros((1, 3, 26)), saveat=[0.0, 0.5, 1.0], sensealg=U.InterpolatingAdjoint(checkpointing=True))
