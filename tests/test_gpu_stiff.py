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


def test_checkpointed_adjoint_is_refused_where_no_instance_exists():
    """the one combination left without it: a distributed state whose recomputed stage vectors do not fit the registers -- the
    1024-point Fisher-KPP UDE with Vern7 (16 stage vectors of 4 points per lane) -- and per-trajectory time grids"""
    nx = 1024
    f = models.nn_ode(nx)
    thk = models.kpp_theta(models.kpp_chain(), np.random.default_rng(0))
    probk = U.ODEProblem(f, models.rho0(nx), (0.0, 0.02), thk)
    with pytest.raises(Exception, match="checkpointed adjoint"):
        U.loss_and_gradient(probk, U.Vern7(), np.zeros((1, 3, nx)), saveat=[0.0, 0.01, 0.02], sensealg=U.InterpolatingAdjoint(checkpointing=True))


def _lv_case(golden, chain=None, trainable=None):
    g = golden("Scenario_1_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    rng = np.random.default_rng(11)
    N = 13
    u0 = X[0][None, :] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    data = np.repeat(X[None], N, axis=0)
    if chain is None:
        f, th = models.ude_dynamics(), np.array(g["initial_parameters"])
    else:
        f = models.ude_dynamics(chain, trainable=trainable) if trainable else models.ude_dynamics(chain)
        th = 0.3 * chain.glorot_uniform(rng)
        if trainable == "both":
            th = np.concatenate([[1.3, 1.8], th])
    return f, th, u0, data, t


@pytest.mark.parametrize("alg", ["Tsit5", "Vern7"])
@pytest.mark.parametrize("case", ["lv_s1", "lv_s1_lanes1", "lv_hudson", "lv_tanh32", "lv_generic", "seir", "node", "kpp_vern7_26"])
def test_checkpointed_adjoint_every_model_equals_dense_store_bit_for_bit(golden, case, alg):
    """round 4: `InterpolatingAdjoint(checkpointing = true)` for every model kind and both algorithms (SURVEY.md 8(b) names
    `store dense|recompute` as a general option; seir_exposure.jl:138-140 is Vern7 + InterpolatingAdjoint).  The forward store keeps
    (t, t_end, dt, u); the adjoint kernel recomputes the interval's stages -- for Vern7 its six lazy dense-output stages too -- with
    the forward pass's operation sequence: loss, gradient, dL/du0 and all step counts identical to the dense-store mode."""
    A = getattr(U, alg)
    ck = U.InterpolatingAdjoint(checkpointing=True)
    kw = dict(abstol=1e-6, reltol=1e-6)
    ealg = None
    if case.startswith("lv"):
        chain, tr = {"lv_s1": (None, None), "lv_s1_lanes1": (None, None), "lv_hudson": (models.hudson_chain(), "both"),
                     "lv_tanh32": (models.tanh32_chain(), None),
                     "lv_generic": (models.Chain(models.Dense(2, 6, "tanh"), models.Dense(6, 7, "rbf"), models.Dense(7, 2)), None)}[case]
        f, th, u0, data, t = _lv_case(golden, chain, tr)
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), u0)
        if case == "lv_s1_lanes1":
            ealg = U.EnsembleMI355(1)
        kw["saveat"] = t
    elif case in ("seir", "node"):
        rng = np.random.default_rng(5)
        node = case == "node"
        f = models.dudt_node() if node else models.dudt_()
        chain = models.seir_node_chain() if node else models.seir_chain()
        th = chain.glorot_uniform(rng)
        N, S0 = 5, 1e4
        u0 = np.zeros((N, 7)); u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0; u0[:, 1] = rng.uniform(0, 5, N); u0[:, 2] = rng.uniform(0, 2, N); u0[:, 4] = S0
        t = np.linspace(0.0, 6.0, 7)
        data = u0[:, None, :] * (1 + 0.05 * rng.standard_normal((N, len(t), 7)))
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 6.0), th), u0)
        kw.update(saveat=t, row_mask=[0, 1, 1, 0, 0, 1, 0])
    else:
        if alg == "Tsit5":
            pytest.skip("covered by test_checkpointed_adjoint_equals_dense_store_bit_for_bit")
        rng = np.random.default_rng(3)
        nx = 26
        f = models.nn_ode(nx)
        th = models.kpp_theta(models.kpp_chain(), rng)
        u0 = models.rho0(nx)[None, :] * (1 + 0.05 * rng.uniform(-1, 1, (3, 1)))
        t = np.arange(6) * 0.5
        data = np.repeat(u0[:, None, :], len(t), axis=1) * 0.9
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 2.5), th), u0)
        kw["saveat"] = t
    dense = U.loss_and_gradient(ens, A(), data, ensemblealg=ealg, **kw)
    rec = U.loss_and_gradient(ens, A(), data, sensealg=ck, ensemblealg=ealg, **kw)
    assert (rec.retcode == 0).all() and rec.stats[:, 5].min() > 3
    for k in ("loss", "grad_theta", "grad_u0", "stats", "u", "loss_per_traj", "retcode"):
        a_, b_ = np.asarray(getattr(rec, k)), np.asarray(getattr(dense, k))
        if k == "stats":
            # (columns 3 / 7 count Vern7's lazy dense-output stages of the FORWARD pass: the dense store builds them for every accepted
            # step, the checkpointed store only where a save point lies inside the step -- the adjoint recomputes them itself)
            a_, b_ = a_[:, [0, 1, 2, 4, 5, 6]], b_[:, [0, 1, 2, 4, 5, 6]]
        if k in ("loss", "grad_theta") and case in ("seir", "node"):
            # (the dense-store default of these two models is the lock-step kernel, one gradient row per trajectory: the sum over
            # trajectories is associated differently; everything per trajectory is bit-identical)
            assert np.linalg.norm(a_ - b_) <= REL_GRAD_SUM * np.linalg.norm(b_), k
        else:
            assert_bitwise(a_, b_, "%s %s: %s" % (case, alg, k))
