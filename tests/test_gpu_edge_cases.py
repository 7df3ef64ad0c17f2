"""Edge cases of the boundary on the GPU: degenerate sizes, save grids, error paths, retcodes."""
import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models, training

pytestmark = pytest.mark.gpu
S1 = "Scenario_1_recovery_0.005"


def bitwise(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


def test_single_trajectory_single_save_point(golden):
    g = golden(S1)
    th = np.array(g["trained_parameters"])
    u0 = [0.44249296, 4.6280594]
    for ts in ([3.0], [0.0], [0.0, 3.0], [1.234], np.linspace(0, 3, 7)[1:-1]):   # with/without t0 and tf in the grid
        ts = np.array(ts, dtype=float)
        sol = U.solve(U.ODEProblem(models.ude_dynamics(), u0, (0.0, 3.0), th), U.Tsit5(), saveat=ts, abstol=1e-8, reltol=1e-8)
        out, st, rc = O.solve_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-8, 1e-8), u0, [0.0, 3.0], th, ts)
        assert sol.retcode == "Success" and bitwise(np.asarray(sol).T, out[0])
        assert (sol.destats.nf, sol.destats.naccept, sol.destats.nreject) == tuple(st[0][:3])


def test_gradient_with_sparse_save_grid_and_odd_ensemble_size(golden):
    g = golden(S1)
    th = np.array(g["initial_parameters"])
    rng = np.random.default_rng(0)
    N = 13                                    # not a multiple of the trajectories per wavefront
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    ts = np.array([0.0, 0.7, 1.9, 3.0])
    data = rng.uniform(0.5, 4.0, (N, len(ts), 2))
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (0.0, 3.0), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=ts, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [0.0, 3.0], th, ts, data)
    assert bitwise(r.stats, ref["stats"]) and bitwise(r.u, ref["u"]) and bitwise(r.grad_u0, ref["grad_u0"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < 1e-12 * np.linalg.norm(ref["grad_theta"])
    # save grid that does not contain tf: the adjoint starts with lambda = 0 at tf
    ts2 = np.array([0.5, 1.5, 2.5])
    r = U.loss_and_gradient(ens, U.Vern7(), data[:, :3], saveat=ts2, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.VERN7, 1e-6, 1e-6), u0, [0.0, 3.0], th, ts2, data[:, :3])
    assert bitwise(r.stats, ref["stats"]) and bitwise(r.grad_u0, ref["grad_u0"])


def test_retcodes_maxiters_and_dense_overflow(golden):
    g = golden(S1)
    th = np.array(g["initial_parameters"])
    u0 = [0.44249296, 4.6280594]
    t = np.arange(31) * 0.1
    prob = U.ODEProblem(models.ude_dynamics(), u0, (0.0, 3.0), th)
    sol = U.solve(prob, U.Tsit5(), saveat=t, abstol=1e-10, reltol=1e-10, maxiters=5)
    out, st, rc = O.solve_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-10, 1e-10, maxiters=5), u0, [0.0, 3.0], th, t)
    assert sol.retcode == "MaxIters" and rc[0] == 1
    X = np.repeat(np.array(u0)[None, None], 31, axis=1)
    # a pinned dense-store capacity that is too small: loud by default ...
    with pytest.raises(U.UdeError, match="retcode 4"):
        U.loss_and_gradient(prob, U.Tsit5(), X, saveat=t, abstol=1e-12, reltol=1e-12, ensemblealg=U.EnsembleMI355(0, 8))
    # ... and, when the caller opts in, reported with the gradient not polluted and the loss +Inf
    r = U.loss_and_gradient(prob, U.Tsit5(), X, saveat=t, abstol=1e-12, reltol=1e-12, ensemblealg=U.EnsembleMI355(0, 8),
                            allow_failures=True)
    assert r.retcode[0] == 4 and np.all(r.grad_theta == 0) and r.loss == np.inf
    # automatic capacity: the forward pass outgrows the initial 256 steps, the store grows x4 and the pass is repeated
    r = U.loss_and_gradient(prob, U.Tsit5(), X, saveat=t, abstol=1e-12, reltol=1e-12)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-12, 1e-12), np.array([u0]), [0.0, 3.0], th, t, X)
    assert r.retcode[0] == 0 and r.stats[0, 1] > 256 and bitwise(r.stats, ref["stats"]) and bitwise(r.grad_u0, ref["grad_u0"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < 1e-12 * np.linalg.norm(ref["grad_theta"])


def test_scalar_saveat_ends_at_tf_and_grid_is_validated():
    prob = U.ODEProblem(models.lotka(), [1.0, 1.0], (0.0, 1.0), [1.3, 0.9, 0.8, 1.8])
    sol = U.solve(prob, U.Tsit5(), saveat=0.3)          # SciML: save_end = true for a Number saveat
    assert np.allclose(sol.t, [0.0, 0.3, 0.6, 0.9, 1.0]) and np.asarray(sol).shape == (2, 5)
    for bad in ([0.5, 0.2], [0.5, 1.5], [-0.1, 0.5], [0.2, 0.2]):
        with pytest.raises(ValueError):
            U.solve(prob, U.Tsit5(), saveat=bad)


def test_invalid_arguments_fail_loudly():
    prob = U.ODEProblem(models.lotka(), [1.0, 1.0], (0.0, 1.0), [1.3, 0.9, 0.8, 1.8])
    with pytest.raises(AssertionError):
        U.solve(U.remake(prob, p=[1.0, 2.0]), U.Tsit5(), saveat=0.5)           # wrong theta length
    with pytest.raises(U.sciml.UdeError):
        U.solve(U.remake(prob, tspan=(1.0, 0.0)), U.Tsit5(), saveat=[0.5])     # tspan not increasing
    with pytest.raises(TypeError):
        U.solve(prob, U.Tsit5(), saveat=0.5, callback=print)                   # keyword the path does not implement
    with pytest.raises(U.sciml.UdeError):
        U.loss_and_gradient(U.ODEProblem(models.corona(), np.ones(7), (0.0, 1.0), []), U.Vern7(), np.ones((1, 2, 7)), saveat=[0.0, 1.0])


def test_adam_then_bfgs_reduce_the_scenario1_loss(golden):
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th0 = np.array(g["initial_parameters"])
    prob = U.ODEProblem(models.ude_dynamics(), X[0], (t[0], t[-1]), th0)

    def lg(theta):
        r = U.loss_and_gradient(U.remake(prob, p=np.asarray(theta)), U.Vern7(), X[None], saveat=t, abstol=1e-6, reltol=1e-6)
        return r.loss, r.grad_theta

    th1, l1 = training.adam(lg, th0, eta=0.1, maxiters=60)
    gold = g["losses"]["data_colmajor"]
    assert abs(l1[0] - gold[0]) < 1e-11 * gold[0]
    for k in (1, 2, 3, 10, 30):
        assert abs(l1[k] - gold[k]) < 1e-4 * gold[k], (k, l1[k], gold[k])      # the stored ADAM trajectory (ForwardDiff gradients upstream)
    th2, l2 = training.bfgs(lg, th1, initial_stepnorm=0.01, maxiters=40)
    assert l2[-1] < l2[0] and l2[-1] < 0.5 * l1[-1]


def test_scenario2_segment_loss_as_one_ensemble(golden):
    """scenario_2.jl:113-124 through the ensemble API (examples/scenario_2.py): stored losses[0] = 5298.020541174686."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("scenario_2", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "scenario_2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g, XS, TS, YS = mod.build()
    assert XS.shape == (5, 13)
    lg = mod.make_loss(XS, TS, YS)
    th = np.array(g["initial_parameters"])
    l0, g0 = lg(th)
    gold = g["losses"]["data_colmajor"]
    assert abs(l0 - gold[0]) < 1e-11 * gold[0]
    # gradient of the mixed abs2/abs loss + regulariser: directional finite difference
    rng = np.random.default_rng(0)
    d = rng.normal(size=th.size)
    d /= np.linalg.norm(d)
    h = 1e-6
    fd = (lg(th + h * d)[0] - lg(th - h * d)[0]) / (2 * h)
    assert abs(fd - g0 @ d) < 2e-5 * abs(fd)


def test_scenario2_adam_follows_the_stored_losses(golden):
    """scenario_2.jl:139-145 on the device: 200 iterations of ADAM(0.1) on the five-segment loss (per-trajectory time grids, the generic
    pullback, trainable delta) against the stored `losses` of Scenario_2_recovery_0.005: the first ten iterations to 5e-6, every one
    of the 200 to 5 % (the loss has an abs() term; BFGS from a point 4e-3 away from the reference's does not follow its losses)."""
    import importlib.util
    import os
    from universal_differential_equations_amd import training
    spec = importlib.util.spec_from_file_location("scenario_2", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "scenario_2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g, XS, TS, YS = mod.build()
    gold = np.array(g["losses"]["data_colmajor"])
    assert gold[199] == gold[200] == gold[201]
    for sense in (U.ForwardDiffSensitivity(), None):       # the script's sensealg (frozen-step sweep), and the interpolating adjoint
        lg = mod.make_loss(XS, TS, YS, sensealg=sense)
        _, la = training.adam(lg, np.array(g["initial_parameters"]), eta=0.1, maxiters=200)
        dev = np.abs(np.array(la) - gold[:200]) / gold[:200]
        print("scenario_2 ADAM (%s): worst relative deviation from the stored losses %.2e at %d, first ten %.1e, last %.1e"
              % (type(sense).__name__, dev.max(), int(dev.argmax()), dev[:10].max(), dev[-1]))
        # (the loss has an abs() term: where a residual changes sign the gradient jumps, and a 1e-7 difference decides on which side)
        assert dev[0] < 1e-11 and dev[:10].max() < 5e-6 and dev.max() < 5e-2, (dev.max(), int(dev.argmax()))


def test_multiple_shoot_on_device_matches_oracle_backend(golden):
    """hudson_bay.jl:108-118: the groups of DiffEqFlux.multiple_shoot as one ensemble through libudecore."""
    from universal_differential_equations_amd import training
    from test_multiple_shoot_cpu import OracleBackend
    g = golden("Scenario_1_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2).T
    t = np.array(g["t"])
    th = np.array(g["initial_parameters"])
    prob = U.ODEProblem(models.ude_dynamics(), X[:, 0], (0.0, 3.0), th)
    dev = training.EngineBackend(prob, U.Vern7(), abstol=1e-6, reltol=1e-6)
    ref = OracleBackend(O.lv_ude_s1(), O.opts(O.VERN7, 1e-6, 1e-6))
    l1, g1, p1 = training.multiple_shoot(th, X, t, dev, 5, continuity_term=200.0)
    l2, g2, p2 = training.multiple_shoot(th, X, t, ref, 5, continuity_term=200.0)
    assert abs(l1 - l2) <= 1e-13 * abs(l2)
    assert np.linalg.norm(g1 - g2) <= 1e-11 * np.linalg.norm(g2)
    for a, b in zip(p1, p2):
        assert (a == b).all()


def test_rhs_ensemble_matches_oracle(golden):
    """One right-hand-side evaluation per state (`U.rhs`): LV UDE, SEIR UDE, Fisher-KPP UDE (26 and 1024 points: VALU and
    matrix-core forward passes) against the oracle's udeo_rhs_f64, bit for bit."""
    rng = np.random.default_rng(9)
    g = golden("Scenario_1_recovery_0.005")
    th = np.array(g["trained_parameters"])
    u = np.abs(rng.normal(size=(70, 2))) + 0.1
    orhs = lambda m, th_, uu: np.stack([O.rhs(m, th_, row) for row in uu])
    assert (U.rhs(models.ude_dynamics(), u, th) == orhs(O.lv_ude_s1(), th, u)).all()
    ths = models.seir_chain().glorot_uniform(rng)
    us = np.abs(rng.normal(size=(9, 7))) * 1e5 + 1.0
    us[:, 4] = 14e6
    assert (U.rhs(models.dudt_(), us, ths) == orhs(O.seir_ude(), ths, us)).all()
    for nx in (26, 1024):
        thk = models.kpp_theta(models.kpp_chain(), rng)
        uk = rng.uniform(0, 1, size=(3, nx))
        assert (U.rhs(models.nn_ode(nx), uk, thk) == orhs(O.kpp_ude(nx), thk, uk)).all()


def test_per_trajectory_tspan_and_save_grids(golden):
    """scenario_2.jl:104-124: every shooting segment is solved on ITS OWN tspan = (T[1], T[end]) with saveat = T -- as one
    ensemble through the boundary's per-trajectory grids (no caller-side time shift); per member bit-identical to the
    oracle solving that member alone, for the plain solve, the interpolating adjoint and the discrete sweep."""
    g = golden("Scenario_2_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(61, 2)
    t = np.array(g["t"]["data_colmajor"]) if isinstance(g["t"], dict) else np.array(g["t"])
    th = np.array(g["initial_parameters"])
    idx = [np.arange(12 * k, 12 * k + 13) for k in range(5)]           # 5 segments of 13 points sharing their end points
    u0 = np.stack([X[i[0]] for i in idx])
    tspans = np.stack([[t[i[0]], t[i[-1]]] for i in idx])
    grids = np.stack([t[i] for i in idx])
    data = np.stack([X[i] for i in idx])
    f = models.ude_dynamics(trainable="delta")
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], tuple(tspans[0]), th), u0, tspans=tspans)
    sol = U.solve(ens, U.Vern7(), saveat=grids, abstol=1e-6, reltol=1e-6)
    assert np.array_equal(sol[3].t, grids[3])
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
        r = U.loss_and_gradient(ens, U.Vern7(), data, row_mask=[1, 0], saveat=grids, abstol=1e-6, reltol=1e-6, sensealg=sense)
        tot, gsum = 0.0, np.zeros(th.size)
        for k in range(5):
            o = O.opts(O.VERN7, 1e-6, 1e-6, sensealg=osense)
            out, st, rc = O.solve_ensemble(O.lv_ude_s2(), o, u0[k], tspans[k], th, grids[k])
            ref = O.loss_grad_ensemble(O.lv_ude_s2(), o, u0[k], tspans[k], th, grids[k], data[k][None], row_mask=[1, 0])
            assert bitwise(sol.u[k], out[0]) and bitwise(sol.stats[k, :4], st[0, :4])
            assert bitwise(r.u[k], ref["u"][0]) and bitwise(r.stats[k], ref["stats"][0]) and bitwise(r.grad_u0[k], ref["grad_u0"][0])
            assert r.loss_per_traj[k] == ref["loss"]
            tot += ref["loss"]
            gsum += ref["grad_theta"]
        assert abs(r.loss - tot) < 1e-12 * tot and np.linalg.norm(r.grad_theta - gsum) < 1e-12 * np.linalg.norm(gsum)
    # one shared grid with per-member spans is refused unless it lies inside every span; a bad member grid is refused
    with pytest.raises(ValueError):
        U.solve(ens, U.Vern7(), saveat=0.1)
    bad = grids.copy()
    bad[2, 5] = bad[2, 4]
    with pytest.raises(ValueError):
        U.solve(ens, U.Vern7(), saveat=bad)


def test_empty_ensemble_and_nan_member(golden):
    """an empty ensemble is refused; a member with a NaN initial state is reported (Unstable), makes the loss +Inf, and
    leaves every other member's results and the rest of the gradient untouched"""
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    f = models.ude_dynamics()
    with pytest.raises((U.UdeError, AssertionError, ValueError)):
        U.solve(U.EnsembleProblem(U.ODEProblem(f, X[0], (t[0], t[-1]), th), np.zeros((0, 2))), U.Tsit5(), saveat=t)
    rng = np.random.default_rng(4)
    N = 13
    u0 = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    data = np.repeat(X[None], N, axis=0)
    good = U.loss_and_gradient(U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), u0), U.Tsit5(), data, saveat=t,
                               abstol=1e-6, reltol=1e-6)
    bad_u0 = u0.copy()
    bad_u0[5, 1] = np.nan
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), bad_u0)
    with pytest.raises(U.UdeError):
        U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
    assert r.retcode[5] == 3 and (np.delete(r.retcode, 5) == 0).all() and np.isinf(r.loss)
    keep = np.arange(N) != 5
    assert np.array_equal(r.stats[keep], good.stats[keep]) and np.array_equal(r.grad_u0[keep], good.grad_u0[keep])
    assert np.array_equal(r.loss_per_traj[keep], good.loss_per_traj[keep])
    # the failed member is left out of the parameter gradient: it equals the gradient of the 12 healthy members
    rest = U.loss_and_gradient(U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), u0[keep]), U.Tsit5(), data[keep], saveat=t,
                               abstol=1e-6, reltol=1e-6)
    assert np.isfinite(r.grad_theta).all()
    assert np.linalg.norm(r.grad_theta - rest.grad_theta) < 1e-12 * np.linalg.norm(rest.grad_theta)


def test_full_size_ensemble_duplicate_and_linearity_properties(golden):
    """BASELINE configs[1] size (10 000 trajectories): size-independent properties -- a duplicated half gives duplicated
    per-trajectory bits, the ensemble loss / gradient are the sums of the halves', the cotangent pullback is linear"""
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    f = models.ude_dynamics()
    rng = np.random.default_rng(10)
    half = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (5000, 2)))
    u0 = np.concatenate([half, half])
    data = np.repeat(X[None], 10000, axis=0)
    prob = U.ODEProblem(f, u0[0], (t[0], t[-1]), th)
    full = U.loss_and_gradient(U.EnsembleProblem(prob, u0), U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    assert (full.retcode == 0).all()
    for a in (full.stats, full.u, full.grad_u0, full.loss_per_traj):
        assert np.array_equal(a[:5000], a[5000:])
    h = U.loss_and_gradient(U.EnsembleProblem(prob, half), U.Tsit5(), data[:5000], saveat=t, abstol=1e-6, reltol=1e-6)
    assert abs(full.loss - 2 * h.loss) < 1e-12 * full.loss
    assert np.linalg.norm(full.grad_theta - 2 * h.grad_theta) < 1e-12 * np.linalg.norm(full.grad_theta)
    # linearity of the pullback in the cotangent (one trajectory, the backward step sequence depends on the cotangent only
    # through error control: compare at the level the tolerance allows)
    p1 = U.ODEProblem(f, half[0], (t[0], t[-1]), th)
    c1, c2 = rng.standard_normal((1, 31, 2)), rng.standard_normal((1, 31, 2))
    ga = U.adjoint_pullback(p1, U.Tsit5(), c1, saveat=t, abstol=1e-9, reltol=1e-9).grad_theta
    gb = U.adjoint_pullback(p1, U.Tsit5(), c2, saveat=t, abstol=1e-9, reltol=1e-9).grad_theta
    gab = U.adjoint_pullback(p1, U.Tsit5(), 2.0 * c1 - 0.5 * c2, saveat=t, abstol=1e-9, reltol=1e-9).grad_theta
    assert np.linalg.norm(gab - (2.0 * ga - 0.5 * gb)) < 1e-6 * np.linalg.norm(gab)


def test_device_gradient_is_capturable_into_a_hip_graph(golden):
    """the `_dev` gradient call inside a hipGraph capture (torch.cuda.graph): replay == the plain call bit for bit, and an
    in-place update of theta is seen by the next replay"""
    import torch
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    rng = np.random.default_rng(12)
    N = 300
    dev = torch.device("cuda:0")
    u0 = torch.tensor(X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2))), device=dev)
    data = torch.tensor(np.repeat(X[None], N, axis=0), device=dev)
    ens = U.DeviceEnsemble(models.ude_dynamics(), U.Tsit5(), (t[0], t[-1]), t, u0, data=data, abstol=1e-6, reltol=1e-6)
    theta = torch.tensor(th, device=dev)
    plain = ens.loss_grad(theta).clone()
    replay = ens.graph(theta)
    for _ in range(3):
        assert torch.equal(replay(), plain)
    theta.mul_(1.01)                                   # a training step updates theta in place
    stepped = replay().clone()
    torch.cuda.synchronize()
    assert not torch.equal(stepped, plain)
    assert torch.equal(ens.loss_grad(theta), stepped)   # and the plain path agrees at the new theta


def test_wide_gradient_matrix_reduction_path():
    """>= 256 wavefronts x >= 512 parameters take the coalesced finishing kernel (finish_wide_kernel): the SEIR exposure UDE
    with 300 trajectories -- ensemble gradient and loss against the oracle's sums"""
    from test_gpu_parity import seir_inputs
    N = 300
    u0, t = seir_inputs(N, seed=8)
    t = t[:8]
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 7.0], [], t, nthreads=8)
    th = models.seir_chain().glorot_uniform(np.random.default_rng(3))
    th[-65:-1] *= 10.0
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 7.0), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), truth, row_mask=[0, 1, 1, 1, 0, 0, 0], saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [0.0, 7.0], th, t, truth, row_mask=[0, 1, 1, 1, 0, 0, 0], nthreads=8)
    assert (r.retcode == 0).all() and np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < 1e-12 * np.linalg.norm(ref["grad_theta"])
    assert abs(r.loss - ref["loss"]) < 1e-12 * ref["loss"]


@pytest.mark.parametrize("case", ["s1_adjoint", "s1_discrete", "hudson_adjoint", "hudson_f32_discrete", "s1_vern7_pt_grids", "tanh32_adjoint",
                                  "tanh32_discrete"])
def test_per_member_parameters_are_n_independent_recoveries(golden, case):
    """LotkaVolterra/run_loops.jl:55-62 runs 500 INDEPENDENT recoveries -- every member its own data and its own network -- one after
    the other; `EnsembleProblem(prob, u0s, ps = thetas)` (UDE_PT_THETA) runs them as one ensemble: member j reads its own parameter
    column and gets its own gradient row, nothing is summed over trajectories.  Every member bit-identical to the oracle solving
    that member alone with its parameters (states, step counts forward and backward, loss, dL/du0 AND every gradient entry)."""
    rng = np.random.default_rng(77)
    f32 = np.float32
    sense, osense = (U.ForwardDiffSensitivity(), 1) if "discrete" in case else (None, 0)
    if case.startswith("hudson"):
        dt = f32 if "f32" in case else np.float64
        g = golden("Hudson_Bay_recovery")
        X = np.array(g["X"]["data_colmajor"], dtype=dt).reshape(21, 2)[:9]
        t = np.arange(9, dtype=dt) * dt(0.25)
        f = models.ude_dynamics(models.hudson_chain(), trainable="both", dtype="float32" if dt is f32 else "float64")
        om = O.lv_ude_hudson(1 if dt is f32 else 0)
        base = np.concatenate([[1.3, 1.8], 0.3 * models.hudson_chain().glorot_uniform(rng)])
        alg, oalg, tol = U.Vern7, O.VERN7, 1e-5
    else:
        dt = np.float64
        g = golden("Scenario_1_recovery_0.005")
        X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
        t = np.array(g["solution"]["t"])
        f, om = models.ude_dynamics(), O.lv_ude_s1()
        base = np.array(g["initial_parameters"])
        if case.startswith("tanh32"):                        # the 2-32-2 net reads its weights at every use: the member's column in HBM
            f, om = models.ude_dynamics(models.tanh32_chain()), O.lv_ude_tanh32()
            base = 0.3 * models.tanh32_chain().glorot_uniform(rng) + 0.02 * rng.standard_normal(162)
        alg, oalg, tol = (U.Vern7, O.VERN7, 1e-7) if "vern7" in case else (U.Tsit5, O.TSIT5, 1e-6)
    N = 17                                                   # two wavefronts of 12 (5 lanes) / three of 8 (8 lanes), the last partial
    thetas = (base[None, :] * (1 + 0.2 * rng.standard_normal((N, base.size)))).astype(dt)
    u0 = (X[0][None, :] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))).astype(dt)
    data = (X[None] * (1 + 0.05 * rng.standard_normal((N,) + X.shape))).astype(dt)
    tspans, grids = None, t
    if "pt_grids" in case:                                   # ... combined with per-member spans and save grids (UDE_PT_TSPAN | SAVEAT | THETA)
        ends = rng.uniform(1.5, 3.0, N)
        tspans = np.stack([np.zeros(N), ends], axis=1)
        grids = np.stack([np.linspace(0.0, e, len(t)) for e in ends])
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (float(t[0]), float(t[-1])), thetas[0]), u0, tspans=tspans, ps=thetas)
    sol = U.solve(ens, alg(), saveat=grids, abstol=tol, reltol=tol)
    r = U.loss_and_gradient(ens, alg(), data, saveat=grids, abstol=tol, reltol=tol, sensealg=sense)
    assert r.grad_theta.shape == (N, base.size) and (r.retcode == 0).all()
    total = 0.0
    for j in range(N):
        tj = grids[j] if tspans is not None else t
        sp = [tspans[j, 0], tspans[j, 1]] if tspans is not None else [t[0], t[-1]]
        ref = O.loss_grad_ensemble(om, O.opts(oalg, tol, tol, sensealg=osense), u0[j:j + 1], sp, thetas[j], tj, data[j:j + 1], dtype=dt)
        what = "%s member %d" % (case, j)
        assert np.array_equal(sol.u[j], ref["u"][0]) and np.array_equal(sol.stats[j, :3], ref["stats"][0, :3]), what   # (plain solve: lazy stages count in column 3, gradient calls: 7)
        assert np.array_equal(r.u[j], ref["u"][0]) and np.array_equal(r.stats[j], ref["stats"][0]), what
        assert np.array_equal(r.grad_u0[j], ref["grad_u0"][0]) and r.loss_per_traj[j] == ref["loss_per_traj"][0], what
        assert np.array_equal(r.grad_theta[j], ref["grad_theta"]), what      # one trajectory, one row: no sum, so bit for bit
        total += float(ref["loss_per_traj"][0])
    assert abs(float(r.loss) - total) <= 1e-6 * abs(total)


@pytest.mark.parametrize("case", ["small15", "small15_discrete", "cnn26_vern7", "runtime_shape", "cnn1024", "cnn300"])
def test_per_member_parameters_fisher_kpp_repeated_trainings(case):
    """round 5: FisherKPP/Fisher-KPP-CNN-Small.jl:311-391 trains the same model five times from five initial networks, one run after
    the other (the only wall-clock numbers the reference publishes).  With UDE_PT_THETA on the Fisher-KPP kinds the runs are the
    members of ONE ensemble: member j reads its own theta column (the compiled 1-3-1 / 1-10-20-10-1 instances, the runtime-shape
    reaction chain, the 1024-point matrix-core kernel) and gets its own gradient row -- every member bit-identical to the oracle
    solving that member alone (states, counts, dL/du0, every gradient entry)."""
    rng = np.random.default_rng(len(case))
    sense, osense = (U.ForwardDiffSensitivity(), 1) if "discrete" in case else (None, 0)
    alg, oalg, kw = (U.Vern7, O.VERN7, dict(abstol=1e-6, reltol=1e-6)) if "vern7" in case else (U.Tsit5, O.TSIT5, {})
    if case.startswith("small15"):
        nx, chain, om, N = 26, models.kpp_small_chain(3), O.kpp_ude(26, (1, 3, 1), ("tanh", "identity")), 5
    elif case == "runtime_shape":
        dims, acts = [1, 6, 4, 1], ["tanh", "rbf", "identity"]
        chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(3)])
        nx, om, N = 19, O.kpp_ude(19, tuple(dims), tuple(acts)), 4
    elif case == "cnn1024":
        nx, chain, om, N = 1024, models.kpp_chain(), O.kpp_ude(1024), 3
    elif case == "cnn300":
        nx, chain, om, N = 300, models.kpp_chain(), O.kpp_ude(300), 2
    else:
        nx, chain, om, N = 26, models.kpp_chain(), O.kpp_ude(26), 5
    f = models.nn_ode(nx, chain)
    thetas = np.stack([models.kpp_theta(chain, rng) for _ in range(N)])
    if nx <= 100:
        thetas[:, f.d0_offset] = rng.uniform(1.0, 6.0, N)
    else:   # (dx stays 0.04 on the large grids, SURVEY 8(d) C4: a large D0 there is the stiff regime)
        thetas[:, f.d0_offset] *= 1 + 0.1 * rng.uniform(-1, 1, N)
    u0 = np.clip(models.rho0(26)[None, :] * (1 + 0.1 * rng.uniform(-1, 1, (N, 1))) + 0.01 * rng.uniform(0, 1, (N, 26)), 0, None)
    u0 = np.tile(u0, (1, 40))[:, :nx]
    tf = 1.0 if nx > 100 else 5.0
    t = np.linspace(0.0, tf, 6)
    data = rng.uniform(0.0, 1.0, (N, len(t), nx))
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), thetas[0]), u0, ps=thetas)
    sol = U.solve(ens, alg(), saveat=t, **kw)
    r = U.loss_and_gradient(ens, alg(), data, saveat=t, sensealg=sense, **kw)
    assert r.grad_theta.shape == thetas.shape and (r.retcode == 0).all()
    o = O.opts(oalg, kw.get("abstol", 0.0), kw.get("reltol", 0.0), sensealg=osense)
    for j in range(N):
        ref = O.loss_grad_ensemble(om, o, u0[j:j + 1], [0.0, tf], thetas[j], t, data[j:j + 1])
        what = "%s member %d" % (case, j)
        assert bitwise(sol.u[j], ref["u"][0]) and bitwise(sol.stats[j, :3], ref["stats"][0, :3]), what
        assert bitwise(r.u[j], ref["u"][0]) and bitwise(r.stats[j], ref["stats"][0]), what
        assert bitwise(r.grad_u0[j], ref["grad_u0"][0]), what
        # (the loss of a distributed state is summed per lane, then over the lanes: not the oracle's order over the points)
        assert abs(r.loss_per_traj[j] - ref["loss_per_traj"][0]) <= 1e-13 * abs(ref["loss_per_traj"][0]), what
        assert bitwise(r.grad_theta[j], ref["grad_theta"]), what


def test_per_member_parameters_are_refused_where_no_kernel_takes_them():
    f = models.dudt_()                                      # SEIR exposure UDE: theta lives in register fragments of the whole block
    th = models.seir_theta(np.random.default_rng(0)) if hasattr(models, "seir_theta") else 0.1 * np.random.default_rng(0).standard_normal(4481)
    u0 = np.array([[12e6, 0, 0, 0, 14e6, 0, 0], [12.5e6, 0, 0, 0, 14e6, 0, 0]], dtype=np.float64)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 1.0), th), u0, ps=np.stack([th, th]))
    with pytest.raises(U.UdeError, match="per-member parameters"):
        U.loss_and_gradient(ens, U.Tsit5(), np.zeros((2, 3, 7)), saveat=[0.0, 0.5, 1.0])
    # ude_model_supported gives the same answer before any solve is attempted (and 0 / 1 for the kinds that have a per-member kernel)
    import ctypes as C
    from universal_differential_equations_amd import _lib
    eng = U.Engine.get(0)
    eng.set_launch()
    o = _lib.SolveOpts()
    o.per_trajectory = 4   # UDE_PT_THETA
    assert eng.L.ude_model_supported(eng.h, C.byref(f), C.byref(o), 1) == -2   # UDE_ERR_UNSUPPORTED
    assert eng.L.ude_model_supported(eng.h, C.byref(models.nn_ode(26, models.kpp_small_chain(3))), C.byref(o), 1) == 0
    assert eng.L.ude_model_supported(eng.h, C.byref(models.ude_dynamics()), C.byref(o), 1) == 0
