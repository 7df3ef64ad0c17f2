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
REL_GRAD_SUM = 1e-12   # ensemble-summed gradient / loss: only the order of the sum over trajectories differs


def assert_bitwise(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    same = (a == b) | (np.isnan(a) & np.isnan(b)) if a.dtype.kind == "f" else (a == b)
    assert same.all(), "%s: %d of %d entries differ (max |diff| %.3e)" % (
        what, (~same).sum(), same.size, np.nanmax(np.abs(a.astype(float) - b.astype(float))))


def check_per_trajectory(r, ref, with_adjoint=True, loss_bitwise=True):
    """ARITH-SPEC: the kernels and the oracle evaluate every trajectory with the same sequence of
    IEEE operations, so everything that belongs to ONE trajectory is bit-identical -- step counts forward and
    backward, saved states, per-trajectory loss and dL/du0.  (The only association that differs is the sum of
    squares inside the error norm, which moves EEst by ~1e-16 relative and is invisible after the controller's
    Float32 quantisation except with probability ~1e-9 per step.)"""
    assert_bitwise(r.retcode, ref["retcode"], "retcode")
    assert_bitwise(r.stats[:, :4], ref["stats"][:, :4], "forward nf/naccept/nreject/nf_lazy")
    assert_bitwise(r.u, ref["u"], "saved states")
    if with_adjoint:
        assert_bitwise(r.stats[:, 4:8], ref["stats"][:, 4:8], "backward nf/naccept/nreject + lazy stages")
        assert_bitwise(r.grad_u0, ref["grad_u0"], "dL/du0")


def s1_data(golden):
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    return g, X, t


def ensemble_u0(X, N, seed=1234):
    rng = np.random.default_rng(seed)
    return np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))


def test_arith_spec_primitives_bitwise_equal_to_oracle():
    import ctypes as C
    L = O.lib()
    eng = U.Engine.get(0)
    rng = np.random.default_rng(1)
    cases = {1: ("udeo_exp", np.concatenate([rng.uniform(-40, 40, 30000), rng.uniform(-1, 1, 10000), [0.0, -800.0, 720.0]])),
             2: ("udeo_tanh", np.concatenate([rng.uniform(-25, 25, 30000), rng.uniform(-0.5, 0.5, 20000), 10.0 ** rng.uniform(-12, -1, 2000), [0.0]])),
             3: ("udeo_log10", 10.0 ** rng.uniform(-18, 18, 40000)),
             4: ("udeo_pow10", rng.uniform(-12, 3, 40000))}
    for op, (name, x) in cases.items():
        f = getattr(L, name)
        f.restype, f.argtypes = C.c_double, [C.c_double]
        ref = np.array([f(float(v)) for v in x])
        assert_bitwise(eng.math(op, x), ref, name)
    L.udeo_log.restype, L.udeo_log.argtypes = C.c_double, [C.c_double]
    L.udeo_pow.restype, L.udeo_pow.argtypes = C.c_double, [C.c_double, C.c_double]
    x = 10.0 ** rng.uniform(-18, 18, 20000)
    assert_bitwise(eng.math(8, x), np.array([L.udeo_log(float(v)) for v in x]), "udeo_log")
    xb, yb = rng.uniform(0.9, 1.0, 20000), rng.uniform(1.0, 1200.0, 20000)
    assert_bitwise(eng.math(9, xb, yb), np.array([L.udeo_pow(float(a_), float(b_)) for a_, b_ in zip(xb, yb)]), "udeo_pow")
    x = 10.0 ** rng.uniform(-200, 200, 40000)
    y = 10.0 ** rng.uniform(-200, 200, 40000) * rng.choice([-1.0, 1.0], 40000)
    assert_bitwise(eng.math(5, x), np.sqrt(x), "sqrt is correctly rounded")
    with np.errstate(over="ignore", under="ignore"):
        assert_bitwise(eng.math(6, x, y), x / y, "division is correctly rounded")
    a, b = rng.normal(size=40000), rng.normal(size=40000)
    import math
    if hasattr(math, "fma"):
        assert_bitwise(eng.math(7, a, b), np.array([math.fma(p, q, p) for p, q in zip(a, b)]), "fma")


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
    assert_bitwise(got, out[0], "states vs oracle")        # same arithmetic => same bits, even over t in [0,50]
    if tol is not None:
        assert np.abs(got - np.array(s["u"])).max() < 1e-12   # and the golden itself


def test_scenario1_loss_known_answers_on_gpu(golden):
    g, X, t = s1_data(golden)
    f = models.ude_dynamics()
    for th, want, counts in ((g["initial_parameters"], g["losses"]["data_colmajor"][0], (142, 14, 0)),
                             (g["trained_parameters"], g["losses"]["data_colmajor"][-1], (202, 18, 2))):
        sol = U.solve(U.ODEProblem(f, X[0], (t[0], t[-1]), th), U.Vern7(), saveat=t, abstol=1e-6, reltol=1e-6)
        loss = float(((X - np.asarray(sol).T) ** 2).sum())
        assert (sol.destats.nf, sol.destats.naccept, sol.destats.nreject) == counts, sol.destats
        assert abs(loss - want) < 1e-8 * want, (loss, want)
        out, st, rc = O.solve_ensemble(O.lv_ude_s1(), O.opts(O.VERN7, 1e-6, 1e-6), X[0], [t[0], t[-1]], th, t)
        assert_bitwise(np.asarray(sol).T, out[0], "states vs oracle")


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
    assert_bitwise(sol.retcodes, rc, "retcode")
    ok = rc == 0
    assert ok.sum() > 0.9 * N
    assert_bitwise(sol.stats[ok, :4], st[ok, :4], "nf/naccept/nreject/nf_lazy")
    assert_bitwise(sol.u[ok], out[ok], "saved states")


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
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")
    assert abs(r.loss - ref["loss"]) < REL_GRAD_SUM * abs(ref["loss"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])


@pytest.mark.parametrize("lanes", [1, 4, 5, 8])
def test_lanes_per_trajectory_variants_agree(golden, lanes):
    g, X, t = s1_data(golden)
    th = np.array(g["initial_parameters"])
    N = 80
    u0 = ensemble_u0(X, N, 3)
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(lanes))
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, nthreads=4)
    check_per_trajectory(r, ref)      # every lane-group size reproduces the same bits
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])


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
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    data = np.repeat(X[None], N, axis=0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, row_mask=[0, 1], saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, row_mask=[0, 1], nthreads=4)
    check_per_trajectory(r, ref)
    assert abs(r.loss - ref["loss"]) < REL_GRAD_SUM * ref["loss"]
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    # masked-out rows are ignored entirely (the reference slices them away): NaN there must not reach loss or gradient
    data_nan = data.copy()
    data_nan[:, :, 0] = np.nan
    r2 = U.loss_and_gradient(ens, U.Tsit5(), data_nan, row_mask=[0, 1], saveat=t, abstol=1e-6, reltol=1e-6)
    assert r2.loss == r.loss and np.array_equal(r2.grad_theta, r.grad_theta) and np.array_equal(r2.grad_u0, r.grad_u0)


def test_adam_trajectory_known_answer_on_gpu(golden):
    """scenario_1.jl:99-114 with the GPU gradient: the stored losses[0..199] -- every one of the reference's 200 ADAM iterations
     -- with the interpolating adjoint and with the discretise-then-optimise sweep the script itself requests."""
    g, X, t = s1_data(golden)
    gold = g["losses"]["data_colmajor"]
    th = np.array(g["initial_parameters"])
    f = models.ude_dynamics()
    eta, b1, b2, eps = 0.1, 0.9, 0.999, np.finfo(float).eps
    for sense in (None, U.ForwardDiffSensitivity()):
        th = np.array(g["initial_parameters"])
        mt, vt, b1t, b2t = np.zeros_like(th), np.zeros_like(th), b1, b2
        worst = 0.0
        for k in range(200):
            r = U.loss_and_gradient(U.ODEProblem(f, X[0], (t[0], t[-1]), th), U.Vern7(), X[None], saveat=t, abstol=1e-6, reltol=1e-6, sensealg=sense)
            dev = abs(r.loss - gold[k]) / gold[k]
            worst = max(worst, dev)
            assert dev < (1e-11 if k < 2 else 5e-6), (sense, k, r.loss, gold[k])
            gr = r.grad_theta
            mt = b1 * mt + (1 - b1) * gr
            vt = b2 * vt + (1 - b2) * gr * gr
            th = th - eta * (mt / (1 - b1t)) / (np.sqrt(vt / (1 - b2t)) + eps)
            b1t *= b1
            b2t *= b2
        print("ADAM trajectory, 200 iterations, worst relative deviation from the stored losses:", worst)


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
    assert_bitwise(full.stats[idx], ref["stats"], "stats of the subsample")
    assert_bitwise(full.loss_per_traj[idx], ref["loss_per_traj"], "per-trajectory loss")
    assert_bitwise(full.grad_u0[idx], ref["grad_u0"], "dL/du0")
    assert_bitwise(full.u[idx], ref["u"], "saved states")


def seir_inputs(N, seed=3):
    """u0_j = (f_j*S0, 0,0,0, S0, 0,0), S0 = 14e6, f_j in U(0.8,0.95); tspan (0,21), saveat 0:1:21;
    data from corona! (seir_exposure.jl:16-37) solved by the engine itself at tol 1e-12."""
    rng = np.random.default_rng(seed)
    S0 = 14e6
    u0 = np.zeros((N, 7))
    u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
    u0[:, 4] = S0
    t = np.arange(22.0)
    return u0, t


def test_seir_true_matches_oracle():
    u0, t = seir_inputs(40)
    ens = U.EnsembleProblem(U.ODEProblem(models.corona(), u0[0], (0.0, 21.0), []), u0)
    sol = U.solve(ens, U.Vern7(), saveat=t, abstol=1e-12, reltol=1e-12)
    out, st, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    assert_bitwise(sol.retcodes, rc, "retcode")
    assert_bitwise(sol.stats[:, :4], st[:, :4], "counts")
    assert_bitwise(sol.u, out, "states")
    assert (rc == 0).all() and out[:, -1, 2].min() > 0          # the epidemic actually develops


@pytest.mark.parametrize("alg,oalg,lanes", [(U.Vern7, O.VERN7, 0), (U.Tsit5, O.TSIT5, 0), (U.Vern7, O.VERN7, 256), (U.Tsit5, O.TSIT5, 256),
                                            (U.Vern7, O.VERN7, 16), (U.Tsit5, O.TSIT5, 16), (U.Vern7, O.VERN7, 64)])
def test_seir_ude_forward_and_adjoint_match_oracle(alg, oalg, lanes, N=12):
    """dudt_ (seir_exposure.jl:114-147): 3-64-64-1 tanh exposure network, loss on rows 2:4 (E, I, R).
    lanes = 64: one wavefront per trajectory (deferred parameter cotangent); 256: four wavefronts per trajectory
    (register-resident weight slices); 16: the lock-step backward kernel (16 trajectories per block as columns of FP64
    matrix-core products, csrc/ude_seir_ls.h) -- all bit-identical to the oracle's wide-dot arithmetic."""
    kw = {"ensemblealg": U.EnsembleMI355(lanes)} if lanes else {}
    u0, t = seir_inputs(N)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    rng = np.random.default_rng(11)
    th = models.seir_chain().glorot_uniform(rng)
    th[-65:-1] *= 10.0      # a larger exposure term (stronger coupling into S and E) without leaving the non-stiff regime
    mask = [0, 1, 1, 1, 0, 0, 0]
    f = models.dudt_()
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 21.0), th), u0)
    sol = U.solve(ens, alg(), saveat=t, abstol=1e-6, reltol=1e-6, **kw)
    out, st, rc = O.solve_ensemble(O.seir_ude(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, 21.0], th, t)
    assert (rc == 0).all()
    assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts")
    assert_bitwise(sol.u, out, "forward states")
    r = U.loss_and_gradient(ens, alg(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, **kw)
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, 21.0], th, t, truth, row_mask=mask, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    assert np.linalg.norm(ref["grad_theta"]) > 0
    if N == 1:
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta (single trajectory: no sum over trajectories)")


@pytest.mark.parametrize("N", [1, 37])
def test_seir_lockstep_kernel_partial_blocks_and_single_trajectory(N):
    """the lock-step kernel with one live slot of sixteen (every one of the 4481 gradient entries bitwise) and with 37
    trajectories = two full blocks and one of five slots"""
    test_seir_ude_forward_and_adjoint_match_oracle(U.Vern7, O.VERN7, 16, N=N)


def _seir_ls_setup(N, seed=11):
    u0, t = seir_inputs(N)
    th = models.seir_chain().glorot_uniform(np.random.default_rng(seed))
    th[-65:-1] *= 10.0
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
    return u0, t, th, ens


def test_seir_lockstep_kernel_user_cotangent_and_fixed_initial_dt():
    """the lock-step backward kernel behind ude_vjp_ensemble (a user cotangent instead of data: Zygote's pullback of
    concrete_solve, seir_exposure.jl:138-140) and with `dt = ...` given (no initial-dt evaluations: the slots start in stage 0)"""
    N = 21
    u0, t, th, ens = _seir_ls_setup(N)
    cot = np.random.default_rng(5).normal(size=(N, len(t), 7)) * 1e-6
    kw = {"ensemblealg": U.EnsembleMI355(16)}
    r = U.adjoint_pullback(ens, U.Vern7(), cot, saveat=t, abstol=1e-6, reltol=1e-6, **kw)
    ref = O.vjp_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6), u0, [0.0, 21.0], th, t, cot, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    mask = [0, 1, 1, 1, 0, 0, 0]
    r = U.loss_and_gradient(ens, U.Tsit5(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, dt=0.05, **kw)
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.TSIT5, 1e-6, 1e-6, dt0=0.05), u0, [0.0, 21.0], th, t, truth, row_mask=mask, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")


def test_user_dt_reaches_the_adjoint_solve_in_every_kernel_family(golden):
    """`dt = ...` is a keyword of the solve; _concrete_solve_adjoint hands the keywords on to the adjoint solve [UP?]: both passes
    start from it (the backward one with dt = tdir * |dt|).  LV (lane groups), SEIR one wavefront per trajectory, Fisher-KPP."""
    g, X, t = s1_data(golden)
    th = np.array(g["trained_parameters"])
    u0 = ensemble_u0(X, 24, 5)
    data = np.repeat(X[None], 24, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6, dt=0.01)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6, dt0=0.01), u0, [t[0], t[-1]], th, t, data, nthreads=4)
    check_per_trajectory(r, ref)
    us, ts_, ths, enss = _seir_ls_setup(6)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), us, [0.0, 21.0], [], ts_)
    mask = [0, 1, 1, 1, 0, 0, 0]
    r = U.loss_and_gradient(enss, U.Vern7(), truth, row_mask=mask, saveat=ts_, abstol=1e-6, reltol=1e-6, dt=0.02, ensemblealg=U.EnsembleMI355(64))
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6, dt0=0.02), us, [0.0, 21.0], ths, ts_, truth, row_mask=mask, nthreads=4)
    check_per_trajectory(r, ref)
    thk, u0k, tk, truthk = kpp_case(26, 3, models.kpp_chain(), None)
    ensk = U.EnsembleProblem(U.ODEProblem(models.nn_ode(26), u0k[0], (0.0, 5.0), thk), u0k)
    r = U.loss_and_gradient(ensk, U.Tsit5(), truthk, saveat=tk, dt=0.003)
    ref = O.loss_grad_ensemble(O.kpp_ude(26), O.opts(O.TSIT5, dt0=0.003), u0k, [0.0, 5.0], thk, tk, truthk)
    check_per_trajectory(r, ref)


@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
def test_seir_lockstep_forward_kernel_plain_solves(alg, oalg):
    """the forward lock-step kernel (csrc/ude_seir_ls_fwd.h) without a dense store: a ragged save grid that contains neither end
    point (Vern7 builds its six lazy stages only in steps with a save point strictly inside), a given dt (Tsit5: the FSAL
    evaluation comes first), 41 trajectories on 16-slot blocks; and solves that stop at maxiters"""
    N = 41
    u0, _, th, _ = _seir_ls_setup(N)
    grid = np.array([0.37, 2.0, 2.5, 7.25, 11.0, 19.99])
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
    for kw, okw in (({}, {}), ({"dt": 0.03}, {"dt0": 0.03}), ({"maxiters": 7}, {"maxiters": 7})):
        sol = U.solve(ens, alg(), saveat=grid, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(16), **kw)
        out, st, rc = O.solve_ensemble(O.seir_ude(), O.opts(oalg, 1e-6, 1e-6, **okw), u0, [0.0, 21.0], th, grid)
        assert_bitwise(sol.retcodes, rc, "retcodes %s" % kw)
        assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts incl. lazy stages %s" % kw)
        if "maxiters" in kw:
            assert (rc != 0).all()
        else:
            assert (rc == 0).all()
            assert_bitwise(sol.u, out, "states %s" % kw)


def test_seir_lockstep_forward_kernel_dense_overflow_matches_the_wavefront_kernel():
    """a dense store of 4 steps: every trajectory stops with DenseOverflow after its fourth accepted step, with the same
    counters whichever forward kernel ran"""
    N = 19
    u0, t, th, ens = _seir_ls_setup(N)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    mask = [0, 1, 1, 1, 0, 0, 0]
    res = [U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6,
                               ensemblealg=U.EnsembleMI355(lanes, 4), allow_failures=True) for lanes in (16, 64)]
    assert (res[0].retcode != 0).all()
    assert_bitwise(res[0].retcode, res[1].retcode, "retcodes")
    assert_bitwise(res[0].stats, res[1].stats, "counters")
    assert np.isinf(res[0].loss) and not res[0].grad_theta.any()


def test_seir_lockstep_kernel_backward_failures_are_reported_per_trajectory():
    """maxiters small enough that every backward solve stops early: retcode MaxIters for each trajectory, zero gradient rows,
    loss +Inf -- and the block's slots still hand themselves on to the rest of the ensemble (40 trajectories on 16 slots)"""
    N = 40
    u0, t, th, ens = _seir_ls_setup(N)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    mask = [0, 1, 1, 1, 0, 0, 0]
    r = U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, maxiters=25,
                            ensemblealg=U.EnsembleMI355(16), allow_failures=True)
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6, maxiters=25), u0, [0.0, 21.0], th, t, truth, row_mask=mask, nthreads=4)
    assert_bitwise(r.retcode, ref["retcode"], "retcodes")
    assert (r.retcode != 0).all() and np.isinf(r.loss) and not r.grad_theta.any()
    assert_bitwise(r.stats[:, [0, 1, 2, 4, 5, 6]], ref["stats"][:, [0, 1, 2, 4, 5, 6]], "work counts of the stopped solves")


def kpp_case(nx, N, chain, omodel, seed=4):
    rng = np.random.default_rng(seed)
    th = models.kpp_theta(chain, rng)
    u0 = np.clip(models.rho0(nx)[None, :] * (1 + 0.1 * rng.uniform(-1, 1, (N, 1))) + 0.01 * rng.uniform(0, 1, (N, nx)), 0, None)
    t = np.arange(11) * 0.5
    truth, st, rc = O.solve_ensemble(O.kpp_true(nx), O.opts(O.TSIT5), u0, [0.0, 5.0], [], t)
    assert (rc == 0).all()
    return th, u0, t, truth


def test_kpp_true_matches_oracle():
    """rc_ode (Fisher-KPP-CNN.jl:51-66): Tsit5 at default tolerances, 26 points, periodic."""
    th, u0, t, truth = kpp_case(26, 6, models.kpp_chain(), None)
    ens = U.EnsembleProblem(U.ODEProblem(models.rc_ode(26), u0[0], (0.0, 5.0), []), u0)
    sol = U.solve(ens, U.Tsit5(), saveat=t)
    out, st, rc = O.solve_ensemble(O.kpp_true(26), O.opts(O.TSIT5), u0, [0.0, 5.0], [], t)
    assert_bitwise(sol.stats[:, :4], st[:, :4], "counts")
    assert_bitwise(sol.u, out, "states")


KPP_CASES = [
    ("cnn26", 26, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Tsit5, O.TSIT5, {}),                       # Fisher-KPP-CNN.jl:136 (default tol)
    ("cnn26_vern7", 26, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Vern7, O.VERN7, dict(abstol=1e-6, reltol=1e-6)),
    ("s3_26", 26, models.kpp_s3_chain, lambda nx: O.kpp_ude_s3(0), U.Vern7, O.VERN7, {}),                    # scenario_3.jl:123 (Float64 here)
    ("cnn1024", 1024, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Tsit5, O.TSIT5, {}),                    # BASELINE C4 size
    ("cnn1024_vern7", 1024, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Vern7, O.VERN7, dict(abstol=1e-6, reltol=1e-6)),   # scenario_3.jl:123-125's solver on the C4 grid
    ("cnn300_vern7", 300, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Vern7, O.VERN7, dict(abstol=1e-5, reltol=1e-5)),
    ("cnn300", 300, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Tsit5, O.TSIT5, {}),   # ragged: 2 blocks (256 + 44 points), partial column tile, empty wavefronts
    ("cnn33", 33, models.kpp_chain, lambda nx: O.kpp_ude(nx), U.Tsit5, O.TSIT5, {}),     # smallest grid of the 4-wavefront kernels
]


@pytest.mark.parametrize("name,nx,chain,omk,alg,oalg,kw", KPP_CASES)
def test_kpp_ude_forward_and_adjoint_match_oracle(name, nx, chain, omk, alg, oalg, kw):
    N = 3 if nx > 100 else 5
    th, u0, t, truth = kpp_case(nx, N, chain(), omk)
    if nx > 100:   # dx kept at 0.04: tile the 26-point bump, the UDE stays in the non-stiff regime
        u0 = np.tile(u0[:, :26], (1, 40))[:, :nx]
        truth, _, rc = O.solve_ensemble(O.kpp_true(nx), O.opts(O.TSIT5), u0, [0.0, 5.0], [], t)
    f = models.nn_ode(nx, chain())
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 5.0), th), u0)
    o = O.opts(oalg, kw.get("abstol", 0.0), kw.get("reltol", 0.0))
    sol = U.solve(ens, alg(), saveat=t, **kw)
    out, st, rc = O.solve_ensemble(omk(nx), o, u0, [0.0, 5.0], th, t)
    assert (rc == 0).all()
    assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts")
    assert_bitwise(sol.u, out, "forward states")
    r = U.loss_and_gradient(ens, alg(), truth, saveat=t, **kw)
    ref = O.loss_grad_ensemble(omk(nx), o, u0, [0.0, 5.0], th, t, truth, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    # the state is distributed over lanes: the per-trajectory loss is a lane-parallel sum (order differs from the oracle)
    assert np.abs(r.loss_per_traj - ref["loss_per_traj"]).max() < 1e-13 * ref["loss_per_traj"].max()
    gn = np.linalg.norm(ref["grad_theta"])
    assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn
    assert r.grad_theta[f.stencil_offset + 3] == 0.0          # the unused conv bias never receives a gradient


DISCRETE_CASES = [
    ("s1_tsit5", lambda: models.ude_dynamics(), O.lv_ude_s1, U.Tsit5, O.TSIT5),
    ("s1_vern7", lambda: models.ude_dynamics(), O.lv_ude_s1, U.Vern7, O.VERN7),
    ("hudson_vern7", lambda: models.ude_dynamics(models.hudson_chain(), trainable="both"), O.lv_ude_hudson, U.Vern7, O.VERN7),
    ("s2_tsit5", lambda: models.ude_dynamics(trainable="delta"), O.lv_ude_s2, U.Tsit5, O.TSIT5),
]


@pytest.mark.parametrize("name,mk,omk,alg,oalg", DISCRETE_CASES)
def test_discrete_gradient_matches_oracle(golden, name, mk, omk, alg, oalg):
    """sensealg = ForwardDiffSensitivity() (scenario_1.jl:86): frozen-step reverse sweep, bit-identical per trajectory."""
    g, X, t = s1_data(golden)
    th = theta_for(name.split("_")[0], golden, mk().n_param)
    N = 50
    u0 = ensemble_u0(X, N, 21)
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(mk(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=U.ForwardDiffSensitivity())
    ref = O.loss_grad_ensemble(omk(), O.opts(oalg, 1e-6, 1e-6, sensealg=1), u0, [t[0], t[-1]], th, t, data, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)                      # incl. stats[4] = number of VJPs and dL/du0
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    # consistent with the continuous adjoint to the solver tolerance
    ra = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    assert np.linalg.norm(r.grad_theta - ra.grad_theta) < 1e-4 * np.linalg.norm(ra.grad_theta)


def test_discrete_gradient_seir_and_kpp_match_oracle():
    u0, t = seir_inputs(6)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    th = models.seir_chain().glorot_uniform(np.random.default_rng(11))
    mask = [0, 1, 1, 1, 0, 0, 0]
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
    r = U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=U.ForwardDiffSensitivity())
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=1), u0, [0.0, 21.0], th, t, truth, row_mask=mask, nthreads=4)
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    th, u0, t, truth = kpp_case(26, 4, models.kpp_chain(), None)
    f = models.nn_ode(26)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 5.0), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), truth, saveat=t, sensealg=U.ForwardDiffSensitivity())
    ref = O.loss_grad_ensemble(O.kpp_ude(26), O.opts(O.TSIT5, sensealg=1), u0, [0.0, 5.0], th, t, truth, nthreads=4)
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    # 1024 points: the reverse sweep in the 4-wavefront matrix-core layout (stage derivatives read from the dense store),
    # and the 64-lane layout (16 points per lane, blocked fused parameter sums on the VALU) against the same oracle;
    # the 64-lane reverse sweep does not fit the LDS and says so
    th, u0, t, truth = kpp_case(1024, 3, models.kpp_chain(), None)
    f = models.nn_ode(1024)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 5.0), th), u0)
    r = U.loss_and_gradient(ens, U.Tsit5(), truth, saveat=t, sensealg=U.ForwardDiffSensitivity())
    ref = O.loss_grad_ensemble(O.kpp_ude(1024), O.opts(O.TSIT5, sensealg=1), u0, [0.0, 5.0], th, t, truth, nthreads=3)
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    r64 = U.loss_and_gradient(ens, U.Tsit5(), truth, saveat=t, ensemblealg=U.EnsembleMI355(64))
    ref64 = O.loss_grad_ensemble(O.kpp_ude(1024), O.opts(O.TSIT5), u0, [0.0, 5.0], th, t, truth, nthreads=3)
    check_per_trajectory(r64, ref64)
    with pytest.raises(U.UdeError, match="LDS"):
        U.loss_and_gradient(ens, U.Tsit5(), truth, saveat=t, sensealg=U.ForwardDiffSensitivity(), ensemblealg=U.EnsembleMI355(64))


def test_failed_trajectory_is_reported_not_summed(golden):
    g, X, t = s1_data(golden)
    th = np.array(g["initial_parameters"])
    u0 = ensemble_u0(X, 8, 1)
    u0[3] = [np.nan, 1.0]
    data = np.repeat(X[None], 8, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    with pytest.raises(U.UdeError, match="trajectory 3"):        # loud by default
        U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
    assert r.retcode[3] == 3 and (np.delete(r.retcode, 3) == 0).all() and r.loss == np.inf
    keep = [0, 1, 2, 4, 5, 6, 7]
    ens2 = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0[keep])
    r2 = U.loss_and_gradient(ens2, U.Tsit5(), data[keep], saveat=t, abstol=1e-6, reltol=1e-6)
    assert np.isfinite(r.grad_theta).all()
    assert np.linalg.norm(r.grad_theta - r2.grad_theta) < 1e-12 * np.linalg.norm(r2.grad_theta)


def test_unsupported_descriptor_is_an_error():
    """a chain no compiled instance covers runs on the runtime-shape fallback (tests/test_gpu_generic.py); one outside the
    fallback too (a layer wider than a wavefront) is refused loudly"""
    prob = U.ODEProblem(models.ude_dynamics(models.Chain(models.Dense(2, 7, "tanh"), models.Dense(7, 2))), [1.0, 1.0], (0.0, 1.0), np.zeros(37))
    assert U.solve(prob, U.Tsit5(), saveat=0.5).retcode == "Success"
    wide = models.Chain(models.Dense(2, 65, "tanh"), models.Dense(65, 2))
    with pytest.raises(U.sciml.UdeError, match="no kernel for model"):
        U.solve(U.ODEProblem(models.ude_dynamics(wide), [1.0, 1.0], (0.0, 1.0), np.zeros(wide.n_param)), U.Tsit5(), saveat=0.5)


def test_full_size_seir_and_kpp_properties():
    """BASELINE configs[2] per-GPU share (6250 SEIR trajectories) and configs[3] (256 PDEs x 1024 points): the
    size-independent properties -- determinism, additivity over shards (what the multi-GPU all-reduce relies on), the
    work-count identities -- plus the oracle on a subsample."""
    # ---- SEIR ----
    N = 6250
    u0, t = seir_inputs(N)
    th = models.seir_chain().glorot_uniform(np.random.default_rng(11))
    mask = [0, 1, 1, 1, 0, 0, 0]
    f = models.dudt_()
    truth = np.asarray(U.solve(U.EnsembleProblem(U.ODEProblem(models.corona(), u0[0], (0.0, 21.0), []), u0), U.Vern7(),
                               saveat=t, abstol=1e-12, reltol=1e-12).u)

    def run(sl):
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 21.0), th), u0[sl])
        return U.loss_and_gradient(ens, U.Vern7(), truth[sl], row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6)

    full, again = run(slice(0, N)), run(slice(0, N))
    assert (full.retcode == 0).all()
    assert np.array_equal(full.grad_theta, again.grad_theta) and full.loss == again.loss
    a, b = run(slice(0, 4000)), run(slice(4000, N))
    gsum = a.grad_theta + b.grad_theta
    assert np.linalg.norm(full.grad_theta - gsum) < 1e-12 * np.linalg.norm(gsum)
    assert np.array_equal(full.stats[:4000], a.stats) and np.array_equal(full.stats[4000:], b.stats)
    assert np.array_equal(full.stats[:, 0], 2 + 10 * (full.stats[:, 1] + full.stats[:, 2]))      # nf identity (Vern7: 10 stages, no FSAL)
    assert np.array_equal(full.stats[:, 4], 2 + 10 * (full.stats[:, 5] + full.stats[:, 6]))      # ... and backward
    idx = np.arange(0, N, 625)
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6), u0[idx], [0.0, 21.0], th, t, truth[idx], row_mask=mask, nthreads=4)
    assert_bitwise(full.stats[idx][:, :7], ref["stats"][:, :7], "SEIR stats of the subsample")
    assert_bitwise(full.grad_u0[idx], ref["grad_u0"], "SEIR dL/du0")
    assert_bitwise(full.u[idx], ref["u"], "SEIR saved states")
    # ---- Fisher-KPP, 1024 points ----
    B = 256
    thk, uk, tk, truthk = kpp_case(1024, B, models.kpp_chain(), None)
    fk = models.nn_ode(1024, models.kpp_chain())

    def runk(sl):
        ens = U.EnsembleProblem(U.ODEProblem(fk, uk[0], (0.0, 5.0), thk), uk[sl])
        return U.loss_and_gradient(ens, U.Tsit5(), truthk[sl], saveat=tk)

    fullk, againk = runk(slice(0, B)), runk(slice(0, B))
    assert (fullk.retcode == 0).all()
    assert np.array_equal(fullk.grad_theta, againk.grad_theta) and fullk.loss == againk.loss
    ak, bk = runk(slice(0, 100)), runk(slice(100, B))
    gk = ak.grad_theta + bk.grad_theta
    assert np.linalg.norm(fullk.grad_theta - gk) < 1e-12 * np.linalg.norm(gk)
    assert np.array_equal(fullk.stats[:, 0], 3 + 6 * (fullk.stats[:, 1] + fullk.stats[:, 2]))    # nf identity (Tsit5, FSAL)
    idk = np.array([0, 131, 255])
    refk = O.loss_grad_ensemble(O.kpp_ude(1024), O.opts(O.TSIT5), uk[idk], [0.0, 5.0], thk, tk, truthk[idk], nthreads=3)
    assert_bitwise(fullk.stats[idk][:, :7], refk["stats"][:, :7], "KPP stats of the subsample")
    assert_bitwise(fullk.u[idk], refk["u"], "KPP saved states")
    assert_bitwise(fullk.grad_u0[idk], refk["grad_u0"], "KPP dL/du0")


def test_mfma_f64_is_the_ascending_fused_chain():
    """The Fisher-KPP kernels put ARITH-SPEC fma chains on the matrix cores.  That is only bit-exact if
    v_mfma_f64_16x16x4 computes d = fma(a_k, b_k, d) for k = 0..3 in ascending order starting from C: checked here
    against exact rational arithmetic (Fraction -> float is correctly rounded) on random operands."""
    from fractions import Fraction
    rng = np.random.default_rng(5)
    nb = 40
    x = rng.normal(size=(nb, 64)) * np.exp2(rng.integers(-6, 6, size=(nb, 64)))
    y = rng.normal(size=(nb, 64)) * np.exp2(rng.integers(-6, 6, size=(nb, 64)))
    out = U.Engine.get(0).math(10, x.ravel(), y.ravel()).reshape(nb, 64)

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    for b in range(nb):
        A = lambda i, k: float(x[b, i + 16 * k])      # lane l = i + 16 k supplies A[i][k]
        B = lambda k, j: float(y[b, j + 16 * k])      # lane l = j + 16 k supplies B[k][j]
        for lane in range(64):
            i, j = lane // 16, lane % 16                # register 0 of lane l holds D[l/16][l%16]
            d = 0.0
            for k in range(4):
                d = fma(A(i, k), B(k, j), d)
            assert d == out[b, lane], (b, lane)


def test_tanh32_wide_lane_group_variant():
    """BASELINE's '2-layer tanh' (2-32-2): every compiled lane-group width (8, 16, 32 lanes per trajectory = 4, 2, 1 hidden neurons per
    lane; 0 = the library's default) must return the same bits -- the output layer and the input cotangent are 32-term tree sums
    (ARITH-SPEC wide-dot rule) formed by the group's butterfly plus register pairs, a different split per width."""
    rng = np.random.default_rng(3)
    N = 24
    t = np.arange(31) * 0.1
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    th = theta_tanh32 = models.tanh32_chain().glorot_uniform(rng) * 0.5
    data, _, rc = O.solve_ensemble(O.lv_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 3.0], [1.3, 0.9, 0.8, 1.8], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(models.tanh32_chain()), u0[0], (0.0, 3.0), th), u0)
    ref = O.loss_grad_ensemble(O.lv_ude_tanh32(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [0.0, 3.0], th, t, data, nthreads=4)
    for lanes in (0, 8, 16, 32):
        r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6,
                                **({"ensemblealg": U.EnsembleMI355(lanes)} if lanes else {}))
        check_per_trajectory(r, ref)


def test_fisher_kpp_small_variant_matches_oracle():
    """Fisher-KPP-CNN-Small.jl:89-124: the 15-parameter variant (1-3-1 tanh), the only one the reference publishes timings for"""
    rng = np.random.default_rng(8)
    nx = 26
    chain = models.kpp_small_chain(3)
    th = models.kpp_theta(chain, rng)
    assert th.size == 15
    u0 = np.stack([models.rho0(nx) * s for s in (1.0, 0.8, 0.6)])
    t = np.arange(11) * 0.5
    truth = U.solve(U.EnsembleProblem(U.ODEProblem(models.rc_ode(nx), u0[0], (0.0, 5.0), []), u0), U.Tsit5(), saveat=t).u
    f = models.nn_ode(nx, chain)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 5.0), th), u0)
    mo = O.kpp_ude(nx, (1, 3, 1), ("tanh", "identity"))
    for alg, oalg in ((U.Tsit5(), O.TSIT5), (U.Vern7(), O.VERN7)):
        r = U.loss_and_gradient(ens, alg, truth, saveat=t)
        ref = O.loss_grad_ensemble(mo, O.opts(oalg), u0, [0.0, 5.0], th, t, truth, nthreads=3)
        assert_bitwise(r.stats, ref["stats"], "stats")
        assert_bitwise(r.u, ref["u"], "u")
        assert_bitwise(r.grad_u0, ref["grad_u0"], "grad_u0")
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
        rd = U.loss_and_gradient(ens, alg, truth, saveat=t, sensealg=U.ForwardDiffSensitivity())
        refd = O.loss_grad_ensemble(mo, O.opts(oalg, sensealg=1), u0, [0.0, 5.0], th, t, truth, nthreads=3)
        assert np.linalg.norm(rd.grad_theta - refd["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(refd["grad_theta"])
