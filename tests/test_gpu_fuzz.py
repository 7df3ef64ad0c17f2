"""Randomised differential test: the HIP path against the oracle over random corners of the boundary -- algorithm,
tolerances, time span, ragged save grids (with and without the end points), ensemble size, lanes per trajectory, network
variant, sensitivity algorithm, row mask, per-trajectory spans.  Every draw must agree per trajectory bit for bit (step
counts forward and backward, saved states, dL/du0, per-trajectory loss) and to summation order in the ensemble sums."""
import os

import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_parity import REL_GRAD_SUM, assert_bitwise, check_per_trajectory

pytestmark = pytest.mark.gpu

# UDE_FUZZ_SCALE=k multiplies the number of seeds of every randomised test below (soak runs of a final binary: profiles/r06_fuzz_soak.log)
SCALE = int(os.environ.get("UDE_FUZZ_SCALE", "1"))

NETS = {
    "s1": (lambda: models.ude_dynamics(), O.lv_ude_s1, lambda rng, g: np.array(g["initial_parameters"]) * (1 + 0.1 * rng.standard_normal(87)), (0, 1, 4, 5, 8)),
    "s2": (lambda: models.ude_dynamics(trainable="delta"), O.lv_ude_s2,
           lambda rng, g: np.concatenate([[1.8 * (1 + 0.1 * rng.standard_normal())], np.array(g["initial_parameters"])]), (0,)),
    "hudson": (lambda: models.ude_dynamics(models.hudson_chain(), trainable="both"), O.lv_ude_hudson,
               lambda rng, g: np.concatenate([[1.3, 1.8], 0.3 * rng.standard_normal(87)]), (0, 8)),
    "tanh32": (lambda: models.ude_dynamics(models.tanh32_chain()), O.lv_ude_tanh32,
               lambda rng, g: 0.1 * models.tanh32_chain().glorot_uniform(rng), (0, 8, 16, 32)),
}


@pytest.mark.parametrize("seed", range(40 * SCALE))
def test_random_corner_matches_oracle(golden, seed):
    g = golden("Scenario_1_recovery_0.005")
    rng = np.random.default_rng(1000 + seed)
    name = list(NETS)[seed % 4]
    mk, omk, mkth, lanes_opts = NETS[name]
    f, om, th = mk(), omk(), mkth(rng, g)
    alg, oalg = ((U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7))[int(rng.integers(2))]
    tol = float(10.0 ** rng.uniform(-9, -4))
    t0 = float(rng.uniform(-1.0, 1.0))
    tf = t0 + float(rng.uniform(0.3, 3.0))
    ns = int(rng.integers(1, 24))
    inner = np.sort(rng.uniform(t0, tf, ns))
    inner = inner[np.concatenate([[True], np.diff(inner) > 1e-6])]
    grid = inner.copy()
    if rng.random() < 0.5:
        grid[0] = t0                                     # a save at the initial time
    if rng.random() < 0.5:
        grid[-1] = tf                                    # ... and / or exactly at the end
    grid = np.unique(grid)
    N = int(rng.integers(1, 40))
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.3 * rng.uniform(-1, 1, (N, 2)))
    data = rng.uniform(0.0, 5.0, (N, len(grid), 2))
    mask = [(1, 1), (1, 0), (0, 1)][int(rng.integers(3))]
    lanes = int(rng.choice(lanes_opts))
    sense, osense = [(None, 0), (U.ForwardDiffSensitivity(), 1), (U.FastInterpolatingAdjoint(), 2)][int(rng.integers(3))]
    kw = dict(saveat=grid, abstol=tol, reltol=tol)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t0, tf), th), u0)
    ealg = U.EnsembleMI355(lanes) if lanes else None
    try:
        r = U.loss_and_gradient(ens, alg(), data, row_mask=mask, sensealg=sense, ensemblealg=ealg, **kw)
    except U.UdeError as e:
        if "no kernel instance" in str(e):               # that lane count is not compiled for this (network, algorithm)
            r = U.loss_and_gradient(ens, alg(), data, row_mask=mask, sensealg=sense, **kw)
        else:
            raise
    ref = O.loss_grad_ensemble(om, O.opts(oalg, tol, tol, sensealg=osense), u0, [t0, tf], th, grid, data, row_mask=list(mask), nthreads=8)
    what = "seed %d: %s %s tol %.1e N %d ns %d lanes %d sense %d" % (seed, name, alg.__name__, tol, N, len(grid), lanes, osense)
    assert (r.retcode == 0).all(), what
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], what)
    gn = np.linalg.norm(ref["grad_theta"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) <= REL_GRAD_SUM * gn, what
    assert abs(r.loss - ref["loss"]) <= 1e-12 * abs(ref["loss"]), what


KPP = {
    "kpp": (models.kpp_chain, lambda nx, dt: O.kpp_ude(nx), "float64"),
    "small": (lambda: models.kpp_small_chain(3), lambda nx, dt: O.kpp_ude(nx, (1, 3, 1), ("tanh", "identity")), "float64"),
    "s3": (models.kpp_s3_chain, lambda nx, dt: O.kpp_ude(nx, (1, 5, 5, 5, 1), ("rbf", "rbf", "rbf", "identity"), dtype=dt), "float64"),
    "s3f32": (models.kpp_s3_chain, lambda nx, dt: O.kpp_ude(nx, (1, 5, 5, 5, 1), ("rbf", "rbf", "rbf", "identity"), dtype=dt), "float32"),
}


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_random_fisher_kpp_grid_matches_oracle(seed):
    """ragged grids (3 ... 32 points, periodic stencil), random stencil / D0 / tolerances, Float64 and the Float32 instance"""
    rng = np.random.default_rng(2000 + seed)
    name = list(KPP)[seed % 4]
    mkchain, omk, dtype = KPP[name]
    rt = np.float32 if dtype == "float32" else np.float64
    nx = 26 if dtype == "float32" else int(rng.integers(3, 33))   # (the Float32 instance is compiled for scenario_3's descriptor)
    chain = mkchain()
    th = models.kpp_theta(chain, rng).astype(rt)
    f = models.nn_ode(nx, chain, dtype=dtype)
    th[f.stencil_offset:f.stencil_offset + 3] = (np.array([1.0, -2.0, 1.0]) + 0.1 * rng.standard_normal(3)).astype(rt)
    th[f.d0_offset] = rt(rng.uniform(1.0, 7.0))
    N = int(rng.integers(1, 9))
    u0 = np.clip(models.rho0(nx)[None, :] * (1 + 0.2 * rng.uniform(-1, 1, (N, 1))) + 0.02 * rng.uniform(0, 1, (N, nx)), 0, None).astype(rt)
    tf = float(rng.uniform(0.5, 4.0))
    grid = np.unique(np.concatenate([[0.0], np.sort(rng.uniform(0.0, tf, int(rng.integers(1, 9)))), [tf]])).astype(rt)
    tf = float(grid[-1])
    data = rng.uniform(0.0, 1.0, (N, len(grid), nx)).astype(rt)
    alg, oalg = ((U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7))[int(rng.integers(2))]
    tol = float(10.0 ** rng.uniform(-7 if dtype == "float64" else -5, -3))
    sense, osense = [(None, 0), (U.ForwardDiffSensitivity(), 1)][int(rng.integers(2))]
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    r = U.loss_and_gradient(ens, alg(), data, saveat=grid, abstol=tol, reltol=tol, sensealg=sense)
    ref = O.loss_grad_ensemble(omk(nx, 1 if dtype == "float32" else 0), O.opts(oalg, tol, tol, sensealg=osense), u0, [0.0, tf], th, grid, data,
                               dtype=rt, nthreads=8)
    what = "seed %d: %s nx %d %s tol %.1e N %d ns %d sense %d" % (seed, name, nx, alg.__name__, tol, N, len(grid), osense)
    assert (r.retcode == 0).all(), what
    check_per_trajectory(r, ref)
    rel = 2e-5 if dtype == "float32" else REL_GRAD_SUM
    assert np.linalg.norm(r.grad_theta.astype(float) - ref["grad_theta"].astype(float)) <= rel * np.linalg.norm(ref["grad_theta"].astype(float)), what


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_random_seir_and_neural_ode_match_oracle(seed):
    """64-wide tanh networks (SEIR exposure UDE 3-64-64-1, neural ODE 7-64-64-64-7): random population scale, horizon, save
    grid, row mask, tolerance, algorithm and sensitivity mode"""
    rng = np.random.default_rng(3000 + seed)
    node = seed % 2 == 1
    f, om = (models.dudt_node(), O.seir_node()) if node else (models.dudt_(), O.seir_ude())
    chain = models.seir_node_chain() if node else models.seir_chain()
    th = chain.glorot_uniform(rng) * float(rng.uniform(0.5, 2.0))
    S0 = float(10.0 ** rng.uniform(2, 7))
    N = int(rng.integers(1, 6))
    u0 = np.zeros((N, 7))
    u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
    u0[:, 1] = rng.uniform(0.0, 5.0, N)
    u0[:, 2] = rng.uniform(0.0, 2.0, N)
    u0[:, 4] = S0
    tf = float(rng.uniform(2.0, 12.0))
    grid = np.unique(np.concatenate([[0.0], np.sort(rng.uniform(0.0, tf, int(rng.integers(1, 12)))), [tf]]))
    data = u0[:, None, :] * (1 + 0.05 * rng.standard_normal((N, len(grid), 7)))
    mask = [int(b) for b in rng.integers(0, 2, 7)]
    mask[1 + int(rng.integers(3))] = 1
    alg, oalg = ((U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7))[int(rng.integers(2))]
    tol = float(10.0 ** rng.uniform(-8, -5))
    sense, osense = [(None, 0), (U.ForwardDiffSensitivity(), 1), (U.FastInterpolatingAdjoint(), 2)][int(rng.integers(3))]
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    r = U.loss_and_gradient(ens, alg(), data, row_mask=mask, saveat=grid, abstol=tol, reltol=tol, sensealg=sense)
    ref = O.loss_grad_ensemble(om, O.opts(oalg, tol, tol, sensealg=osense), u0, [0.0, tf], th, grid, data, row_mask=mask, nthreads=8)
    what = "seed %d: %s S0 %.1e %s tol %.1e N %d ns %d sense %d" % (seed, "node" if node else "seir", S0, alg.__name__, tol, N, len(grid), osense)
    assert (r.retcode == 0).all(), what
    check_per_trajectory(r, ref)
    gn = np.linalg.norm(ref["grad_theta"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) <= REL_GRAD_SUM * max(gn, 1e-300), what


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_random_deep_bsde_step_matches_oracle(seed):
    """highdim_pde/lambaem.jl: random Philox seed / training iteration / ensemble size (below, at and above one 32-slot block,
    so the slot queue is exercised) / tolerances / controller settings / lambda / horizon / x0, adaptive LambaEM and fixed EM"""
    import _sde_oracle as S
    from universal_differential_equations_amd import pde
    rng = np.random.default_rng(4000 + seed)
    alg = pde.NNPDENS(100, 110, opt=pde.ADAM(0.03))
    th = (alg.init_params(rng) + 0.03 * rng.standard_normal(sum(alg.num_params()))).astype(np.float32)
    lam = float(rng.uniform(0.5, 2.0))
    t1 = float(rng.uniform(0.3, 1.0))
    x0 = (0.2 * rng.standard_normal(100)).astype(np.float32) if rng.random() < 0.5 else np.zeros(100, dtype=np.float32)
    prob = pde.TerminalPDEProblem(pde.hjb(lam), x0, (0.0, t1))
    M = int(rng.choice([3, 31, 32, 33, 70]))
    it = int(rng.integers(0, 500))
    pseed = int(rng.integers(0, 2 ** 62))
    if seed % 4 == 3:
        dt = float(rng.uniform(0.01, 0.05))
        r = pde.loss_and_gradient(prob, alg, pde.EM(), th, M, it=it, dt=dt, seed=pseed)
        ref = S.loss_grad(S.desc(lam=lam, tspan=(0.0, t1), adaptive=0, dt=dt, seed=pseed), M, prob.x0, th, it=it, nthreads=8)
    else:
        tol = float(rng.uniform(0.05, 0.3))
        kw = dict(abstol=tol, reltol=tol, seed=pseed, qmax=float(rng.choice([1.125, 2.0, 10.0])), gamma=float(rng.uniform(0.8, 0.95)))
        r = pde.loss_and_gradient(prob, alg, pde.LambaEM(), th, M, it=it, **kw)
        ref = S.loss_grad(S.desc(lam=lam, tspan=(0.0, t1), **kw), M, prob.x0, th, it=it, nthreads=8)
    assert np.array_equal(r.retcode, ref["retcode"]) and (r.retcode == 0).all()
    assert np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.XT, ref["XT"]) and np.array_equal(r.uT, ref["uT"])
    assert np.array_equal(r.loss_traj, ref["loss_traj"]) and r.u0 == ref["u0"]
    assert np.linalg.norm(r.grad - ref["grad"]) < 1e-5 * np.linalg.norm(ref["grad"])


@pytest.mark.parametrize("nx", [65, 257, 777, 1000])
def test_ragged_large_fisher_kpp_grids_match_oracle(nx):
    """grids that end inside a 16-point matrix-core tile / a wavefront's 256-point block of the 1024-point kernels"""
    rng = np.random.default_rng(nx)
    th = models.kpp_theta(models.kpp_chain(), rng)
    f = models.nn_ode(nx)
    th[f.stencil_offset:f.stencil_offset + 3] = [1.05, -2.1, 1.0]
    x = np.arange(nx) / nx
    u0 = (0.5 * (np.tanh((x - 0.3) / 0.05) - np.tanh((x - 0.7) / 0.05)))[None, :] * np.array([[1.0], [0.9]])
    t = np.array([0.0, 0.4, 1.0])
    data = rng.uniform(0, 1, (2, 3, nx))
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 1.0), th), u0)
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
        r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, sensealg=sense)
        ref = O.loss_grad_ensemble(O.kpp_ude(nx), O.opts(O.TSIT5, sensealg=osense), u0, [0.0, 1.0], th, t, data, nthreads=2)
        assert (r.retcode == 0).all()
        check_per_trajectory(r, ref)
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) <= REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])


@pytest.mark.parametrize("seed", range(18 * SCALE))
def test_random_solver_keywords_match_oracle(golden, seed):
    """the solve keywords the scripts can pass besides the tolerances -- dt, dtmax, qmin / qmax / gamma, beta1 / beta2, qoldinit --
    over every kernel family: LV lane groups, the SEIR kernels (one wavefront per trajectory, and the lock-step matrix-core
    backward kernel), the runtime-shape fallback, Fisher-KPP with the dense store and with the checkpointed adjoint; loss + gradient
    or a user cotangent.  Both solves of a gradient receive the keywords (the backward one with dt = tdir |dt|)."""
    rng = np.random.default_rng(5000 + seed)
    fam = ["lv", "seir64", "seirls", "generic", "kpp", "kppck"][seed % 6]
    okw, kw = {}, {}
    if rng.random() < 0.6:
        kw["dt"] = okw["dt0"] = float(10.0 ** rng.uniform(-3, -1))
    if rng.random() < 0.5:
        kw["dtmax"] = okw["dtmax"] = float(rng.uniform(0.05, 0.5))
    if rng.random() < 0.5:
        kw["qmax"] = okw["qmax"] = float(rng.choice([2.0, 5.0, 20.0]))
        kw["qmin"] = okw["qmin"] = float(rng.choice([0.1, 0.2, 0.5]))
    if rng.random() < 0.5:
        kw["gamma"] = okw["gamma"] = float(rng.uniform(0.7, 0.95))
    if rng.random() < 0.4:
        kw["beta1"] = okw["beta1"] = float(rng.uniform(0.08, 0.2))
        kw["beta2"] = okw["beta2"] = float(rng.uniform(0.0, 0.08)) or 0.04
    if rng.random() < 0.3:
        kw["qoldinit"] = okw["qoldinit"] = float(10.0 ** rng.uniform(-5, -2))
    alg, oalg = ((U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7))[int(rng.integers(2))]
    tol = float(10.0 ** rng.uniform(-8, -5))
    ealg, sense, osense, mask = None, None, 0, None
    if fam == "lv":
        g = golden("Scenario_1_recovery_0.005")
        f, om = models.ude_dynamics(), O.lv_ude_s1()
        th = np.array(g["trained_parameters"]) * (1 + 0.05 * rng.standard_normal(87))
        N = int(rng.integers(1, 30))
        u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.3 * rng.uniform(-1, 1, (N, 2)))
        tf = float(rng.uniform(0.5, 3.0))
    elif fam in ("seir64", "seirls", "generic"):
        if fam == "generic":
            dims, acts = [3, 16, 33, 1], ["tanh", "rbf", "identity"]
            chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(3)])
            f, om = models.dudt_(chain), O.make_model(O.KIND_SEIR_UDE, 7, dims, acts, consts=O.SEIR_P)
            th = chain.glorot_uniform(rng)
        else:
            f, om = models.dudt_(), O.seir_ude()
            th = models.seir_chain().glorot_uniform(rng) * float(rng.uniform(0.5, 2.0))
            ealg = U.EnsembleMI355(64 if fam == "seir64" else 16)
        S0 = float(10.0 ** rng.uniform(2, 7))
        N = int(rng.integers(1, 20))
        u0 = np.zeros((N, 7))
        u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
        u0[:, 1] = rng.uniform(0.0, 5.0, N)
        u0[:, 4] = S0
        tf = float(rng.uniform(2.0, 8.0))
        mask = [0, 1, 1, 1, 0, 0, 0]
    else:
        nx = 26
        chain = models.kpp_chain()
        f, om = models.nn_ode(nx, chain), O.kpp_ude(nx)
        th = models.kpp_theta(chain, rng)
        N = int(rng.integers(1, 5))
        u0 = np.clip(models.rho0(nx)[None, :] * (1 + 0.2 * rng.uniform(-1, 1, (N, 1))), 0, None)
        tf = float(rng.uniform(0.5, 3.0))
        alg, oalg = U.Tsit5, O.TSIT5
        tol = float(10.0 ** rng.uniform(-7, -3))
        if fam == "kppck":
            sense = U.InterpolatingAdjoint(checkpointing=True)     # (the oracle has one mode: the results are the dense store's)
    n = u0.shape[1]
    grid = np.unique(np.concatenate([[0.0], np.sort(rng.uniform(0.0, tf, int(rng.integers(1, 8)))), [tf]]))
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    oo = O.opts(oalg, tol, tol, sensealg=osense, **okw)
    what = "seed %d: %s %s tol %.1e N %d ns %d kw %s" % (seed, fam, alg.__name__, tol, N, len(grid), kw)
    if rng.random() < 0.35:
        cot = rng.standard_normal((N, len(grid), n)) * (1e-6 if fam.startswith("seir") or fam == "generic" else 1.0)
        r = U.adjoint_pullback(ens, alg(), cot, saveat=grid, abstol=tol, reltol=tol, sensealg=sense, ensemblealg=ealg, **kw)
        ref = O.vjp_ensemble(om, oo, u0, [0.0, tf], th, grid, cot, nthreads=8)
    else:
        data = u0[:, None, :] * (1 + 0.05 * rng.standard_normal((N, len(grid), n)))
        r = U.loss_and_gradient(ens, alg(), data, row_mask=mask, saveat=grid, abstol=tol, reltol=tol, sensealg=sense, ensemblealg=ealg, **kw)
        ref = O.loss_grad_ensemble(om, oo, u0, [0.0, tf], th, grid, data, row_mask=mask, nthreads=8)
        if not fam.startswith("kpp"):   # (a distributed state sums its loss over lanes: association differs from the oracle's)
            assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], what)
    assert (r.retcode == 0).all() and (ref["retcode"] == 0).all(), what
    check_per_trajectory(r, ref)
    gn = np.linalg.norm(ref["grad_theta"])
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) <= REL_GRAD_SUM * max(gn, 1e-300), what
