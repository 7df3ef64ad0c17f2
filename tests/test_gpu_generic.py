"""The runtime-shape fallback kernel (csrc/ude_model_generic.h): any ude_model_desc of the replicated-state kinds with <= 8
Dense layers of width <= 64 and activations identity / tanh / rbf / relu -- the reference accepts any chain
(`U = Lux.Chain(...)` /root/reference/LotkaVolterra/scenario_1.jl:62-64 is a script variable; `FastChain` of
SEIR_exposure/seir_exposure.jl:53,114) -- against the oracle's generic dense chain: per trajectory bit-identical
(forward and backward step counts, states, dL/du0, per-trajectory loss; for a single trajectory every gradient entry).
Plus Fisher-KPP-CNN-Small.jl:88 with n_weights in {1, 2, 3} (compiled instances of the pointwise-network kernel)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import _lib, models
from test_gpu_parity import REL_GRAD_SUM, assert_bitwise, check_per_trajectory

pytestmark = pytest.mark.gpu
WIDTHS = [1, 2, 3, 5, 8, 16, 17, 31, 32, 48, 63, 64]
ACTS = ["tanh", "rbf", "relu", "identity"]


def random_chain(rng, n_in, n_out, max_hidden=7, widths=WIDTHS):
    nh = int(rng.integers(1, max_hidden + 1))
    dims = [n_in] + [int(rng.choice(widths)) for _ in range(nh)] + [n_out]
    acts = [str(rng.choice(ACTS[:3] if rng.random() < 0.8 else ACTS)) for _ in range(nh)] + ["identity"]
    return dims, acts


def chain_of(dims, acts):
    return models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(dims) - 1)])


def theta_for(chain, rng, scale):
    th = chain.glorot_uniform(rng)
    th = scale * th + 0.02 * rng.standard_normal(th.size)      # non-zero biases: every parameter gets a cotangent
    return th


def supported(f, alg=0, sense=0):
    eng = U.Engine.get(0)
    eng.set_launch()   # library defaults (the engine is shared: a previous call's lanes_per_traj = 64 would report the runtime-shape kernel)
    o = _lib.SolveOpts()
    o.alg, o.sensealg = alg, sense
    return eng.L.ude_model_supported(eng.h, C.byref(f), C.byref(o), 1)


def test_model_supported_reports_fast_instance_or_generic_kernel():
    assert supported(models.ude_dynamics()) == 0                                          # scenario_1's shape: compiled instance
    assert supported(models.ude_dynamics(chain_of([2, 6, 5, 2], ["rbf", "tanh", "identity"]))) == 1   # hidden width 6: the fallback
    assert supported(models.dudt_(chain_of([3, 16, 1], ["tanh", "identity"]))) == 1
    assert supported(models.dudt_node(chain_of([7, 32, 32, 7], ["relu", "tanh", "identity"]))) == 1
    assert supported(models.ude_dynamics(chain_of([2, 65, 2], ["tanh", "identity"]))) == -2   # wider than a wavefront: UDE_ERR_UNSUPPORTED
    assert supported(models.nn_ode(26, chain_of([1, 7, 9, 1], ["tanh", "rbf", "identity"]))) == 1   # round 4: any pointwise reaction chain
    assert supported(models.nn_ode(26, chain_of([1, 32, 32, 32, 1], ["tanh", "tanh", "tanh", "identity"]))) == -2   # > 768 parameters
    assert supported(models.ude_dynamics(chain_of([2, 6, 2], ["tanh", "identity"])), sense=1) == 1   # round 4: the discrete sweep too
    assert supported(models.ude_dynamics(chain_of([2, 6, 2], ["tanh", "identity"]), dtype="float32")) == 1   # ... and Float32 LV-kind problems
    assert supported(models.ude_dynamics(chain_of([2, 6, 2], ["tanh", "identity"])), sense=3) == 1   # ... and the checkpointed adjoint


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_lv_kind_random_shapes(golden, seed):
    rng = np.random.default_rng(100 + seed)
    dims, acts = random_chain(rng, 2, 2)
    chain = chain_of(dims, acts)
    trainable = [None, "delta", "both"][seed % 3]
    f = models.ude_dynamics(chain, trainable=trainable)
    nn_off = {None: 0, "delta": 1, "both": 2}[trainable]
    om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, nn_offset=nn_off,
                      lin_idx={None: (-1, -1), "delta": (-1, 0), "both": (0, 1)}[trainable],
                      lin_sign={None: (1.0, 1.0), "delta": (1.0, -1.0), "both": (1.0, -1.0)}[trainable],
                      lin_const={None: (1.3, -1.8), "delta": (1.3, 0.0), "both": (0.0, 0.0)}[trainable])
    lead = {None: [], "delta": [1.8], "both": [1.3, 1.8]}[trainable]       # scenario_2: theta = [delta; ude]; hudson_bay: [p1; p2; ude]
    th = np.concatenate([lead, theta_for(chain, rng, 0.3)]).astype(np.float64)
    assert supported(f) == 1 and f.n_param == th.size == om.n_param
    g = golden("Scenario_1_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    N = 1 if seed % 2 == 0 else 5
    u0 = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    data = np.repeat(X[None], N, axis=0)
    alg, oalg = (U.Tsit5, O.TSIT5) if seed % 4 < 2 else (U.Vern7, O.VERN7)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), u0)
    sol = U.solve(ens, alg(), saveat=t, abstol=1e-6, reltol=1e-6)
    out, st, rc = O.solve_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t)
    assert (rc == 0).all()
    assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts %s %s" % (dims, acts))
    assert_bitwise(sol.u, out, "forward states")
    r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")
    if N == 1:
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta %s %s" % (dims, acts))
    else:
        gn = np.linalg.norm(ref["grad_theta"])
        assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn


LV_RT_CASES = [([2, 8, 8, 8, 2], ["tanh", "tanh", "tanh", "identity"], None), ([2, 6, 5, 7, 2], ["rbf", "tanh", "relu", "identity"], "both"),
               ([2, 5, 5, 5, 2], ["tanh", "tanh", "tanh", "identity"], "delta"), ([2, 1, 1, 1, 2], ["rbf", "rbf", "rbf", "identity"], None),
               ([2, 8, 8, 2], ["tanh", "rbf", "identity"], "both"), ([2, 3, 8, 2], ["relu", "tanh", "identity"], None),
               ([2, 7, 2, 2], ["identity", "tanh", "identity"], "delta"), ([2, 8, 1, 8, 2], ["tanh", "identity", "rbf", "identity"], None),
               ([2, 4, 5, 2], ["tanh", "relu", "identity"], "both"), ([2, 5, 3, 4, 2], ["relu", "rbf", "tanh", "identity"], None),
               # widths 9 .. 16: sixteen lanes per trajectory, the weights read from the block's LDS copy of theta at every use
               ([2, 16, 16, 16, 2], ["tanh", "tanh", "tanh", "identity"], None), ([2, 9, 12, 2], ["rbf", "tanh", "identity"], "both"),
               ([2, 3, 16, 7, 2], ["tanh", "relu", "rbf", "identity"], "delta"), ([2, 16, 1, 2], ["tanh", "tanh", "identity"], None),
               # ONE hidden layer (BASELINE's "2-layer MLP" with an edited width)
               ([2, 8, 2], ["tanh", "identity"], None), ([2, 3, 2], ["rbf", "identity"], "both"), ([2, 16, 2], ["tanh", "identity"], "delta"),
               ([2, 11, 2], ["relu", "identity"], None)]


@pytest.mark.parametrize("case", range(len(LV_RT_CASES)))
def test_lv_kind_edited_network_on_the_lane_group_kernels(golden, case):
    """round 5: the LV scripts' chain with EDITED widths / activations (`U = Lux.Chain(Dense(2,5,rbf), ...)` is a script variable:
    scenario_1.jl:62-64) -- one, two or three hidden layers of width <= 16, linear output layer -- runs on the lane-group kernels of the
    compiled instances (CoopMlp over NetCfgRt: the register copy of the weights zero-padded to width 8 or 5, eight or five lanes per
    trajectory; widths 9 .. 16: sixteen lanes, masked reads of the LDS copy of theta)
    instead of one wavefront per trajectory: forward solve, interpolating adjoint, discrete sweep, checkpointed adjoint, per-member
    parameters.  Per trajectory every number is the oracle's (a single trajectory: every gradient entry), and the same bits as the
    wavefront-per-trajectory runtime-shape kernel (lanes_per_traj = 64)."""
    dims, acts, trainable = LV_RT_CASES[case]
    rng = np.random.default_rng(4000 + case)
    chain = chain_of(dims, acts)
    f = models.ude_dynamics(chain, trainable=trainable)
    nn_off = {None: 0, "delta": 1, "both": 2}[trainable]
    om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, nn_offset=nn_off,
                      lin_idx={None: (-1, -1), "delta": (-1, 0), "both": (0, 1)}[trainable],
                      lin_sign={None: (1.0, 1.0), "delta": (1.0, -1.0), "both": (1.0, -1.0)}[trainable],
                      lin_const={None: (1.3, -1.8), "delta": (1.3, 0.0), "both": (0.0, 0.0)}[trainable])
    lead = {None: [], "delta": [1.8], "both": [1.3, 1.8]}[trainable]
    th = np.concatenate([lead, theta_for(chain, rng, 0.3)]).astype(np.float64)
    assert supported(f) == 1 and f.n_param == th.size == om.n_param
    g = golden("Scenario_1_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    alg, oalg = (U.Tsit5, O.TSIT5) if case % 2 == 0 else (U.Vern7, O.VERN7)
    W64 = U.EnsembleMI355(lanes_per_traj=64)
    for N in (1, 13):
        u0 = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
        data = np.repeat(X[None], N, axis=0)
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), th), u0)
        sol = U.solve(ens, alg(), saveat=t, abstol=1e-6, reltol=1e-6)
        out, st, rc = O.solve_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t)
        assert (rc == 0).all()
        assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts %s %s" % (dims, acts))
        assert_bitwise(sol.u, out, "forward states")
        assert_bitwise(U.rhs(f, u0, th), np.array([O.rhs(om, th, u) for u in u0]), "rhs")
        for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1), (U.InterpolatingAdjoint(checkpointing=True), 0)):
            r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=sense)
            ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6, sensealg=osense), u0, [t[0], t[-1]], th, t, data, nthreads=4)
            what = "%s %s N %d sense %s" % (dims, acts, N, type(sense).__name__)
            assert (r.retcode == 0).all(), what
            if isinstance(sense, U.InterpolatingAdjoint):
                # (the checkpointed forward pass builds Vern7's lazy stages only in steps that hold a save point: column 7 counts those)
                assert_bitwise(r.stats[:, [0, 1, 2, 4, 5, 6]], ref["stats"][:, [0, 1, 2, 4, 5, 6]], what)
                assert_bitwise(r.u, ref["u"], what)
                assert_bitwise(r.grad_u0, ref["grad_u0"], what)
            else:
                check_per_trajectory(r, ref)
            assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss " + what)
            if N == 1:
                assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta " + what)
            else:
                gn = np.linalg.norm(ref["grad_theta"])
                assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn, what
            if sense is None and max(dims[1:-1]) <= 5:   # width <= 5: five lanes per trajectory by default; eight on request -- the same bits
                w8 = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(lanes_per_traj=8))
                assert_bitwise(r.grad_u0, w8.grad_u0, "five lanes vs eight lanes: dL/du0")
                assert_bitwise(r.stats, w8.stats, "five lanes vs eight lanes: counts")
                if N == 1:
                    assert_bitwise(r.grad_theta, w8.grad_theta, "five lanes vs eight lanes: dL/dtheta")
            if sense is None:
                w64 = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=W64)
                assert_bitwise(r.grad_u0, w64.grad_u0, "lane groups vs wavefront per trajectory: dL/du0")
                assert_bitwise(r.stats, w64.stats, "lane groups vs wavefront per trajectory: counts")
                if N == 1:
                    assert_bitwise(r.grad_theta, w64.grad_theta, "lane groups vs wavefront per trajectory: dL/dtheta")
    # per-member parameters (UDE_PT_THETA): every member its own weights, its own gradient row
    N = 5
    thetas = th[None, :] * (1 + 0.2 * rng.standard_normal((N, th.size)))
    u0 = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (t[0], t[-1]), thetas[0]), u0, ps=thetas)
    r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    for j in range(N):
        ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0[j:j + 1], [t[0], t[-1]], thetas[j], t, data[j:j + 1])
        assert_bitwise(r.stats[j], ref["stats"][0], "member %d counts" % j)
        assert_bitwise(r.grad_theta[j], ref["grad_theta"], "member %d dL/dtheta" % j)


@pytest.mark.parametrize("dims,nx", [([1, 16, 16, 16, 1], 1024), ([1, 8, 12, 5, 1], 300), ([1, 1, 1, 1, 1], 33), ([1, 10, 16, 10, 1], 257), ([1, 3, 2, 16, 1], 64)],
                         ids=lambda v: "-".join(map(str, v)) if isinstance(v, list) else str(v))
def test_fisher_kpp_runtime_shape_reaction_network_on_large_grids(dims, nx):
    """round 5: `nn_ode` (Fisher-KPP-CNN.jl:92-126) with an EDITED reaction network 1 -> a -> b -> c -> 1 (tanh hidden layers of width <= 16) on
    grids of 33 .. 1024 points: the run-time-shape instance of the 1024-point matrix-core kernel (KppUdeW over NetCfgRt: operand tables
    padded to width 16, block sums written to the true chain's parameter indices) -- forward solve, right-hand side, interpolating
    adjoint, discrete sweep, per-member parameters; per PDE the oracle's bits (a single PDE: every gradient entry)."""
    acts = ["tanh", "tanh", "tanh", "identity"]
    rng = np.random.default_rng(sum(dims) + nx)
    chain = chain_of(dims, acts)
    f = models.nn_ode(nx, chain)
    assert supported(f) == 1 and supported(f, alg=1) == -2       # (Tsit5; the Vern7 stage storage does not fit the LDS next to the padded tiles)
    om = O.kpp_ude(nx, tuple(dims), tuple(acts))
    th = models.kpp_theta(chain, rng)
    for N in (1, 2):
        u0 = np.clip(models.rho0(26)[None, :] * (1 + 0.1 * rng.uniform(-1, 1, (N, 1))) + 0.01 * rng.uniform(0, 1, (N, 26)), 0, None)
        u0 = np.tile(u0, (1, 40))[:, :nx]
        tf = 1.0
        t = np.linspace(0.0, tf, 6)
        data = rng.uniform(0.0, 1.0, (N, len(t), nx))
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
        sol = U.solve(ens, U.Tsit5(), saveat=t)
        out, st, rc = O.solve_ensemble(om, O.opts(O.TSIT5), u0, [0.0, tf], th, t)
        assert (rc == 0).all()
        assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts %s nx %d" % (dims, nx))
        assert_bitwise(sol.u, out, "forward states")
        assert_bitwise(U.rhs(f, u0, th), np.array([O.rhs(om, th, u) for u in u0]), "rhs")
        for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
            r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, sensealg=sense)
            ref = O.loss_grad_ensemble(om, O.opts(O.TSIT5, sensealg=osense), u0, [0.0, tf], th, t, data, nthreads=2)
            what = "%s nx %d N %d sense %d" % (dims, nx, N, osense)
            assert (r.retcode == 0).all(), what
            assert_bitwise(r.stats[:, [0, 1, 2, 4, 5, 6]], ref["stats"][:, [0, 1, 2, 4, 5, 6]], what)
            assert_bitwise(r.u, ref["u"], what)
            assert_bitwise(r.grad_u0, ref["grad_u0"], what)
            if N == 1:
                assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta " + what)
            else:
                gn = np.linalg.norm(ref["grad_theta"])
                assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn, what
    # per-member parameters
    thetas = np.stack([models.kpp_theta(chain, rng) for _ in range(2)])
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), thetas[0]), u0, ps=thetas)
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t)
    for j in range(2):
        ref = O.loss_grad_ensemble(om, O.opts(O.TSIT5), u0[j:j + 1], [0.0, tf], thetas[j], t, data[j:j + 1])
        assert_bitwise(r.stats[j], ref["stats"][0], "member %d counts" % j)
        assert_bitwise(r.grad_theta[j], ref["grad_theta"], "member %d dL/dtheta" % j)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_discrete_sweep_random_shapes(golden, seed):
    """round 4: `sensealg = ForwardDiffSensitivity()` -- what scenario_1.jl:86 / scenario_2.jl:108 request -- for ANY chain: the
    runtime-shape kernel's reverse sweep defers the parameter cotangent (factors of up to 8 VJPs in the stage storage, added to
    the accumulators in VJP order) and is bit-identical to the oracle's discrete_sweep: VJP count, dL/du0, and for a single
    trajectory every gradient entry.  LV kind (seeds 0-4) and the SEIR kinds (5-7); Vern7 steps with interior save points
    exercise the 16-VJP steps (two flushes)."""
    rng = np.random.default_rng(700 + seed)
    if seed < 5:
        dims, acts = random_chain(rng, 2, 2)
        chain = chain_of(dims, acts)
        trainable = [None, "delta", "both"][seed % 3]
        f = models.ude_dynamics(chain, trainable=trainable)
        om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, nn_offset={None: 0, "delta": 1, "both": 2}[trainable],
                          lin_idx={None: (-1, -1), "delta": (-1, 0), "both": (0, 1)}[trainable],
                          lin_sign={None: (1.0, 1.0), "delta": (1.0, -1.0), "both": (1.0, -1.0)}[trainable],
                          lin_const={None: (1.3, -1.8), "delta": (1.3, 0.0), "both": (0.0, 0.0)}[trainable])
        th = np.concatenate([{None: [], "delta": [1.8], "both": [1.3, 1.8]}[trainable], theta_for(chain, rng, 0.3)]).astype(np.float64)
        g = golden("Scenario_1_recovery_0.005")
        X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
        t = np.array(g["solution"]["t"])
        N = 1 if seed % 2 == 0 else 5
        u0 = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
        data = np.repeat(X[None], N, axis=0)
        tspan, mask = (t[0], t[-1]), None
    else:
        node = seed % 2 == 1
        dims, acts = random_chain(rng, 7 if node else 3, 7 if node else 1, max_hidden=3, widths=[4, 16, 33, 64])
        chain = chain_of(dims, acts)
        f = (models.dudt_node if node else models.dudt_)(chain)
        om = O.make_model(O.KIND_SEIR_NODE if node else O.KIND_SEIR_UDE, 7, dims, acts, consts=O.SEIR_P)
        th = theta_for(chain, rng, 0.5)
        N = 1 if seed < 7 else 3
        u0 = np.zeros((N, 7)); u0[:, 0] = rng.uniform(0.8, 0.95, N) * 100.0; u0[:, 1] = rng.uniform(0.5, 2.0, N); u0[:, 2] = rng.uniform(0.2, 1.0, N); u0[:, 4] = 100.0
        t = np.arange(0.0, 4.5, 1.0)
        data, _, _ = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 4.0], [], t)
        tspan, mask = (0.0, 4.0), [0, 1, 1, 1, 0, 0, 0]
    alg, oalg = (U.Vern7, O.VERN7) if seed % 2 else (U.Tsit5, O.TSIT5)
    assert supported(f, sense=1) == 1
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], tspan, th), u0)
    r = U.loss_and_gradient(ens, alg(), data, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=U.ForwardDiffSensitivity())
    ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6, sensealg=1), u0, list(tspan), th, t, data, row_mask=mask, nthreads=4)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")
    if N == 1:
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta %s %s" % (dims, acts))
    else:
        gn = np.linalg.norm(ref["grad_theta"])
        assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_seir_kinds_random_shapes(seed):
    rng = np.random.default_rng(200 + seed)
    node = seed % 2 == 1
    dims, acts = random_chain(rng, 7 if node else 3, 7 if node else 1, max_hidden=4, widths=[4, 16, 33, 64])
    chain = chain_of(dims, acts)
    f = (models.dudt_node if node else models.dudt_)(chain)
    om = O.make_model(O.KIND_SEIR_NODE if node else O.KIND_SEIR_UDE, 7, dims, acts, consts=O.SEIR_P)
    th = theta_for(chain, rng, 0.5)
    assert supported(f) == 1
    N = 1 if seed < 4 else 4
    S0 = 100.0
    u0 = np.zeros((N, 7))
    u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
    u0[:, 1] = rng.uniform(0.5, 2.0, N)
    u0[:, 2] = rng.uniform(0.2, 1.0, N)
    u0[:, 4] = S0
    tf = 4.0
    t = np.arange(0.0, tf + 0.5, 1.0)
    mask = [0, 1, 1, 1, 0, 0, 0]
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
    alg, oalg = (U.Vern7, O.VERN7) if seed % 4 < 2 else (U.Tsit5, O.TSIT5)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    du = U.rhs(f, u0, th)
    assert_bitwise(du, np.array([O.rhs(om, th, u) for u in u0]), "rhs %s %s" % (dims, acts))
    r = U.loss_and_gradient(ens, alg(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6)
    ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=mask, nthreads=4)
    assert (r.retcode == 0).all() and (ref["retcode"] == 0).all()
    check_per_trajectory(r, ref)
    assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")
    if N == 1:
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta %s %s" % (dims, acts))
    else:
        gn = np.linalg.norm(ref["grad_theta"])
        assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn


@pytest.mark.parametrize("dims", [[3, 64, 63, 1], [3, 16, 16, 1], [3, 33, 64, 1], [3, 64, 16, 1], [3, 17, 50, 1], [3, 63, 32, 1],
                                  # ... more corners of the family: H2 one above a tile, both lengths between tiles, the widest chain input
                                  # with the narrowest output, H2 = 32 with a narrow H1 (forward on the wavefront kernel), and H1 = 32 (the
                                  # tree case the lock-step instances exclude: both passes on the wavefront kernel, same oracle)
                                  [3, 64, 33, 1], [3, 48, 48, 1], [3, 31, 47, 1], [3, 49, 64, 1], [3, 16, 64, 1], [3, 64, 17, 1], [3, 20, 32, 1],
                                  [3, 32, 40, 1],
                                  # widths below a tile: served unless a 32- / 64-term product has fewer than 16 results (64-8 and 8-64 stay on
                                  # the wavefront kernel, like 32-40)
                                  [3, 8, 8, 1], [3, 5, 40, 1], [3, 63, 3, 1], [3, 1, 1, 1], [3, 64, 8, 1], [3, 8, 64, 1]],
                         ids=lambda d: "-".join(map(str, d)))
@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
def test_runtime_shape_exposure_chain_on_the_lockstep_matrix_core_kernel(dims, alg, oalg):
    """round 5: an exposure-UDE chain 3 -> H1 -> H2 -> 1 (tanh, tanh, identity; H1, H2 <= 64 without a 32- / 64-term product of fewer than 16 results) WITHOUT a compiled instance
    runs its forward and backward passes on the lock-step matrix-core kernels (csrc/ude_seir_ls_fwd.h / ude_seir_ls2.h, GEN: weights zero-padded to 64 x 64, a 64-term
    product in four chains, a shorter one in ONE ascending chain, the input cotangent a tree for H1 = 64 and a chain otherwise) -- every
    number per trajectory as the oracle has it, every gradient entry for a single trajectory, and the same bits as the
    wavefront-per-trajectory runtime-shape kernel (lanes_per_traj = 64), which is 4x slower on the configs[2] share (47.6 vs 11 ms)"""
    acts = ["tanh", "tanh", "identity"]
    rng = np.random.default_rng(sum(dims))
    chain = chain_of(dims, acts)
    f = models.dudt_(chain)
    om = O.make_model(O.KIND_SEIR_UDE, 7, dims, acts, consts=O.SEIR_P)
    th = theta_for(chain, rng, 0.5)
    assert supported(f) == 1
    mask = [0, 1, 1, 1, 0, 0, 0]
    tf = 4.0
    t = np.arange(0.0, tf + 0.5, 1.0)
    for N in (1, 21):
        S0 = 100.0
        u0 = np.zeros((N, 7))
        u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
        u0[:, 1] = rng.uniform(0.5, 2.0, N)
        u0[:, 2] = rng.uniform(0.2, 1.0, N)
        u0[:, 4] = S0
        truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
        # the plain solve: the lock-step forward kernel's runtime-shape instance (ude_seir_ls_fwd.h, GEN; H2 = 32 stays on the wavefront kernel)
        sol = U.solve(ens, alg(), saveat=t, abstol=1e-6, reltol=1e-6)
        out, st, rc = O.solve_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t)
        assert (rc == 0).all() and (np.asarray(sol.retcodes) == 0).all()
        assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts %s" % dims)
        assert_bitwise(sol.u, out, "forward states %s" % dims)
        r = U.loss_and_gradient(ens, alg(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6)
        ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=mask, nthreads=8)
        assert (r.retcode == 0).all() and (ref["retcode"] == 0).all()
        check_per_trajectory(r, ref)
        w64 = U.loss_and_gradient(ens, alg(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(lanes_per_traj=64))
        assert_bitwise(r.grad_u0, w64.grad_u0, "lock-step vs wavefront-per-trajectory: dL/du0")
        assert_bitwise(r.stats, w64.stats, "lock-step vs wavefront-per-trajectory: counts")
        if dims in ([3, 64, 63, 1], [3, 32, 40, 1]) and N == 21:
            # an EXPLICIT lanes_per_traj = 16 (the lock-step request; advisor, round 5: it used to end in "no kernel instance ...
            # lanes_per_traj 16" for a runtime-shape chain): the same kernels the default selects -- or, for an excluded shape, the
            # wavefront-per-trajectory ones --, the same bits
            w16 = U.loss_and_gradient(ens, alg(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(lanes_per_traj=16))
            assert_bitwise(r.grad_u0, w16.grad_u0, "lanes_per_traj = 16: dL/du0")
            assert_bitwise(r.stats, w16.stats, "lanes_per_traj = 16: counts")
            assert_bitwise(r.grad_theta, w16.grad_theta, "lanes_per_traj = 16: dL/dtheta")
        gn = np.linalg.norm(ref["grad_theta"])
        if N == 1:
            assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta %s" % dims)
            assert_bitwise(r.grad_theta, w64.grad_theta, "lock-step vs wavefront-per-trajectory: dL/dtheta")
        else:
            assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn


def test_generic_kernel_agrees_with_the_compiled_instance_of_the_same_shape():
    """The fallback is chosen by shape alone.  scenario_1's own chain runs on its compiled instance, a chain one neuron away from it
    on the fallback: both bit-identical to the same oracle, i.e. the two kernels implement one specification."""
    rng = np.random.default_rng(7)
    for dims in ([2, 5, 5, 5, 2], [2, 5, 6, 5, 2]):
        acts = ["rbf", "rbf", "rbf", "identity"]
        chain = chain_of(dims, acts)
        f = models.ude_dynamics(chain)
        om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, lin_const=(1.3, -1.8))
        th = theta_for(chain, rng, 0.5)
        assert supported(f) == (0 if dims[2] == 5 else 1)
        u0 = np.array([[0.44249296, 4.6280594]])
        t = np.linspace(0.0, 3.0, 31)
        data = np.ones((1, 31, 2))
        r = U.loss_and_gradient(U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 3.0), th), u0), U.Vern7(), data, saveat=t, abstol=1e-6, reltol=1e-6)
        ref = O.loss_grad_ensemble(om, O.opts(O.VERN7, 1e-6, 1e-6), u0, [0.0, 3.0], th, t, data)
        assert_bitwise(r.stats, ref["stats"], "stats")
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta")


@pytest.mark.parametrize("n_weights", [1, 2, 3])
def test_fisher_kpp_small_n_weights(n_weights):
    """Fisher-KPP-CNN-Small.jl:88 `n_weights` (timing log :343-391 for 1, 2, 3): reaction network 1 -> n -> 1 tanh"""
    rng = np.random.default_rng(40 + n_weights)
    nx = 26
    chain = models.kpp_small_chain(n_weights)
    f = models.nn_ode(nx, chain)
    th = models.kpp_theta(chain, rng)
    assert th.size == 3 * n_weights + 6 and supported(f) == 0
    om = O.kpp_ude(nx, (1, n_weights, 1), ("tanh", "identity"))
    rho = models.rho0(nx)
    t = np.arange(11) * 0.5
    truth, _, rc = O.solve_ensemble(O.kpp_true(nx), O.opts(O.TSIT5), rho, [0.0, 5.0], [], t)
    prob = U.ODEProblem(f, rho, (0.0, 5.0), th)
    r = U.loss_and_gradient(prob, U.Tsit5(), truth, saveat=t)
    ref = O.loss_grad_ensemble(om, O.opts(O.TSIT5), rho[None], [0.0, 5.0], th, t, truth)
    assert (r.retcode == 0).all()
    assert_bitwise(r.stats, ref["stats"], "stats")
    assert_bitwise(r.u, ref["u"], "states")
    assert_bitwise(r.grad_u0, ref["grad_u0"], "dL/du0")
    gn = np.linalg.norm(ref["grad_theta"])
    assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < 1e-12 * gn


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_fisher_kpp_runtime_shape_reaction_network(seed):
    """round 4: `nn_ode` (Fisher-KPP-CNN.jl:92-126) with ANY pointwise reaction chain 1 -> ... -> 1 of <= 4 layers and width <= 32 --
    the scripts' `FastChain(...)` / `n_weights` are variables -- on the reference's grid sizes (<= 32 points, ragged), through the
    runtime-shape kernel csrc/ude_model_kpp_generic.h: forward solve, interpolating adjoint, discrete sweep and the checkpointed
    adjoint; per trajectory bit-identical to the oracle (a single PDE: every gradient entry)."""
    rng = np.random.default_rng(900 + seed)
    nh = int(rng.integers(1, 4))
    widths = [1, 2, 4, 7, 8, 13, 16] if nh == 3 else [1, 3, 6, 12, 17, 24, 32]
    while True:
        dims = [1] + [int(rng.choice(widths)) for _ in range(nh)] + [1]
        if sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1)) + 5 <= 768:
            break
    if seed == 0:
        dims = [1, 16, 16, 16, 1]                     # 598 parameters: close to the 768 the kernel takes
    acts = [str(rng.choice(["tanh", "rbf", "relu"])) for _ in range(len(dims) - 2)] + ["identity"]
    chain = chain_of(dims, acts)
    nx = int(rng.integers(3, 33))
    f = models.nn_ode(nx, chain)
    assert f.n_param <= 768 and supported(f) == 1, (dims, f.n_param)
    om = O.kpp_ude(nx, tuple(dims), tuple(acts))
    th = models.kpp_theta(chain, rng)
    th[f.stencil_offset:f.stencil_offset + 3] = np.array([1.0, -2.0, 1.0]) + 0.1 * rng.standard_normal(3)
    th[f.d0_offset] = rng.uniform(1.0, 6.0)
    N = 1 if seed % 2 == 0 else 3
    u0 = np.clip(models.rho0(nx)[None, :] * (1 + 0.2 * rng.uniform(-1, 1, (N, 1))) + 0.02 * rng.uniform(0, 1, (N, nx)), 0, None)
    tf = float(rng.uniform(0.5, 3.0))
    t = np.unique(np.concatenate([[0.0], np.sort(rng.uniform(0.0, tf, 5)), [tf]]))
    data = rng.uniform(0.0, 1.0, (N, len(t), nx))
    alg, oalg = (U.Vern7, O.VERN7) if seed % 3 == 0 else (U.Tsit5, O.TSIT5)
    tol = float(10.0 ** rng.uniform(-7, -4))
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    sol = U.solve(ens, alg(), saveat=t, abstol=tol, reltol=tol)
    out, st, rc = O.solve_ensemble(om, O.opts(oalg, tol, tol), u0, [0.0, tf], th, t)
    assert (rc == 0).all()
    assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts %s %s" % (dims, acts))
    assert_bitwise(sol.u, out, "forward states")
    assert_bitwise(U.rhs(f, u0, th), np.array([O.rhs(om, th, u) for u in u0]), "rhs")
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1), (U.InterpolatingAdjoint(checkpointing=True), 0)):
        r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=tol, reltol=tol, sensealg=sense)
        ref = O.loss_grad_ensemble(om, O.opts(oalg, tol, tol, sensealg=osense), u0, [0.0, tf], th, t, data, nthreads=3)
        what = "%s %s nx %d sense %s" % (dims, acts, nx, type(sense).__name__)
        assert (r.retcode == 0).all(), what
        assert_bitwise(r.retcode, ref["retcode"], what)
        assert_bitwise(r.stats[:, [0, 1, 2, 4, 5, 6]], ref["stats"][:, [0, 1, 2, 4, 5, 6]], what)
        assert_bitwise(r.u, ref["u"], what)
        assert_bitwise(r.grad_u0, ref["grad_u0"], what)
        if N == 1:
            assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta " + what)
        else:
            gn = np.linalg.norm(ref["grad_theta"])
            assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn, what


def test_tanh_bits_across_the_argument_range():
    """ARITH-SPEC tanh = em / (em + 2), em = expm1(2|x|): the kernels form that quotient without the operand scaling and special-case
    fix-up of the general division (csrc/ude_math.h, div_em) -- it must still be the correctly rounded quotient for EVERY argument:
    zeros, subnormal and barely normal arguments (the divisor is then exactly 2 and the quotient may be subnormal), the range where
    em + 2 starts to differ from 2, moderate and saturating arguments.  A 2-2-2 chain with identity weights makes the right-hand
    side return tanh(u) itself: compared bit for bit with the oracle, which divides with the C operator."""
    dims, acts = [2, 2, 2], ["tanh", "identity"]
    chain = chain_of(dims, acts)
    f = models.ude_dynamics(chain, p_true=(0.0, 0.9, 0.8, 0.0))
    om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, lin_const=(0.0, -0.0))
    th = np.array([1, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0], dtype=np.float64)          # W1 = I, b1 = 0, W2 = I, b2 = 0
    assert f.n_param == th.size == om.n_param
    rng = np.random.default_rng(7)
    tiny = np.float64(5e-324)
    special = [0.0, tiny, 3 * tiny, 1e-320, 2.2250738585072014e-308, 2.2250738585072009e-308, 4.4501477170144023e-308, 1e-300, 1e-292,
               1e-280, 2.0 ** -54, 2.0 ** -53, 2.0 ** -52, 2.0 ** -51, 1e-10, 1e-3, 0.34657359027997264, 0.5, 1.0, 10.0, 19.999999999999996, 20.0,
               25.0, 700.0, np.nan]
    x = np.concatenate([special, 10.0 ** rng.uniform(-323, 1.5, 4000), rng.uniform(0, 2, 2000), np.ldexp(rng.uniform(1, 2, 1000), rng.integers(-1074, -1000, 1000))])
    x = np.concatenate([x, -x])
    if x.size % 2:
        x = np.append(x, 0.0)
    u = x.reshape(-1, 2)
    du = U.rhs(f, u, th)
    ref = np.array([O.rhs(om, th, ui) for ui in u])
    assert_bitwise(du, ref, "tanh over %d arguments" % x.size)
    assert np.nanmax(np.abs(ref)) == 1.0 and (np.abs(ref[np.abs(u) < 1e-300]) <= np.abs(u[np.abs(u) < 1e-300])).all()
    assert np.isnan(du[np.isnan(u)]).all() and np.isnan(du).sum() == 4   # (0 * NaN: the row's other component is NaN as well)
