"""The pure neural ODE of the SEIR script on the GPU: dudt_node, FastChain 7-64-64-64-7 tanh, 9287 parameters
(/root/reference/SEIR_exposure/seir_exposure.jl:53-83; the call site of InterpolatingAdjoint at :69-73).
Device (SeirNode<64>: one wavefront per trajectory, two 64x64 layers in LDS, deferred parameter cotangent) against the
oracle's generic dense-chain restatement: per trajectory bit-identical, ensemble sums to summation order."""
import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_parity import REL_GRAD_SUM, assert_bitwise, check_per_trajectory

pytestmark = pytest.mark.gpu
MASK = [0, 1, 1, 1, 0, 0, 0]


def node_case(N, S0, seed=21):
    rng = np.random.default_rng(seed)
    u0 = np.zeros((N, 7))
    u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
    u0[:, 1] = rng.uniform(0.5, 2.0, N)
    u0[:, 2] = rng.uniform(0.2, 1.0, N)
    u0[:, 4] = S0
    th = models.seir_node_chain().glorot_uniform(rng)
    return u0, th


def test_node_rhs_matches_oracle():
    u0, th = node_case(9, 100.0)
    f = models.dudt_node()
    assert f.n_param == 9287
    du = U.rhs(f, u0, th)
    ref = np.array([O.rhs(O.seir_node(), th, u) for u in u0])
    assert_bitwise(du, ref, "dudt_node")
    assert np.abs(ref[:, :4]).min() > 0 and np.array_equal(ref[:, 4], -0.02 * u0[:, 4])


@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
@pytest.mark.parametrize("S0,tf", [(100.0, 6.0), (14e6, 21.0)])
def test_node_forward_and_adjoint_match_oracle(alg, oalg, S0, tf):
    """S0 = 100: every layer alive, > 8000 of the 9287 parameter cotangents nonzero (the two unused output rows and the
    rows of saturated first-layer neurons are exactly zero); S0 = 14e6: the script's own scale (N enters the first layer unscaled and saturates it)."""
    N = 6
    u0, th = node_case(N, S0)
    t = np.arange(0.0, tf + 0.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
    f = models.dudt_node()
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    sol = U.solve(ens, alg(), saveat=t, abstol=1e-6, reltol=1e-6)
    out, st, rc = O.solve_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t)
    assert (rc == 0).all()
    assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts")
    assert_bitwise(sol.u, out, "forward states")
    ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
    gn = np.linalg.norm(ref["grad_theta"])
    for ealg in (None, U.EnsembleMI355(64)):    # the default (the lock-step kernel, csrc/ude_node_ls.h) and the wavefront-per-trajectory kernel
        r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=ealg)
        assert (r.retcode == 0).all()
        check_per_trajectory(r, ref)
        assert_bitwise(r.loss_per_traj, ref["loss_per_traj"], "per-trajectory loss")
        assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn
    W4g = r.grad_theta[-455:-7].reshape(64, 7)
    assert np.all(W4g[:, 5:] == 0) and np.all(r.grad_theta[-2:] == 0) and np.abs(W4g[:, 1:4]).min() > 0   # rows of dE, dI, dR: the masked loss rows
    if S0 == 100.0:
        assert np.count_nonzero(r.grad_theta) > 8000       # (first-layer neurons saturated by N = 100 have exactly zero rows)


def test_node_single_trajectory_gradient_is_bitwise():
    """one trajectory: no sum over trajectories, so every one of the 9287 slots must carry the oracle's bits"""
    u0, th = node_case(1, 100.0, seed=5)
    t = np.arange(0.0, 4.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 4.0], [], t)
    prob = U.ODEProblem(models.dudt_node(), u0[0], (0.0, 4.0), th)
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
        r = U.loss_and_gradient(prob, U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=sense)
        ref = O.loss_grad_ensemble(O.seir_node(), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=osense), u0, [0.0, 4.0], th, t, truth, row_mask=MASK)
        assert_bitwise(r.stats, ref["stats"], "stats")
        assert_bitwise(r.grad_u0, ref["grad_u0"], "dL/du0")
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta")


@pytest.mark.parametrize("N", [1, 6])
def test_node_adjoint_is_reproducible_run_to_run(N):
    """Regression guard: earlier builds of these kernels produced a garbage backward solve depending on what had run before
    on the device (a register copy the compiler placed under a narrowed EXEC kept stale lanes: DESIGN.md 8b; the
    deterministic check is tests/test_gpu_poison.py); alternate the two algorithms several times and demand the oracle's
    bits every time."""
    u0, th = node_case(N, 100.0)
    t = np.arange(0.0, 6.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 6.0], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, 6.0), th), u0)
    refs = {a: O.loss_grad_ensemble(O.seir_node(), O.opts(oa, 1e-6, 1e-6), u0, [0.0, 6.0], th, t, truth, row_mask=MASK, nthreads=4)
            for a, oa in (("t", O.TSIT5), ("v", O.VERN7))}
    for a in "ttvttvvt":
        r = U.loss_and_gradient(ens, U.Tsit5() if a == "t" else U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6)
        assert_bitwise(r.stats, refs[a]["stats"], "stats (%s)" % a)
        assert_bitwise(r.grad_u0, refs[a]["grad_u0"], "dL/du0 (%s)" % a)


@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
@pytest.mark.parametrize("N,S0,tf", [(1, 100.0, 6.0), (21, 100.0, 6.0), (5, 14e6, 21.0)])
def test_node_lockstep_kernel_matches_oracle(alg, oalg, N, S0, tf):
    """the lock-step matrix-core backward kernel of the neural ODE (csrc/ude_node_ls.h, lanes_per_traj = 16): backward step counts,
    dL/du0 and -- for a single trajectory -- every one of the 9287 gradient entries bit-identical to the oracle"""
    u0, th = node_case(N, S0)
    t = np.arange(0.0, tf + 0.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, tf), th), u0)
    r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(16))
    ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    gn = np.linalg.norm(ref["grad_theta"])
    assert gn > 0 and np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * gn
    if N == 1:
        assert_bitwise(r.grad_theta, ref["grad_theta"], "dL/dtheta (single trajectory)")


def test_node_lockstep_kernel_user_cotangent_fixed_dt_and_failures():
    """the lock-step neural-ODE kernel behind ude_vjp_ensemble (a user cotangent), with `dt = ...` given, and with backward solves that
    stop at maxiters (retcodes and work counts per trajectory, zero gradient rows, the slots moving on to the rest of the ensemble)"""
    N, S0, tf = 37, 100.0, 6.0
    u0, th = node_case(N, S0)
    t = np.arange(0.0, tf + 0.5, 1.0)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, tf), th), u0)
    kw = {"ensemblealg": U.EnsembleMI355(16)}
    cot = np.random.default_rng(3).normal(size=(N, len(t), 7))
    r = U.adjoint_pullback(ens, U.Vern7(), cot, saveat=t, abstol=1e-6, reltol=1e-6, **kw)
    ref = O.vjp_ensemble(O.seir_node(), O.opts(O.VERN7, 1e-6, 1e-6), u0, [0.0, tf], th, t, cot, nthreads=6)
    assert (r.retcode == 0).all()
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
    r = U.loss_and_gradient(ens, U.Tsit5(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, dt=0.02, **kw)
    ref = O.loss_grad_ensemble(O.seir_node(), O.opts(O.TSIT5, 1e-6, 1e-6, dt0=0.02), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
    check_per_trajectory(r, ref)
    r = U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, maxiters=9, allow_failures=True, **kw)
    ref = O.loss_grad_ensemble(O.seir_node(), O.opts(O.VERN7, 1e-6, 1e-6, maxiters=9), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
    assert_bitwise(r.retcode, ref["retcode"], "retcodes")
    assert (r.retcode != 0).any() and np.isinf(r.loss)
    assert_bitwise(r.stats[:, [0, 1, 2, 4, 5, 6]], ref["stats"][:, [0, 1, 2, 4, 5, 6]], "work counts")


@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
def test_node_lockstep_forward_kernel_plain_solves(alg, oalg):
    """the forward lock-step kernel of the neural ODE (csrc/ude_node_ls_fwd.h) without a dense store: a ragged save grid that
    contains neither end point (Vern7's lazy stages only in steps with a save point strictly inside), a given dt, 41 trajectories
    on 16-slot blocks, solves that stop at maxiters -- and the same calls on the wavefront-per-trajectory kernel"""
    N, tf = 41, 6.0
    u0, th = node_case(N, 100.0)
    grid = np.array([0.37, 2.0, 2.5, 4.25, 5.99])
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, tf), th), u0)
    for kw, okw in (({}, {}), ({"dt": 0.03}, {"dt0": 0.03}), ({"maxiters": 5}, {"maxiters": 5})):
        out, st, rc = O.solve_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6, **okw), u0, [0.0, tf], th, grid)
        for lanes in (16, 64):
            sol = U.solve(ens, alg(), saveat=grid, abstol=1e-6, reltol=1e-6, ensemblealg=U.EnsembleMI355(lanes), **kw)
            assert_bitwise(sol.retcodes, rc, "retcodes %s" % kw)
            assert_bitwise(sol.stats[:, :4], st[:, :4], "forward counts incl. lazy stages %s" % kw)
            if "maxiters" in kw:
                assert (rc != 0).any()          # (a mixed block: some slots stop at maxiters, the others finish)
            else:
                assert (rc == 0).all()
            assert_bitwise(sol.u[rc == 0], out[rc == 0], "states %s" % kw)


def test_node_lockstep_forward_kernel_dense_overflow_matches_the_wavefront_kernel():
    """a dense store of 3 steps: every trajectory stops with DenseOverflow after its third accepted step, with the same counters
    whichever forward kernel ran"""
    N, tf = 19, 6.0
    u0, th = node_case(N, 100.0)
    t = np.arange(0.0, tf + 0.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, tf), th), u0)
    res = [U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6,
                               ensemblealg=U.EnsembleMI355(lanes, 3), allow_failures=True) for lanes in (16, 64)]
    assert (res[0].retcode != 0).all()
    assert_bitwise(res[0].retcode, res[1].retcode, "retcodes")
    assert_bitwise(res[0].stats, res[1].stats, "counters")
    assert np.isinf(res[0].loss) and not res[0].grad_theta.any()
