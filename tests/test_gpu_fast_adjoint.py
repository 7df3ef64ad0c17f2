"""UDE_SENSE_INTERPOLATING_ADJOINT_FAST (SURVEY.md 8(b) `fast` mode): the interpolating adjoint with only lambda under
error control, the parameter cotangent a quadrature on the accepted steps.  Device against the oracle's same mode: per
trajectory bit-identical (backward step counts, dL/du0, and for one trajectory every gradient entry); against the default
(parity) mode: same forward pass, gradients equal to the solver tolerance."""
import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_parity import REL_GRAD_SUM, assert_bitwise, check_per_trajectory, seir_inputs, kpp_case
from test_gpu_node import node_case, MASK

pytestmark = pytest.mark.gpu
S1 = "Scenario_1_recovery_0.005"
FAST = U.FastInterpolatingAdjoint


@pytest.mark.parametrize("alg,oalg", [(U.Tsit5, O.TSIT5), (U.Vern7, O.VERN7)])
def test_lv_fast_mode_matches_oracle_and_parity_mode(golden, alg, oalg):
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    rng = np.random.default_rng(9)
    N = 25
    u0 = X[0] * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
    r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST())
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(oalg, 1e-6, 1e-6, sensealg=2), u0, [t[0], t[-1]], th, t, data, nthreads=8)
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    full = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6)
    assert_bitwise(r.u, full.u, "forward pass") and None
    # (not necessarily fewer backward steps: the parity mode's RMS over n + np components dilutes lambda's error, the
    # lambda-only norm does not -- what the fast mode saves is the per-slot error estimate, its divisions and accumulators)
    assert np.linalg.norm(r.grad_theta - full.grad_theta) < 2e-4 * np.linalg.norm(full.grad_theta)
    # one trajectory: every entry of the gradient carries the oracle's bits
    one = U.loss_and_gradient(U.ODEProblem(models.ude_dynamics(), u0[3], (t[0], t[-1]), th), alg(), data[:1], saveat=t,
                              abstol=1e-6, reltol=1e-6, sensealg=FAST())
    ref1 = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(oalg, 1e-6, 1e-6, sensealg=2), u0[3], [t[0], t[-1]], th, t, data[:1])
    assert_bitwise(one.grad_theta, ref1["grad_theta"], "dL/dtheta")


def _seir_fast_case(N, seed=11, scale=10.0):
    u0, t = seir_inputs(N)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    th = models.seir_chain().glorot_uniform(np.random.default_rng(seed))
    th[-65:-1] *= scale
    return u0, t, truth, th


def test_seir_fast_mode_matches_oracle():
    """the wavefront-per-trajectory kernel (lanes_per_traj = 64): deferred parameter cotangent, the slot sums are formed on accepted
    steps only (commit_slots) -- the oracle's UDEO_SENSE_FAST association"""
    N = 8
    u0, t, truth, th = _seir_fast_case(N)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
    W64 = U.EnsembleMI355(lanes_per_traj=64)
    for alg, oalg in ((U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)):
        r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), ensemblealg=W64)
        ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(oalg, 1e-6, 1e-6, sensealg=2), u0, [0.0, 21.0], th, t, truth, row_mask=MASK, nthreads=8)
        check_per_trajectory(r, ref)
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
        full = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6)
        assert np.linalg.norm(r.grad_theta - full.grad_theta) < 1e-3 * np.linalg.norm(full.grad_theta)


@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
@pytest.mark.parametrize("kw", [{}, {"dt": 0.9}], ids=["auto-dt", "dt0.9-rejections"])
def test_seir_fast_mode_block_level_matrix_core_accumulation(alg, oalg, kw):
    """round 5: the DEFAULT fast-mode kernel of the SEIR exposure UDE (csrc/ude_seir_lsf.h): 16 trajectory slots in lock-step, the
    parameter cotangent accumulated per trip on v_mfma_f64_16x16x4 into block-resident registers -- no mu in HBM, no per-trajectory
    gradient row.  The lambda solve is the fast mode's (backward step counts and dL/du0 bit-identical to the oracle per trajectory);
    the gradient is the oracle's UDEO_SENSE_FAST_MM association: ONE trajectory -> every one of the 4481 entries bit for bit (a
    given dt = 0.9 makes the first attempts fail: rejected attempts are replayed with negated weights, oracle and device alike);
    several trajectories interleave in the block's chains -> <= 1e-12; and against the parity mode: the solver tolerance."""
    okw = {"dt0": kw["dt"]} if kw else {}
    # one trajectory: bitwise, all entries
    u0, t, truth, th = _seir_fast_case(3)
    for j in (0, 2):
        one = U.loss_and_gradient(U.ODEProblem(models.dudt_(), u0[j], (0.0, 21.0), th), alg(), truth[j:j + 1], row_mask=MASK, saveat=t,
                                  abstol=1e-6, reltol=1e-6, sensealg=FAST(), **kw)
        ref1 = O.loss_grad_ensemble(O.seir_ude(), O.opts(oalg, 1e-6, 1e-6, sensealg=4, **okw), u0[j], [0.0, 21.0], th, t, truth[j:j + 1], row_mask=MASK)
        check_per_trajectory(one, ref1)
        if kw:
            assert ref1["stats"][:, 6].sum() > 0      # the case does contain rejected backward steps
        assert_bitwise(one.grad_theta, ref1["grad_theta"], "dL/dtheta, single trajectory")
    # partial block, one full block + a refill, several blocks
    for N in (5, 37, 300):
        u0, t, truth, th = _seir_fast_case(N)
        ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
        r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), **kw)
        ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(oalg, 1e-6, 1e-6, sensealg=4, **okw), u0, [0.0, 21.0], th, t, truth, row_mask=MASK, nthreads=8)
        check_per_trajectory(r, ref)
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
        r2 = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), **kw)
        assert_bitwise(r.grad_theta, r2.grad_theta, "two runs, same bits (no queue: the trajectories are dealt round-robin)")
        if N == 37 and not kw:
            full = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6)
            assert np.linalg.norm(r.grad_theta - full.grad_theta) < 1e-3 * np.linalg.norm(full.grad_theta)


@pytest.mark.parametrize("dims", [[3, 64, 63, 1], [3, 16, 16, 1], [3, 33, 64, 1], [3, 64, 17, 1], [3, 31, 47, 1], [3, 49, 32, 1], [3, 8, 8, 1], [3, 5, 40, 1]],
                         ids=lambda d: "-".join(map(str, d)))
@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
def test_runtime_shape_exposure_chain_fast_mode_on_the_lockstep_kernel(dims, alg, oalg):
    """round 5: an exposure-UDE chain 3 -> H1 -> H2 -> 1 without a compiled instance in the `fast` mode: the runtime-shape instance of
    csrc/ude_seir_lsf.h (weights zero-padded to 64 x 64, the block's accumulators hold the padded gradient).  One trajectory: every
    gradient entry bit-identical to the oracle's UDEO_SENSE_FAST_MM association, also with rejected (replayed) attempts under a given
    dt; ensembles: per trajectory the oracle's step counts and dL/du0, gradient <= 1e-12, two runs the same bits."""
    acts = ["tanh", "tanh", "identity"]
    chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(3)])
    f = models.dudt_(chain)
    om = O.make_model(O.KIND_SEIR_UDE, 7, dims, acts, consts=O.SEIR_P)
    rng = np.random.default_rng(sum(dims))
    th = chain.glorot_uniform(rng)
    th[-(dims[2] + 1):-1] *= 10.0
    for kw in ({}, {"dt": 0.9}):
        okw = {"dt0": kw["dt"]} if kw else {}
        u0, t = seir_inputs(3)
        truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
        one = U.loss_and_gradient(U.ODEProblem(f, u0[1], (0.0, 21.0), th), alg(), truth[1:2], row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6,
                                  sensealg=FAST(), **kw)
        ref1 = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6, sensealg=4, **okw), u0[1], [0.0, 21.0], th, t, truth[1:2], row_mask=MASK)
        check_per_trajectory(one, ref1)
        assert_bitwise(one.grad_theta, ref1["grad_theta"], "dL/dtheta, single trajectory %s %s" % (dims, kw))
    u0, t = seir_inputs(37)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 21.0], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, 21.0), th), u0)
    r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST())
    ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6, sensealg=4), u0, [0.0, 21.0], th, t, truth, row_mask=MASK, nthreads=8)
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    r2 = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST())
    assert_bitwise(r.grad_theta, r2.grad_theta, "two runs, same bits")
    full = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6)
    assert np.linalg.norm(r.grad_theta - full.grad_theta) < 1e-3 * np.linalg.norm(full.grad_theta)


@pytest.mark.parametrize("alg,oalg", [(U.Vern7, O.VERN7), (U.Tsit5, O.TSIT5)])
@pytest.mark.parametrize("kw", [{}, {"dt": 0.9}], ids=["auto-dt", "dt0.9-rejections"])
def test_node_fast_mode_block_level_matrix_core_accumulation(alg, oalg, kw):
    """the neural ODE 7-64-64-64-7 (seir_exposure.jl:53-83) through csrc/ude_node_lsf.h, the default of its fast mode: one trajectory ->
    every one of the 9287 gradient entries bit-identical to the oracle's UDEO_SENSE_FAST_MM association (with rejected, replayed
    attempts under a given dt); ensembles: per trajectory bit-identical step counts and dL/du0, gradient <= 1e-12, two runs the same bits"""
    okw = {"dt0": kw["dt"]} if kw else {}
    scale = 4.0 if kw else 1.0     # (the rejection case: four times the Glorot weights, a right-hand side the given dt does not fit)
    t = np.arange(0.0, 6.5, 1.0)
    u0, th = node_case(3, 100.0)
    th = th * scale
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 6.0], [], t)
    for j in (0, 2):
        one = U.loss_and_gradient(U.ODEProblem(models.dudt_node(), u0[j], (0.0, 6.0), th), alg(), truth[j:j + 1], row_mask=MASK, saveat=t,
                                  abstol=1e-6, reltol=1e-6, sensealg=FAST(), **kw)
        ref1 = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6, sensealg=4, **okw), u0[j], [0.0, 6.0], th, t, truth[j:j + 1], row_mask=MASK)
        check_per_trajectory(one, ref1)
        if kw:
            assert ref1["stats"][:, 6].sum() > 0      # the case does contain rejected backward steps
        assert_bitwise(one.grad_theta, ref1["grad_theta"], "dL/dtheta, single trajectory")
    for N in (7, 40, 300):
        u0, th = node_case(N, 100.0)
        th = th * scale
        truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 6.0], [], t, nthreads=8)
        ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, 6.0), th), u0)
        r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), **kw)
        ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6, sensealg=4, **okw), u0, [0.0, 6.0], th, t, truth, row_mask=MASK, nthreads=8)
        check_per_trajectory(r, ref)
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
        r2 = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), **kw)
        assert_bitwise(r.grad_theta, r2.grad_theta, "two runs, same bits")
        if N == 40 and not kw:
            full = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6)
            assert np.linalg.norm(r.grad_theta - full.grad_theta) < 1e-3 * np.linalg.norm(full.grad_theta)


def test_seir_fast_block_mode_backward_solves_that_stop_early():
    """maxiters = 40: every forward solve succeeds (35 steps), every backward solve stops at MaxIters.  The block-level kernel must end
    (no hang), report the failures loudly (UDE_ERR_TRAJECTORY, infinite loss) and hand back the oracle's return codes, step counts and
    the lambda each solve had reached.  The gradient of such a call cannot be the sum over the successful members (evaluated stages of a
    stopped solve are already inside the block's accumulators): since round 6 the block refuses it -- NaN in every entry -- instead of
    handing back a sum polluted by partial adjoints (include/udecore.h: UDE_SENSE_INTERPOLATING_ADJOINT_FAST, ude_last_failures)"""
    N = 6
    u0, t, truth, th = _seir_fast_case(N)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
    with pytest.raises(U.UdeError, match="trajector"):
        U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), maxiters=40)
    r = U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), maxiters=40, allow_failures=True)
    ref = O.loss_grad_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=4, maxiters=40), u0, [0.0, 21.0], th, t, truth, row_mask=MASK, nthreads=6)
    assert (ref["retcode"] == 1).all() and (ref["stats"][:, 1] < 40).all()
    assert_bitwise(r.retcode, ref["retcode"], "retcode (MaxIters in the backward solve)")
    assert_bitwise(r.stats[:, 4:7], ref["stats"][:, 4:7], "backward counts of the stopped solves")
    assert_bitwise(r.grad_u0, ref["grad_u0"], "lambda where the solves stopped")
    assert np.isinf(r.loss) and np.isnan(r.grad_theta).all()


def test_seir_fast_block_mode_user_cotangent():
    """the pullback entry point (a user cotangent instead of data) through the block-level kernel"""
    N = 20
    u0, t, truth, th = _seir_fast_case(N)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_(), u0[0], (0.0, 21.0), th), u0)
    cot = np.random.default_rng(3).standard_normal((N, len(t), 7)) * 1e-3
    r = U.adjoint_pullback(ens, U.Vern7(), cot, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST())
    ref = O.vjp_ensemble(O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=4), u0, [0.0, 21.0], th, t, cot, nthreads=8)
    assert_bitwise(r.grad_u0, ref["grad_u0"], "dL/du0")
    assert_bitwise(r.stats[:, 4:8], ref["stats"][:, 4:8], "backward counts")
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])


def test_node_and_kpp_fast_mode_match_oracle():
    u0, th = node_case(3, 100.0)
    t = np.arange(0.0, 6.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 6.0], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, 6.0), th), u0)
    # the wavefront-per-trajectory kernel (lanes_per_traj = 64): the oracle's UDEO_SENSE_FAST association
    r = U.loss_and_gradient(ens, U.Vern7(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=FAST(), ensemblealg=U.EnsembleMI355(lanes_per_traj=64))
    ref = O.loss_grad_ensemble(O.seir_node(), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=2), u0, [0.0, 6.0], th, t, truth, row_mask=MASK, nthreads=3)
    check_per_trajectory(r, ref)
    assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(ref["grad_theta"])
    # Fisher-KPP, 26 points and the 1024-point matrix-core kernels
    for nx, N in ((26, 4), (1024, 2)):
        thk, u0k, tk, truthk = kpp_case(nx, N, models.kpp_chain(), None)
        ensk = U.EnsembleProblem(U.ODEProblem(models.nn_ode(nx), u0k[0], (0.0, 5.0), thk), u0k)
        rk = U.loss_and_gradient(ensk, U.Tsit5(), truthk, saveat=tk, sensealg=FAST())
        refk = O.loss_grad_ensemble(O.kpp_ude(nx), O.opts(O.TSIT5, sensealg=2), u0k, [0.0, 5.0], thk, tk, truthk, nthreads=4)
        check_per_trajectory(rk, refk)
        assert np.linalg.norm(rk.grad_theta - refk["grad_theta"]) < REL_GRAD_SUM * np.linalg.norm(refk["grad_theta"])


def test_fast_mode_with_per_trajectory_grids_is_refused(golden):
    g = golden(S1)
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    u0 = np.stack([X[0], X[0]])
    ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0, tspans=np.array([[0.0, 3.0], [0.0, 3.0]]))
    with pytest.raises(U.UdeError, match="per-trajectory"):
        U.loss_and_gradient(ens, U.Tsit5(), np.repeat(X[None], 2, axis=0), saveat=t, sensealg=FAST())
