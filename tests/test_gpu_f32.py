"""Float32 problems on the GPU (ude_model_desc.dtype = 1): LotkaVolterra/scenario_3.jl:26-57 (true Fisher-KPP, Tsit5),
scenario_3.jl:83-134 (its UDE: Vern7, ForwardDiffSensitivity) and hudson_bay.jl:77-123 (LV UDE, tanh third layer).
The kernels are the Float64 ones compiled with real = float; the bar is the same: per trajectory bit-identical to the
oracle's Float32 instantiation (step counts, saved states, dL/du0), the golden known answers of the reference's Float32
artifacts reproduced on the device, ensemble-summed gradients to Float32 summation order."""
import numpy as np
import pytest

import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models

pytestmark = pytest.mark.gpu
S3 = "Scenario_3_recovery_0.005"
HB = "Hudson_Bay_recovery"
f32 = np.float32


def test_true_fisher_kpp_f32_golden_destats_on_device(golden):
    """scenario_3.jl:56-57: Tsit5, Float32, Nx = 26, saveat 0.5 -- golden DEStats 243 / 39 / 1"""
    s = golden(S3)["solution"]
    prob = U.ODEProblem(models.rc_ode(26, 0.01, 1.0, 0.04, dtype="float32"), np.array(s["u0"], dtype=f32), tuple(s["tspan"]), [])
    sol = U.solve(prob, U.Tsit5(), saveat=np.array(s["t"], dtype=f32))
    out, st, rc = O.solve_ensemble(O.kpp_true(26, 0.01, 1.0, 0.04, dtype=1), O.opts(O.TSIT5), s["u0"], s["tspan"], [], s["t"], dtype=f32)
    assert sol.retcode == "Success" and np.asarray(sol).dtype == f32
    assert np.array_equal(np.asarray(sol).T, out[0])                       # device == oracle, bit for bit
    d = sol.destats
    assert (d.nf, d.naccept, d.nreject) == tuple(st[0][:3])
    # the reference's artifact: DEStats reproduced exactly on the device (the solve runs at Tsit5's stability limit; which
    # Float32 arithmetic shape gives the stored counts: tests/test_oracle_golden.py::test_kpp_true_f32,
    # profiles/r03_f32_golden_search.md), states to Float32 rounding before that phase and to 2e-4 after it
    assert (d.nf, d.naccept, d.nreject) == (s["destats"]["nf"], s["destats"]["naccept"], s["destats"]["nreject"]) == (243, 39, 1)
    Ug = np.array(s["u"], dtype=f32)
    assert np.abs(np.asarray(sol).T[:2] - Ug[:2]).max() < 1e-6 and np.abs(np.asarray(sol).T - Ug).max() < 2e-4   # (the reference's two artifacts of this solve differ by 7.1e-4)


def test_scenario3_ude_f32_loss_known_answer_and_gradients(golden):
    """scenario_3.jl:121-134: objective at theta_init = losses[0] = 2967.0867 (Float32), on the device; gradient by the
    discrete sweep (the script's ForwardDiffSensitivity) and by the interpolating adjoint against the oracle's Float32 runs"""
    g = golden(S3)
    X = np.array(g["X"]["data_colmajor"], dtype=f32).reshape(11, 26)
    t = np.array(g["t"], dtype=f32)
    th = np.array(g["initial_parameters"], dtype=f32)
    f = models.nn_ode(26, models.kpp_s3_chain(), dtype="float32")
    prob = U.ODEProblem(f, X[0], (float(t[0]), float(t[-1])), th)
    for sense, osense in ((U.ForwardDiffSensitivity(), 1), (None, 0)):
        r = U.loss_and_gradient(prob, U.Vern7(), X[None], saveat=t, sensealg=sense)
        ref = O.loss_grad_ensemble(O.kpp_ude_s3(1), O.opts(O.VERN7, sensealg=osense), X[0], [t[0], t[-1]], th, t, X[None], dtype=f32)
        assert r.u.dtype == f32 and r.grad_theta.dtype == f32
        assert np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.u, ref["u"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
        assert abs(float(r.loss) - float(ref["loss"])) < 1e-5 * float(ref["loss"])     # (per-lane partial sums of the loss)
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < 2e-5 * np.linalg.norm(ref["grad_theta"])
    loss = float(r.loss) + abs(float(th[-5:-2].sum()))                               # + the script's penalty term
    gold = g["losses"]["data_colmajor"][0]
    assert abs(loss - gold) < 2e-5 * gold                                            # 2967.0867


@pytest.mark.parametrize("name", [S3, "Scenario_3_recovery_0.025"])
def test_scenario3_f32_adam_trajectory_follows_the_stored_losses(golden, name):
    """scenario_3.jl:148-154: `Optimization.solve(optprob, ADAM(0.1), maxiters = 10)` on the Float32 Fisher-KPP UDE -- objective =
    sum(abs2, pred .- Xn) + abs(sum(p[end-4:end-2])) -- with the Float32 kernels' gradient (the script's ForwardDiffSensitivity as the
    frozen-step sweep, and the interpolating adjoint): all ten stored losses of BOTH artifacts (noise 0.005 and 0.025) to Float32
    accuracy.  Optimisers' ADAM keeps its moments in the parameters' type (Float32) and computes with the Float64 eta / betas."""
    g = golden(name)
    X = np.array(g["X"]["data_colmajor"], dtype=f32).reshape(11, 26)
    t = np.array(g["t"], dtype=f32)
    gold = np.array(g["losses"]["data_colmajor"])
    f = models.nn_ode(26, models.kpp_s3_chain(), dtype="float32")
    for sense in (U.ForwardDiffSensitivity(), None):
        th = np.array(g["initial_parameters"], dtype=f32)
        mt, vt, b1t, b2t = np.zeros_like(th), np.zeros_like(th), 0.9, 0.999
        for k in range(10):
            r = U.loss_and_gradient(U.ODEProblem(f, X[0], (float(t[0]), float(t[-1])), th), U.Vern7(), X[None], saveat=t, sensealg=sense)
            sw = float(th[-5:-2].astype(np.float64).sum())
            loss = float(r.loss) + abs(sw)
            assert abs(loss - gold[k]) < 5e-5 * gold[k], (name, type(sense).__name__, k, loss, gold[k])
            gr = r.grad_theta.astype(np.float64)
            gr[-5:-2] += np.sign(sw)
            mt = (0.9 * mt.astype(np.float64) + 0.1 * gr).astype(f32)
            vt = (0.999 * vt.astype(np.float64) + 0.001 * gr * gr).astype(f32)
            step = 0.1 * (mt.astype(np.float64) / (1 - b1t)) / (np.sqrt(vt.astype(np.float64) / (1 - b2t)) + np.finfo(np.float64).eps)
            th = (th.astype(np.float64) - step).astype(f32)
            b1t *= 0.9
            b2t *= 0.999


def test_hudson_bay_f32_trained_loss_and_gradient(golden):
    """hudson_bay.jl:98-123 in Float32: loss(trained_parameters) = losses[end] = 0.00357905; gradient vs the oracle"""
    g = golden(HB)
    X = np.array(g["X"]["data_colmajor"], dtype=f32).reshape(21, 2)
    t = np.array(g["t"], dtype=f32)
    th = np.array(g["trained_parameters"], dtype=f32)
    f = models.ude_dynamics(models.hudson_chain(), trainable="both", dtype="float32")
    u0 = np.stack([X[0], X[0] * f32(1.05), X[0] * f32(0.9)])
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (float(t[0]), float(t[-1])), th), u0)
    data = np.repeat(X[None], 3, axis=0)
    for sense, osense in ((U.ForwardDiffSensitivity(), 1), (None, 0)):
        r = U.loss_and_gradient(ens, U.Vern7(), data, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=sense)
        ref = O.loss_grad_ensemble(O.lv_ude_hudson(1), O.opts(O.VERN7, 1e-6, 1e-6, sensealg=osense), u0, [t[0], t[-1]], th, t, data,
                                   dtype=f32, nthreads=3)
        assert np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.u, ref["u"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
        assert np.array_equal(r.loss_per_traj, ref["loss_per_traj"])
        assert np.linalg.norm(r.grad_theta - ref["grad_theta"]) < 2e-6 * np.linalg.norm(ref["grad_theta"])
    loss = float(r.loss_per_traj[0]) / 21 + 1e-3 * float(np.sum(th[2:].astype(np.float64) ** 2)) / len(th[2:])
    gold = g["losses"]["data_colmajor"][-1]
    assert abs(loss - gold) < 3e-5 * gold                                            # 0.00357905
    # the right-hand side itself
    du = U.rhs(f, X[:5], th)
    ref_du = np.array([O.rhs(O.lv_ude_hudson(1), th, X[i], dtype=f32) for i in range(5)])
    assert du.dtype == f32 and np.array_equal(du, ref_du)


def test_f32_descriptor_outside_the_fallback_is_refused():
    f = models.nn_ode(26, models.kpp_chain(), dtype="float32")          # Fisher-KPP's tanh chain has no Float32 instance (scenario_3's has)
    with pytest.raises(U.UdeError, match="no kernel for model"):
        U.solve(U.ODEProblem(f, models.rho0(26).astype(f32), (0.0, 1.0), np.zeros(f.n_param, dtype=f32)), U.Tsit5(), saveat=0.5)


@pytest.mark.parametrize("seed", range(6))
def test_f32_lv_kind_any_chain_matches_oracle(golden, seed):
    """round 4: the runtime-shape kernel in Float32 -- hudson_bay.jl:77-79 is `FastChain(...)`, a script variable, and its problem is
    Float32 (`:85-104`): any chain of <= 8 Dense layers of width <= 64, interpolating adjoint and the discrete sweep the script requests"""
    from test_gpu_generic import chain_of, random_chain, theta_for
    rng = np.random.default_rng(500 + seed)
    dims, acts = random_chain(rng, 2, 2, max_hidden=4)
    if seed == 0:
        dims, acts = [2, 5, 5, 5, 2], ["rbf", "rbf", "rbf", "identity"]   # scenario_1's chain as a Float32 problem (no compiled instance)
    chain = chain_of(dims, acts)
    trainable = [None, "both"][seed % 2]
    f = models.ude_dynamics(chain, trainable=trainable, dtype="float32")
    om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, nn_offset={None: 0, "both": 2}[trainable], lin_idx={None: (-1, -1), "both": (0, 1)}[trainable],
                      lin_sign=(1.0, -1.0) if trainable else (1.0, 1.0), lin_const={None: (1.3, -1.8), "both": (0.0, 0.0)}[trainable], dtype=1)
    th = np.concatenate([[1.3, 1.8] if trainable else [], theta_for(chain, rng, 0.3)]).astype(f32)
    g = golden(HB)
    X = np.array(g["X"]["data_colmajor"], dtype=f32).reshape(21, 2)[:13]
    t = np.linspace(0.0, 3.0, 13).astype(f32)      # (an untrained random chain blows up over the script's 20 years: a short horizon)
    N = 1 if seed < 3 else 4
    u0 = (X[0][None, :] * (1 + 0.1 * rng.uniform(-1, 1, (N, 2)))).astype(f32)
    data = np.repeat(X[None], N, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (float(t[0]), float(t[-1])), th), u0)
    alg, oalg = (U.Vern7, O.VERN7) if seed % 3 else (U.Tsit5, O.TSIT5)
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
        r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-5, reltol=1e-5, sensealg=sense)
        ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-5, 1e-5, sensealg=osense), u0, [t[0], t[-1]], th, t, data, dtype=f32, nthreads=4)
        what = "%s %s sense %d" % (dims, acts, osense)
        assert (r.retcode == 0).all() and r.u.dtype == f32, what
        assert np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.u, ref["u"]) and np.array_equal(r.grad_u0, ref["grad_u0"]), what
        assert np.array_equal(r.loss_per_traj, ref["loss_per_traj"]), what
        if N == 1:
            assert np.array_equal(r.grad_theta, ref["grad_theta"]), what
        else:
            assert np.linalg.norm(r.grad_theta.astype(float) - ref["grad_theta"].astype(float)) < 2e-6 * np.linalg.norm(ref["grad_theta"].astype(float)), what


@pytest.mark.parametrize("case", range(5))
def test_f32_lv_kind_edited_network_on_the_lane_group_kernels(golden, case):
    """round 5: hudson_bay.jl:77-104 with an EDITED FastChain (two / three hidden layers of width <= 8) as a Float32 problem: the
    run-time-shape instances of the lane-group kernels (NetCfgRt, five or eight lanes per trajectory) on `real = float`; per trajectory
    the Float32 oracle's bits, a single trajectory: every gradient entry; the same bits as the wavefront-per-trajectory Float32 kernel"""
    from test_gpu_generic import chain_of, theta_for
    dims, acts, trainable = [([2, 8, 8, 8, 2], ["tanh", "tanh", "tanh", "identity"], "both"), ([2, 5, 5, 5, 2], ["tanh", "tanh", "tanh", "identity"], None),
                             ([2, 6, 4, 2], ["rbf", "tanh", "identity"], "both"), ([2, 3, 5, 2], ["relu", "rbf", "identity"], None),
                             ([2, 7, 1, 8, 2], ["rbf", "identity", "tanh", "identity"], "both")][case]
    rng = np.random.default_rng(700 + case)
    chain = chain_of(dims, acts)
    f = models.ude_dynamics(chain, trainable=trainable, dtype="float32")
    om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, nn_offset={None: 0, "both": 2}[trainable], lin_idx={None: (-1, -1), "both": (0, 1)}[trainable],
                      lin_sign=(1.0, -1.0) if trainable else (1.0, 1.0), lin_const={None: (1.3, -1.8), "both": (0.0, 0.0)}[trainable], dtype=1)
    th = np.concatenate([[1.3, 1.8] if trainable else [], theta_for(chain, rng, 0.3)]).astype(f32)
    g = golden(HB)
    X = np.array(g["X"]["data_colmajor"], dtype=f32).reshape(21, 2)[:13]
    t = np.linspace(0.0, 3.0, 13).astype(f32)
    alg, oalg = (U.Vern7, O.VERN7) if case % 2 else (U.Tsit5, O.TSIT5)
    for N in (1, 9):
        u0 = (X[0][None, :] * (1 + 0.1 * rng.uniform(-1, 1, (N, 2)))).astype(f32)
        data = np.repeat(X[None], N, axis=0)
        ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (float(t[0]), float(t[-1])), th), u0)
        for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
            r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-5, reltol=1e-5, sensealg=sense)
            ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-5, 1e-5, sensealg=osense), u0, [t[0], t[-1]], th, t, data, dtype=f32, nthreads=4)
            what = "%s %s N %d sense %d" % (dims, acts, N, osense)
            assert (r.retcode == 0).all() and r.u.dtype == f32, what
            assert np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.u, ref["u"]) and np.array_equal(r.grad_u0, ref["grad_u0"]), what
            assert np.array_equal(r.loss_per_traj, ref["loss_per_traj"]), what
            if N == 1:
                assert np.array_equal(r.grad_theta, ref["grad_theta"]), what
            else:
                assert np.linalg.norm(r.grad_theta.astype(float) - ref["grad_theta"].astype(float)) < 2e-6 * np.linalg.norm(ref["grad_theta"].astype(float)), what
            if sense is None:
                w64 = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-5, reltol=1e-5, ensemblealg=U.EnsembleMI355(lanes_per_traj=64))
                assert np.array_equal(r.stats, w64.stats) and np.array_equal(r.grad_u0, w64.grad_u0), what
                if N == 1:
                    assert np.array_equal(r.grad_theta, w64.grad_theta), what


def test_device_resident_f32_ensemble_matches_host_buffer_path(golden):
    """DeviceEnsemble on float32 CUDA tensors (the `_dev` entry points with dtype = 1) == the host-buffer entry points"""
    import torch
    g = golden(S3)
    X = np.array(g["X"]["data_colmajor"], dtype=f32).reshape(11, 26)
    t = np.array(g["t"], dtype=f32)
    th = np.array(g["initial_parameters"], dtype=f32)
    f = models.nn_ode(26, models.kpp_s3_chain(), dtype="float32")
    u0 = np.stack([X[0], X[0] * f32(0.97), X[0] * f32(1.02)])
    data = np.repeat(X[None], 3, axis=0)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (float(t[0]), float(t[-1])), th), u0)
    ref = U.loss_and_gradient(ens, U.Vern7(), data, saveat=t)
    dev = torch.device("cuda:0")
    de = U.DeviceEnsemble(f, U.Vern7(), (float(t[0]), float(t[-1])), t, torch.tensor(u0, device=dev), data=torch.tensor(data, device=dev))
    gd = de.loss_grad(torch.tensor(th, device=dev))
    torch.cuda.synchronize()
    assert gd.dtype == torch.float32 and de.u.dtype == torch.float32
    assert np.array_equal(gd[:-1].cpu().numpy(), ref.grad_theta) and np.array_equal(de.u.cpu().numpy(), ref.u)
    assert float(gd[-1]) == float(ref.loss) and np.array_equal(de.grad_u0.cpu().numpy(), ref.grad_u0)
    with pytest.raises(AssertionError):
        U.DeviceEnsemble(f, U.Vern7(), (0.0, 5.0), t, torch.tensor(u0.astype(np.float64), device=dev))
