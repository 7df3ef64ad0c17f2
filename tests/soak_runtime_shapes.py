#!/usr/bin/env python3
"""Soak of the run-time-shape instances against the CPU oracle: random LV chains of width <= 16 (lane-group instances), random
exposure chains 3-H1-H2-1 (lock-step instances) and random Fisher-KPP reaction chains on grids of 33 .. 199 points, one trajectory each, every gradient entry compared bit for bit, parity and -- exposure
chains -- fast mode.  test_gpu_soak_runtime_shapes.py runs a short one under `-m gpu`; a long one:
    python tests/soak_runtime_shapes.py [rounds] [seed]      (needs a GPU)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O                                                  # noqa: E402
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models              # noqa: E402

ACTS = ["tanh", "rbf", "relu", "identity"]


def soak(n, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        bad += one_round(rng, it)
    return bad


def one_round(rng, it):
    bad = 0
    # ---- LV kind ----
    nh = int(rng.integers(1, 4))
    wmax = int(rng.choice([5, 8, 16]))
    dims = [2] + [int(rng.integers(1, wmax + 1)) for _ in range(nh)] + [2]
    acts = [str(rng.choice(ACTS)) for _ in range(nh)] + ["identity"]
    chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(dims) - 1)])
    f = models.ude_dynamics(chain)
    om = O.make_model(O.KIND_LV_UDE, 2, dims, acts, lin_const=(1.3, -1.8))
    th = 0.3 * chain.glorot_uniform(rng)
    u0 = np.array([[0.44249296, 4.6280594]]) * (1 + 0.2 * rng.uniform(-1, 1, (1, 2)))
    t = np.linspace(0.0, 3.0, 16)
    data = rng.uniform(0.3, 4.0, (1, 16, 2))
    alg, oalg = (U.Tsit5, O.TSIT5) if it % 2 else (U.Vern7, O.VERN7)
    tol = float(10.0 ** rng.uniform(-8, -4))
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1)):
        r = U.loss_and_gradient(U.ODEProblem(f, u0[0], (0.0, 3.0), th), alg(), data, saveat=t, abstol=tol, reltol=tol, sensealg=sense, allow_failures=True)
        ref = O.loss_grad_ensemble(om, O.opts(oalg, tol, tol, sensealg=osense), u0, [0.0, 3.0], th, t, data)
        ok = np.array_equal(r.retcode, ref["retcode"]) and np.array_equal(r.stats[:, :7], ref["stats"][:, :7]) and np.array_equal(r.u, ref["u"], equal_nan=True)
        if (ref["retcode"] == 0).all():
            ok = ok and np.array_equal(r.grad_theta, ref["grad_theta"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
        if not ok:
            bad += 1
            print("MISMATCH lv", dims, acts, alg.__name__, tol, "sense", osense)
    # ---- exposure UDE ----
    while True:
        h1, h2 = int(rng.integers(1, 65)), int(rng.integers(1, 65))
        if not (h1 == 64 and h2 == 64):
            break
    dims = [3, h1, h2, 1]
    acts = ["tanh", "tanh", "identity"]
    chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(3)])
    f = models.dudt_(chain)
    om = O.make_model(O.KIND_SEIR_UDE, 7, dims, acts, consts=O.SEIR_P)
    th = chain.glorot_uniform(rng) * float(rng.choice([0.5, 1.0]))
    S0 = 100.0
    u0 = np.zeros((1, 7))
    u0[0, 0], u0[0, 1], u0[0, 2], u0[0, 4] = rng.uniform(0.8, 0.95) * S0, rng.uniform(0.5, 2.0), rng.uniform(0.2, 1.0), S0
    tf = 4.0
    t = np.arange(0.0, tf + 0.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
    mask = [0, 1, 1, 1, 0, 0, 0]
    served = h1 != 32 and not (h1 == 64 and h2 < 16) and not (h2 in (32, 64) and h1 < 16)   # (udecore.hip: seir_gen_ls_shape)
    for sense, osense in ((None, 0), (U.FastInterpolatingAdjoint(), 4 if served else 2)):   # (the wavefront kernel's fast mode: the oracle's association 2)
        r = U.loss_and_gradient(U.ODEProblem(f, u0[0], (0.0, tf), th), alg(), truth, row_mask=mask, saveat=t, abstol=1e-6, reltol=1e-6, sensealg=sense, allow_failures=True)
        ref = O.loss_grad_ensemble(om, O.opts(oalg, 1e-6, 1e-6, sensealg=osense), u0, [0.0, tf], th, t, truth, row_mask=mask)
        ok = np.array_equal(r.retcode, ref["retcode"]) and np.array_equal(r.stats[:, :7], ref["stats"][:, :7]) and np.array_equal(r.u, ref["u"], equal_nan=True)
        if (ref["retcode"] == 0).all():
            ok = ok and np.array_equal(r.grad_theta, ref["grad_theta"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
        if not ok:
            bad += 1
            print("MISMATCH seir", dims, alg.__name__, "sense", osense)
    # ---- Fisher-KPP on a grid of more than 32 points, edited reaction network ----
    dims = [1] + [int(rng.integers(1, 17)) for _ in range(3)] + [1]
    acts = ["tanh", "tanh", "tanh", "identity"]
    nx = int(rng.integers(33, 200))
    chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(4)])
    f = models.nn_ode(nx, chain)
    om = O.kpp_ude(nx, tuple(dims), tuple(acts))
    th = models.kpp_theta(chain, rng)
    u0 = np.tile(np.clip(models.rho0(26) * (1 + 0.1 * rng.uniform(-1, 1)) + 0.01 * rng.uniform(0, 1, 26), 0, None), 8)[None, :nx]
    t = np.linspace(0.0, 0.6, 4)
    data = rng.uniform(0.0, 1.0, (1, len(t), nx))
    for sense, osense in ((None, 0), (U.ForwardDiffSensitivity(), 1), (U.InterpolatingAdjoint(checkpointing=True), 0)):
        r = U.loss_and_gradient(U.ODEProblem(f, u0[0], (0.0, 0.6), th), U.Tsit5(), data, saveat=t, sensealg=sense, allow_failures=True)
        ref = O.loss_grad_ensemble(om, O.opts(O.TSIT5, sensealg=osense), u0, [0.0, 0.6], th, t, data)
        ok = np.array_equal(r.retcode, ref["retcode"]) and np.array_equal(r.stats[:, [0, 1, 2, 4, 5, 6]], ref["stats"][:, [0, 1, 2, 4, 5, 6]]) and np.array_equal(r.u, ref["u"], equal_nan=True)
        if (ref["retcode"] == 0).all():
            ok = ok and np.array_equal(r.grad_theta, ref["grad_theta"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
        if not ok:
            bad += 1
            print("MISMATCH kpp", dims, nx, "sense", type(sense).__name__)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    bad = soak(n, int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    print("soak: %d rounds, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)
