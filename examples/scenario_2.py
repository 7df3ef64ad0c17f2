#!/usr/bin/env python3
"""LotkaVolterra/scenario_2.jl, lines 57-148: partial observations, 5 shooting segments sharing theta = [delta; ude].

The five `predict(theta, [XS[i,1], YS[i,1]], TS[i,:])` calls of the loss (scenario_2.jl:113-124) are ONE 5-trajectory
ensemble here, every member on its own tspan = (T[1], T[end]) and save grid T (per-trajectory time grids of the boundary).  The loss
mixes abs2 on x with abs on the last y, so the gradient goes through the generic pullback (user cotangent) instead of the
built-in sum-of-squares loss; the regulariser stays on the host.  Data X, t and the initial theta come from the
reference's artifact (tests/golden/Scenario_2_recovery_0.005.json).   Needs a GPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models, training     # noqa: E402


def build(golden_path=os.path.join(ROOT, "tests", "golden", "Scenario_2_recovery_0.005.json")):
    g = json.load(open(golden_path))
    X = np.array(g["X"]["data_colmajor"]).reshape(61, 2)
    t = np.array(g["t"])
    ty = np.arange(t[0], t[-1] + 1e-9, 6 / 5)                       # scenario_2.jl:59
    segs = []
    for i in range(len(ty) - 1):                                     # scenario_2.jl:66-71
        idx = np.nonzero((ty[i] - 1e-9 <= t) & (t <= ty[i + 1] + 1e-9))[0]
        segs.append(idx)
    XS = np.array([X[idx, 0] for idx in segs])                       # 5 x 13
    TS = np.array([t[idx] for idx in segs])
    YS = np.array([[X[idx[0], 1], X[idx[-1], 1]] for idx in segs])   # y only at the segment ends
    return g, XS, TS, YS


def make_loss(XS, TS, YS, sensealg=None):
    f = models.ude_dynamics(trainable="delta")                       # scenario_2.jl:87-95
    u0s = np.stack([XS[:, 0], YS[:, 0]], axis=1)
    tspans = np.stack([TS[:, 0], TS[:, -1]], axis=1)                 # remake(prob; tspan = (T[1], T[end])), scenario_2.jl:105
    tau = TS                                                         # saveat = T per segment, scenario_2.jl:107
    nseg, npts = XS.shape

    def loss_grad(theta):
        theta = np.asarray(theta, dtype=float)
        # the 5 segments as ONE ensemble, every member on its own tspan / save grid (no time shift)
        ens = U.EnsembleProblem(U.ODEProblem(f, u0s[0], tuple(tspans[0]), theta), u0s, tspans=tspans)
        sol = U.solve(ens, U.Vern7(), saveat=tau, abstol=1e-6, reltol=1e-6)
        Xh = sol.u                                                   # (5, 13, 2)
        reg = 1e-3 * np.sum(theta[1:] ** 2) / (theta.size - 1)       # scenario_2.jl:115
        l = reg + np.sum((XS - Xh[:, :, 0]) ** 2) + np.sum(np.abs(YS[:, 1] - Xh[:, -1, 1]))
        cot = np.zeros_like(Xh)
        cot[:, :, 0] = 2.0 * (Xh[:, :, 0] - XS)
        cot[:, -1, 1] = np.sign(Xh[:, -1, 1] - YS[:, 1])
        r = U.adjoint_pullback(ens, U.Vern7(), cot, saveat=tau, abstol=1e-6, reltol=1e-6, sensealg=sensealg)
        grad = r.grad_theta.copy()
        grad[1:] += 2e-3 * theta[1:] / (theta.size - 1)
        return l, grad

    return loss_grad


if __name__ == "__main__":
    g, XS, TS, YS = build()
    loss_grad = make_loss(XS, TS, YS)
    p0 = np.array(g["initial_parameters"])
    l0, _ = loss_grad(p0)
    gold = g["losses"]["data_colmajor"]
    print("loss(theta_init) = %.15g   (reference artifact losses[0] = %.15g)" % (l0, gold[0]))
    n_adam = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    p1, hist = training.adam(loss_grad, p0, eta=0.1, maxiters=n_adam)
    print("after %d ADAM(0.1) iterations: %g   (reference losses[%d] = %g)" % (n_adam, hist[-1], n_adam - 1, gold[n_adam - 1]))
