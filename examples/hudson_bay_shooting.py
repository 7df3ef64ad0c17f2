#!/usr/bin/env python3
"""LotkaVolterra/hudson_bay.jl, lines 62-149: hare / lynx pelts 1900-1920 (normalised, X and t from the reference's
artifact), UDE with trainable linear rates theta = [p1, p2, FastChain(2-5-5-5-2: rbf, rbf, tanh, linear)], trained with
MULTIPLE SHOOTING (group_size 5, continuity_term 200, hudson_bay.jl:106-118) and ADAM(0.1); the callback reports the
script's equivalent L2 loss (hudson_bay.jl:120-123).  The five groups are one 5-trajectory ensemble per loss evaluation.
Needs a GPU:  python examples/hudson_bay_shooting.py [adam_iters]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models, training     # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "Hudson_Bay_recovery.json")))
t = np.array(g["t"], dtype=float)
Xn = np.array(g["X"]["data_colmajor"], dtype=float).reshape(len(t), 2).T          # 2 x 21
chain = models.hudson_chain()
f = models.ude_dynamics(chain, trainable="both")                                    # hudson_bay.jl:82-91
rng = np.random.default_rng(1234)
p = np.concatenate([rng.uniform(0, 1, 2), chain.glorot_uniform(rng)])               # hudson_bay.jl:80 (rand(2); initial_params(U))
prob_nn = U.ODEProblem(f, Xn[:, 0], (t[0], t[-1]), p)
backend = training.EngineBackend(prob_nn, U.Vern7(), abstol=1e-6, reltol=1e-6)      # hudson_bay.jl:97-103


def shooting_loss_grad(theta):                                                      # hudson_bay.jl:114-117
    l, grad, _ = training.multiple_shoot(theta, Xn, t, backend, 5, continuity_term=200.0)
    return l, grad


def l2_loss(theta):                                                                 # hudson_bay.jl:120-123
    Xh = np.asarray(U.solve(U.remake(prob_nn, u0=Xn[:, 0], tspan=(0.0, t[-1] - t[0]), p=theta), U.Vern7(),
                            saveat=t - t[0], abstol=1e-6, reltol=1e-6))
    return np.sum((Xn - Xh) ** 2) / Xn.shape[1] + 1e-3 * np.sum(theta[2:] ** 2) / (theta.size - 2)


hist = []


def callback(theta, l):
    hist.append(l2_loss(np.asarray(theta)))
    if len(hist) % 25 == 0:
        print("iteration %d: shooting loss %.5g, L2 loss %.5g" % (len(hist), l, hist[-1]))
    return False


n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
theta, sl = training.adam(shooting_loss_grad, p, eta=0.1, maxiters=n, callback=callback)
print("shooting loss %.5g -> %.5g; equivalent L2 loss %.5g -> %.5g" % (sl[0], sl[-1], hist[0], hist[-1]))
