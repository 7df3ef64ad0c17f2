#!/usr/bin/env python3
"""LotkaVolterra/scenario_1.jl, lines 29-126, against the MI355X core: same data, same network, same solver settings,
ADAM(0.1) x 200 then BFGS; the noisy data X and the initial parameters come from the reference's own artifact
(tests/golden/Scenario_1_recovery_0.005.json) because Julia's RNG stream cannot be reproduced.
Needs a GPU:  python examples/scenario_1.py [adam_iters] [bfgs_iters]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models, training     # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "Scenario_1_recovery_0.005.json")))
Xn = np.array(g["X"]["data_colmajor"]).reshape(31, 2)                 # Xₙ (2 x 31 in Julia)
t = np.array(g["solution"]["t"])
p0 = np.array(g["initial_parameters"])                                # Lux.setup(rng, U)
prob_nn = U.ODEProblem(models.ude_dynamics(), Xn[0], (t[0], t[-1]), p0)          # scenario_1.jl:78


def predict(theta, X=Xn[0], T=t):                                                  # scenario_1.jl:82-88
    _prob = U.remake(prob_nn, u0=X, tspan=(T[0], T[-1]), p=theta)
    return np.asarray(U.solve(_prob, U.Vern7(), saveat=T, abstol=1e-6, reltol=1e-6))


def loss_grad(theta):                                                               # scenario_1.jl:91-94 + gradient
    r = U.loss_and_gradient(U.remake(prob_nn, p=np.asarray(theta)), U.Vern7(), Xn[None], saveat=t, abstol=1e-6, reltol=1e-6)
    return r.loss, r.grad_theta


losses = []


def callback(p, l):                                                                 # scenario_1.jl:99-105
    losses.append(l)
    if len(losses) % 50 == 0:
        print("Current loss after %d iterations: %g" % (len(losses), losses[-1]))
    return False


n_adam = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_bfgs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
p1, _ = training.adam(loss_grad, p0, eta=0.1, maxiters=n_adam, callback=callback, result="evaluated")   # (res1.u: the last evaluated parameters)
print("Training loss after %d iterations: %g" % (len(losses), losses[-1]))
p2, _ = training.bfgs_hagerzhang(loss_grad, p1, initial_stepnorm=0.01, maxiters=n_bfgs, callback=callback)   # Optim.BFGS + HagerZhang
print("Final training loss after %d iterations: %g" % (len(losses), losses[-1]))
gold = g["losses"]["data_colmajor"]
print("reference artifact: losses[0..3] = %s ; ours = %s" % (gold[:4], losses[:4]))
print("reference final loss %g after %d iterations" % (gold[-1], len(gold)))

# Save the results (scenario_1.jl:210-213) in the reference's own file format: the arrays of its `save(...)` call
from universal_differential_equations_amd import io                                   # noqa: E402
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "Scenario_1_recovery_0.005_mi355x.jld2")
io.save_jld2(out, X=Xn, t=t, initial_parameters=p0, trained_parameters=np.asarray(p2), losses=np.asarray(losses))
print("saved", out)
