#!/usr/bin/env python3
"""SEIR_exposure/seir_exposure.jl, lines 16-38 and 111-161 ("Universal ODE Part 1"): the exposure term of a 7-state
SEIR model replaced by ann = FastChain(FastDense(3,64,tanh), FastDense(64,64,tanh), FastDense(64,1)); Vern7 at 1e-6,
InterpolatingAdjoint, loss on rows 2:4, ADAM(0.01).  Data = corona! solved at 1e-12 by the same engine (+1e-5 noise).
Needs a GPU:  python examples/seir_exposure.py [adam_iters] [fast]
`fast` (an opt-in, not the script's sensealg): U.FastInterpolatingAdjoint() -- lambda-only error control; for this model the parameter
cotangent is then accumulated on the matrix cores per block (csrc/ude_seir_lsf.h), no mu column in HBM."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models, training     # noqa: E402

S0 = 14e6
u0 = np.array([0.9 * S0, 0.0, 0.0, 0.0, S0, 0.0, 0.0])                # seir_exposure.jl:31-32
tspan = (0.0, 21.0)
prob = U.ODEProblem(models.corona(), u0, tspan, [])
solution = U.solve(prob, U.Vern7(), abstol=1e-12, reltol=1e-12, saveat=1)          # seir_exposure.jl:36-37
tsdata = np.asarray(solution)                                                       # 7 x 22
rng = np.random.default_rng(0)
noisy_data = tsdata + 1e-5 * rng.standard_normal(tsdata.shape)                      # seir_exposure.jl:46

ann = models.seir_chain()                                                           # seir_exposure.jl:114
p = ann.glorot_uniform(rng)
prob_nn = U.ODEProblem(models.dudt_(ann), u0, tspan, p)                             # seir_exposure.jl:117-131


SENSEALG = U.FastInterpolatingAdjoint() if "fast" in sys.argv[2:] else U.InterpolatingAdjoint(autojacvec=U.ReverseDiffVJP())


def loss_grad(theta):                                                               # seir_exposure.jl:137-147
    r = U.loss_and_gradient(U.remake(prob_nn, p=np.asarray(theta)), U.Vern7(), noisy_data.T[None], row_mask=[0, 1, 1, 1, 0, 0, 0],
                            saveat=solution.t, abstol=1e-6, reltol=1e-6, sensealg=SENSEALG)
    return r.loss, r.grad_theta


losses = []


def callback(theta, l):                                                             # seir_exposure.jl:151-158
    losses.append(l)
    if len(losses) % 50 == 0:
        print(losses[-1])
    return False


n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
res1, _ = training.adam(loss_grad, p, eta=0.01, maxiters=n, callback=callback)      # seir_exposure.jl:160
print("loss %g -> %g after %d ADAM(0.01) iterations" % (losses[0], losses[-1], len(losses)))
