#!/usr/bin/env python3
"""FisherKPP/Fisher-KPP-CNN.jl, lines 16-143 and 236-238: 1-D reaction-diffusion, reaction term = pointwise network
1-10-20-10-1 tanh, diffusion = 3-tap periodic stencil (1.1, -2.5, 1.0) times D0 = 6.5; Tsit5 at default tolerances,
InterpolatingAdjoint, loss = sum(abs2, ode_data - pred) + 100*abs(w1+w2+w3) (penalty on the host), ADAM(0.001).
Needs a GPU:  python examples/fisher_kpp.py [adam_iters]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models, training     # noqa: E402

D, r, X, T, dx = 0.01, 1.0, 1.0, 5.0, 0.04                            # Fisher-KPP-CNN.jl:16-21
dt = T / 10
Nx = int(X / dx + 1)
rho0 = models.rho0(Nx, dx)                                            # IC-1, Fisher-KPP-CNN.jl:27-31
prob = U.ODEProblem(models.rc_ode(Nx, D, r, dx), rho0, (0.0, T), [], saveat=dt)    # Fisher-KPP-CNN.jl:65
sol = U.solve(prob, U.Tsit5())                                                       # Fisher-KPP-CNN.jl:66
ode_data = np.asarray(sol)                                                           # Nx x 11

rx_nn = models.kpp_chain()                                                           # Fisher-KPP-CNN.jl:92-96
rng = np.random.default_rng(0)
p = models.kpp_theta(rx_nn, rng)                                                     # [p1; p2; D0], Fisher-KPP-CNN.jl:98-109
f = models.nn_ode(Nx, rx_nn)
prob_nn = U.ODEProblem(f, rho0, (0.0, T), p)                                         # Fisher-KPP-CNN.jl:131


def loss_rd(theta):                                                                  # Fisher-KPP-CNN.jl:134-143
    theta = np.asarray(theta)
    res = U.loss_and_gradient(U.remake(prob_nn, p=theta), U.Tsit5(), ode_data.T[None], saveat=dt,
                              sensealg=U.InterpolatingAdjoint(autojacvec=U.ReverseDiffVJP()))
    w = theta[f.stencil_offset:f.stencil_offset + 3]
    grad = res.grad_theta.copy()
    grad[f.stencil_offset:f.stencil_offset + 3] += 100.0 * np.sign(w.sum())
    return res.loss + 100.0 * abs(w.sum()), grad


seen = []   # the callback's own iteration count (Fisher-KPP-CNN.jl:163-233 prints every 100th loss)


def cb(th, l):
    if len(seen) % 100 == 0:
        print("iter %d loss %g" % (len(seen), l))
    seen.append(l)
    return False


n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pstar, hist = training.adam(loss_rd, p, eta=1e-3, maxiters=n, callback=cb)
print("loss %g -> %g ; D0 = %g, stencil = %s" % (hist[0], hist[-1], pstar[f.d0_offset], pstar[f.stencil_offset:f.stencil_offset + 3]))

# `@save @sprintf("%s/model.bson", save_folder) pstar`  (Fisher-KPP-CNN.jl:243)
from universal_differential_equations_amd import io                                   # noqa: E402
io.save_bson(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model.bson"), pstar=np.asarray(pstar))
