#!/usr/bin/env python3
"""highdim_pde/lambaem.jl, the script's own call:

    ans = solve(prob, pdealg, verbose = true, maxiters = 500, trajectories = m,
                alg = LambaEM(), pabstol = 1f-2, reltol = 1e-4, abstol = 1e-4)            (lambaem.jl:33-34)

with d = 100, x0 = 0, tspan = (0, 1), m = 100, lambda = 1, hls = 110, Flux.ADAM(0.03) (lambaem.jl:8-31), then the script's
comparison with the Monte-Carlo reference value and its gate `error_l2 < 0.2` (lambaem.jl:36-48).  Random numbers are
Philox streams (Julia's MersenneTwister cannot be reproduced), the estimator is the restatement of oracle/sde_oracle.h.
Needs a GPU:  python examples/highdim_pde_lambaem.py [maxiters] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from universal_differential_equations_amd import pde     # noqa: E402

d = 100                                                   # lambaem.jl:8
x0 = np.zeros(d, dtype=np.float32)                        # :9
tspan = (0.0, 1.0)                                        # :10
m = 100                                                   # :11
lam = 1.0                                                 # :12
prob = pde.TerminalPDEProblem(pde.hjb(lam), x0, tspan)    # :14-18
hls = 10 + d                                              # :20
pdealg = pde.NNPDENS(d, hls, opt=pde.ADAM(0.03))          # :21-31
maxiters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
theta0 = pdealg.init_params(np.random.default_rng(0))     # Flux.Dense defaults: glorot_uniform weights, zero biases

hist = []
t0 = time.perf_counter()


def cb(it, loss, u0):
    hist.append((it, loss, u0, time.perf_counter() - t0))
    if it % 25 == 0:
        print("iteration %4d  loss %-10.4g u0(x0) %.4f   %.1f s" % (it, loss, u0, hist[-1][3]), flush=True)
    return False


ans, theta, losses = pde.solve(prob, pdealg, theta0, verbose=False, maxiters=maxiters, trajectories=m, alg=pde.LambaEM(),
                               pabstol=1e-2, reltol=1e-4, abstol=1e-4, seed=0, callback=cb)
elapsed = time.perf_counter() - t0
analytical_ans = pde.u_analytical(x0, lam, tspan[1], np.random.default_rng(1))    # lambaem.jl:36-41 (MC = 10^5)
error_l2 = float(np.sqrt((ans - analytical_ans) ** 2 / ans ** 2))                 # :43
print("Hamilton Jacobi Bellman Equation")
print("error_l2 = ", error_l2)
rec = {"script": "highdim_pde/lambaem.jl:33-34 (maxiters = %d, trajectories = %d, LambaEM, abstol = reltol = 1e-4, pabstol = 1e-2)" % (maxiters, m),
       "numerical": ans, "analytical": float(analytical_ans), "error_l2": error_l2, "gate_error_l2_lt_0.2": bool(error_l2 < 0.2),
       "iterations_run": len(losses), "loss_first": losses[0], "loss_last": losses[-1], "elapsed_s": elapsed,
       "u0_every_50": [round(h[2], 4) for h in hist[::50]]}
print(json.dumps(rec))
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        json.dump(rec, fh, indent=1)
