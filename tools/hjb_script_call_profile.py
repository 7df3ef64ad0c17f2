#!/usr/bin/env python3
"""Where the time of highdim_pde/lambaem.jl's own call goes, iteration by iteration (round-4 review: bench.py's `hjb_script_tol` leg
-- ONE loss + gradient at theta_init, 100 trajectories, abstol = reltol = 1e-4 -- takes 2.2 s, while the 500-iteration training of
examples/highdim_pde_lambaem.py averaged 86 ms per iteration: the gap has to be explained by data, not by prose).

Runs the script's training (pde.solve's loop: Flux.ADAM(0.03), fresh Philox noise per iteration) and records, at iterations
0, 1, 2, 5, 10, 25, 50, 100, ..., 450 and the last one: accepted steps per trajectory (mean / max), network evaluations, rejected
steps, the wall-clock of the iteration (synchronised) and the two kernel times.  The adaptive LambaEM step count depends on theta
through |z| = |sigma^T grad u|: at the Glorot initialisation the diffusion term of u is large and the estimator asks for 1e4 .. 5e4
steps; a few ADAM iterations later z is small and a trajectory needs a few hundred.
    python tools/hjb_script_call_profile.py [maxiters] [out.json]        (needs a GPU)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from universal_differential_equations_amd import pde     # noqa: E402

d, m, lam, hls = 100, 100, 1.0, 110
maxiters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
x0 = np.zeros(d, dtype=np.float32)
prob = pde.TerminalPDEProblem(pde.hjb(lam), x0, (0.0, 1.0))
alg = pde.NNPDENS(d, hls, opt=pde.ADAM(0.03))
theta = torch.tensor(alg.init_params(np.random.default_rng(0)), device="cuda:0")
bs = pde.DeviceBSDE(prob, alg, pde.LambaEM(), m, device=torch.device("cuda", 0), abstol=1e-4, reltol=1e-4, seed=0)
opt = alg.opt
mm, vv = torch.zeros_like(theta), torch.zeros_like(theta)
b1p, b2p = opt.beta
marks = {0, 1, 2, 5, 10, 25} | set(range(50, maxiters, 50)) | {maxiters - 1}
rows, t_all = [], time.perf_counter()
for it in range(maxiters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, g = bs.loss_grad(theta, it=it)
    lval = float(loss.item())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    if it in marks:
        st = bs.stats.cpu().numpy()
        f, b = bs.kernel_ms()
        rows.append({"iteration": it, "loss": lval, "u0": float(bs.u0.item()), "ms": ms, "fwd_kernel_ms": f, "bwd_kernel_ms": b,
                     "accepted_steps_mean": float(st[:, 1].mean()), "accepted_steps_max": int(st[:, 1].max()),
                     "rejected_steps_mean": float(st[:, 2].mean()), "net_evals": int(st[:, 0].sum())})
        print(rows[-1], flush=True)
    if lval < 1e-2:
        break
    mm.mul_(opt.beta[0]).add_(g, alpha=1 - opt.beta[0])
    vv.mul_(opt.beta[1]).addcmul_(g, g, value=1 - opt.beta[1])
    theta = theta - (mm / (1 - b1p)) / (torch.sqrt(vv / (1 - b2p)) + opt.eps) * opt.eta
    b1p *= opt.beta[0]
    b2p *= opt.beta[1]
elapsed = time.perf_counter() - t_all
rec = {"script": "highdim_pde/lambaem.jl:33-34 (maxiters = %d, trajectories = %d, LambaEM, abstol = reltol = 1e-4, ADAM(0.03))" % (maxiters, m),
       "elapsed_s": elapsed, "mean_ms_per_iteration": elapsed / maxiters * 1e3,
       "note": "the step count of the adaptive solve depends on theta: iteration 0 (theta_init, what bench.py's hjb_script_tol leg measures) "
               "is the most expensive evaluation of the whole training", "iterations": rows}
print(json.dumps(rec)[:400])
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as fh:
        json.dump(rec, fh, indent=1)
