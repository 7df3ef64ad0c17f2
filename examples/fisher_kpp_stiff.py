#!/usr/bin/env python3
"""SURVEY.md 8(d) "stiff C4": the Fisher-KPP UDE of FisherKPP/Fisher-KPP-CNN.jl on 1024 points with the DOMAIN kept at
X = 1 (dx = 1/1023, D/dx^2 = 1.05e4) instead of the grid spacing kept at 0.04.  Tsit5 is then stability-limited: ~6e4 accepted
steps forward (the oracle takes 59 683 for the true model), and the interpolating adjoint walks a dense store of
~6e4 steps x 8195 fields per PDE (4.6 GB per PDE: sized for the 288 GB of an MI355X).  With `checkpointing` as second argument the
gradient is taken with InterpolatingAdjoint(checkpointing = true): the store keeps (t, dt, u) per step (1027 fields, 0.58 GB per
PDE) and the adjoint kernel recomputes the stages of an interval when it enters it -- bit-identical results, 8x the PDEs per GPU.
One loss + gradient of B PDEs, timed by the library's HIP events.
Needs a GPU:  python examples/fisher_kpp_stiff.py [B=8] [checkpointing]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models               # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CKPT = len(sys.argv) > 2 and sys.argv[2].startswith("checkpoint")
nx, D, r, X, T = 1024, 0.01, 1.0, 1.0, 5.0
dx = X / (nx - 1)
x = np.arange(nx) * dx
rho0 = 0.5 * (np.tanh((x - 0.3) / 0.2) - np.tanh((x - 0.7) / 0.2))            # Fisher-KPP-CNN.jl:31 on the X = 1 domain
rng = np.random.default_rng(0)
u0 = rho0[None, :] * (1 + 0.05 * rng.uniform(-1, 1, (B, 1)))
t = np.arange(11) * 0.5
dev = torch.device("cuda:0")
u0_d = torch.tensor(u0, device=dev)
CAP = 70000                                                                    # dense-store capacity (steps per PDE)
truth = U.DeviceEnsemble(models.rc_ode(nx, D, r, dx), U.Tsit5(), (0.0, T), t, u0_d, maxiters=400000)
t0 = time.perf_counter()
data = truth.solve(torch.zeros(1, dtype=torch.float64, device=dev)).clone()
torch.cuda.synchronize()
t_true = time.perf_counter() - t0
assert int((truth.retcode != 0).sum()) == 0
th = models.kpp_theta(models.kpp_chain(), rng)
f = models.nn_ode(nx)
th[f.d0_offset] = 0.95 * D / dx ** 2                                           # D0 near the true D/dx^2 (the script's 6.5 belongs to dx = 0.04)
th[f.stencil_offset:f.stencil_offset + 3] = [1.01, -2.0, 0.99]
ens = U.DeviceEnsemble(f, U.Tsit5(), (0.0, T), t, u0_d, data=data, max_dense_steps=CAP, maxiters=400000,
                       sensealg=U.InterpolatingAdjoint(checkpointing=True) if CKPT else None)
theta = torch.tensor(th, device=dev)
t0 = time.perf_counter()
g = ens.loss_grad(theta, check=False)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
nfail = ens.check()
fwd_ms, bwd_ms = ens.kernel_ms()
st = ens.stats.cpu().numpy()
dense_gb = CAP * (3 + nx + (0 if CKPT else 7 * nx)) * ((B + 7) // 8 * 8) * 8 / 1e9
evals = int(st[:, 0].sum() + st[:, 4].sum())
print(json.dumps({"variant": "stiff C4: Fisher-KPP UDE, 1024 points on X = 1 (D/dx^2 = %.4g), Tsit5 default tol, loss + InterpolatingAdjoint%s gradient" % (D / dx ** 2, "(checkpointing = true)" if CKPT else ""),
                  "pdes": B, "failed": int(nfail), "true_model_forward_s": t_true, "true_model_steps": int(truth.stats[0, 1]),
                  "forward_steps_per_pde": [int(st[:, 1].min()), int(st[:, 1].max())], "backward_steps_per_pde": [int(st[:, 5].min()), int(st[:, 5].max())],
                  "rhs_evals": evals, "fwd_kernel_s": fwd_ms / 1e3, "adj_kernel_s": bwd_ms / 1e3, "wall_s": wall, "evals_per_s": evals / wall,
                  "dense_store_GB": dense_gb, "dense_bytes_read_by_adjoint_GB_est": (0.0 if CKPT else int(st[:, 4].sum()) * 8 * nx * 8 / 1e9),
                  "loss": float(g[-1]), "grad_norm": float(torch.linalg.norm(g[:-1]))}))
