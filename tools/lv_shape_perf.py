import sys, os, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
dev = torch.device('cuda', 0)
for name in ("lv_shape8", "lv_tanh32"):
    r = bench.quick_measure(name, dev)
    print(name, {k: r[k] for k in ("ms_per_step", "kernel_ms", "fwd_kernel_ms", "frac", "failed_trajectories")})
import numpy as np
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
theta_h, u0_d, t, data = bench.synth_inputs(10000, 0, dev)
chain8 = models.Chain(models.Dense(2, 8, "tanh"), models.Dense(8, 8, "tanh"), models.Dense(8, 8, "tanh"), models.Dense(8, 2, "identity"))
th = torch.tensor(0.1 * chain8.glorot_uniform(np.random.default_rng(7)), dtype=torch.float64, device=dev)
ens = U.DeviceEnsemble(models.ude_dynamics(chain8), U.Tsit5(), (0.0, 3.0), t, u0_d, data=data, abstol=1e-6, reltol=1e-6, lanes_per_traj=64)
for _ in range(3):
    ens.loss_grad(th); torch.cuda.synchronize()
print("2-8-8-8-2 on the wavefront-per-trajectory kernel (lanes 64): kernel ms", ens.kernel_ms())

chain5 = models.Chain(models.Dense(2, 5, "tanh"), models.Dense(5, 5, "tanh"), models.Dense(5, 5, "tanh"), models.Dense(5, 2, "identity"))
th5 = torch.tensor(0.3 * chain5.glorot_uniform(np.random.default_rng(7)), dtype=torch.float64, device=dev)
for lanes in (0, 8, 64):
    ens = U.DeviceEnsemble(models.ude_dynamics(chain5), U.Tsit5(), (0.0, 3.0), t, u0_d, data=data, abstol=1e-6, reltol=1e-6, lanes_per_traj=lanes)
    for _ in range(3):
        ens.loss_grad(th5); torch.cuda.synchronize()
    print("2-5-5-5-2 tanh (the scripts' widths, edited activations), lanes_per_traj", lanes or "default (5)", ": kernel ms", ens.kernel_ms())

chain16 = models.Chain(models.Dense(2, 16, "tanh"), models.Dense(16, 16, "tanh"), models.Dense(16, 16, "tanh"), models.Dense(16, 2, "identity"))
th16 = torch.tensor(0.1 * chain16.glorot_uniform(np.random.default_rng(7)), dtype=torch.float64, device=dev)
for lanes in (0, 64):
    ens = U.DeviceEnsemble(models.ude_dynamics(chain16), U.Tsit5(), (0.0, 3.0), t, u0_d, data=data, abstol=1e-6, reltol=1e-6, lanes_per_traj=lanes)
    for _ in range(3):
        ens.loss_grad(th16); torch.cuda.synchronize()
    print("2-16-16-16-2 tanh, lanes_per_traj", lanes or "default (16)", ": kernel ms", ens.kernel_ms())
