"""configs[3] (1024 points x 256 PDEs, Tsit5) with the reaction network edited to 1-16-16-16-1 / 1-8-8-8-1 (run-time-shape instance of the
1024-point matrix-core kernel) next to the compiled 1-10-20-10-1: kernel ms forward, backward.  Needs a GPU."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                          # noqa: E402
import universal_differential_equations_amd as U                      # noqa: E402
from universal_differential_equations_amd import models              # noqa: E402

dev = torch.device('cuda', 0)
w = bench.synth_inputs_other('kpp', 256, 0, dev)
for dims in ([1, 10, 20, 10, 1], [1, 16, 16, 16, 1], [1, 8, 8, 8, 1]):
    chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], "tanh" if i < 3 else "identity") for i in range(4)])
    f = models.nn_ode(1024, chain)
    th = torch.tensor(models.kpp_theta(chain, np.random.default_rng(0)), dtype=torch.float64, device=dev)
    ens = U.DeviceEnsemble(f, w['alg'], w['tspan'], w['t'], w['u0'], data=w['data'], row_mask=w['mask'], **w['tol'])
    for _ in range(3):
        ens.loss_grad(th); torch.cuda.synchronize()
    print(dims, 'kernel ms (fwd, bwd)', ens.kernel_ms(), 'failed', int((ens.retcode != 0).sum()))
