import sys, numpy as np, torch, time
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
dev = torch.device('cuda', 0)
w = bench.synth_inputs_other('seir', 6250, 0, dev)
for dims, acts in (([3, 64, 63, 1], ["tanh", "tanh", "identity"]), ([3, 16, 16, 1], ["tanh", "tanh", "identity"])):
    chain = models.Chain(*[models.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(dims) - 1)])
    f = models.dudt_(chain)
    th = torch.tensor(chain.glorot_uniform(np.random.default_rng(0)), dtype=torch.float64, device=dev)
    for sense in ("adjoint", "fast"):
        ens = U.DeviceEnsemble(f, w['alg'], w['tspan'], w['t'], w['u0'], data=w['data'], row_mask=w['mask'], sensealg=bench.SENSE_OBJ(U, sense), **w['tol'])
        for _ in range(3):
            ens.loss_grad(th); torch.cuda.synchronize()
        print(dims, sense, 'kernel ms (fwd, bwd)', ens.kernel_ms(), 'failed', int((ens.retcode != 0).sum()))
