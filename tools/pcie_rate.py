"""PCIe-inclusive rate of the host-buffer C entry point on the C2 workload (DESIGN.md quotes it; it is never bench.py's value)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
g = json.load(open("tests/golden/Scenario_1_recovery_0.005.json"))
X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
t = np.array(g["solution"]["t"]); th = np.array(g["initial_parameters"])
N = 10000
rng = np.random.default_rng(1234)
u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
data = np.repeat(X[None], N, axis=0)
ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (0.0, 3.0), th), u0)
for _ in range(3):
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6)
t0 = time.perf_counter(); K = 10
for _ in range(K):
    r = U.loss_and_gradient(ens, U.Tsit5(), data, saveat=t, abstol=1e-6, reltol=1e-6)
dt = (time.perf_counter() - t0) / K
ev = int(r.stats[:, 0].sum() + r.stats[:, 4].sum())
print(json.dumps({"host_buffer_ms_per_gradient": dt * 1e3, "evals": ev, "evals_per_s_pcie_inclusive": ev / dt,
                  "bytes_h2d": u0.nbytes + data.nbytes + th.nbytes, "bytes_d2h": r.u.nbytes + r.stats.nbytes + r.grad_u0.nbytes}))
