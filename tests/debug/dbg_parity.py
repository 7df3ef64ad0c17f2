"""Find trajectories whose backward step counts differ between the HIP path and the oracle and print both traces."""
import ctypes, json, os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models

g = json.load(open("tests/golden/Scenario_1_recovery_0.005.json"))
X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
t = np.array(g["solution"]["t"])
th = np.array(g["trained_parameters"])
alg, oalg = (U.Vern7, O.VERN7) if len(sys.argv) < 2 or sys.argv[1] == "vern7" else (U.Tsit5, O.TSIT5)
N = 96
rng = np.random.default_rng(7)
u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
data = np.repeat(X[None], N, axis=0)
ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (t[0], t[-1]), th), u0)
r = U.loss_and_gradient(ens, alg(), data, saveat=t, abstol=1e-6, reltol=1e-6)
ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(oalg, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data, nthreads=4)
bad = np.nonzero((r.stats[:, 4:7] != ref["stats"][:, 4:7]).any(axis=1))[0]
print("mismatching trajectories:", bad, "fwd mismatch:", np.nonzero((r.stats[:, :3] != ref["stats"][:, :3]).any(axis=1))[0])
if len(bad):
    j = int(bad[0])
    print("gpu", r.stats[j], "oracle", ref["stats"][j])
    eng = U.Engine.get(0)
    eng.set_trace(0, 256)
    one = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[j], (t[0], t[-1]), th), u0[j:j + 1])
    r1 = U.loss_and_gradient(one, alg(), data[j:j + 1], saveat=t, abstol=1e-6, reltol=1e-6)
    tr = eng.get_trace()
    print("GPU backward trace (t, dt, EEst, q, acc):")
    for row in tr[1][: int(r1.stats[0, 5] + r1.stats[0, 6]) + 1]:
        print("  t=%.17g dt=%.17g EEst=%.9g q=%.9g acc=%d" % tuple(row))
    ctypes.c_int.in_dll(O.lib(), "udeo_debug").value = 1
    sys.stderr.flush()
    O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(oalg, 1e-6, 1e-6), u0[j], [t[0], t[-1]], th, t, data[j:j + 1])
