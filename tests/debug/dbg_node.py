"""debug: run-to-run determinism of the neural-ODE adjoint (device vs oracle, alternating algorithms)"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_node import node_case, MASK
N = 6
u0, th = node_case(N, 100.0)
tf = 6.0
t = np.arange(0.0, tf + 0.5, 1.0)
truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
f = models.dudt_node()
ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
refs = {}
for name, alg, oalg in (("vern7", U.Vern7, O.VERN7), ("tsit5", U.Tsit5, O.TSIT5)):
    refs[name] = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for name, alg in (("vern7", U.Vern7), ("tsit5", U.Tsit5)):
        try:
            r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
        except Exception as e:
            print(it, name, "EXC", e); bad += 1; continue
        ref = refs[name]
        ok = np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.grad_u0, ref["grad_u0"])
        if not ok:
            bad += 1
            rows = [j for j in range(N) if not np.array_equal(r.stats[j], ref["stats"][j])]
            print(it, name, "MISMATCH rows", rows, "dev", r.stats[rows].tolist(), "ref", ref["stats"][rows].tolist(), "retcode", r.retcode.tolist())
print("bad", bad)
