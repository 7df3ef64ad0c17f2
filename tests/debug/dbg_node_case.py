"""debug: the S0 = 14e6, tf = 21 node case per trajectory (alone and in the ensemble) for one library (UDE_EXP_LIB)"""
import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import universal_differential_equations_amd._lib as _L
if os.environ.get("UDE_EXP_LIB"):
    _L.LIB_PATH = os.environ["UDE_EXP_LIB"]
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_node import node_case, MASK
S0, tf = 14e6, 21.0
u0, th = node_case(6, S0)
t = np.arange(0.0, tf + 0.5, 1.0)
truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
f = models.dudt_node()
for nm, alg, oalg in (("t5", U.Tsit5, O.TSIT5), ("v7", U.Vern7, O.VERN7)):
    ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
    ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
    r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
    print(nm, "ensemble: stats ok per trajectory", [int(np.array_equal(r.stats[i], ref["stats"][i])) for i in range(6)],
          "grad_u0 ok", [int(np.array_equal(r.grad_u0[i], ref["grad_u0"][i])) for i in range(6)])
    print("   dev", r.stats[:3, 4:8].tolist(), "ref", ref["stats"][:3, 4:8].tolist())
    for i in range(6):
        e1 = U.EnsembleProblem(U.ODEProblem(f, u0[i], (0.0, tf), th), u0[i:i + 1])
        r1 = U.loss_and_gradient(e1, alg(), truth[i:i + 1], row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
        print("   alone %d: stats ok %d grad_u0 ok %d  dev %s ref %s" % (i, np.array_equal(r1.stats[0], ref["stats"][i]),
              np.array_equal(r1.grad_u0[0], ref["grad_u0"][i]), r1.stats[0, 4:8].tolist(), ref["stats"][i, 4:8].tolist()))
