"""debug: node adjoint, good/bad counts over repeated single-trajectory calls against the oracle's step counts
(env: UDE_EXP_LIB, UDE_EXP_WS_FILL, UDE_EXP_POISON, UDE_DBG_ALG = t5 | v7, UDE_DBG_S0)"""
import os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import universal_differential_equations_amd._lib as _L
if os.environ.get("UDE_EXP_LIB"):
    _L.LIB_PATH = os.environ["UDE_EXP_LIB"]
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_node import node_case, MASK
S0 = float(os.environ.get("UDE_DBG_S0", "14e6"))
tf = 21.0 if S0 > 1e3 else 6.0
alg, oalg = (U.Vern7, O.VERN7) if os.environ.get("UDE_DBG_ALG", "t5") == "v7" else (U.Tsit5, O.TSIT5)
u0, th = node_case(6, S0)
t = np.arange(0.0, tf + 0.5, 1.0)
truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=6)
f = models.dudt_node()
res = []
for rep in range(3):
    for i in range(6):
        e1 = U.EnsembleProblem(U.ODEProblem(f, u0[i], (0.0, tf), th), u0[i:i + 1])
        r1 = U.loss_and_gradient(e1, alg(), truth[i:i + 1], row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
        ok = np.array_equal(r1.stats[0], ref["stats"][i]) and np.array_equal(r1.grad_u0[0], ref["grad_u0"][i])
        res.append("G" if ok else "b")
print("lib=%s alg=%s S0=%g fill=%s poison=%s : %s" % (os.path.basename(os.environ.get("UDE_EXP_LIB", "default")), os.environ.get("UDE_DBG_ALG", "t5"), S0,
      os.environ.get("UDE_EXP_WS_FILL"), os.environ.get("UDE_EXP_POISON"), "".join(res)))
