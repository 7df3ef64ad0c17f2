"""debug: determinism of the node Vern7 adjoint for one variant library (UDE_EXP_LIB)"""
import hashlib, os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import universal_differential_equations_amd._lib as _L
if os.environ.get("UDE_EXP_LIB"):
    _L.LIB_PATH = os.environ["UDE_EXP_LIB"]
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_node import node_case, MASK
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:6]
out = []
for N in (1, 6):
    u0, th = node_case(N, 100.0)
    t = np.arange(0.0, 6.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 6.0], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, 6.0), th), u0)
    for nm, alg, oalg in (("v7", U.Vern7, O.VERN7), ("t5", U.Tsit5, O.TSIT5)):
        ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, 6.0], th, t, truth, row_mask=MASK, nthreads=4)
        ok = 0
        for rep in range(8):
            r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
            ok += int(np.array_equal(r.stats, ref["stats"]) and np.array_equal(r.grad_u0, ref["grad_u0"]))
        out.append("N=%d %s %d/8" % (N, nm, ok))
print(os.environ.get("UDE_EXP_LIB", "default"), " ".join(out))
