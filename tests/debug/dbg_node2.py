import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import os
import universal_differential_equations_amd._lib as _L
if os.environ.get('UDE_EXP_LIB'):
    _L.LIB_PATH = os.environ['UDE_EXP_LIB']
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_node import node_case, MASK
N = int(sys.argv[2]) if len(sys.argv) > 2 else 6
u0, th = node_case(N, 100.0)
tf = 6.0
t = np.arange(0.0, tf + 0.5, 1.0)
truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
f = models.dudt_node()
ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
refs = {}
algs = {"v": ("vern7", U.Vern7, O.VERN7), "t": ("tsit5", U.Tsit5, O.TSIT5)}
for k, (name, alg, oalg) in algs.items():
    refs[k] = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK, nthreads=4)
seq = sys.argv[1]
res = ""
for k in seq:
    name, alg, oalg = algs[k]
    r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
    ok = np.array_equal(r.stats, refs[k]["stats"]) and np.array_equal(r.grad_u0, refs[k]["grad_u0"])
    res += k.upper() if ok else "x"
print(seq, "->", res)
