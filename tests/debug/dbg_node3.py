import sys, ctypes as C, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models, sciml
from test_gpu_node import node_case, MASK
N = 1
u0, th = node_case(N, 100.0)
tf = 6.0
t = np.arange(0.0, tf + 0.5, 1.0)
truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, tf], [], t)
f = models.dudt_node()
ens = U.EnsembleProblem(U.ODEProblem(f, u0[0], (0.0, tf), th), u0)
eng = sciml.Engine.get(0)
eng.set_trace(0, 64)
C.c_int.in_dll(O.lib(), "udeo_debug").value = 1
ref = O.loss_grad_ensemble(O.seir_node(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [0.0, tf], th, t, truth, row_mask=MASK)
C.c_int.in_dll(O.lib(), "udeo_debug").value = 0
print("oracle stats", ref["stats"].tolist())
for it in range(3):
    r = U.loss_and_gradient(ens, U.Tsit5(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
    tr = eng.get_trace()
    print("device stats", r.stats.tolist(), "retcode", r.retcode.tolist())
    print("bwd trace rows:"); print(tr[1][:4])
