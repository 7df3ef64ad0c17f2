"""debug: hashes of device and oracle results of the neural-ODE adjoint (to tell a device difference from an oracle difference across GPU boxes)"""
import hashlib, sys, subprocess, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from test_gpu_node import node_case, MASK
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
try:
    print(subprocess.run("rocm-smi --showuniqueid | grep -i 'unique id:' | head -1", shell=True, capture_output=True, text=True).stdout.strip())
    print(subprocess.run("lscpu | grep 'Model name' | head -1", shell=True, capture_output=True, text=True).stdout.strip())
except Exception:
    pass
for N in (1, 6):
    u0, th = node_case(N, 100.0)
    t = np.arange(0.0, 6.5, 1.0)
    truth, _, rc = O.solve_ensemble(O.seir_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 6.0], [], t)
    ens = U.EnsembleProblem(U.ODEProblem(models.dudt_node(), u0[0], (0.0, 6.0), th), u0)
    for name, alg, oalg in (("vern7", U.Vern7, O.VERN7), ("tsit5", U.Tsit5, O.TSIT5)):
        ref = O.loss_grad_ensemble(O.seir_node(), O.opts(oalg, 1e-6, 1e-6), u0, [0.0, 6.0], th, t, truth, row_mask=MASK, nthreads=4)
        devs = []
        for rep in range(4):
            r = U.loss_and_gradient(ens, alg(), truth, row_mask=MASK, saveat=t, abstol=1e-6, reltol=1e-6, allow_failures=True)
            devs.append((h(r.stats), h(r.grad_u0), h(r.u)))
        print("N=%d %s oracle stats %s gu0 %s u %s truth %s | device" % (N, name, h(ref["stats"]), h(ref["grad_u0"]), h(ref["u"]), h(truth)), devs)
