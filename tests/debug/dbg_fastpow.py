import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import _oracle as O
import universal_differential_equations_amd as U
rng = np.random.default_rng(0)
x = np.concatenate([10.0 ** rng.uniform(-14, 8, 20000), [1.0, 1e-4, 0.5, 2.0, 1.5, 0.75]])
eng = U.Engine.get(0)
L = O.lib()
for y in (0.14, 0.08):
    dev = eng.fastpow(x, y)
    ref = np.array([L.udeo_fastpow(float(v), y) for v in x])
    bad = np.nonzero(dev != ref)[0]
    print("y", y, "mismatches", len(bad), "of", len(x))
    for i in bad[:12]:
        print("  x=%r dev=%r ref=%r  ulps(f32)=%d" % (x[i], dev[i], ref[i], int(np.float32(dev[i]).view(np.int32)) - int(np.float32(ref[i]).view(np.int32))))
