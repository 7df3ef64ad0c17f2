"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the checker must be clean itself).  `make -C oracle
asan` builds libude_oracle_asan.so from the same sources; a child process preloads the sanitizer runtimes, loads it through
tests/_oracle.py (UDE_ORACLE_LIB) and runs one representative pass of every path the parity tests rely on: forward solves, the
interpolating adjoint (parity and fast), the discrete sweep, Tsit5 and Vern7, Float64 and Float32, the replicated and the
distributed-state models, a runtime-shape chain, and the deep-BSDE (LambaEM) step.  Any report aborts the child."""
import glob
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import _oracle as O
import _sde_oracle as S
rng = np.random.default_rng(0)
t = np.linspace(0.0, 1.5, 7)
# LV UDE (scenario_1.jl:59-94): adjoint, fast, discrete; Tsit5 and Vern7
m = O.lv_ude_s1(); th = 0.3 * rng.standard_normal(m.n_param)
u0 = np.array([[0.44, 4.6], [0.5, 4.2]]); data = rng.uniform(0, 5, (2, 7, 2))
for alg in (O.TSIT5, O.VERN7):
    for sense in (0, 1, 2):
        r = O.loss_grad_ensemble(m, O.opts(alg, 1e-6, 1e-6, sensealg=sense), u0, [0.0, 1.5], th, t, data, row_mask=[1, 0], nthreads=2)
        assert (r["retcode"] == 0).all() and np.isfinite(r["grad_theta"]).all()
# Float32 (hudson_bay.jl:77-104)
mh = O.lv_ude_hudson(1); thh = np.concatenate([[1.3, 1.8], 0.3 * rng.standard_normal(87)]).astype(np.float32)
r = O.loss_grad_ensemble(mh, O.opts(O.VERN7, 1e-5, 1e-5), u0.astype(np.float32), [0.0, 1.5], thh, t.astype(np.float32), data.astype(np.float32), dtype=np.float32)
assert (r["retcode"] == 0).all()
# SEIR exposure UDE + neural ODE (seir_exposure.jl:53-147), a runtime-shape chain
for mk in (O.seir_ude(), O.seir_node(), O.make_model(O.KIND_SEIR_UDE, 7, (3, 17, 5, 1), ("tanh", "rbf", "identity"), consts=O.SEIR_P)):
    ths = 0.1 * rng.standard_normal(mk.n_param)
    us = np.zeros((1, 7)); us[0, 0] = 90.0; us[0, 1] = 2.0; us[0, 2] = 1.0; us[0, 4] = 100.0
    ds = us[:, None, :] * np.ones((1, 4, 1))
    r = O.loss_grad_ensemble(mk, O.opts(O.VERN7, 1e-6, 1e-6), us, [0.0, 3.0], ths, np.linspace(0, 3, 4), ds, row_mask=[0, 1, 1, 0, 0, 1, 0])
    assert (r["retcode"] == 0).all()
# Fisher-KPP (Fisher-KPP-CNN.jl:51-143): true model and the UDE on a ragged grid
out, st, rc = O.solve_ensemble(O.kpp_true(26), O.opts(O.TSIT5), np.full((1, 26), 0.3), [0.0, 1.0], [], np.array([0.0, 0.5, 1.0]))
assert (rc == 0).all()
mk = O.kpp_ude(11); thk = 0.2 * rng.standard_normal(mk.n_param); thk[-5:-2] = [1.0, -2.0, 1.0]; thk[-1] = 2.0
r = O.loss_grad_ensemble(mk, O.opts(O.TSIT5, 1e-5, 1e-5, sensealg=1), np.full((1, 11), 0.3), [0.0, 0.5], thk, np.array([0.0, 0.25, 0.5]), np.full((1, 3, 11), 0.2))
assert (r["retcode"] == 0).all()
# deep-BSDE step (highdim_pde/lambaem.jl:18-34)
D = S.desc(d=8, hls=12, abstol=0.05, reltol=0.05, seed=3)
thb = S.glorot_params(8, 12, rng)
r = S.loss_grad(D, 3, np.zeros(8, dtype=np.float32), thb, it=1)
assert np.isfinite(r["loss"]) and np.isfinite(r["grad"]).all()
print("sanitized ok")
'''


def test_oracle_is_clean_under_asan_and_ubsan():
    gcc_dir = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(gcc_dir) and os.path.exists(gcc_dir) and os.path.exists(ubsan)):
        pytest.skip("no sanitizer runtimes next to this gcc")
    env = dict(os.environ, UDE_ORACLE_LIB="libude_oracle_asan.so", LD_PRELOAD=gcc_dir + ":" + ubsan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", CHILD % HERE], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
