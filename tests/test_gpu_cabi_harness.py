"""The boundary driven from plain C++ (no Python, no torch in the calling process), as SURVEY.md 4.5 asks: build
tests/cabi_harness.cpp against include/udecore.h + libudecore.so, run it, compare what it printed with the oracle."""
import json
import os
import subprocess

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_harness_through_the_c_abi(tmp_path):
    libdir = os.path.join(ROOT, "universal_differential_equations_amd")
    exe = str(tmp_path / "cabi_harness")
    subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cabi_harness.cpp"), "-L" + libdir,
                           "-l:libudecore.so", "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["version"] == 100 and r["rc_unsupported"] == -2
    assert r["fd_worst"] < 1e-6 and abs(r["loss"] - r["loss_direct"]) < 1e-12 * r["loss"]
    # the same call on the oracle
    u0 = np.array([[0.44249296, 4.6280594], [0.5, 4.0], [0.4, 5.0]])
    t = 0.5 * np.arange(7)
    o = O.opts(O.VERN7, 1e-10, 1e-10)
    truth, _, _ = O.solve_ensemble(O.lv_true(), o, u0, [0.0, 3.0], np.array([1.3, 0.9, 0.8, 1.8]), t)
    ref = O.loss_grad_ensemble(O.lv_true(), o, u0, [0.0, 3.0], np.array([1.2, 1.0, 0.7, 1.9]), t, truth)
    assert r["nf0"] == ref["stats"][0, 0] and r["naccept0"] == ref["stats"][0, 1]
    assert r["pred00"] == ref["u"][0, 0, 0]
    assert abs(r["loss"] - ref["loss"]) < 1e-12 * ref["loss"]
    assert np.linalg.norm(np.array(r["grad"]) - ref["grad_theta"]) < 1e-12 * np.linalg.norm(ref["grad_theta"])
