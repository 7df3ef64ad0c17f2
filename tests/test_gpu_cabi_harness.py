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
    # the Float32 solve (udeo f32 instantiation) and the deep-BSDE call (SDE oracle) through the same binary
    import _sde_oracle as S
    xx = np.arange(26, dtype=np.float32) / np.float32(25.0)
    om = np.float32(1.0) - xx
    u0f = ((np.float32(16.0) * xx) * xx) * (om * om)
    assert u0f.dtype == np.float32
    outf, stf, rcf = O.solve_ensemble(O.kpp_true(26, 0.01, 1.0, 0.04, dtype=1), O.opts(O.TSIT5), u0f, [0.0, 5.0], [], [0.0, 2.5, 5.0], dtype=np.float32)
    assert r["f32_rc"] == 0 and [r["f32_nf"], r["f32_naccept"], r["f32_nreject"]] == list(stf[0][:3])
    assert np.float32(r["f32_u_end_13"]) == outf[0][2, 13]
    np0, np1 = S.num_params(100, 110)
    assert r["hjb_np"] == np0 + np1 == 70171
    lcg, th = 12345, np.zeros(np0 + np1, dtype=np.float32)
    for i in range(th.size):
        lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
        th[i] = (np.float32((lcg >> 8) & 0xFFFF) / np.float32(65536.0) - np.float32(0.5)) * np.float32(0.2)
    refh = S.loss_grad(S.desc(abstol=0.1, reltol=0.1, seed=77), 5, np.zeros(100, dtype=np.float32), th, it=3)
    assert r["hjb_loss"] == refh["loss"] and np.float32(r["hjb_u0"]) == refh["u0"] and np.float32(r["hjb_uT0"]) == refh["uT"][0]
    assert [r["hjb_naccept0"], r["hjb_nreject0"]] == list(refh["stats"][0][1:3])
    assert abs(r["hjb_grad_norm"] - np.linalg.norm(refh["grad"].astype(np.float64))) < 1e-5 * np.linalg.norm(refh["grad"].astype(np.float64))
