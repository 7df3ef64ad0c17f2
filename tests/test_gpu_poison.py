"""Stale-register check.  The DEBUG build of the library (libudecore_dbg.so: the same kernel objects, host side compiled with
-DUDE_DEBUG_HOOKS; `UDE_LIB_VARIANT=dbg`) runs, with UDE_EXP_POISON=3,5, in front of the forward kernel and between the forward
and the backward kernel of every call a kernel that leaves different garbage in every lane of every VGPR and AGPR of the chip
(udecore.hip: ude_poison_chip_dbg).  A kernel that reads a register lane it never wrote -- round 2 found one: a
compiler-inserted VGPR->AGPR copy in front of the EXEC restore of a join block in the neural-ODE adjoint -- then
fails on every run instead of on some runs of some GPUs; one that does not is bit-identical to the oracle as always.

The shipping library contains no such hook (and reads no environment variable on its launch path), so the parity tests are
re-run in a child process that loads the debug library; nothing here has its own expected values."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

# Every oracle-comparing GPU test file is repeated under the poison kernel (pytest -k expressions per file; "" = the whole
# file).  Round 4: the fuzz corners, the fast-adjoint, Float32, stiff and edge-case files joined (the round-3 review); what the
# poison test can and cannot see is stated in DESIGN.md 2a -- a register that keeps an OLDER VALUE OF THE SAME KERNEL is not
# garbage and stays invisible here; that class is closed at build time (tools/isa_endcf_fix.py, tests/test_build_gate_cpu.py).
SELECTION = [
    ("test_gpu_node.py", "forward_and_adjoint_match_oracle or reproducible_run_to_run or rhs_matches_oracle"),
    ("test_gpu_parity.py", "test_seir_ude_forward_and_adjoint_match_oracle or test_adjoint_gradient_matches_oracle or "
                           "test_discrete_gradient_seir_and_kpp_match_oracle or test_kpp_ude_forward_and_adjoint_match_oracle or "
                           "test_seir_true_matches_oracle or test_kpp_true_matches_oracle or test_forward_ensemble_matches_oracle"),
    ("test_gpu_hjb.py", "test_adaptive_loss_and_gradient_match_oracle or test_rejections_and_stack_match_oracle"),
    ("test_gpu_generic.py", "fuzz"),
    ("test_gpu_fuzz.py", ""),
    ("test_gpu_fast_adjoint.py", ""),
    ("test_gpu_f32.py", ""),
    ("test_gpu_stiff.py", ""),
    ("test_gpu_edge_cases.py", ""),
]


def _run(files_k, extra_env):
    env = dict(os.environ, UDE_LIB_VARIANT="dbg", **extra_env)
    for fname, kexpr in files_k:
        path = os.path.join(HERE, fname)
        if not os.path.exists(path):
            continue
        r = subprocess.run([sys.executable, "-m", "pytest", path, "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + (["-k", kexpr] if kexpr else []),
                           env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
        assert r.returncode == 0, "%s under %s:\n%s\n%s" % (fname, extra_env, r.stdout[-3000:], r.stderr[-2000:])
        assert " passed" in r.stdout, r.stdout[-500:]


def test_debug_library_really_poisons():
    """positive control: the child loads libudecore_dbg.so AND its poison kernel reaches the register file -- after a poison
    launch with a given pattern a fresh wavefront finds that pattern in a VGPR and an AGPR it never wrote (a debug library built
    without the hook, or a hook that no longer runs, fails here instead of turning this file into a plain re-run)"""
    code = ("import ctypes as C, numpy as np; from universal_differential_equations_amd import _lib; import universal_differential_equations_amd as U; "
            "L = _lib.load(); assert _lib.LIB_PATH.endswith('libudecore_dbg.so'), _lib.LIB_PATH; eng = U.Engine.get(0); "
            "L.ude_dbg_poison_selftest.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]; out = np.zeros(128, dtype=np.uint32); "
            "rc = L.ude_dbg_poison_selftest(eng.h, 0x5EED1234, out.ctypes.data); assert rc == 0, rc; "
            "hits = int((out == 0x5EED1234).sum()); print('hits', hits); assert hits == 128, out; print('dbg ok')")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UDE_LIB_VARIANT="dbg"), capture_output=True, text=True,
                       cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and "dbg ok" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]


@pytest.mark.parametrize("idx", range(len(SELECTION)), ids=[s[0][9:-3] for s in SELECTION])
def test_parity_with_garbage_registers(idx):
    # kind 3 (different garbage in every lane and register), what = 1 | 4: registers, also in front of the forward kernels
    _run([SELECTION[idx]], {"UDE_EXP_POISON": "3,5"})
