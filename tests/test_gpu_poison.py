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

# the parity tests that are repeated under the poison kernel (pytest -k expressions per file)
SELECTION = [
    ("test_gpu_node.py", "forward_and_adjoint_match_oracle or reproducible_run_to_run or rhs_matches_oracle"),
    ("test_gpu_parity.py", "test_seir_ude_forward_and_adjoint_match_oracle or test_adjoint_gradient_matches_oracle or "
                           "test_discrete_gradient_seir_and_kpp_match_oracle or test_kpp_ude_forward_and_adjoint_match_oracle or "
                           "test_seir_true_matches_oracle or test_kpp_true_matches_oracle or test_forward_ensemble_matches_oracle"),
    ("test_gpu_hjb.py", "test_adaptive_loss_and_gradient_match_oracle or test_rejections_and_stack_match_oracle"),
    ("test_gpu_generic.py", "fuzz"),
]


def _run(files_k, extra_env):
    env = dict(os.environ, UDE_LIB_VARIANT="dbg", **extra_env)
    for fname, kexpr in files_k:
        path = os.path.join(HERE, fname)
        if not os.path.exists(path):
            continue
        r = subprocess.run([sys.executable, "-m", "pytest", path, "-x", "-q", "-m", "gpu", "-k", kexpr, "-p", "no:cacheprovider"],
                           env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE))
        assert r.returncode == 0, "%s under %s:\n%s\n%s" % (fname, extra_env, r.stdout[-3000:], r.stderr[-2000:])
        assert " passed" in r.stdout, r.stdout[-500:]


def test_debug_library_really_poisons():
    """the hook is live in the debug build: with NaN garbage in LDS *and* a deliberately too-small check it still passes, i.e.
    the child loads libudecore_dbg.so (a missing debug library must not turn this file into a no-op)"""
    code = ("import os, ctypes; from universal_differential_equations_amd import _lib; L = _lib.load(); "
            "assert _lib.LIB_PATH.endswith('libudecore_dbg.so'), _lib.LIB_PATH; print('dbg ok')")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UDE_LIB_VARIANT="dbg"), capture_output=True, text=True,
                       cwd=os.path.dirname(HERE))
    assert r.returncode == 0 and "dbg ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("idx", range(len(SELECTION)), ids=[s[0][9:-3] for s in SELECTION])
def test_parity_with_garbage_registers(idx):
    # kind 3 (different garbage in every lane and register), what = 1 | 4: registers, also in front of the forward kernels
    _run([SELECTION[idx]], {"UDE_EXP_POISON": "3,5"})
