"""Cost-ordered adjoint launch (round 6; SURVEY.md 7 "sort / bucket trajectories by expected cost"; csrc/udecore.hip sort kernels,
KParams::perm): a multi-round ensemble on the lane-group kernels runs its backward solves with the wavefronts filled in the order of
what the members' backward solves cost in the previous call.  What must hold: every per-trajectory number is the one the identity order gives (bit for bit, and therefore
the oracle's), the gradient -- now a sum of N per-trajectory rows in trajectory order -- agrees with the oracle to the summation
tolerance, and two runs give the same bits although the counting sort's atomics fill a bucket in any order.

`UDE_COST_SORT` is read once per process, so every configuration runs in a child process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
import _oracle as O
N, alg_name, sense = int(sys.argv[1]), sys.argv[2], sys.argv[3]
g = json.load(open(os.path.join(%r, "tests", "golden", "Scenario_1_recovery_0.005.json")))
th = np.array(g["initial_parameters"])
rng = np.random.default_rng(99)
u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
t = np.arange(31) * 0.1
truth, _, rc = O.solve_ensemble(O.lv_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 3.0], [1.3, 0.9, 0.8, 1.8], t)
data = truth + 5e-3 * truth.mean(axis=1, keepdims=True) * rng.standard_normal((N, 31, 2))
data[7] *= 1.5                      # a few members far from their data: the expensive backward solves
data[N // 2] *= 0.4
ens = U.EnsembleProblem(U.ODEProblem(models.ude_dynamics(), u0[0], (0.0, 3.0), th), u0)
alg = U.Vern7() if alg_name == "vern7" else U.Tsit5()
kw = dict(sensealg=U.FastInterpolatingAdjoint()) if sense == "fast" else {}
r = U.loss_and_gradient(ens, alg, data, saveat=t, abstol=1e-6, reltol=1e-6, **kw)
r2 = U.loss_and_gradient(ens, alg, data, saveat=t, abstol=1e-6, reltol=1e-6, **kw)
r3 = U.loss_and_gradient(ens, alg, data, saveat=t, abstol=1e-6, reltol=1e-6, **kw)
np.savez(sys.argv[4], stats=r.stats, grad_u0=r.grad_u0, lpt=r.loss_per_traj, grad=r.grad_theta, grad2=r2.grad_theta, grad3=r3.grad_theta, loss=r.loss, retcode=r.retcode,
         stats2=r2.stats, grad_u0_2=r2.grad_u0, lpt2=r2.loss_per_traj, loss2=r2.loss, u2=r2.u, u=r.u, kernel_ms=np.array(r.kernel_ms))
"""


def run_child(tmp_path, tag, N, alg, sense, sort):
    out = str(tmp_path / ("%s.npz" % tag))
    env = dict(os.environ, UDE_COST_SORT=str(sort))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, ROOT, ROOT), str(N), alg, sense, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("alg,sense,N", [("tsit5", "adjoint", 1501), ("vern7", "adjoint", 700), ("tsit5", "fast", 700)])
def test_cost_ordered_launch_changes_no_trajectory_and_is_deterministic(tmp_path, alg, sense, N):
    a = run_child(tmp_path, "identity", N, alg, sense, 0)
    b = run_child(tmp_path, "sorted", N, alg, sense, 1)
    assert (a["retcode"] == 0).all() and (b["retcode"] == 0).all()
    for key in ("stats", "grad_u0", "lpt", "u"):     # per trajectory: the same solves, wherever they ran
        assert np.array_equal(a[key], b[key]), key
    # the FIRST call of the mode has no costs to sort by (identity order through the mode's kernels); the second and third are ordered by the
    # previous call's backward attempts -- forward AND backward kernel: every member's numbers are still its own
    for key, key2 in (("stats", "stats2"), ("grad_u0", "grad_u0_2"), ("lpt", "lpt2"), ("u", "u2")):
        assert np.array_equal(a[key], b[key2]), key2
    assert a["loss"] == b["loss"] == b["loss2"]
    gn = np.linalg.norm(a["grad"])
    assert gn > 0 and np.linalg.norm(a["grad"] - b["grad"]) < 1e-12 * gn        # another association of the same N numbers per parameter
    assert np.array_equal(b["grad"], b["grad2"]) and np.array_equal(b["grad"], b["grad3"])   # identity order, sorted order, sorted again: same bits (the row sum does not see the order)
    assert np.array_equal(a["grad"], a["grad2"])


def test_cost_ordered_launch_agrees_with_the_oracle(tmp_path):
    """... and with the oracle itself (N = 300: the oracle finishes in seconds)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    N = 300
    b = run_child(tmp_path, "sorted", N, "tsit5", "adjoint", 1)
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "Scenario_1_recovery_0.005.json")))
    th = np.array(g["initial_parameters"])
    rng = np.random.default_rng(99)
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))
    t = np.arange(31) * 0.1
    truth, _, rc = O.solve_ensemble(O.lv_true(), O.opts(O.VERN7, 1e-12, 1e-12), u0, [0.0, 3.0], [1.3, 0.9, 0.8, 1.8], t)
    data = truth + 5e-3 * truth.mean(axis=1, keepdims=True) * rng.standard_normal((N, 31, 2))
    data[7] *= 1.5
    data[N // 2] *= 0.4
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [0.0, 3.0], th, t, data, nthreads=8)
    assert np.array_equal(b["stats"], ref["stats"]) and np.array_equal(b["grad_u0"], ref["grad_u0"]) and np.array_equal(b["lpt"], ref["loss_per_traj"])
    assert np.array_equal(b["stats2"], ref["stats"]) and np.array_equal(b["grad_u0_2"], ref["grad_u0"]) and np.array_equal(b["lpt2"], ref["loss_per_traj"])   # (the sorted call)
    assert np.array_equal(b["u2"], ref["u"])
    assert np.linalg.norm(b["grad"] - ref["grad_theta"]) < 1e-12 * np.linalg.norm(ref["grad_theta"])
