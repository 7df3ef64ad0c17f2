"""The N1 oracle (highdim_pde/lambaem.jl restated, oracle/sde_oracle.*) checked against what CAN be pinned on a CPU:
the Random123 known-answer vectors of Philox4x32-10, the accuracy of the fixed-order sin/cos kernels, the statistics of
the normals and of the Brownian increments that rejection sampling with memory hands to the stepper, and the
reverse-sweep gradient against central finite differences.  (The reference ships no artifact of this script and its
Julia RNG stream is not reproducible: parity with upstream is unpinned, see oracle/sde_oracle.h.)"""
import numpy as np

import _sde_oracle as S


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors: philox4x32 10
    assert S.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert S.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert S.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_sincos_kernel_accuracy():
    rng = np.random.default_rng(0)
    for u in np.concatenate([rng.uniform(0, 1, 2000), [0.0, 0.25, 0.5, 0.75, 1 - 2.0 ** -32, 2.0 ** -32]]):
        s, c = S.sincos2pi(float(u))
        a = 2 * np.longdouble("3.14159265358979323846264338327950288") * np.longdouble(u)   # 80-bit reference
        assert abs(s - np.sin(a)) < 4e-16 and abs(c - np.cos(a)) < 4e-16


def test_normals_moments_and_independence_of_counters():
    z = np.array([S.normals(11, 3, j, 0, 100) for j in range(1500)])
    assert abs(z.mean()) < 0.01 and abs(z.var() - 1) < 0.02 and abs((z ** 4).mean() - 3) < 0.1
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.08           # within a Box-Muller pair
    assert abs(np.corrcoef(z[:, 0], z[:, 2])[0, 1]) < 0.08           # across pairs of a Philox block
    a, b = S.normals(11, 3, 5, 0, 100), S.normals(11, 3, 5, 1, 100)  # another event / trajectory / iteration / seed
    assert not np.array_equal(a, b) and not np.array_equal(a, S.normals(11, 4, 5, 0, 100))
    assert not np.array_equal(a, S.normals(12, 3, 5, 0, 100)) and np.array_equal(a, S.normals(11, 3, 5, 0, 100))
    assert np.array_equal(S.normals(11, 3, 5, 0, 100)[:7], S.normals(11, 3, 5, 0, 7))  # prefix property


def test_parameter_counts_of_the_two_chains():
    # lambaem.jl:23-30 with d = 100, hls = 110
    assert S.num_params(100, 110) == (100 * 110 + 110 + 110 * 110 + 110 + 110 + 1, 101 * 110 + 110 + 2 * (110 * 110 + 110) + 110 * 100 + 100)


def test_gradient_matches_central_differences_f64_fixed_steps():
    rng = np.random.default_rng(1)
    d, H = 12, 16
    th = S.glorot_params(d, H, rng, np.float64) + 0.05 * rng.standard_normal(sum(S.num_params(d, H)))
    x0 = 0.3 * rng.standard_normal(d)
    D = S.desc(d=d, hls=H, adaptive=0, dt=0.05, seed=7)
    r = S.loss_grad(D, 6, x0, th, dtype=np.float64)
    assert r["ret"] == 0 and (r["stats"][:, 1] == 20).all()
    for i in rng.choice(th.size, 30, replace=False):
        e = 1e-6
        tp, tm = th.copy(), th.copy()
        tp[i] += e
        tm[i] -= e
        fd = (S.loss_grad(D, 6, x0, tp, want_grad=False, dtype=np.float64)["loss"] -
              S.loss_grad(D, 6, x0, tm, want_grad=False, dtype=np.float64)["loss"]) / (2 * e)
        assert abs(fd - r["grad"][i]) < 1e-6 * max(1.0, abs(fd))


def test_adaptive_solve_statistics_with_rejections():
    """Rejection sampling with memory must hand the stepper increments of the SAME Brownian path: whatever the
    accept/reject history, X_T - x0 = sigma W_T ~ N(0, sigma^2 T) per component, and the path is reproducible."""
    rng = np.random.default_rng(2)
    d, H = 100, 110
    th = S.glorot_params(d, H, rng)
    D = S.desc(abstol=0.1, reltol=0.1, seed=5, qmax=10.0)      # aggressive growth -> many rejections
    M = 48
    r = S.loss_grad(D, M, np.zeros(d), th, want_grad=False, nthreads=8)
    assert (r["retcode"] == 0).all() and r["stats"][:, 2].sum() > M       # rejections did happen
    v = (r["XT"].astype(np.float64) ** 2).mean() / D.sigma ** 2
    assert abs(v - 1.0) < 0.06, v                                          # 4800 samples: std of the estimate 0.02
    r2 = S.loss_grad(D, M, np.zeros(d), th, want_grad=False, nthreads=3)
    assert np.array_equal(r["XT"], r2["XT"]) and np.array_equal(r["stats"], r2["stats"]) and r["loss"] == r2["loss"]
    # u_T = u0 + sum(lambda |z|^2 dt + z . dW): recomputed from the recorded path of one trajectory
    p = S.path(D, np.zeros(d), th, traj=3)
    assert p["n"] == r["stats"][3, 1] and np.isclose(p["t"][-1] + p["dt"][-1], 1.0)
    assert np.allclose(p["X"][-1] + np.float32(D.sigma) * p["dW"][-1], r["XT"][3], atol=1e-5)
    np_u0 = S.num_params(d, H)[0]
    u = 0.0
    for n in range(p["n"]):
        z = S.net(d, H, th[np_u0:], np.append(p["X"][n], p["t"][n])).astype(np.float64)
        u += D.lam * (z ** 2).sum() * p["dt"][n] + z @ p["dW"][n]
    assert abs(u - r["uT"][3]) < 1e-3 * max(1.0, abs(u))
