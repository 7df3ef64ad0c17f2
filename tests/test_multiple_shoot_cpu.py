"""Host orchestration of SURVEY 8(f) N3: DiffEqFlux.multiple_shoot as called in hudson_bay.jl:108-118, checked on the
CPU with the oracle as the solve / pullback backend (the product binds the same two calls to libudecore)."""
import numpy as np

import _oracle as O
from universal_differential_equations_amd import training


class OracleBackend:
    def __init__(self, model, opts):
        self.m, self.o = model, opts

    def solve(self, p, u0s, tau):
        out, st, rc = O.solve_ensemble(self.m, self.o, u0s, [0.0, float(tau[-1])], p, tau)
        assert (rc == 0).all()
        return out

    def pullback(self, p, u0s, tau, cot):
        r = O.vjp_ensemble(self.m, self.o, u0s, [0.0, float(tau[-1])], p, tau, cot)
        return r["grad_theta"]


def test_group_ranges_match_diffeqflux():
    # datasize 21, group_size 5 (hudson_bay.jl:106): 1:5, 5:9, 9:13, 13:17, 17:21 (1-based) -> overlapping by one point
    rg = training.group_ranges(21, 5)
    assert [(r[0], r[-1]) for r in rg] == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20)]
    rg = training.group_ranges(10, 4)   # last group shorter: 1:4, 4:7, 7:10
    assert [(r[0], r[-1]) for r in rg] == [(0, 3), (3, 6), (6, 9)]
    rg = training.group_ranges(11, 4)   # 1:4, 4:7, 7:10, 10:11
    assert [(r[0], r[-1]) for r in rg] == [(0, 3), (3, 6), (6, 9), (9, 10)]


def test_multiple_shoot_loss_and_gradient(golden):
    g = golden("Scenario_1_recovery_0.005")
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2).T        # (2, 31)
    t = np.array(g["t"])
    th = np.array(g["initial_parameters"])
    be = OracleBackend(O.lv_ude_s1(), O.opts(O.VERN7, 1e-8, 1e-8))
    loss, grad, preds = training.multiple_shoot(th, X, t, be, 5, continuity_term=200.0)
    # the same loss, composed by hand group by group
    ref = 0.0
    rgs = training.group_ranges(31, 5)
    for i, rg in enumerate(rgs):
        rg = list(rg)
        out, st, rc = O.solve_ensemble(O.lv_ude_s1(), O.opts(O.VERN7, 1e-8, 1e-8), X[:, rg[0]], [t[rg[0]], t[rg[-1]]], th, t[rg])
        ref += np.sum((X[:, rg].T - out[0]) ** 2)
        if i + 1 < len(rgs):
            ref += 200.0 * np.sum(np.abs(out[0][-1] - X[:, rgs[i + 1][0]]))
        assert np.allclose(preds[i].T, out[0], rtol=1e-9, atol=1e-12)
    assert abs(loss - ref) < 1e-9 * abs(ref)
    # gradient: directional finite differences of the multiple-shooting loss
    rng = np.random.default_rng(0)
    for _ in range(2):
        d = rng.normal(size=th.size)
        d /= np.linalg.norm(d)
        h = 1e-6
        lp = training.multiple_shoot(th + h * d, X, t, be, 5, continuity_term=200.0, want_grad=False)[0]
        lm = training.multiple_shoot(th - h * d, X, t, be, 5, continuity_term=200.0, want_grad=False)[0]
        fd = (lp - lm) / (2 * h)
        assert abs(fd - grad @ d) < 1e-5 * max(1.0, abs(fd))
