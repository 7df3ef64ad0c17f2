"""Host-side optimiser loops (scenario_1.jl:111-118): the corner cases of the ADAM hand-over and of the
restated HagerZhang line search that no stored trajectory of the reference exercises."""
import numpy as np

from universal_differential_equations_amd import training


def quad(theta):
    return float(theta @ theta), 2.0 * theta


def test_adam_evaluated_result_is_the_last_point_the_objective_saw():
    th0 = np.array([1.0, -2.0])
    # the callback stops the very first iteration: the evaluated point is the start (was an UnboundLocalError)
    th, losses = training.adam(quad, th0, maxiters=5, callback=lambda t, l: True, result="evaluated")
    assert np.array_equal(th, th0) and losses == [5.0]
    # ... and at iteration k > 0 it is theta_k, the argument of the last callback -- not theta_{k-1}
    seen = []
    def cb(t, l):
        seen.append(np.array(t))
        return len(seen) == 3
    th, losses = training.adam(quad, th0, maxiters=10, callback=cb, result="evaluated")
    assert len(losses) == 3 and np.array_equal(th, seen[-1]) and abs(losses[-1] - float(th @ th)) < 1e-15
    # without an early stop: theta_{maxiters-1} (what Optimization.solve hands to BFGS: stored losses[199] == losses[200])
    th, losses = training.adam(quad, th0, maxiters=4, result="evaluated")
    assert abs(losses[-1] - float(th @ th)) < 1e-15


def test_hagerzhang_flat_slopes_and_non_finite_objectives_do_not_raise():
    # equal slopes at both ends of the bracket (a linear objective): secant() has nothing to divide by
    a, p = training.hagerzhang(lambda al: (1.0 - al, -1.0), 1.0, 1.0, -1.0)
    assert np.isfinite(a) and np.isfinite(p)
    # an objective that is non-finite for every positive step: alpha = 0, bounded work
    calls = []
    def bad(al):
        calls.append(al)
        return float("inf"), float("nan")
    a, p = training.hagerzhang(bad, 1.0, 3.0, -1.0)
    assert a == 0.0 and p == 3.0 and len(calls) <= training._HZ.iterfinitemax + 1
    # non-finite beyond alpha = 2 while the bracket phase keeps expanding (a descending line never brackets): every expansion
    # backs off to a finite point in bounded work and the search ends with a finite step instead of looping forever
    calls.clear()
    def wall(al):
        calls.append(al)
        return (float("inf"), float("nan")) if al > 2.0 else (1.0 - 0.4 * al, -0.4)
    a, p = training.hagerzhang(wall, 1.0, 1.0, -0.4)
    assert 0.0 <= a <= 2.0 and np.isfinite(p) and len(calls) < 60 * 64


def test_hagerzhang_expansion_against_a_wall_of_non_finite_values_shrinks_alphamax():
    """LineSearches.jl's bracket phase when the objective is not finite beyond some step (a solve that diverges for a too-long step):
    the expansion c <- 5 c lands behind the wall, is pulled back by bisection towards the last good point, and EVERY non-finite point
    found becomes the search's alphamax: a later expansion is clamped to it (upstream's `alphamax = c` inside the loop; without it the
    next expansion probes 5 c again -- other trial points, another BFGS trajectory).  phi(a) = -a for a < 3.9, +inf behind: from c = 1
    upstream's sequence of trial points is 1, 5 (inf), 3 (finite, downhill), then min(15, alphamax = 5) = 5 (inf), 4 (inf), 3.5, ...;
    the search must return a finite point below the wall with a lower value than every earlier one and never evaluate beyond 5."""
    from universal_differential_equations_amd.training import hagerzhang
    seen = []

    def phidphi(a):
        seen.append(a)
        return (-a, -1.0) if a < 3.9 else (float("inf"), float("nan"))

    alpha, phi = hagerzhang(phidphi, 1.0, 0.0, -1.0)
    assert seen[:3] == [1.0, 5.0, 3.0]
    assert seen[3] == 5.0 and seen[4] == 4.0 and seen[5] == 3.5          # clamped to the shrunken alphamax, then bisected towards 3
    assert max(seen) == 5.0                                              # never beyond the first point found non-finite
    assert np.isfinite(phi) and 3.0 <= alpha < 3.9 and phi == -alpha
