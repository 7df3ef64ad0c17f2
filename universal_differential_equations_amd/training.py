"""Host-side training loop around the fused loss/gradient kernels (the optimiser stays on the host).

    res1 = Optimization.solve(optprob, ADAM(0.1), callback=callback, maxiters=200)        scenario_1.jl:111-115
    res2 = Optimization.solve(optprob2, Optim.BFGS(initial_stepnorm=0.01), ...)           scenario_1.jl:117-118
    DiffEqFlux.sciml_train(loss, p, ADAM(0.01), cb=callback, maxiters=500)                seir_exposure.jl:160

`loss_grad(theta) -> (loss, grad)` is any callable (numpy or torch tensors; the device-resident ensemble keeps theta,
gradient and optimiser state in HBM, nothing crosses PCIe inside the loop).  ADAM is Optimisers.jl's rule
(eta, beta=(0.9,0.999), eps=eps(Float64)); the callback sees loss(theta_k) BEFORE the update, as upstream's does
(SURVEY App. A.6).  BFGS here is a plain inverse-Hessian BFGS with Armijo backtracking: it plays Optim.BFGS's role in the
scripts but is not a restatement of Optim.jl's HagerZhang line search.
"""
import numpy as np


def _xp(x):
    return __import__("torch") if type(x).__module__.startswith("torch") else np


def adam(loss_grad, theta, eta=0.1, beta=(0.9, 0.999), maxiters=200, callback=None, eps=None):
    xp = _xp(theta)
    eps = np.finfo(np.float64).eps if eps is None else eps
    theta = theta.clone() if xp is not np else np.array(theta, dtype=np.float64)
    m = xp.zeros_like(theta)
    v = xp.zeros_like(theta)
    b1t, b2t = beta
    losses = []
    for _ in range(maxiters):
        loss, g = loss_grad(theta)
        losses.append(float(loss))
        if callback is not None and callback(theta, losses[-1]):
            break
        m = beta[0] * m + (1 - beta[0]) * g
        v = beta[1] * v + (1 - beta[1]) * g * g
        theta = theta - eta * (m / (1 - b1t)) / (xp.sqrt(v / (1 - b2t)) + eps)
        b1t *= beta[0]
        b2t *= beta[1]
    return theta, losses


def bfgs(loss_grad, theta, initial_stepnorm=0.01, maxiters=1000, gtol=1e-8, callback=None, c1=1e-4):
    theta = np.array(theta, dtype=np.float64)
    n = theta.size
    f, g = loss_grad(theta)
    f, g = float(f), np.asarray(g, dtype=np.float64)
    H = np.eye(n) * (initial_stepnorm / max(np.linalg.norm(g, np.inf), 1e-300))
    losses = [f]
    for _ in range(maxiters):
        if callback is not None and callback(theta, f):
            break
        if np.linalg.norm(g, np.inf) < gtol:
            break
        d = -H @ g
        gd = g @ d
        if gd >= 0:                       # lost positive definiteness: restart
            H = np.eye(n) * (initial_stepnorm / max(np.linalg.norm(g, np.inf), 1e-300))
            d = -H @ g
            gd = g @ d
        a = 1.0
        for _ls in range(40):
            fn, gn = loss_grad(theta + a * d)
            fn = float(fn)
            if np.isfinite(fn) and fn <= f + c1 * a * gd:
                break
            a *= 0.5
        else:
            break
        gn = np.asarray(gn, dtype=np.float64)
        s, y = a * d, gn - g
        sy = s @ y
        if sy > 1e-12 * np.linalg.norm(s) * np.linalg.norm(y):
            rho = 1.0 / sy
            I = np.eye(n)
            H = (I - rho * np.outer(s, y)) @ H @ (I - rho * np.outer(y, s)) + rho * np.outer(s, s)
        theta, f, g = theta + s, fn, gn
        losses.append(f)
    return theta, losses


# ---------------------------------------------------------------------------------------------------------------
# Multiple shooting: DiffEqFlux.multiple_shoot as the reference calls it
#     multiple_shoot(p, Xn, t, prob_nn, loss, Vern7(), group_size; continuity_term)        hudson_bay.jl:108-118
# Published algorithm (DiffEqFlux, multiple_shooting.jl): the save grid is cut into overlapping groups
#     ranges = [i : min(datasize, i + group_size - 1)  for i in 1 : group_size - 1 : datasize - 1]
# every group is solved from the DATA point at its first time, and
#     loss = sum_i loss_function(data[:, rg_i], pred_i) + continuity_term * sum_{i>1} sum(abs, pred_{i-1}[:, end] - data[:, first(rg_i)])
# Here the groups are ONE ensemble per distinct relative time grid (the right-hand sides on this path are autonomous,
# so a group is integrated on [0, t_last - t_first]); the gradient with respect to p goes through the fused adjoint
# pullback with the cotangent 2 (pred - data) + continuity_term * sign(pred_end - data_next) on the last point.
# ---------------------------------------------------------------------------------------------------------------
def group_ranges(datasize, group_size):
    if group_size < 2 or group_size > datasize:
        raise ValueError("group_size must lie in [2, datasize]")
    return [range(i, min(datasize - 1, i + group_size - 1) + 1) for i in range(0, datasize - 1, group_size - 1)]


class EngineBackend:
    """solve / pullback through libudecore (MI355X)."""

    def __init__(self, prob, alg, **solve_kw):
        from . import sciml as U
        self.U, self.prob, self.alg, self.kw = U, prob, alg, solve_kw

    def solve(self, p, u0s, tau):
        U = self.U
        ens = U.EnsembleProblem(U.remake(self.prob, u0=u0s[0], tspan=(0.0, float(tau[-1])), p=p), u0s)
        return np.asarray(U.solve(ens, self.alg, saveat=tau, **self.kw).u)

    def pullback(self, p, u0s, tau, cot):
        U = self.U
        ens = U.EnsembleProblem(U.remake(self.prob, u0=u0s[0], tspan=(0.0, float(tau[-1])), p=p), u0s)
        return np.asarray(U.adjoint_pullback(ens, self.alg, cot, saveat=tau, **self.kw).grad_theta)


def multiple_shoot(p, ode_data, tsteps, backend, group_size, continuity_term=100.0, want_grad=True):
    """ode_data: (n, datasize) as in the scripts.  Returns (loss, grad or None, group_predictions [list of (n, len) arrays])."""
    p = np.asarray(p, dtype=np.float64)
    X = np.asarray(ode_data, dtype=np.float64)
    t = np.asarray(tsteps, dtype=np.float64)
    n, T = X.shape
    ranges = group_ranges(T, group_size)
    # groups with the same relative grid are solved together
    buckets = {}
    for gi, rg in enumerate(ranges):
        tau = t[list(rg)] - t[rg[0]]
        buckets.setdefault(tuple(np.round(tau, 12)), []).append(gi)
    preds = [None] * len(ranges)
    loss = 0.0
    grad = np.zeros_like(p) if want_grad else None
    for key, gis in buckets.items():
        tau = np.array(key)
        u0s = np.stack([X[:, ranges[gi][0]] for gi in gis])
        P = backend.solve(p, u0s, tau)                                   # (groups, len, n)
        cot = np.zeros_like(P)
        for b, gi in enumerate(gis):
            rg = list(ranges[gi])
            D = X[:, rg].T                                               # (len, n)
            preds[gi] = P[b].T
            loss += float(np.sum((D - P[b]) ** 2))
            cot[b] = 2.0 * (P[b] - D)
            if gi + 1 < len(ranges):                                     # continuity with the next group's initial data point
                nxt = X[:, ranges[gi + 1][0]]
                loss += continuity_term * float(np.sum(np.abs(P[b][-1] - nxt)))
                cot[b][-1] += continuity_term * np.sign(P[b][-1] - nxt)
        if want_grad:
            grad += backend.pullback(p, u0s, tau, cot)
    return loss, grad, preds
