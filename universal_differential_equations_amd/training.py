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
