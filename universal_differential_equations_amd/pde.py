"""Host-side mirror of the NeuralNetDiffEq surface that `highdim_pde/lambaem.jl` uses:

    prob   = TerminalPDEProblem(g, f, mu, sigma, x0, tspan)                          lambaem.jl:18
    u0     = Flux.Chain(Dense(d,hls,relu), Dense(hls,hls,relu), Dense(hls,1))        lambaem.jl:23-25
    sg     = Flux.Chain(Dense(d+1,hls,relu), ... , Dense(hls,d))                     lambaem.jl:27-30
    pdealg = NNPDENS(u0, sg, opt=Flux.ADAM(0.03))                                    lambaem.jl:21,31
    ans    = solve(prob, pdealg, verbose=true, maxiters=500, trajectories=m,
                   alg=LambaEM(), pabstol=1f-2, reltol=1e-4, abstol=1e-4)            lambaem.jl:33-34

g, f, mu, sigma are Julia closures upstream; here the problem is the declarative `hjb(lambda_)` family of the script
(g(X) = log(0.5 + 0.5|X|^2), f = -lambda |sigma^T grad u|^2, mu = 0, sigma = sqrt(2) I).  Every loss/gradient evaluation
is one call of libudecore's fused forward + backward kernels (ude_hjb_loss_grad_dev); parameters, gradient and ADAM
state stay in HBM (torch tensors), nothing crosses PCIe inside the training loop except the scalar loss the callback prints.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HjbDesc, UdeError
from .sciml import Engine, _ptr


class LambaEM:
    """StochasticDiffEq.LambaEM(): adaptive Euler-Maruyama with Lamba's error estimate (lambaem.jl:33)"""
    adaptive = 1


class EM:
    """StochasticDiffEq.EM(): fixed-step Euler-Maruyama (needs dt)"""
    adaptive = 0


class HJB:
    """The terminal-value problem of lambaem.jl:12-17: g, f, mu, sigma as a declarative family"""

    def __init__(self, lam=1.0, sigma=float(np.sqrt(np.float32(2.0)))):
        self.lam, self.sigma = float(lam), float(sigma)


def hjb(lam=1.0):
    return HJB(lam)


class TerminalPDEProblem:
    """TerminalPDEProblem(g, f, mu, sigma, x0, tspan) with (g, f, mu, sigma) = an HJB family member"""

    def __init__(self, family, x0, tspan):
        assert isinstance(family, HJB)
        self.family, self.x0, self.tspan = family, np.asarray(x0, dtype=np.float32), (float(tspan[0]), float(tspan[1]))


class ADAM:
    """Flux.ADAM(eta, (0.9, 0.999)) as Flux 0.9 applies it (eps = 1e-8)"""

    def __init__(self, eta=0.001, beta=(0.9, 0.999)):
        self.eta, self.beta, self.eps = eta, beta, 1e-8


class NNPDENS:
    """NNPDENS(u0, sigma^T grad u; opt): the two chains are given by their sizes (d, hls) -- Flux.Chain(Dense(d,hls,relu),
    Dense(hls,hls,relu), Dense(hls,1)) and Flux.Chain(Dense(d+1,hls,relu), Dense(hls,hls,relu), Dense(hls,hls,relu), Dense(hls,d))"""

    def __init__(self, d, hls, opt=None):
        self.d, self.hls, self.opt = d, hls, opt or ADAM(0.001)

    def num_params(self):
        a, b = C.c_int32(0), C.c_int32(0)
        _lib.load().ude_hjb_num_params(self.d, self.hls, C.byref(a), C.byref(b))
        return a.value, b.value

    def init_params(self, rng):
        """Flux.Dense default initialisation: glorot_uniform weights, zero biases; Flux.params(u0, sg) order"""
        out = []
        d, h = self.d, self.hls
        for dims in ((d, h, h, 1), (d + 1, h, h, h, d)):
            for i in range(len(dims) - 1):
                fin, fout = dims[i], dims[i + 1]
                lim = np.sqrt(6.0 / (fin + fout))
                out.append(rng.uniform(-lim, lim, fin * fout))
                out.append(np.zeros(fout))
        return np.concatenate(out).astype(np.float32)


def make_desc(prob, pdealg, alg, abstol=1e-6, reltol=1e-3, dt=0.0, seed=0, maxiters_sde=0, max_steps=0, **kw):
    D = HjbDesc()
    D.d, D.hls, D.adaptive = pdealg.d, pdealg.hls, alg.adaptive
    D.maxiters, D.max_steps, D.seed = maxiters_sde, max_steps, seed
    D.lam, D.sigma, D.t0, D.t1 = prob.family.lam, prob.family.sigma, prob.tspan[0], prob.tspan[1]
    D.abstol, D.reltol, D.dt = abstol, reltol, dt
    for k, v in kw.items():
        setattr(D, k, v)
    return D


class DeviceBSDE:
    """One loss/gradient evaluation of loss_n_sde() (NNPDENS) for M trajectories with everything resident in HBM."""

    def __init__(self, prob, pdealg, alg, trajectories, device=None, **kw):
        import torch
        self.torch = torch
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.eng = Engine.get(dev.index or 0)
        self.D = make_desc(prob, pdealg, alg, **kw)
        self.M = int(trajectories)
        np0, np1 = pdealg.num_params()
        self.np = np0 + np1
        self.x0 = torch.tensor(prob.x0, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float64, device=dev)
        self.grad = torch.zeros(self.np, dtype=torch.float32, device=dev)
        self.u0 = torch.zeros(1, dtype=torch.float32, device=dev)
        self.uT = torch.zeros(self.M, dtype=torch.float32, device=dev)
        self.XT = torch.zeros((self.M, pdealg.d), dtype=torch.float32, device=dev)
        self.loss_traj = torch.zeros(self.M, dtype=torch.float64, device=dev)
        self.stats = torch.zeros((self.M, 4), dtype=torch.int64, device=dev)
        self.retcode = torch.zeros(self.M, dtype=torch.int32, device=dev)

    def loss_grad(self, theta, it=0, want_grad=True, check_store=True):
        e = self.eng
        e.set_stream(self.torch.cuda.current_stream().cuda_stream)
        assert theta.dtype == self.torch.float32 and theta.numel() == self.np
        while True:
            e.check(e.L.ude_hjb_loss_grad_dev(e.h, C.byref(self.D), self.M, _ptr(self.x0), _ptr(theta), it, _ptr(self.loss),
                                              _ptr(self.grad) if want_grad else None, _ptr(self.u0), _ptr(self.uT), _ptr(self.XT),
                                              _ptr(self.loss_traj), _ptr(self.stats), _ptr(self.retcode)))
            if not (want_grad and check_store):
                break
            # a trajectory that outgrew the automatic accepted-step store: the library has grown it, repeat the call
            nfail, grown = C.c_int32(0), C.c_int32(0)
            e.check(e.L.ude_hjb_last_failures(e.h, _ptr(self.retcode), self.M, C.byref(nfail), C.byref(grown)))
            if not grown.value:
                break
        return self.loss, self.grad

    def kernel_ms(self):
        f, b = C.c_float(0), C.c_float(0)
        self.eng.check(self.eng.L.ude_hjb_last_kernel_ms(self.eng.h, C.byref(f), C.byref(b)))
        return f.value, b.value


def loss_and_gradient(prob, pdealg, alg, theta, trajectories, it=0, want_grad=True, device=0, allow_failures=False, **kw):
    """numpy in / out through the host-buffer C entry point (what a Julia ccall binds)"""
    eng = Engine.get(device)
    D = make_desc(prob, pdealg, alg, **kw)
    M = int(trajectories)
    theta = np.ascontiguousarray(theta, dtype=np.float32)
    x0 = np.ascontiguousarray(prob.x0, dtype=np.float32)
    loss = C.c_double(0)

    class R:
        pass
    r = R()
    r.grad = np.zeros(theta.size, dtype=np.float32) if want_grad else None
    u0 = np.zeros(1, dtype=np.float32)
    r.uT = np.zeros(M, dtype=np.float32)
    r.XT = np.zeros((M, pdealg.d), dtype=np.float32)
    r.loss_traj = np.zeros(M)
    r.stats = np.zeros((M, 4), dtype=np.int64)
    r.retcode = np.zeros(M, dtype=np.int32)
    rc = eng.L.ude_hjb_loss_grad(eng.h, C.byref(D), M, _ptr(x0), _ptr(theta), it, C.byref(loss), _ptr(r.grad), _ptr(u0), _ptr(r.uT),
                                 _ptr(r.XT), _ptr(r.loss_traj), _ptr(r.stats), _ptr(r.retcode))
    eng.check(rc, allow_traj=allow_failures)
    r.loss, r.u0 = loss.value, float(u0[0])
    f, b = C.c_float(0), C.c_float(0)
    eng.L.ude_hjb_last_kernel_ms(eng.h, C.byref(f), C.byref(b))
    r.kernel_ms = (f.value, b.value)
    return r


def solve(prob, pdealg, theta0, verbose=False, maxiters=300, trajectories=100, alg=None, pabstol=1e-6, callback=None, device=None,
          **kw):
    """solve(prob::TerminalPDEProblem, pdealg::NNPDENS; verbose, maxiters, trajectories, alg, pabstol, abstol, reltol)
    (lambaem.jl:33-34): Flux.train! with the callback evaluating the loss first and stopping below pabstol; returns
    (u0(x0) after training -- the PDE solution estimate the script compares with the analytical value --, theta, losses)."""
    import torch
    alg = alg or LambaEM()
    bs = DeviceBSDE(prob, pdealg, alg, trajectories, device=device, **kw)
    dev = bs.x0.device
    theta = torch.tensor(np.asarray(theta0, dtype=np.float32), device=dev)
    opt = pdealg.opt
    m = torch.zeros_like(theta)
    v = torch.zeros_like(theta)
    b1p, b2p = opt.beta
    losses = []
    for it in range(maxiters):
        loss, g = bs.loss_grad(theta, it=it)
        lval = float(loss.item())
        losses.append(lval)
        if verbose:
            print("Current loss is: %g" % lval)
        if callback is not None and callback(it, lval, float(bs.u0.item())):
            break
        if not np.isfinite(lval):
            raise UdeError(_lib.UDE_ERR_TRAJECTORY, "a trajectory failed (retcodes %s)" % sorted(set(bs.retcode.tolist())))
        if lval < pabstol:
            break
        # Flux.ADAM (0.9): mt = b1 mt + (1-b1) g; vt = b2 vt + (1-b2) g^2; theta -= mt/(1-b1^t) / (sqrt(vt/(1-b2^t)) + eps) * eta
        m.mul_(opt.beta[0]).add_(g, alpha=1 - opt.beta[0])
        v.mul_(opt.beta[1]).addcmul_(g, g, value=1 - opt.beta[1])
        theta = theta - (m / (1 - b1p)) / (torch.sqrt(v / (1 - b2p)) + opt.eps) * opt.eta
        b1p *= opt.beta[0]
        b2p *= opt.beta[1]
    bs.loss_grad(theta, it=maxiters, want_grad=False)
    return float(bs.u0.item()), theta, losses


def u_analytical(x0, lam, T, rng, MC=10 ** 5):
    """the reference value the script compares with (lambaem.jl:37-41): -(1/lambda) log(mean(exp(-lambda g(x + sqrt(2) |T - t| W))))"""
    x0 = np.asarray(x0, dtype=np.float64)
    acc = 0.0
    for _ in range(10):
        W = rng.standard_normal((MC // 10, x0.size))
        X = x0[None, :] + np.sqrt(2.0) * abs(T) * W
        acc += np.exp(-lam * np.log(0.5 + 0.5 * (X ** 2).sum(axis=1))).sum()
    return -(1.0 / lam) * np.log(acc / MC)
