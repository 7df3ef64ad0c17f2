"""Host-side mirror of the SciML surface the reference scripts use for the hot path.

    prob  = ODEProblem(f, u0, tspan, p)                      scenario_1.jl:78, seir_exposure.jl:131
    _prob = remake(prob; u0, tspan, p)                       scenario_1.jl:83
    sol   = solve(_prob, Vern7(); saveat, abstol, reltol)    scenario_1.jl:84-87     -> Array(sol), sol.t, sol.destats
    concrete_solve(prob, alg, u0, p; saveat, sensealg)       seir_exposure.jl:138-140, Fisher-KPP-CNN.jl:136
    EnsembleProblem(prob; u0s) + solve(ens, alg, EnsembleMI355(); ...)   (SciML's ensemble slot; BASELINE C2/C3)

`f` is a declarative RHS descriptor from .models (closures cannot cross the C ABI).  All arithmetic runs in
libudecore.so on the GPU; numpy in/out uses the host-buffer C entry points, torch CUDA tensors use the
device-resident ones on torch's current stream.  Arrays follow Julia's layout: a solution is
(n, ns) per trajectory, an ensemble solution (N, ns, n) in C order == n x ns x N column-major.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ALG_TSIT5, ALG_VERN7, LaunchOpts, ModelDesc, NSTATS, RETCODES, SolveOpts, UdeError


# ---- algorithm / sensealg / ensemble tags (dispatch types in Julia) -------------------------------------
class Tsit5:
    alg = ALG_TSIT5
    order = 5


class Vern7:
    alg = ALG_VERN7
    order = 7


class InterpolatingAdjoint:
    """sensealg = InterpolatingAdjoint(autojacvec = ReverseDiffVJP())  (seir_exposure.jl:140).
    autojacvec is accepted and ignored: the VJP is hand-derived inside the fused kernel.
    checkpointing = True selects UDE_SENSE_INTERPOLATING_ADJOINT_CHECKPOINTED: the forward pass stores (t, dt, u) of every accepted
    step only, the adjoint kernel recomputes a step's stages when it enters its interval -- results bit-identical to the dense
    store, 1 / (1 + stages) of its memory.  Instances: the Fisher-KPP UDEs with Tsit5 (the stiff 1024-point variant is what needs
    it); anything else fails with UDE_ERR_UNSUPPORTED."""

    def __init__(self, autojacvec=None, checkpointing=False):
        self.checkpointing = bool(checkpointing)


class ReverseDiffVJP:
    pass


class ForwardDiffSensitivity:
    """sensealg = ForwardDiffSensitivity()  (scenario_1.jl:86, scenario_2.jl:108, scenario_3.jl:124, hudson_bay.jl:102):
    discretise-then-optimise.  Here: the exact reverse-mode derivative of the discrete Tsit5/Vern7 map of the primal
    solve with its step sequence frozen -- one VJP per stage, no second adaptive solve."""

    def __init__(self, convert_tspan=None):
        pass


class FastInterpolatingAdjoint:
    """SURVEY.md 8(b) `fast` mode (UDE_SENSE_INTERPOLATING_ADJOINT_FAST): the interpolating adjoint with only lambda under
    error control; the parameter cotangent rides along as a quadrature on the accepted steps.  NOT an upstream sensealg
    (upstream's InterpolatingAdjoint keeps the parameter cotangent in the error norm, which is the default here): an
    opt-in that drops the per-parameter error estimate (its accumulators and IEEE divisions): 5-12 % less time per gradient
    on the reference's problem sizes, not fewer steps (the parity mode's RMS over n + np components dilutes lambda's error);
    gradients agree to the tolerance."""


class EnsembleMI355:
    """ensemble algorithm tag: trajectories run as lane groups of the fused HIP kernels"""

    def __init__(self, lanes_per_traj=0, max_dense_steps=0):
        self.lanes_per_traj, self.max_dense_steps = lanes_per_traj, max_dense_steps


class DEStats:
    def __init__(self, row):
        (self.nf, self.naccept, self.nreject, self.nf_lazy, self.nf_bwd, self.naccept_bwd, self.nreject_bwd,
         self.nf_fwd_lazy_adj) = (int(x) for x in row)

    def __repr__(self):
        return "DEStats(nf=%d, naccept=%d, nreject=%d)" % (self.nf, self.naccept, self.nreject)


# ---- engine context ---------------------------------------------------------------------------------------
class Engine:
    """One ude_ctx (device + stream).  Created lazily per device."""

    _by_device = {}
    _lock = __import__("threading").Lock()

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.ude_create(device, C.byref(h))
        if rc != 0:
            raise UdeError(rc, "ude_create(device=%d) failed (no MI355X visible?)" % device)
        self.h = h
        self.device = device

    @classmethod
    def get(cls, device=0):
        """one context per (device, host thread), as include/udecore.h requires"""
        import threading
        key = (device, threading.get_ident())
        with cls._lock:
            if key not in cls._by_device:
                cls._by_device[key] = Engine(device)
            return cls._by_device[key]

    def close(self):
        if self.h is not None:
            self.L.ude_destroy(self.h)
            self.h = None

    @classmethod
    def close_all(cls):
        with cls._lock:
            for e in cls._by_device.values():
                e.close()
            cls._by_device.clear()

    def failures(self, retcode_dev=None, N=0):
        """(number of failed trajectories, capacity grown?) of the most recent gradient call; blocks on the stream"""
        nf, gr = C.c_int32(0), C.c_int32(0)
        self.check(self.L.ude_last_failures(self.h, _ptr(retcode_dev), N, C.byref(nf), C.byref(gr)))
        return nf.value, bool(gr.value)

    def check(self, rc, allow_traj=False):
        if rc != 0 and not (allow_traj and rc == _lib.UDE_ERR_TRAJECTORY):
            raise UdeError(rc, self.L.ude_last_error(self.h).decode())
        return rc

    def set_launch(self, lanes_per_traj=0, max_dense_steps=0, waves_per_simd=0):
        lo = LaunchOpts(lanes_per_traj, 0, max_dense_steps, waves_per_simd)
        self.check(self.L.ude_set_launch_opts(self.h, C.byref(lo)))

    def set_stream(self, stream_ptr):
        self.check(self.L.ude_set_stream(self.h, C.c_void_p(stream_ptr)))

    def kernel_ms(self):
        f, b = C.c_float(0), C.c_float(0)
        self.check(self.L.ude_last_kernel_ms(self.h, C.byref(f), C.byref(b)))
        return f.value, b.value

    def set_trace(self, traj, cap=512):
        self._trace_cap = cap
        self.check(self.L.ude_set_trace(self.h, traj, cap))

    def get_trace(self):
        out = np.zeros((2, self._trace_cap, 5))
        self.check(self.L.ude_get_trace(self.h, out.ctypes.data))
        return out

    def math(self, op, x, y=None):
        """ARITH-SPEC primitive `op` on the device: 0 fastpow, 1 exp, 2 tanh, 3 log10, 4 pow10, 5 sqrt, 6 div, 7 fma(x,y,x)"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(np.broadcast_to(1.0 if y is None else y, x.shape), dtype=np.float64)
        out = np.empty_like(x)
        self.check(self.L.ude_math_dev(self.h, op, x.size, x.ctypes.data, y.ctypes.data, out.ctypes.data))
        return out

    def fastpow(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(np.broadcast_to(y, x.shape), dtype=np.float64)
        out = np.empty_like(x)
        self.check(self.L.ude_fastpow_dev(self.h, x.size, x.ctypes.data, y.ctypes.data, out.ctypes.data))
        return out


def _opts(alg, abstol=None, reltol=None, dtmax=None, dt=None, maxiters=None, sensealg=None, **kw):
    o = SolveOpts()
    o.alg = alg.alg if not isinstance(alg, int) else alg
    o.sensealg = (1 if isinstance(sensealg, ForwardDiffSensitivity) else 2 if isinstance(sensealg, FastInterpolatingAdjoint)
                  else 3 if getattr(sensealg, "checkpointing", False) else 0)
    o.abstol = abstol or 0.0
    o.reltol = reltol or 0.0
    o.dtmax = dtmax or 0.0
    o.dt0 = dt or 0.0
    o.maxiters = int(maxiters or 0)
    for k in ("qmin", "qmax", "gamma", "qoldinit", "beta1", "beta2"):
        setattr(o, k, kw.pop(k, 0.0) or 0.0)
    if kw:
        raise TypeError("unsupported solve keyword(s): %s" % sorted(kw))
    return o


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


def _np(x, dtype=np.float64):
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


def _rt(f):
    """numpy scalar type of a problem: ude_model_desc.dtype = 1 -> every real-valued array of the call is Float32
    (scenario_3.jl:26-57,121-126; hudson_bay.jl:77-104); tspan stays a pair of host doubles"""
    return np.float32 if f.dtype == 1 else np.float64


# ---- problems and solutions ------------------------------------------------------------------------------
class ODEProblem:
    """ODEProblem(f, u0, tspan, p; saveat=...) -- f is a ModelDesc from .models"""

    def __init__(self, f, u0, tspan, p=None, **kwargs):
        assert isinstance(f, ModelDesc), "f must be an RHS descriptor from universal_differential_equations_amd.models"
        self.f, self.u0, self.tspan, self.p, self.kwargs = f, u0, (float(tspan[0]), float(tspan[1])), p, kwargs


def remake(prob, u0=None, tspan=None, p=None):
    """remake(prob; u0, tspan, p)  (scenario_1.jl:83)"""
    return ODEProblem(prob.f, prob.u0 if u0 is None else u0, prob.tspan if tspan is None else tspan,
                      prob.p if p is None else p, **prob.kwargs)


class EnsembleProblem:
    """EnsembleProblem(prob; u0s[, tspans][, ps]): N trajectories sharing prob.p (theta); u0s is (N, n).
    tspans (N, 2): every member has its own time span -- prob_func = remake(prob; u0 = X[:, i0], tspan = (T[1], T[end]))
    per shooting segment (scenario_2.jl:104-124); pass the members' save grids as a 2-D saveat (N, ns).
    ps (N, np): every member has its OWN parameter vector -- prob_func = remake(prob; u0, p = p_i): the 500 independent
    recoveries of LotkaVolterra/run_loops.jl:55-62 as one ensemble; loss_and_gradient then returns grad_theta (N, np), one
    gradient per member (UDE_PT_THETA)."""

    def __init__(self, prob, u0s, tspans=None, ps=None):
        self.prob, self.u0s, self.tspans, self.ps = prob, u0s, tspans, ps


class ODESolution:
    def __init__(self, t, u, stats, retcode):
        self.t, self._u = t, u          # u: (ns, n)
        self.destats = DEStats(stats)
        self.retcode = RETCODES.get(int(retcode), "Unknown")

    @property
    def u(self):
        return [self._u[i] for i in range(self._u.shape[0])]

    def __array__(self, dtype=None, copy=None):
        a = self._u.T                    # Array(sol) is n x ns
        return a.astype(dtype) if dtype else a

    def __call__(self, tvals, deriv=0):
        """sol(t) / sol(t, Val{1})  (scenario_1.jl:46, scenario_2.jl:48, seir_exposure.jl:176: `DX = Array(solution(solution.t, Val{1}))`).
        A solve with `saveat` keeps no dense output upstream (dense = false), so the solution interpolates LINEARLY between
        the saved points [UP: SciMLBase LinearInterpolation, continuity = :left]: the value at t in (t[i-1], t[i]] is
        u[i-1] + theta (u[i] - u[i-1]); Val{1} is the slope (u[i] - u[i-1]) / (t[i] - t[i-1]) of that interval, the first
        interval serving t = t[1].  Returns n x len(tvals) like Array(sol(ts)).  (The model's exact derivative at the saved
        states is `rhs(f, Array(sol)', p)`.)"""
        t = np.asarray(self.t, dtype=float)
        u = self._u
        scalar = np.isscalar(tvals)
        tv = np.atleast_1d(np.asarray(tvals, dtype=float))
        if tv.min() < t[0] or tv.max() > t[-1]:
            raise ValueError("Solution interpolation cannot extrapolate past the saved time points")
        if len(t) < 2:
            raise ValueError("interpolation needs at least two saved points")
        out = np.empty((u.shape[1], tv.size))
        for k, x in enumerate(tv):
            i = int(np.searchsorted(t, x, side="left"))     # first index with t[i] >= x
            i = max(i, 1)
            dt = t[i] - t[i - 1]
            if deriv == 0:
                th = (x - t[i - 1]) / dt
                out[:, k] = u[i] if x == t[i] else u[i - 1] + th * (u[i] - u[i - 1])
            elif deriv == 1:
                out[:, k] = (u[i] - u[i - 1]) / dt
            else:
                raise ValueError("a linear interpolant has derivatives of order 0 and 1 only")
        return out[:, 0] if scalar else out


class EnsembleSolution:
    def __init__(self, t, u, stats, retcode):
        self.t, self.u, self.stats, self.retcodes = t, u, stats, retcode   # u: (N, ns, n)

    def __getitem__(self, j):
        t = self.t[j] if getattr(self.t, "ndim", 1) == 2 else self.t     # per-trajectory save grids
        return ODESolution(t, self.u[j], self.stats[j], self.retcodes[j])

    def __len__(self):
        return self.u.shape[0]


def _time_grids(prob, saveat, o):
    """(tspan array, save grid array, ns) for the C ABI; sets o.per_trajectory for per-member spans / grids"""
    ens = isinstance(prob, EnsembleProblem)
    base = prob.prob if ens else prob
    saveat = base.kwargs.get("saveat") if saveat is None else saveat
    tspans = getattr(prob, "tspans", None) if ens else None
    flags = 0
    if tspans is not None:
        tspan = _np(tspans)
        assert tspan.ndim == 2 and tspan.shape[1] == 2
        flags |= 1
    else:
        tspan = _np(base.tspan)
    if saveat is not None and not np.isscalar(saveat) and np.ndim(saveat) == 2:
        ts = _np(saveat)
        spans = tspan if tspan.ndim == 2 else np.repeat(tspan[None], ts.shape[0], axis=0)
        for j in range(ts.shape[0]):
            if spans[j, 1] > spans[j, 0] and (np.any(np.diff(ts[j]) <= 0) or ts[j, 0] < spans[j, 0] or ts[j, -1] > spans[j, 1]):
                raise ValueError("saveat[%d] must be strictly increasing and inside its tspan" % j)
        flags |= 2
        ns = ts.shape[1]
    else:
        if tspan.ndim == 2 and (saveat is None or np.isscalar(saveat)):
            raise ValueError("per-trajectory tspans need explicit save grids (2-D saveat, or one shared grid inside every span)")
        ts = _saveat_grid(saveat, (float(tspan[0]), float(tspan[1])) if tspan.ndim == 1 else (float(tspan[:, 0].max()), float(tspan[:, 1].min())))
        ns = len(ts)
    if ens and getattr(prob, "ps", None) is not None:
        flags |= 4   # UDE_PT_THETA
    o.per_trajectory = flags
    return tspan, ts, ns


def _saveat_grid(saveat, tspan):
    if saveat is None:
        raise NotImplementedError("solve without saveat (save every step) is not part of the hot path; pass saveat")
    if np.isscalar(saveat):
        n = int(np.floor((tspan[1] - tspan[0]) / saveat * (1 + 1e-12))) + 1
        ts = tspan[0] + saveat * np.arange(n)
        if ts[-1] < tspan[1] and abs(ts[-1] - tspan[1]) < 1e-12 * max(1.0, abs(tspan[1])):
            ts[-1] = tspan[1]
        elif ts[-1] < tspan[1]:
            ts = np.append(ts, tspan[1])   # SciML: save_end = true for a Number saveat -> Array(sol) ends at tf
        return ts
    ts = _np(saveat)
    if tspan[1] <= tspan[0]:
        return ts   # (the library rejects a non-increasing tspan itself)
    if ts.ndim != 1 or ts.size == 0 or np.any(np.diff(ts) <= 0) or ts[0] < tspan[0] or ts[-1] > tspan[1]:
        raise ValueError("saveat must be strictly increasing and inside tspan = (%g, %g)" % (tspan[0], tspan[1]))
    return ts


def solve(prob, alg, ensemblealg=None, saveat=None, sensealg=None, trajectories=None, device=0, allow_failures=True, **kw):
    """solve(prob, alg; saveat, abstol, reltol, ...) -> ODESolution  /  EnsembleSolution.
    Like upstream, a solve that stops early RETURNS (sol.retcode says why: "MaxIters", "Unstable", ...);
    allow_failures=False turns any retcode other than Success into a UdeError."""
    ens = isinstance(prob, EnsembleProblem)
    base = prob.prob if ens else prob
    eng = Engine.get(device)
    if isinstance(ensemblealg, EnsembleMI355):
        eng.set_launch(ensemblealg.lanes_per_traj, ensemblealg.max_dense_steps)
    else:
        eng.set_launch()  # library defaults (the engine is shared: do not inherit another call's launch options)
    o = _opts(alg, **kw)
    tspan, ts, ns = _time_grids(prob, saveat, o)
    rt = _rt(base.f)
    ts = _np(ts, rt)
    u0 = _np(prob.u0s if ens else base.u0, rt)
    if u0.ndim == 1:
        u0 = u0[None, :]
    N, n = u0.shape
    assert n == base.f.n_state
    if ens and getattr(prob, "ps", None) is not None:       # per-member parameters (EnsembleProblem(..., ps = ...))
        theta = _np(prob.ps, rt)
        # the C side reads N * n_param elements: the FULL shape is checked, not the column count alone (a 1-D ps or a row count
        # other than N would be read out of bounds)
        assert theta.ndim == 2 and theta.shape == (N, base.f.n_param), \
            "ps has shape %s, expected (N, n_param) = (%d, %d)" % (theta.shape, N, base.f.n_param)
    else:
        theta = _np(base.p if base.p is not None else [], rt)
        assert theta.size == base.f.n_param, "theta has %d entries, model expects %d" % (theta.size, base.f.n_param)
    out = np.zeros((N, ns, n), dtype=rt)
    stats = np.zeros((N, NSTATS), dtype=np.int64)
    rc = np.zeros(N, dtype=np.int32)
    eng.check(eng.L.ude_solve_ensemble(eng.h, C.byref(base.f), C.byref(o), N, _ptr(u0), _ptr(tspan), _ptr(theta),
                                       _ptr(ts), ns, _ptr(out), _ptr(stats), _ptr(rc)), allow_traj=allow_failures)
    if ens:
        return EnsembleSolution(ts, out, stats, rc)
    return ODESolution(ts, out[0], stats[0], rc[0])


def rhs(f, u, p, device=0):
    """du = f(u, p) for a batch of states u (N, n) on the device: the right-hand side closure of the scripts
    (`ude_dynamics!`, `dudt_`, `nn_ode`) evaluated once per state."""
    eng = Engine.get(device)
    eng.set_launch()
    rt = _rt(f)
    u = _np(u, rt)
    if u.ndim == 1:
        u = u[None, :]
    N, n = u.shape
    assert n == f.n_state
    p = _np(p, rt)
    du = np.zeros((N, n), dtype=rt)
    eng.check(eng.L.ude_rhs_ensemble(eng.h, C.byref(f), N, _ptr(u), _ptr(p if p.size else np.zeros(1, dtype=rt)), _ptr(du)))
    return du


def concrete_solve(prob, alg, u0, p, saveat=None, sensealg=None, **kw):
    """concrete_solve(prob, alg, u0, p; saveat, abstol, reltol, sensealg)  (seir_exposure.jl:138)"""
    return solve(remake(prob, u0=u0, p=p), alg, saveat=saveat, sensealg=sensealg, **kw)


class GradResult:
    pass


def _grad_common(prob, alg, data, cotangent, row_mask, saveat, device, ensemblealg, kw, sensealg=None, allow_failures=False):
    ens = isinstance(prob, EnsembleProblem)
    base = prob.prob if ens else prob
    eng = Engine.get(device)
    if isinstance(ensemblealg, EnsembleMI355):
        eng.set_launch(ensemblealg.lanes_per_traj, ensemblealg.max_dense_steps)
    else:
        eng.set_launch()  # library defaults (the engine is shared: do not inherit another call's launch options)
    o = _opts(alg, sensealg=sensealg, **kw)
    tspan, ts, ns = _time_grids(prob, saveat, o)
    rt = _rt(base.f)
    ts = _np(ts, rt)
    u0 = _np(prob.u0s if ens else base.u0, rt)
    if u0.ndim == 1:
        u0 = u0[None, :]
    N, n = u0.shape
    per_member = ens and getattr(prob, "ps", None) is not None
    theta = _np(prob.ps if per_member else base.p, rt)
    assert theta.shape == (N, base.f.n_param) if per_member else theta.size == base.f.n_param
    r = GradResult()
    r.t = ts
    r.u = np.zeros((N, ns, n), dtype=rt)
    r.grad_theta = np.zeros(theta.shape if per_member else theta.size, dtype=rt)
    r.grad_u0 = np.zeros((N, n), dtype=rt)
    r.stats = np.zeros((N, NSTATS), dtype=np.int64)
    r.retcode = np.zeros(N, dtype=np.int32)
    if cotangent is not None:
        cot = _np(cotangent, rt).reshape(N, ns, n)
        rc = eng.L.ude_vjp_ensemble(eng.h, C.byref(base.f), C.byref(o), N, _ptr(u0), _ptr(tspan), _ptr(theta),
                                    _ptr(ts), ns, _ptr(cot), _ptr(r.u), _ptr(r.grad_theta), _ptr(r.grad_u0),
                                    _ptr(r.stats), _ptr(r.retcode))
        r.loss = None
    else:
        dat = _np(data, rt).reshape(N, ns, n)
        mask = None if row_mask is None else _np(row_mask, np.uint8)
        loss = np.zeros(1, dtype=rt)
        r.loss_per_traj = np.zeros(N, dtype=rt)
        rc = eng.L.ude_loss_grad_ensemble(eng.h, C.byref(base.f), C.byref(o), N, _ptr(u0), _ptr(tspan), _ptr(theta),
                                          _ptr(ts), ns, _ptr(dat), _ptr(mask), _ptr(loss), _ptr(r.loss_per_traj),
                                          _ptr(r.grad_theta), _ptr(r.grad_u0), _ptr(r.u), _ptr(r.stats),
                                          _ptr(r.retcode))
        r.loss = float(loss[0]) if rt is np.float64 else loss[0]
    # a failed trajectory is dropped from the gradient and the loss is +Inf: never train on a partial objective silently
    eng.check(rc, allow_traj=allow_failures)
    r.kernel_ms = eng.kernel_ms()
    return r


def loss_and_gradient(prob, alg, data, row_mask=None, saveat=None, sensealg=None, ensemblealg=None, device=0,
                      allow_failures=False, **kw):
    """loss(theta) = sum(abs2, data[rows,:] .- Array(solve(...))[rows,:]) and dloss/dtheta by the
    interpolating adjoint (seir_exposure.jl:137-147; Fisher-KPP-CNN.jl:134-143; scenario_1.jl:82-94),
    summed over an ensemble.  data: (N, ns, n).
    allow_failures=False (default): any trajectory whose retcode is not Success raises UdeError.  allow_failures=True: the call returns --
    loss = +Inf, the failed members contribute nothing to grad_theta; one exception: with FastInterpolatingAdjoint() on the SEIR exposure
    UDE / neural ODE (block-level accumulation) a member whose BACKWARD solve stops early makes grad_theta NaN (include/udecore.h)."""
    return _grad_common(prob, alg, data, None, row_mask, saveat, device, ensemblealg, kw, sensealg, allow_failures)


def adjoint_pullback(prob, alg, cotangent, saveat=None, sensealg=None, ensemblealg=None, device=0, allow_failures=False, **kw):
    """The ChainRules pullback of concrete_solve under InterpolatingAdjoint: cotangent (N, ns, n) of
    Array(sol) -> (grad_theta, grad_u0)."""
    return _grad_common(prob, alg, None, cotangent, None, saveat, device, ensemblealg, kw, sensealg, allow_failures)


# ---- device-resident path (torch CUDA tensors; used by bench.py and the training loop) -------------------
class DeviceEnsemble:
    """Ensemble whose u0 / data / theta already live in HBM (torch CUDA tensors of the problem's scalar type: float64, or
    float32 for a descriptor with dtype = 1).  Every call enqueues on torch's current stream and returns torch tensors;
    nothing touches the host."""

    def __init__(self, f, alg, tspan, saveat, u0, data=None, row_mask=None, lanes_per_traj=0, max_dense_steps=0,
                 waves_per_simd=0, sensealg=None, **kw):
        import torch
        self.torch = torch
        self.f, self.o = f, _opts(alg, sensealg=sensealg, **kw)
        dev = u0.device
        rt = torch.float32 if f.dtype == 1 else torch.float64
        self.rt, self.es = rt, (4 if f.dtype == 1 else 8)
        assert dev.type == "cuda" and u0.dtype == rt, "u0 must be a CUDA tensor of the problem's scalar type (%s)" % rt
        assert data is None or data.dtype == rt
        self.eng = Engine.get(dev.index or 0)
        self.launch = (lanes_per_traj, max_dense_steps, waves_per_simd)
        self.N, self.n = u0.shape
        self.u0 = u0.contiguous()
        self.tspan = _np(tspan)
        ts = _saveat_grid(saveat, (float(tspan[0]), float(tspan[1])))
        self.ns = len(ts)
        self.saveat = torch.tensor(ts, dtype=rt, device=dev)
        self.data = None if data is None else data.contiguous()
        self.mask = None if row_mask is None else torch.tensor(list(row_mask), dtype=torch.uint8, device=dev)
        self.u = torch.empty((self.N, self.ns, self.n), dtype=rt, device=dev)
        self.stats = torch.zeros((self.N, NSTATS), dtype=torch.int64, device=dev)
        self.retcode = torch.zeros(self.N, dtype=torch.int32, device=dev)
        self.grad = torch.zeros(f.n_param + 1, dtype=rt, device=dev)  # [grad(np); loss]
        self.grad_u0 = torch.zeros((self.N, self.n), dtype=rt, device=dev)

    def _bind(self):
        self.eng.set_launch(*self.launch)
        self.eng.set_stream(self.torch.cuda.current_stream().cuda_stream)

    def solve(self, theta):
        self._bind()
        L, e = self.eng.L, self.eng
        e.check(L.ude_solve_ensemble_dev(e.h, C.byref(self.f), C.byref(self.o), self.N, _ptr(self.u0), _ptr(self.tspan),
                                         _ptr(theta), _ptr(self.saveat), self.ns, _ptr(self.u), _ptr(self.stats),
                                         _ptr(self.retcode)))
        return self.u

    def check(self):
        """Blocks on the stream: number of failed trajectories of the last loss_grad call.  A DenseOverflow grows
        the dense store (automatic capacity) and returns -1: repeat the call."""
        nfail, grown = self.eng.failures(self.retcode, self.N)
        return -1 if grown else nfail

    def loss_grad(self, theta, check=None):
        """returns a view: grad[:np] = dloss/dtheta, grad[np] = loss (one buffer so a single all-reduce moves both).
        Failed trajectories are dropped from the gradient and make the loss +Inf (nothing is silent).  The FIRST call
        (or check=True) also blocks once to size the dense store: on DenseOverflow the capacity grows x4 and the pass
        is repeated, as the oracle's / upstream's growing solution arrays do."""
        check = (not getattr(self, "_sized", False)) if check is None else check
        if check:
            for _ in range(8):
                g = self.loss_grad(theta, check=False)
                if self.check() >= 0:
                    break
            self._sized = True
            return g
        self._bind()
        L, e = self.eng.L, self.eng
        np_ = self.f.n_param
        loss_ptr = C.c_void_p(self.grad.data_ptr() + self.es * np_)
        e.check(L.ude_loss_grad_ensemble_dev(e.h, C.byref(self.f), C.byref(self.o), self.N, _ptr(self.u0),
                                             _ptr(self.tspan), _ptr(theta), _ptr(self.saveat), self.ns,
                                             _ptr(self.data), _ptr(self.mask), loss_ptr, None, _ptr(self.grad),
                                             _ptr(self.grad_u0), _ptr(self.u), _ptr(self.stats), _ptr(self.retcode)))
        return self.grad

    def graph(self, theta):
        """Capture ONE loss_grad(theta) -- memset, forward kernel, adjoint kernel, the three reduction kernels -- into a
        hipGraph (torch.cuda.CUDAGraph) and return replay() -> the same [grad; loss] view.  `theta` must remain the same
        tensor (update it in place between replays, as a training loop does).  A first, uncaptured call sizes every
        workspace; inside the capture the library leaves out its timing events and refuses to allocate."""
        torch = self.torch
        self.loss_grad(theta)                    # warm-up (+ dense-store sizing): nothing may allocate during the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss_grad(theta, check=False)
        torch.cuda.synchronize()
        self._graph = g

        def replay():
            g.replay()
            return self.grad
        return replay

    def kernel_ms(self):
        return self.eng.kernel_ms()
