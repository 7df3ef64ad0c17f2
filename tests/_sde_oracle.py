"""ctypes binding of the N1 (highdim_pde / LambaEM) part of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C

import numpy as np

import _oracle as O


class HjbDesc(C.Structure):
    """udeo_hjb_desc == ude_hjb_desc (include/udecore.h)"""
    _fields_ = [("d", C.c_int32), ("hls", C.c_int32), ("adaptive", C.c_int32), ("maxiters", C.c_int32),
                ("max_steps", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64),
                ("lam", C.c_double), ("sigma", C.c_double), ("t0", C.c_double), ("t1", C.c_double),
                ("abstol", C.c_double), ("reltol", C.c_double), ("dt", C.c_double),
                ("qmin", C.c_double), ("qmax", C.c_double), ("gamma", C.c_double), ("qoldinit", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("dtmax", C.c_double)]


def desc(d=100, hls=None, adaptive=1, abstol=1e-4, reltol=1e-4, seed=0, lam=1.0, sigma=float(np.sqrt(np.float32(2.0))),
         tspan=(0.0, 1.0), dt=0.0, maxiters=0, max_steps=0, **kw):
    """highdim_pde/lambaem.jl:8-34: d = 100, hls = 10 + d, lambda = 1, sigma = sqrt(2f0), tspan (0, 1), tolerances 1e-4"""
    D = HjbDesc()
    D.d, D.hls, D.adaptive, D.maxiters, D.max_steps = d, (10 + d if hls is None else hls), adaptive, maxiters, max_steps
    D.seed, D.lam, D.sigma, D.t0, D.t1 = seed, lam, sigma, tspan[0], tspan[1]
    D.abstol, D.reltol, D.dt = abstol, reltol, dt
    for k, v in kw.items():
        setattr(D, k, v)
    return D


def num_params(d, hls):
    a, b = C.c_int32(0), C.c_int32(0)
    O.lib().udeo_hjb_num_params(d, hls, C.byref(a), C.byref(b))
    return a.value, b.value


def glorot_params(d, hls, rng, dtype=np.float32):
    """Flux.Dense default init (glorot_uniform weights, zero bias), theta = [u0 chain; sigma^T grad u chain]"""
    out = []
    for dims in ((d, hls, hls, 1), (d + 1, hls, hls, hls, d)):
        for i in range(len(dims) - 1):
            fin, fout = dims[i], dims[i + 1]
            lim = np.sqrt(6.0 / (fin + fout))
            out.append(rng.uniform(-lim, lim, fin * fout))
            out.append(np.zeros(fout))
    return np.concatenate(out).astype(dtype)


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    O.lib().udeo_philox4x32_10((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
    return [int(x) for x in out]


def normals(seed, it, traj, event, d):
    out = np.zeros(d)
    O.lib().udeo_hjb_normals(C.c_uint64(seed), C.c_uint32(it), C.c_uint32(traj), C.c_uint32(event), d, O._p(out))
    return out


def sincos2pi(u):
    L = O.lib()
    L.udeo_sincos2pi.restype = C.c_double
    L.udeo_sincos2pi.argtypes = [C.c_double, C.POINTER(C.c_double)]
    c = C.c_double(0)
    s = L.udeo_sincos2pi(u, C.byref(c))
    return s, c.value


def loss_grad(D, M, x0, theta, it=0, want_grad=True, dtype=np.float32, nthreads=1):
    L = O.lib()
    fn = L.udeo_hjb_loss_grad_f32 if dtype == np.float32 else L.udeo_hjb_loss_grad_f64
    x0 = np.ascontiguousarray(x0, dtype=dtype)
    theta = np.ascontiguousarray(theta, dtype=dtype)
    loss = C.c_double(0)
    grad = np.zeros(theta.size, dtype=dtype) if want_grad else None
    u0 = np.zeros(1, dtype=dtype)
    uT = np.zeros(M, dtype=dtype)
    XT = np.zeros((M, D.d), dtype=dtype)
    lt = np.zeros(M)
    stats = np.zeros((M, 4), dtype=np.int64)
    rc = np.zeros(M, dtype=np.int32)
    ret = fn(C.byref(D), C.c_int64(M), O._p(x0), O._p(theta), C.c_uint32(it), C.byref(loss), O._p(grad), O._p(u0), O._p(uT),
             O._p(XT), O._p(lt), O._p(stats), O._p(rc), nthreads)
    return dict(ret=ret, loss=loss.value, grad=grad, u0=u0[0], uT=uT, XT=XT, loss_traj=lt, stats=stats, retcode=rc)


def net(d, hls, theta_sg, x_in):
    z = np.zeros(d, dtype=np.float32)
    O.lib().udeo_hjb_net_f32(d, hls, O._p(np.ascontiguousarray(theta_sg, dtype=np.float32)),
                             O._p(np.ascontiguousarray(x_in, dtype=np.float32)), O._p(z))
    return z


def path(D, x0, theta, it=0, traj=0, cap=65536):
    t = np.zeros(cap, dtype=np.float32)
    dt = np.zeros(cap, dtype=np.float32)
    X = np.zeros((cap, D.d), dtype=np.float32)
    dW = np.zeros((cap, D.d), dtype=np.float32)
    EE = np.zeros(cap, dtype=np.float32)
    n = O.lib().udeo_hjb_path_f32(C.byref(D), O._p(np.ascontiguousarray(x0, dtype=np.float32)),
                                  O._p(np.ascontiguousarray(theta, dtype=np.float32)), C.c_uint32(it), C.c_uint32(traj), cap,
                                  O._p(t), O._p(dt), O._p(X), O._p(dW), O._p(EE))
    if n < 0:
        raise RuntimeError("trajectory failed with retcode %d" % -n)
    return dict(n=n, t=t[:n], dt=dt[:n], X=X[:n], dW=dW[:n], EEst=EE[:n])
