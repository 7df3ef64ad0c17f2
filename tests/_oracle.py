"""ctypes binding of the CPU oracle (oracle/libude_oracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MAX_LAYERS = 8
NSTATS = 8

KIND_LV_TRUE, KIND_LV_UDE, KIND_SEIR_TRUE, KIND_SEIR_UDE, KIND_KPP_TRUE, KIND_KPP_UDE, KIND_SEIR_NODE = range(7)
ACT = {"identity": 0, "tanh": 1, "rbf": 2, "relu": 3}
TSIT5, VERN7 = 0, 1


class ModelDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32), ("n_state", C.c_int32), ("n_param", C.c_int32),
        ("n_layers", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("act", C.c_int32 * MAX_LAYERS),
        ("nn_offset", C.c_int32), ("lin_idx", C.c_int32 * 2), ("stencil_offset", C.c_int32),
        ("d0_offset", C.c_int32), ("reserved", C.c_int32),
        ("lin_sign", C.c_double * 2), ("lin_const", C.c_double * 2), ("consts", C.c_double * 16),
    ]


class SolveOpts(C.Structure):
    _fields_ = [
        ("alg", C.c_int32), ("maxiters", C.c_int32), ("abstol", C.c_double), ("reltol", C.c_double),
        ("dtmax", C.c_double), ("dt0", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double),
        ("gamma", C.c_double), ("qoldinit", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
        ("sensealg", C.c_int32), ("per_trajectory", C.c_int32),
    ]


def nn_param_count(dims):
    return sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))


def make_model(kind, n_state, dims=(), acts=(), nn_offset=0, n_param=None, lin_idx=(-1, -1),
               lin_sign=(1.0, 1.0), lin_const=(0.0, 0.0), stencil_offset=0, d0_offset=0, consts=(),
               dtype=0):
    m = ModelDesc()
    m.kind, m.dtype, m.n_state = kind, dtype, n_state
    m.n_layers = max(len(dims) - 1, 0)
    for i, d in enumerate(dims):
        m.dims[i] = d
    for i, a in enumerate(acts):
        m.act[i] = ACT[a] if isinstance(a, str) else a
    m.nn_offset = nn_offset
    for i in range(2):
        m.lin_idx[i], m.lin_sign[i], m.lin_const[i] = lin_idx[i], lin_sign[i], lin_const[i]
    m.stencil_offset, m.d0_offset = stencil_offset, d0_offset
    for i, c in enumerate(consts):
        m.consts[i] = c
    if n_param is None:
        n_param = nn_offset + (nn_param_count(dims) if dims else 0)
    m.n_param = n_param
    return m


# ---- the reference's model configurations -------------------------------------------------------
def lv_true():
    """lotka!  (LotkaVolterra/scenario_1.jl:30-34); theta = p_ = [1.3, 0.9, 0.8, 1.8]"""
    return make_model(KIND_LV_TRUE, 2, n_param=4)


def lv_ude_s1():
    """scenario_1.jl:59-73: 2-5-5-5-2 rbf, du1 = 1.3 u1 + NN1, du2 = -1.8 u2 + NN2 (87 params)"""
    return make_model(KIND_LV_UDE, 2, (2, 5, 5, 5, 2), ("rbf", "rbf", "rbf", "identity"),
                      lin_const=(1.3, -1.8))


def lv_ude_s2():
    """scenario_2.jl:87-95: theta = [delta; ude(87)], du2 = -delta u2 + NN2"""
    return make_model(KIND_LV_UDE, 2, (2, 5, 5, 5, 2), ("rbf", "rbf", "rbf", "identity"), nn_offset=1,
                      lin_idx=(-1, 0), lin_sign=(1.0, -1.0), lin_const=(1.3, 0.0))


def lv_ude_hudson(dtype=0):
    """hudson_bay.jl:77-91: theta = [p1, p2, FastChain(87)], rbf/rbf/tanh"""
    return make_model(KIND_LV_UDE, 2, (2, 5, 5, 5, 2), ("rbf", "rbf", "tanh", "identity"), nn_offset=2,
                      lin_idx=(0, 1), lin_sign=(1.0, -1.0), dtype=dtype)


def lv_ude_tanh32():
    """BASELINE.json's "2-layer tanh MLP": 2-32-2 tanh (SURVEY.md 8(d) C2 choice)"""
    return make_model(KIND_LV_UDE, 2, (2, 32, 2), ("tanh", "identity"), lin_const=(1.3, -1.8))


SEIR_P = [10.0, 0.5944, 0.4239, 1117.3, 0.02, 1 / 3, 1 / 5, 0.2, 1 / 11.2]  # seir_exposure.jl:33


def seir_true(p=SEIR_P):
    return make_model(KIND_SEIR_TRUE, 7, n_param=0, consts=p)


def seir_ude(p=SEIR_P):
    """seir_exposure.jl:114-130: NN 3-64-64-1 tanh (4481 params)"""
    return make_model(KIND_SEIR_UDE, 7, (3, 64, 64, 1), ("tanh", "tanh", "identity"), consts=p)


def seir_node(p=SEIR_P):
    """seir_exposure.jl:53-66: the pure neural ODE, FastChain 7-64-64-64-7 tanh (9287 params)"""
    return make_model(KIND_SEIR_NODE, 7, (7, 64, 64, 64, 7), ("tanh", "tanh", "tanh", "identity"), consts=p)


def kpp_true(nx=26, D=0.01, r=1.0, dx=0.04, dtype=0):
    """rc_ode (Fisher-KPP-CNN.jl:51-63 / scenario_3.jl:43-53): consts = entries of D*lap and r,
    formed in the problem's float type the way the script forms them."""
    if dtype == 1:
        f = np.float32
        dx2 = f(f(dx) * f(dx))
        off = f(np.float64(1.0) / np.float64(dx2))       # Float32.(diagm(...) ./ dx^2)
        dia = f(np.float64(-2.0) / np.float64(dx2))
        consts = (float(f(f(D) * off)), float(f(f(D) * dia)), float(f(r)))
    else:
        consts = (D * (1.0 / dx ** 2), D * (-2.0 / dx ** 2), r)
    return make_model(KIND_KPP_TRUE, nx, n_param=0, consts=consts, dtype=dtype)


def kpp_ude(nx=26, dims=(1, 10, 20, 10, 1), acts=("tanh", "tanh", "tanh", "identity"), dtype=0):
    """nn_ode (Fisher-KPP-CNN.jl:92-126): theta = [NN; w1 w2 w3 unused; D0]"""
    nn = nn_param_count(dims)
    return make_model(KIND_KPP_UDE, nx, dims, acts, nn_offset=0, n_param=nn + 5, stencil_offset=nn,
                      d0_offset=nn + 4, dtype=dtype)


def kpp_ude_s3(dtype=1):
    """scenario_3.jl:83-114: NN 1-5-5-5-1 rbf, theta = [ude(76); p2s(4); D0]"""
    return kpp_ude(26, (1, 5, 5, 5, 1), ("rbf", "rbf", "rbf", "identity"), dtype=dtype)


def opts(alg=TSIT5, abstol=0.0, reltol=0.0, **kw):
    o = SolveOpts()
    o.alg, o.abstol, o.reltol = alg, abstol, reltol
    for k, v in kw.items():
        setattr(o, k, v)
    return o


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, os.environ.get("UDE_ORACLE_LIB", "libude_oracle.so")])


def lib():
    global _lib
    if _lib is None:
        name = os.environ.get("UDE_ORACLE_LIB", "libude_oracle.so")   # (libude_oracle_asan.so: the sanitizer build, oracle/Makefile `asan`)
        path = os.path.join(ORACLE_DIR, name)
        if name != "libude_oracle.so" and not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, name])
        src_newer = (not os.path.exists(path)) or any(
            os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(path)
            for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h")))
        if src_newer:
            build()
        L = C.CDLL(path)
        L.udeo_fastpow.restype = C.c_double
        L.udeo_fastpow.argtypes = [C.c_double, C.c_double]
        L.udeo_fastlog2.restype = C.c_float
        L.udeo_fastlog2.argtypes = [C.c_float]
        L.udeo_exp2f.restype = C.c_float
        L.udeo_exp2f.argtypes = [C.c_float]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(x, dt):
    return np.ascontiguousarray(np.asarray(x, dtype=dt))


def rhs(m, theta, u, dtype=np.float64):
    L = lib()
    theta, u = _arr(theta, dtype), _arr(u, dtype)
    du = np.zeros_like(u)
    if dtype == np.float64:
        L.udeo_rhs_f64(C.byref(m), _p(theta), _p(u), C.c_double(0.0), _p(du))
    else:
        L.udeo_rhs_f32(C.byref(m), _p(theta), _p(u), C.c_float(0.0), _p(du))
    return du


def rhs_vjp(m, theta, u, lam):
    L = lib()
    theta, u, lam = _arr(theta, np.float64), _arr(u, np.float64), _arr(lam, np.float64)
    dlam = np.zeros_like(u)
    dth = np.zeros_like(theta)
    rc = L.udeo_rhs_vjp_f64(C.byref(m), _p(theta), _p(u), C.c_double(0.0), _p(lam), _p(dlam), _p(dth))
    assert rc == 0
    return dlam, dth


def solve_ensemble(m, o, u0, tspan, theta, saveat, dtype=np.float64, nthreads=1):
    """u0: (N, n) array (row j = trajectory j).  returns u (N, ns, n), stats (N, 8), retcode (N,)"""
    L = lib()
    u0 = _arr(u0, dtype)
    if u0.ndim == 1:
        u0 = u0[None, :]
    N, n = u0.shape
    tspan, theta, saveat = _arr(tspan, dtype), _arr(theta, dtype), _arr(saveat, dtype)
    ns = len(saveat)
    out = np.zeros((N, ns, n), dtype=dtype)
    stats = np.zeros((N, NSTATS), dtype=np.int64)
    rc = np.zeros(N, dtype=np.int32)
    fn = L.udeo_solve_ensemble_f64 if dtype == np.float64 else L.udeo_solve_ensemble_f32
    fn(C.byref(m), C.byref(o), C.c_int64(N), _p(u0), _p(tspan), _p(theta), _p(saveat), C.c_int32(ns),
       _p(out), _p(stats), _p(rc), C.c_int32(nthreads))
    return out, stats, rc


def loss_grad_ensemble(m, o, u0, tspan, theta, saveat, data, row_mask=None, nthreads=1, dtype=np.float64):
    """data: (N, ns, n).  returns dict(loss, loss_per_traj, grad_theta, grad_u0, u, stats, retcode)"""
    L = lib()
    u0 = _arr(u0, dtype)
    if u0.ndim == 1:
        u0 = u0[None, :]
    N, n = u0.shape
    tspan, theta, saveat = _arr(tspan, dtype), _arr(theta, dtype), _arr(saveat, dtype)
    data = _arr(data, dtype).reshape(N, len(saveat), n)
    ns = len(saveat)
    mask = None if row_mask is None else _arr(row_mask, np.uint8)
    loss = np.zeros(1, dtype=dtype)
    lpt = np.zeros(N, dtype=dtype)
    g = np.zeros(m.n_param, dtype=dtype)
    gu0 = np.zeros((N, n), dtype=dtype)
    out = np.zeros((N, ns, n), dtype=dtype)
    stats = np.zeros((N, NSTATS), dtype=np.int64)
    rc = np.zeros(N, dtype=np.int32)
    fn = L.udeo_loss_grad_ensemble_f64 if dtype == np.float64 else L.udeo_loss_grad_ensemble_f32
    fn(C.byref(m), C.byref(o), C.c_int64(N), _p(u0), _p(tspan), _p(theta), _p(saveat), C.c_int32(ns), _p(data), _p(mask),
       _p(loss), _p(lpt), _p(g), _p(gu0), _p(out), _p(stats), _p(rc), C.c_int32(nthreads))
    return dict(loss=loss[0] if dtype != np.float64 else float(loss[0]), loss_per_traj=lpt, grad_theta=g, grad_u0=gu0, u=out,
                stats=stats, retcode=rc)


def vjp_ensemble(m, o, u0, tspan, theta, saveat, cotangent, nthreads=1):
    L = lib()
    u0 = _arr(u0, np.float64)
    if u0.ndim == 1:
        u0 = u0[None, :]
    N, n = u0.shape
    tspan, theta, saveat = _arr(tspan, np.float64), _arr(theta, np.float64), _arr(saveat, np.float64)
    ns = len(saveat)
    cot = _arr(cotangent, np.float64).reshape(N, ns, n)
    g = np.zeros(m.n_param)
    gu0 = np.zeros((N, n))
    out = np.zeros((N, ns, n))
    stats = np.zeros((N, NSTATS), dtype=np.int64)
    rc = np.zeros(N, dtype=np.int32)
    L.udeo_vjp_ensemble_f64(C.byref(m), C.byref(o), C.c_int64(N), _p(u0), _p(tspan), _p(theta), _p(saveat),
                            C.c_int32(ns), _p(cot), _p(out), _p(g), _p(gu0), _p(stats), _p(rc),
                            C.c_int32(nthreads))
    return dict(grad_theta=g, grad_u0=gu0, u=out, stats=stats, retcode=rc)


def solve_dense(m, o, u0, tspan, theta, cap=4096):
    L = lib()
    u0, tspan, theta = _arr(u0, np.float64), _arr(tspan, np.float64), _arr(theta, np.float64)
    n = len(u0)
    nk = 7 if o.alg == TSIT5 else 16
    t = np.zeros(cap + 1)
    u = np.zeros((cap + 1, n))
    k = np.zeros((cap, nk, n))
    stats = np.zeros(NSTATS, dtype=np.int64)
    ns = L.udeo_solve_dense_f64(C.byref(m), C.byref(o), _p(u0), _p(tspan), _p(theta), C.c_int32(cap),
                                _p(t), _p(u), _p(k), _p(stats))
    assert ns >= 0, ns
    return t[:ns + 1], u[:ns + 1], k[:ns], stats
