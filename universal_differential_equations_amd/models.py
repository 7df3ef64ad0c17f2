"""Declarative RHS descriptors: the reference's right-hand-side closures restated as data.

A Julia (or Python) closure cannot cross the C ABI, so each RHS family of the reference is a
descriptor (include/udecore.h:ude_model_desc).  Constructors are named after the reference functions.
"""
from ._lib import (ACT, KIND_KPP_TRUE, KIND_KPP_UDE, KIND_LV_TRUE, KIND_LV_UDE, KIND_SEIR_NODE, KIND_SEIR_TRUE, KIND_SEIR_UDE,
                   ModelDesc)


class Dense:
    """Lux.Dense(in, out, act) / FastDense / Flux.Dense (scenario_1.jl:62-64, seir_exposure.jl:114)"""

    def __init__(self, n_in, n_out, act="identity"):
        self.n_in, self.n_out, self.act = n_in, n_out, act


class Chain:
    """Lux.Chain / FastChain / Flux.Chain of Dense layers; parameters flatten per layer as
    [vec(W) column-major (out x in); b]."""

    def __init__(self, *layers):
        self.layers = layers
        for a, b in zip(layers[:-1], layers[1:]):
            assert a.n_out == b.n_in, "layer sizes do not chain"

    @property
    def dims(self):
        return [self.layers[0].n_in] + [l.n_out for l in self.layers]

    @property
    def acts(self):
        return [l.act for l in self.layers]

    @property
    def n_param(self):
        return sum(l.n_in * l.n_out + l.n_out for l in self.layers)

    def glorot_uniform(self, rng, dtype="float64"):
        """Lux/Flux default init: glorot_uniform weights (Float32), zero bias -- synthetic stand-in for
        Lux.setup(rng, U) (scenario_1.jl:66); Julia's RNG stream itself cannot be reproduced."""
        import numpy as np
        out = []
        for l in self.layers:
            lim = np.sqrt(6.0 / (l.n_in + l.n_out))
            w = rng.uniform(-lim, lim, size=(l.n_in, l.n_out)).astype(np.float32)  # column-major (out x in)
            out += [w.ravel(), np.zeros(l.n_out, dtype=np.float32)]
        return np.concatenate(out).astype(dtype)


def _desc(kind, n_state, chain=None, nn_offset=0, n_param=None, lin_idx=(-1, -1), lin_sign=(1.0, 1.0),
          lin_const=(0.0, 0.0), stencil_offset=0, d0_offset=0, consts=(), dtype="float64"):
    m = ModelDesc()
    m.kind, m.dtype, m.n_state = kind, {"float64": 0, "float32": 1}[str(dtype)], n_state
    if chain is not None:
        m.n_layers = len(chain.layers)
        for i, d in enumerate(chain.dims):
            m.dims[i] = d
        for i, a in enumerate(chain.acts):
            m.act[i] = ACT[a]
    m.nn_offset = nn_offset
    for i in range(2):
        m.lin_idx[i], m.lin_sign[i], m.lin_const[i] = lin_idx[i], lin_sign[i], lin_const[i]
    m.stencil_offset, m.d0_offset = stencil_offset, d0_offset
    for i, c in enumerate(consts):
        m.consts[i] = c
    m.n_param = n_param if n_param is not None else nn_offset + (chain.n_param if chain else 0)
    return m


def lotka():
    """lotka!(du,u,p,t)  LotkaVolterra/scenario_1.jl:30-34; p = (alpha, beta, gamma, delta)"""
    return _desc(KIND_LV_TRUE, 2, n_param=4)


def lv_chain():
    """U = Lux.Chain(Dense(2,5,rbf), Dense(5,5,rbf), Dense(5,5,rbf), Dense(5,2))  scenario_1.jl:62-64"""
    return Chain(Dense(2, 5, "rbf"), Dense(5, 5, "rbf"), Dense(5, 5, "rbf"), Dense(5, 2))


def ude_dynamics(chain=None, p_true=(1.3, 0.9, 0.8, 1.8), trainable=None, dtype="float64"):
    """ude_dynamics!  du1 = p_true[1]*u1 + U(u)[1];  du2 = -p_true[4]*u2 + U(u)[2]   (scenario_1.jl:69-73).

    trainable: None (scenario_1), "delta" (scenario_2.jl:87-95: theta = [delta; ude], du2 = -delta*u2 + ...),
    "both" (hudson_bay.jl:82-91: theta = [p1, p2, ude], du1 = p1*u1 + ..., du2 = -p2*u2 + ...).
    dtype="float32": the Float32 problem of hudson_bay.jl:77-104 (compiled: hudson_chain(), trainable="both")."""
    chain = chain or lv_chain()
    if trainable is None:
        return _desc(KIND_LV_UDE, 2, chain, lin_const=(p_true[0], -p_true[3]), dtype=dtype)
    if trainable == "delta":
        return _desc(KIND_LV_UDE, 2, chain, nn_offset=1, lin_idx=(-1, 0), lin_sign=(1.0, -1.0), lin_const=(p_true[0], 0.0), dtype=dtype)
    if trainable == "both":
        return _desc(KIND_LV_UDE, 2, chain, nn_offset=2, lin_idx=(0, 1), lin_sign=(1.0, -1.0), dtype=dtype)
    raise ValueError(trainable)


def hudson_chain():
    """FastChain(FastDense(2,5,rbf), FastDense(5,5,rbf), FastDense(5,5,tanh), FastDense(5,2))  hudson_bay.jl:77-79"""
    return Chain(Dense(2, 5, "rbf"), Dense(5, 5, "rbf"), Dense(5, 5, "tanh"), Dense(5, 2))


def tanh32_chain():
    """BASELINE.json's "2-layer tanh MLP" for the LV ensemble: 2 -> 32 -> 2 tanh"""
    return Chain(Dense(2, 32, "tanh"), Dense(32, 2))


SEIR_P = (10.0, 0.5944, 0.4239, 1117.3, 0.02, 1 / 3, 1 / 5, 0.2, 1 / 11.2)  # seir_exposure.jl:33


def corona(p_=SEIR_P):
    """corona!(du,u,p,t)  SEIR_exposure/seir_exposure.jl:16-30"""
    return _desc(KIND_SEIR_TRUE, 7, n_param=0, consts=p_)


def seir_chain():
    """ann = FastChain(FastDense(3,64,tanh), FastDense(64,64,tanh), FastDense(64,1))  seir_exposure.jl:114"""
    return Chain(Dense(3, 64, "tanh"), Dense(64, 64, "tanh"), Dense(64, 1))


def dudt_(chain=None, p_=SEIR_P):
    """dudt_(u,p,t)  seir_exposure.jl:117-130"""
    return _desc(KIND_SEIR_UDE, 7, chain or seir_chain(), consts=p_)


def seir_node_chain():
    """ann_node = FastChain(FastDense(7,64,tanh), FastDense(64,64,tanh), FastDense(64,64,tanh), FastDense(64,7))  seir_exposure.jl:53"""
    return Chain(Dense(7, 64, "tanh"), Dense(64, 64, "tanh"), Dense(64, 64, "tanh"), Dense(64, 7))


def dudt_node(chain=None, p_=SEIR_P):
    """dudt_node(u,p,t)  seir_exposure.jl:55-66: the pure neural ODE; dS,dE,dI,dR,dD = the first five network outputs"""
    return _desc(KIND_SEIR_NODE, 7, chain or seir_node_chain(), consts=p_)


def rc_ode(nx=26, D=0.01, r=1.0, dx=0.04, dtype="float64"):
    """rc_ode(rho,p,t) = D*lap*rho + reaction.(rho)  FisherKPP/Fisher-KPP-CNN.jl:51-63 (periodic);
    dtype="float32": LotkaVolterra/scenario_3.jl:43-53 (the matrix entries formed in Float32 the way the script forms them)"""
    # D * lap with lap = diagm(-2, 1, 1) ./ dx^2: the entries Julia forms are D*(1/dx^2) and D*(-2/dx^2)
    if str(dtype) == "float32":
        import numpy as np
        f = np.float32
        dx2 = f(f(dx) * f(dx))
        off, dia = f(np.float64(1.0) / np.float64(dx2)), f(np.float64(-2.0) / np.float64(dx2))     # Float32.(diagm(...) ./ dx^2)
        return _desc(KIND_KPP_TRUE, nx, n_param=0, consts=(float(f(f(D) * off)), float(f(f(D) * dia)), float(f(r))), dtype=dtype)
    return _desc(KIND_KPP_TRUE, nx, n_param=0, consts=(D * (1.0 / dx ** 2), D * (-2.0 / dx ** 2), r))


def kpp_chain():
    """rx_nn = Chain(Dense(1,10,tanh), Dense(10,20,tanh), Dense(20,10,tanh), Dense(10,1))  Fisher-KPP-CNN.jl:92-96"""
    return Chain(Dense(1, 10, "tanh"), Dense(10, 20, "tanh"), Dense(20, 10, "tanh"), Dense(10, 1))


def kpp_small_chain(n_weights=3):
    """rx_nn = Chain(Dense(1, n_weights, tanh), Dense(n_weights, 1))  Fisher-KPP-CNN-Small.jl:89-94 (15 parameters in all)"""
    return Chain(Dense(1, n_weights, "tanh"), Dense(n_weights, 1))


def kpp_s3_chain():
    """rx_nn = Lux.Chain(Dense(1,5,rbf), Dense(5,5,rbf), Dense(5,5,rbf), Dense(5,1))  LotkaVolterra/scenario_3.jl:83-88"""
    return Chain(Dense(1, 5, "rbf"), Dense(5, 5, "rbf"), Dense(5, 5, "rbf"), Dense(5, 1))


def kpp_theta(chain, rng, stencil=(1.1, -2.5, 1.0), D0=6.5):
    """p = [p1; p2; D0] with the reference's initial stencil and D0 (Fisher-KPP-CNN.jl:98-109)"""
    import numpy as np
    return np.concatenate([chain.glorot_uniform(rng), list(stencil), [0.0], [D0]])


def rho0(nx=26, dx=0.04, amp=1.0, delta=0.2):
    """IC-1 of Fisher-KPP-CNN.jl:27-31 on x = 0:dx:(nx-1)*dx"""
    import numpy as np
    x = dx * np.arange(nx)
    return amp * (np.tanh((x - (0.5 - delta / 2)) / (delta / 10)) - np.tanh((x - (0.5 + delta / 2)) / (delta / 10))) / 2


def nn_ode(nx=26, chain=None, dtype="float64"):
    """nn_ode(u,p,t)  Fisher-KPP-CNN.jl:111-126; theta = [rx_nn params; w1 w2 w3; unused conv bias; D0];
    dtype="float32" with kpp_s3_chain(): scenario_3.jl:103-114"""
    chain = chain or kpp_chain()
    nn = chain.n_param
    return _desc(KIND_KPP_UDE, nx, chain, nn_offset=0, n_param=nn + 5, stencil_offset=nn, d0_offset=nn + 4, dtype=dtype)
