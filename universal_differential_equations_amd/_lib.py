"""ctypes binding of libudecore.so (include/udecore.h).  The HIP library is the product: if it is
missing this module raises -- there is no CPU fallback anywhere in the package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# UDE_LIB_VARIANT=dbg selects the debug build (libudecore_dbg.so: same kernels, host side compiled with -DUDE_DEBUG_HOOKS --
# register poison, workspace fill, phase clocks; tests/test_gpu_poison.py runs the parity tests against it in a subprocess)
# (any other value <name>: libudecore_<name>.so, a developer's timing-experiment build -- build.py: UDE_EXP_VARIANTS)
_variant = os.environ.get("UDE_LIB_VARIANT", "")
LIB_PATH = os.path.join(HERE, "libudecore_%s.so" % _variant if _variant else "libudecore.so")
MAX_LAYERS = 8
NSTATS = 8

KIND_LV_TRUE, KIND_LV_UDE, KIND_SEIR_TRUE, KIND_SEIR_UDE, KIND_KPP_TRUE, KIND_KPP_UDE, KIND_SEIR_NODE = range(7)
ACT = {"identity": 0, "tanh": 1, "rbf": 2, "relu": 3}
ALG_TSIT5, ALG_VERN7 = 0, 1
RETCODES = {0: "Success", 1: "MaxIters", 2: "DtLessThanMin", 3: "Unstable", 4: "DenseOverflow"}
UDE_ERR_TRAJECTORY = -5
UDE_ERR_TIMEOUT = -6


class ModelDesc(C.Structure):
    """ude_model_desc"""
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32), ("n_state", C.c_int32), ("n_param", C.c_int32),
        ("n_layers", C.c_int32), ("dims", C.c_int32 * (MAX_LAYERS + 1)), ("act", C.c_int32 * MAX_LAYERS),
        ("nn_offset", C.c_int32), ("lin_idx", C.c_int32 * 2), ("stencil_offset", C.c_int32),
        ("d0_offset", C.c_int32), ("reserved", C.c_int32),
        ("lin_sign", C.c_double * 2), ("lin_const", C.c_double * 2), ("consts", C.c_double * 16),
    ]


class SolveOpts(C.Structure):
    """ude_solve_opts"""
    _fields_ = [
        ("alg", C.c_int32), ("maxiters", C.c_int32), ("abstol", C.c_double), ("reltol", C.c_double),
        ("dtmax", C.c_double), ("dt0", C.c_double), ("qmin", C.c_double), ("qmax", C.c_double),
        ("gamma", C.c_double), ("qoldinit", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
        ("sensealg", C.c_int32), ("per_trajectory", C.c_int32),
    ]


class LaunchOpts(C.Structure):
    """ude_launch_opts"""
    _fields_ = [("lanes_per_traj", C.c_int32), ("block_threads", C.c_int32), ("max_dense_steps", C.c_int32),
                ("waves_per_simd", C.c_int32)]


class HjbDesc(C.Structure):
    """ude_hjb_desc (highdim_pde/lambaem.jl:8-34)"""
    _fields_ = [("d", C.c_int32), ("hls", C.c_int32), ("adaptive", C.c_int32), ("maxiters", C.c_int32),
                ("max_steps", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64),
                ("lam", C.c_double), ("sigma", C.c_double), ("t0", C.c_double), ("t1", C.c_double),
                ("abstol", C.c_double), ("reltol", C.c_double), ("dt", C.c_double),
                ("qmin", C.c_double), ("qmax", C.c_double), ("gamma", C.c_double), ("qoldinit", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("dtmax", C.c_double)]


class UdeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("udecore error %d: %s" % (code, msg))
        self.code = code


_lib = None

EXPORTS = ["ude_version", "ude_create", "ude_destroy", "ude_last_error", "ude_set_stream", "ude_set_launch_opts",
           "ude_model_supported", "ude_solve_ensemble", "ude_solve_ensemble_dev", "ude_vjp_ensemble",
           "ude_vjp_ensemble_dev", "ude_loss_grad_ensemble", "ude_loss_grad_ensemble_dev", "ude_last_kernel_ms",
           "ude_fastpow_dev", "ude_set_trace", "ude_get_trace", "ude_math_dev", "ude_rhs_ensemble", "ude_rhs_ensemble_dev",
           "ude_last_failures", "ude_hjb_num_params", "ude_hjb_loss_grad_dev", "ude_hjb_loss_grad", "ude_hjb_normals",
           "ude_hjb_net", "ude_hjb_last_kernel_ms", "ude_hjb_debug_read", "ude_hjb_last_failures",
           "ude_comm_unique_id", "ude_comm_create", "ude_comm_create_local", "ude_comm_destroy", "ude_allreduce_grad",
           "ude_allreduce_grad_local", "ude_allreduce_grad_p2p", "ude_comm_create_p2p", "ude_comm_p2p_connect",
           "ude_allreduce_grad_p2p_mp", "ude_comm_p2p_status", "ude_comm_p2p_disconnect", "ude_pack_counters_dev"]


def load():
    """Load libudecore.so; raises if the HIP extension has not been built (python -m ...build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libudecore.so is missing: build it with `python -m universal_differential_equations_amd.build` "
                          "(the HIP library is required; there is no CPU fallback)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64 (same SONAME as
    # /opt/rocm's).  Whichever is mapped first serves both, so torch must come first when it is installed --
    # a system runtime initialised before torch leaves torch without GPUs.  (A non-Python host, e.g. the Julia
    # ccall shim, simply gets /opt/rocm's runtime through libudecore's RUNPATH.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.ude_version.restype = C.c_int
    L.ude_create.argtypes = [i32, C.POINTER(vp)]
    L.ude_destroy.argtypes = [vp]
    L.ude_destroy.restype = None
    L.ude_last_error.argtypes = [vp]
    L.ude_last_error.restype = C.c_char_p
    L.ude_set_stream.argtypes = [vp, vp]
    L.ude_set_launch_opts.argtypes = [vp, C.POINTER(LaunchOpts)]
    L.ude_model_supported.argtypes = [vp, C.POINTER(ModelDesc), C.POINTER(SolveOpts), i32]
    common = [vp, C.POINTER(ModelDesc), C.POINTER(SolveOpts), i64, vp, vp, vp, vp, i32]
    for name in ("ude_solve_ensemble", "ude_solve_ensemble_dev"):
        getattr(L, name).argtypes = common + [vp, vp, vp]
    for name in ("ude_vjp_ensemble", "ude_vjp_ensemble_dev"):
        getattr(L, name).argtypes = common + [vp, vp, vp, vp, vp, vp]
    for name in ("ude_loss_grad_ensemble", "ude_loss_grad_ensemble_dev"):
        getattr(L, name).argtypes = common + [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ude_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ude_fastpow_dev.argtypes = [vp, i64, vp, vp, vp]
    L.ude_math_dev.argtypes = [vp, i32, i64, vp, vp, vp]
    L.ude_rhs_ensemble.argtypes = [vp, vp, i64, vp, vp, vp]
    L.ude_rhs_ensemble_dev.argtypes = [vp, vp, i64, vp, vp, vp]
    L.ude_last_failures.argtypes = [vp, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    u32, u64 = C.c_uint32, C.c_uint64
    L.ude_hjb_num_params.argtypes = [i32, i32, C.POINTER(i32), C.POINTER(i32)]
    for name in ("ude_hjb_loss_grad_dev", "ude_hjb_loss_grad"):
        getattr(L, name).argtypes = [vp, C.POINTER(HjbDesc), i64, vp, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ude_hjb_last_failures.argtypes = [vp, vp, i64, C.POINTER(i32), C.POINTER(i32)]
    L.ude_hjb_normals.argtypes = [vp, u64, u32, u32, u32, i32, vp]
    L.ude_hjb_net.argtypes = [vp, i32, i32, vp, i64, vp, vp]
    L.ude_comm_unique_id.argtypes = [vp]
    L.ude_comm_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    L.ude_comm_create_local.argtypes = [i32, vp, vp]
    L.ude_comm_destroy.argtypes = [vp]
    L.ude_comm_destroy.restype = None
    L.ude_allreduce_grad.argtypes = [vp, vp, i64]
    L.ude_allreduce_grad_local.argtypes = [i32, vp, vp, i64]
    L.ude_allreduce_grad_p2p.argtypes = [i32, vp, vp, i64]
    L.ude_comm_create_p2p.argtypes = [vp, i32, i32, i64, vp, C.POINTER(vp)]
    L.ude_comm_p2p_connect.argtypes = [vp, vp]
    L.ude_allreduce_grad_p2p_mp.argtypes = [vp, vp, i64]
    L.ude_comm_p2p_status.argtypes = [vp, C.POINTER(i32)]
    L.ude_comm_p2p_disconnect.argtypes = [vp]
    L.ude_pack_counters_dev.argtypes = [vp, i64, vp, vp, i32]
    L.ude_hjb_debug_read.argtypes = [vp, i32, i64, i64, vp]
    L.ude_hjb_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ude_set_trace.argtypes = [vp, i64, i32]
    L.ude_get_trace.argtypes = [vp, vp]
    _lib = L
    return L
