"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/udecore.h declares; struct layouts agree with the header; host logic (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "udecore.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ude_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from universal_differential_equations_amd import _lib
    L = _lib.load()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libudecore.so does not export %s" % n
    assert set(names) == set(_lib.EXPORTS)
    assert L.ude_version() == 100


def test_dynamic_symbol_table_is_exactly_the_header():
    """no -fvisibility leak: `nm -D --defined-only` of the shipping library lists the entry points of include/udecore.h and
    nothing else (the instance getters ude_inst_*, the lock-step getters and every C++ symbol are local: build.py links with a
    version script written from the header); the debug library adds its one probe"""
    import subprocess
    from universal_differential_equations_amd import _lib, build
    names = set(declared_functions())
    for lib, extra in ((build.LIB, set()), (build.LIB_DBG, set(build.DBG_EXPORTS))):
        out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
        syms = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
        assert syms == names | extra, "%s: unexpected %s, missing %s" % (os.path.basename(lib), sorted(syms - names - extra)[:8], sorted((names | extra) - syms)[:8])


def test_struct_sizes_match_header():
    from universal_differential_equations_amd import _lib
    # ude_model_desc: 5 + 9 + 8 + 1 + 2 + 3 int32 = 28 int32 = 112 B, then 2+2+16 doubles = 160 B
    assert C.sizeof(_lib.ModelDesc) == 112 + 160
    assert C.sizeof(_lib.SolveOpts) == 8 + 10 * 8 + 8
    assert C.sizeof(_lib.LaunchOpts) == 16
    import _oracle as O
    assert C.sizeof(O.ModelDesc) == C.sizeof(_lib.ModelDesc)
    assert C.sizeof(O.SolveOpts) == C.sizeof(_lib.SolveOpts)
    # ude_hjb_desc: 6 int32 + uint64 + 14 doubles
    import _sde_oracle as S
    assert C.sizeof(_lib.HjbDesc) == 24 + 8 + 14 * 8 == C.sizeof(S.HjbDesc)
    assert [f[0] for f in _lib.HjbDesc._fields_] == [f[0] for f in S.HjbDesc._fields_]


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    prob = U.ODEProblem(models.lotka(), [0.44, 4.6], (0.0, 1.0), [1.3, 0.9, 0.8, 1.8])
    with pytest.raises(U.sciml.UdeError):
        U.solve(prob, U.Tsit5(), saveat=0.1)


def test_model_descriptors_match_reference_parameter_counts():
    from universal_differential_equations_amd import models
    assert models.ude_dynamics().n_param == 87                      # scenario_1.jl:62-66
    assert models.ude_dynamics(trainable="delta").n_param == 88     # scenario_2.jl:87-88
    assert models.ude_dynamics(models.hudson_chain(), trainable="both").n_param == 89   # hudson_bay.jl:82
    assert models.dudt_().n_param == 4481                           # seir_exposure.jl:114
    assert models.nn_ode().n_param == 466                           # Fisher-KPP-CNN.jl:106-109
    assert models.seir_chain().dims == [3, 64, 64, 1]
    import _oracle as O
    for mine, theirs in ((models.ude_dynamics(), O.lv_ude_s1()), (models.ude_dynamics(trainable="delta"), O.lv_ude_s2()),
                         (models.ude_dynamics(models.hudson_chain(), trainable="both"), O.lv_ude_hudson()),
                         (models.dudt_(), O.seir_ude()), (models.nn_ode(), O.kpp_ude()), (models.lotka(), O.lv_true())):
        assert bytes(mine) == bytes(theirs)


def test_saveat_grid():
    from universal_differential_equations_amd.sciml import _saveat_grid
    assert np.allclose(_saveat_grid(0.1, (0.0, 3.0)), np.arange(31) * 0.1)
    assert len(_saveat_grid(0.5, (0.0, 5.0))) == 11
    assert len(_saveat_grid(1, (0.0, 21.0))) == 22


def test_saveat_solution_interpolation_and_derivative():
    """`DX = Array(solution(solution.t, Val{1}))` (scenario_1.jl:46): a saveat-only solution interpolates linearly"""
    from universal_differential_equations_amd.sciml import ODESolution
    t = np.array([0.0, 0.1, 0.3, 0.6])
    u = np.stack([t ** 2, np.sin(t)], axis=1)              # (ns, n)
    sol = ODESolution(t, u, np.zeros(8), 0)
    DX = sol(t, 1)
    assert DX.shape == (2, 4)
    slopes = np.diff(u, axis=0).T / np.diff(t)
    assert np.allclose(DX[:, 0], slopes[:, 0]) and np.allclose(DX[:, 1:], slopes)      # first interval serves t[1]; then the left interval
    assert np.allclose(sol(t), u.T) and np.allclose(sol(0.2), 0.5 * (u[1] + u[2])) and np.allclose(sol(0.2, 1), slopes[:, 1])
    with pytest.raises(ValueError):
        sol(0.7)
