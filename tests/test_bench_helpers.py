"""bench.py's host-side helpers that need no GPU: what `roofline.kernel` names for each command, and that every workload the
headline line reports has its algorithmic flop count (the roofline numerator)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def _args(**kw):
    a = argparse.Namespace(workload="lv", sensealg="adjoint", lanes=0, net="s1")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_roofline_kernel_names_the_commands_own_dominant_kernel():
    assert bench.roofline_kernel_name(_args()).startswith("adj_kernel (interpolating adjoint)")
    assert "valu" in bench.roofline_kernel_name(_args()).lower()
    assert bench.roofline_kernel_name(_args(sensealg="discrete")).startswith("dadj_kernel")
    assert bench.roofline_kernel_name(_args(workload="seir")).startswith("seirls::seir_ls_adj_kernel")
    assert bench.roofline_kernel_name(_args(workload="seir", lanes=16)).startswith("seirls::seir_ls_adj_kernel")
    assert bench.roofline_kernel_name(_args(workload="seir", lanes=64)).startswith("adj_kernel")       # the wavefront-per-trajectory kernel
    assert bench.roofline_kernel_name(_args(workload="node")).startswith("nodels::node_ls_adj_kernel")
    assert "matrix cores" in bench.roofline_kernel_name(_args(workload="kpp"))


def test_every_reported_workload_has_its_flop_count():
    for k in ("lv", "seir", "kpp", "node", "lv_tanh32"):
        fwd, adj = bench.FLOPS[k]
        assert 0 < fwd < adj
