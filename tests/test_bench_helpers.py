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
    assert bench.roofline_kernel_name(_args(workload="seir")).startswith("seirls2::seir_ls2_adj_kernel")
    assert bench.roofline_kernel_name(_args(workload="seir", lanes=16)).startswith("seirls2::seir_ls2_adj_kernel")
    assert bench.roofline_kernel_name(_args(workload="seir", lanes=64)).startswith("adj_kernel")       # the wavefront-per-trajectory kernel
    assert bench.roofline_kernel_name(_args(workload="node")).startswith("nodels2::node_ls2_adj_kernel")
    assert "matrix cores" in bench.roofline_kernel_name(_args(workload="kpp"))


def test_every_reported_workload_has_its_flop_count():
    for k in ("lv", "seir", "kpp", "node", "lv_tanh32"):
        fwd, adj = bench.FLOPS[k]
        assert 0 < fwd < adj


def test_roofline_bound_is_chosen_per_kernel():
    """`roofline.bound` names the roof that bounds the command's dominant kernel -- valu / mfma / hbm -- and achieved / peak / frac are
    against that roof; the HBM-bound lock-step kernels keep their flop fraction beside it"""
    import torch
    stats = torch.zeros(4, 8, dtype=torch.int64)
    stats[:, 5], stats[:, 6] = 40, 3                    # backward: 43 step attempts per trajectory
    a = _args(alg="tsit5", waves=0, traj=0)
    r = bench.headline_roofline(a, "lv", 2.0, 1.3e-3, stats, 87)
    assert r["bound"] == "valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - 2.0 / 78.6) < 1e-12
    r = bench.headline_roofline(_args(workload="kpp", alg="tsit5", waves=0, traj=0), "kpp", 8.0, 25e-3, stats, 466)
    assert r["bound"] == "valu"      # (round 6: the network on the vector unit; FP64 matrix instructions share its issue port and its 78.6 TF)
    r = bench.headline_roofline(_args(workload="seir", alg="tsit5", waves=0, traj=0), "seir", 5.5, 12e-3, stats, 4481)
    want = 4 * 43 * 2 * 4481 * 8 / 12e-3 / 1e9
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["achieved"] - want) < 1e-6 * want and r["peak"] == 8000.0
    assert abs(r["flop_frac"] - 5.5 / 78.6) < 1e-12
    # the wavefront-per-trajectory kernel of the same workload keeps mu in LDS/registers: compute bound
    r = bench.headline_roofline(_args(workload="seir", lanes=64, alg="tsit5", waves=0, traj=0), "seir", 3.0, 18e-3, stats, 4481)
    assert r["bound"] == "valu"
    assert set(bench.BOUND.values()) <= {"valu", "mfma", "hbm"}


def test_every_workload_line_has_its_kernels_inside_its_step():
    """`kernel_within_step`: a workload's dominant kernel (and forward + backward kernel together) cannot take longer than the step that
    contains them -- the check the deep-BSDE script-tolerance line once failed (kernel time of the LAST call beside the median step of
    calls with other Philox iterations); every line of the committed driver-command record of this round must pass it"""
    import glob
    import json
    assert bench.kernel_within_step({"ms_per_step": 10.0, "kernel_ms": 8.0, "fwd_kernel_ms": 1.9})
    assert not bench.kernel_within_step({"ms_per_step": 2210.0, "kernel_ms": 2270.0, "bwd_kernel_ms": 1.0})     # round 5's hjb_script_tol
    assert not bench.kernel_within_step({"ms_per_step": 10.0, "kernel_ms": 8.0, "fwd_kernel_ms": 2.5})
    assert "lv_trained" in bench.OTHER_WORKLOADS and bench.BOUND["lv_trained"] == "valu"
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_bench_default_driver_command.json"))):
        line = json.load(open(f))
        assert line["config"]["bwd_kernel_ms"] + line["config"]["fwd_kernel_ms"] <= 1.02 * line["ms_per_step"]
        for name, e in line["config"]["other_workloads"].items():
            assert "error" not in e, (name, e)
            assert bench.kernel_within_step(e), (name, e["ms_per_step"], e.get("kernel_ms"))
        assert "reference_julia" in line["cpu_baseline"]


def test_reference_julia_probe_reports_what_it_finds():
    r = bench.reference_julia()
    assert r == "absent" or r.startswith("found at ")
