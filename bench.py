#!/usr/bin/env python3
"""bench.py -- north-star metric on BASELINE config 2.

Workload ("step" = one pass of the hot path over one batch): loss + interpolating-adjoint gradient of a
10 000-trajectory Lotka-Volterra UDE ensemble per GPU (scenario_1.jl's 2-5-5-5-2 rbf network, theta_init from
the reference's own artifact, Tsit5 abstol=reltol=1e-6, 31 save points on t in [0,3]); N>1: every rank owns
its own 10k trajectories (weak scaling) and the ranks exchange one RCCL all-reduce of [grad(87); loss].
Metric: ODE RHS evaluations per second, forward (upstream destats.nf) + adjoint (augmented-RHS evals).
Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector == FP64 matrix peak
# algorithmic flop per work unit (forward RHS eval, adjoint eval): SURVEY.md 8(d) roofline table
FLOPS = {"lv": (160.0, 550.0), "seir": (8744.0, 26000.0), "kpp": (0.87e6, 2.6e6)}


def synth_inputs_other(workload, N, rank, device):
    """SEIR (BASELINE configs[2] per-GPU share) and Fisher-KPP (configs[3]) synthetic inputs, SURVEY.md 8(d)."""
    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    rng = np.random.default_rng(1234 + rank)
    if workload == "seir":
        S0 = 14e6
        u0 = np.zeros((N, 7))
        u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
        u0[:, 4] = S0
        t = np.arange(22.0)
        tspan, f_true, f_ude = (0.0, 21.0), models.corona(), models.dudt_()
        theta = models.seir_chain().glorot_uniform(rng)
        mask, alg, tol = [0, 1, 1, 1, 0, 0, 0], U.Vern7(), dict(abstol=1e-6, reltol=1e-6)
        true_alg, true_tol = U.Vern7(), dict(abstol=1e-12, reltol=1e-12)
    else:
        nx = 1024
        base = models.rho0(26)
        u0 = np.tile(base, 40)[None, :nx] * (1 + 0.1 * rng.uniform(-1, 1, (N, 1)))
        t = np.arange(11) * 0.5
        tspan, f_true, f_ude = (0.0, 5.0), models.rc_ode(nx), models.nn_ode(nx)
        theta = models.kpp_theta(models.kpp_chain(), rng)
        mask, alg, tol = None, U.Tsit5(), {}
        true_alg, true_tol = U.Tsit5(), {}
    u0_d = torch.tensor(u0, dtype=torch.float64, device=device)
    truth = U.DeviceEnsemble(f_true, true_alg, tspan, t, u0_d, **true_tol)
    X = truth.solve(torch.zeros(1, dtype=torch.float64, device=device)).clone()
    torch.cuda.synchronize()
    assert int((truth.retcode != 0).sum()) == 0
    return dict(theta=theta, u0=u0_d, t=t, data=X, tspan=tspan, f=f_ude, mask=mask, alg=alg, tol=tol)


def synth_inputs(N, rank, device):
    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "Scenario_1_recovery_0.005.json")))
    theta = np.array(g["initial_parameters"])
    rng = np.random.default_rng(1234 + rank)
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (N, 2)))       # scenario_1.jl:38
    t = np.arange(31) * 0.1
    u0_d = torch.tensor(u0, dtype=torch.float64, device=device)
    # data = true Lotka-Volterra (lotka!, p_ = (1.3,0.9,0.8,1.8)) at tol 1e-12 + 5e-3*mean noise (scenario_1.jl:30-53),
    # generated on the GPU by the same engine (untimed)
    truth = U.DeviceEnsemble(models.lotka(), U.Vern7(), (0.0, 3.0), t, u0_d, abstol=1e-12, reltol=1e-12)
    p_true = torch.tensor([1.3, 0.9, 0.8, 1.8], dtype=torch.float64, device=device)
    X = truth.solve(p_true).clone()
    torch.cuda.synchronize()
    assert int((truth.retcode != 0).sum()) == 0
    noise = torch.tensor(rng.standard_normal((N, 31, 2)), dtype=torch.float64, device=device)
    data = X + 5e-3 * X.mean(dim=1, keepdim=True) * noise
    return theta, u0_d, t, data


def cpu_baseline(theta, u0, t, data, seconds_target=15.0, workload="lv", mask=None):
    """The CPU restatement (oracle, kind "port") on this box's host cores, bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = os.cpu_count() or 1
    if workload == "lv":
        m, o = O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6)
    elif workload == "seir":
        m, o = O.seir_ude(), O.opts(O.VERN7, 1e-6, 1e-6)
    else:
        m, o = O.kpp_ude(u0.shape[1]), O.opts(O.TSIT5)
    n = min(len(u0), cores if workload != "lv" else 64 * cores)
    t0 = time.perf_counter()
    r = O.loss_grad_ensemble(m, o, u0[:n], [t[0], t[-1]], theta, t, data[:n], row_mask=mask, nthreads=cores)
    dt = time.perf_counter() - t0
    per_traj = dt / n
    n2 = int(min(len(u0), max(n, seconds_target / max(per_traj, 1e-9))))
    if n2 > n:
        t0 = time.perf_counter()
        r = O.loss_grad_ensemble(m, o, u0[:n2], [t[0], t[-1]], theta, t, data[:n2], row_mask=mask, nthreads=cores)
        dt = time.perf_counter() - t0
        n = n2
    evals = int(r["stats"][:, 0].sum() + r["stats"][:, 4].sum())
    return {"value": evals / dt, "unit": "RHS-evals/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d trajectories, one loss+adjoint-gradient pass, OpenMP over trajectories (%.1f s)" % (n, len(u0), dt)}


def pmc_traffic(a):
    """HBM bytes per adj_kernel launch of the default C2 command, as collected by `rocprofv3 --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` (tools/prof_pass.sh; counters in KB) and committed under profiles/ -- PMC passes cannot run
    inside the timed bench itself.  None when the run is not that command or the summary is absent."""
    if a.workload != "lv" or a.net != "s1" or a.alg != "tsit5" or a.sensealg != "adjoint" or a.lanes or a.waves or a.traj:
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_v4_pmc.md")
    try:
        txt = open(path).read()
    except OSError:
        return None
    sec = txt.split("### ")
    for blk in sec:
        if blk.startswith("`void adj_kernel<LvUde"):
            vals = {}
            for line in blk.splitlines():
                c = [x.strip() for x in line.split("|")]
                if len(c) >= 5 and c[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[c[1]] = float(c[4])
            if len(vals) == 2:
                return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--traj", type=int, default=0, help="trajectories per GPU (0 = workload default)")
    ap.add_argument("--workload", default="lv", choices=["lv", "seir", "kpp"],
                    help="lv = BASELINE configs[1] (the headline); seir / kpp = configs[2] per-GPU share / configs[3]")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per trajectory (0 = library default)")
    ap.add_argument("--waves", type=int, default=0, help="adjoint kernel variant: waves per SIMD (0 = default)")
    ap.add_argument("--alg", default="tsit5")
    ap.add_argument("--sensealg", default="adjoint", choices=["adjoint", "discrete"],
                    help="adjoint = InterpolatingAdjoint (the north-star path); discrete = frozen-step reverse sweep (a9)")
    ap.add_argument("--net", default="s1", choices=["s1", "tanh32"],
                    help="lv workload: s1 = the reference 2-5-5-5-2 rbf chain (headline), tanh32 = BASELINE's '2-layer tanh' 2-32-2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    from universal_differential_equations_amd.parallel import allreduce_grad

    N = a.traj or {"lv": 10000, "seir": 6250, "kpp": 256}[a.workload]
    mask = None
    if a.workload == "lv":
        theta_h, u0_d, t, data = synth_inputs(N, rank, device)
        alg = U.Tsit5() if a.alg == "tsit5" else U.Vern7()
        f_lv = models.ude_dynamics()
        if a.net == "tanh32":
            f_lv = models.ude_dynamics(models.tanh32_chain())
            theta_h = 0.1 * models.tanh32_chain().glorot_uniform(np.random.default_rng(7))
        ens = U.DeviceEnsemble(f_lv, alg, (0.0, 3.0), t, u0_d, data=data, lanes_per_traj=a.lanes,
                               waves_per_simd=a.waves, abstol=1e-6, reltol=1e-6,
                               sensealg=U.ForwardDiffSensitivity() if a.sensealg == "discrete" else None)
        wl_name = ("BASELINE configs[1]: LV UDE (%s), %d trajectories per GPU, "
                   "%s abstol=reltol=1e-6, 31 save points, loss + InterpolatingAdjoint gradient"
                   % ("2-5-5-5-2 rbf, 87 params, theta_init of scenario_1" if a.net == "s1" else "2-32-2 tanh, 162 params, 0.1 x glorot", N, a.alg))
    else:
        w = synth_inputs_other(a.workload, N, rank, device)
        theta_h, u0_d, t, data, mask = w["theta"], w["u0"], w["t"], w["data"], w["mask"]
        ens = U.DeviceEnsemble(w["f"], w["alg"], w["tspan"], t, u0_d, data=data, row_mask=mask, lanes_per_traj=a.lanes,
                               sensealg=U.ForwardDiffSensitivity() if a.sensealg == "discrete" else None, **w["tol"])
        wl_name = {"seir": "BASELINE configs[2] per-GPU share: SEIR exposure UDE (7 states, NN 3-64-64-1 tanh, 4481 params), %d trajectories "
                           "per GPU, Vern7 abstol=reltol=1e-6, 22 save points, loss rows 2:4 + InterpolatingAdjoint gradient",
                   "kpp": "BASELINE configs[3]: Fisher-KPP UDE, 1024 points (dx = 0.04), NN 1-10-20-10-1 tanh + 3-tap stencil (466 params), "
                          "%d PDEs per GPU, Tsit5 default tol, 11 save points, loss + InterpolatingAdjoint gradient"}[a.workload] % N
    theta = torch.tensor(theta_h, dtype=torch.float64, device=device)

    def step():
        g = ens.loss_grad(theta)
        allreduce_grad(g, dist)
        return g

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    fwd_ms, bwd_ms = [], []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        # HIP events recorded by the library on the launch stream; read after the loop would only see the last pair
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-kernel durations (HIP events on the launch stream), untimed extra passes
    for _ in range(5):
        ens.loss_grad(theta)
        torch.cuda.synchronize()
        f, b = ens.kernel_ms()
        fwd_ms.append(f)
        bwd_ms.append(b)
    nf_fwd = int(ens.stats[:, 0].sum().item())
    nf_bwd = int(ens.stats[:, 4].sum().item())
    nfail = int((ens.retcode != 0).sum().item())
    evals = torch.tensor([nf_fwd + nf_bwd, nfail], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(evals)
    total_evals = float(evals[0].item())
    ms_per_step = elapsed / a.steps * 1e3
    value = total_evals * a.steps / elapsed

    if rank == 0:
        bwd = float(np.mean(bwd_ms)) * 1e-3
        flops_bwd = nf_bwd * FLOPS[a.workload][1]
        achieved = flops_bwd / bwd / 1e12
        out = {
            "metric": "ODE RHS-evals/s (fwd+adjoint)", "value": value, "unit": "RHS-evals/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name,
                       "trajectories_per_gpu": N, "sensealg": a.sensealg, "lanes_per_trajectory": a.lanes or "default",
                       "evals_per_step_fwd": nf_fwd, "evals_per_step_bwd": nf_bwd, "failed_trajectories": int(evals[1].item()),
                       "adjoint_grad_wallclock_ms": ms_per_step, "fwd_kernel_ms": float(np.mean(fwd_ms)),
                       "bwd_kernel_ms": float(np.mean(bwd_ms))},
            "roofline": {"bound": "mfma", "kernel": "adj_kernel (interpolating adjoint)", "achieved": achieved,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": pmc_traffic(a),
                         "note": "FP64 vector/matrix peak (both 78.6 TF); algorithmic %g flop per adjoint eval; "
                                 "the path is FP64-ALU/latency bound, not HBM bound (DESIGN.md); traffic = FETCH_SIZE + "
                                 "WRITE_SIZE bytes per adj_kernel launch from the separate rocprofv3 --pmc passes of this "
                                 "command (profiles/r01_v4_pmc.md), null for other workloads" % FLOPS[a.workload][1]},
        }
        if not a.no_cpu_baseline and world == 1:  # (the CPU leg is a rank-0, N=1 measurement)
            out["cpu_baseline"] = cpu_baseline(theta_h, u0_d.cpu().numpy(), t, data.cpu().numpy(), workload=a.workload, mask=mask)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
