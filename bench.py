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
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 (MI355X guide: 64 FLOP/clk/SIMD)
HJB_FLOP_PER_EVAL = 2.0 * (101 * 110 + 110 * 110 * 2 + 110 * 100)   # one sigma^T grad u chain evaluation (lambaem.jl:27-30)
HJB_FLOP_PER_BWD_COL = 2.0 * (100 * 110 + 110 * 110 * 2) + 2.0 * (102 * 110 + 111 * 110 * 2 + 111 * 100)  # delta chain + outer products
# algorithmic flop per work unit (forward RHS eval, adjoint eval): SURVEY.md 8(d) roofline table
FLOPS = {"lv": (160.0, 550.0), "seir": (8744.0, 26000.0), "kpp": (0.87e6, 2.6e6),
         # neural ODE 7-64-64-64-7: forward 2*(7*64 + 2*64*64 + 64*7) = 18.2 kflop; adjoint = forward + transposed products + outer products
         "node": (18176.0, 54500.0),
         # BASELINE's literal "2-layer tanh MLP" 2-32-2 on the LV ensemble: forward 2*(2*32 + 32*2) = 256 flop (+ 2 for the diagonal terms);
         # adjoint = forward + the two transposed products + the 162 outer-product / bias entries
         "lv_tanh32": (258.0, 840.0),
         # the LV network edited to 2-8-8-8-2: forward 2*(2*8 + 2*8*8 + 8*2) + 2 = 322 flop; adjoint = forward + the transposed products (320) +
         # the 186 outer-product / bias entries (2 flop each)
         "lv_shape8": (322.0, 1000.0)}


HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E (MI355X_MICROARCH.md)
# The roof that bounds each workload's dominant (backward) kernel -- `roofline.bound`:
#   valu  the FP64 vector unit (78.6 TF): the LV kernels (5- and 32-wide layers stay on the vector unit, as north_star says)
#   mfma  the FP64 / FP32 matrix cores: the fast-mode lock-step kernels (78.6 TF), the deep-BSDE step (157.3 TF).  (Fisher-KPP: "valu" since round 6 --
#         its network runs on the vector unit; on gfx950 the FP64 matrix instructions share the vector issue port and the 78.6 TF either way)
#   hbm   the lock-step SEIR / neural-ODE backward kernels in parity mode: the parameter cotangent mu (np doubles per slot) lives in
#         HBM and every step attempt reads the current column and writes the candidate -- 2 * np * 8 B per attempt, the algorithmic
#         stream SURVEY.md 8(d) C3 names ("HBM in parity mode"); their flop fraction is reported beside it (`flop_frac`)
# the headline command at other fillings of the chip / with the wavefront-per-trajectory layout: name -> (trajectories, lanes per trajectory)
LV_VARIANTS = {"lv_sat40k": (40000, 0), "lv_sat160k": (160000, 0), "lv_wave64": (10000, 64)}
# the workloads the default command measures beside the headline (config.other_workloads), in this order
OTHER_WORKLOADS = ("lv_trained", "seir", "seir_fast", "seir_shape63", "node_fast", "kpp", "hjb", "hjb_script_tol", "node", "lv_tanh32", "lv_tanh5", "lv_shape8",
                   "lv_discrete", "lv_sat40k", "lv_sat160k", "lv_wave64")


def kernel_within_step(entry):
    """a workload line is self-consistent when its dominant kernel's time -- and, where reported, forward + backward kernel together --
    does not exceed the step it is part of (2 % for the clocks: HIP events against the host's wall clock)"""
    ks = [entry.get("kernel_ms") or 0.0, (entry.get("kernel_ms") or 0.0) + (entry.get("fwd_kernel_ms") or entry.get("bwd_kernel_ms") or 0.0)]
    return max(ks) <= 1.02 * entry["ms_per_step"]
BOUND = {"lv_trained": "valu", "lv_tanh5": "valu", "lv_shape8": "valu", "seir_shape63": "hbm", "seir_fast": "mfma", "node_fast": "mfma", "lv_sat40k": "valu", "lv_sat160k": "valu", "lv_wave64": "valu", "lv": "valu", "lv_tanh32": "valu", "lv_discrete": "valu", "kpp": "valu", "hjb": "mfma", "seir": "hbm", "node": "hbm"}


# `roofline.traffic` is NOT measured inside this run (PMC passes cannot run inside a timed bench): it is read from the committed
# rocprofv3 --pmc summary of the same command and binary
TRAFFIC_SOURCE = "from profiles/: FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc passes of this command (tools/prof_r06.sh), not a counter read of this run"


def mu_stream_bytes(stats, n_param):
    """algorithmic HBM bytes of one backward launch of a deferred-cotangent kernel: (accepted + rejected) step attempts x (read mu, write candidate)"""
    return float((stats[:, 5].sum() + stats[:, 6].sum()).item()) * 2.0 * n_param * 8.0


def headline_roofline(a, fkey, achieved_tflops, bwd_s, stats, n_param):
    """the `roofline` object of the JSON line for the dominant (backward) kernel: `bound` names the roof that bounds it (BOUND above),
    achieved / peak / frac are against THAT roof; hbm-bound kernels carry their flop fraction beside it"""
    bound = BOUND["lv_tanh32" if fkey == "lv_tanh32" else a.workload]
    if a.workload in ("seir", "node") and a.sensealg == "fast" and a.lanes in (0, 16):
        bound = "mfma"   # the block-level matrix-core accumulation: no mu in HBM at all ("fast mode: FP64 MFMA is the bound")
    elif a.sensealg != "adjoint" or a.lanes not in (0, 16):
        bound = "valu" if bound == "hbm" else bound   # (the wavefront-per-trajectory / fast / discrete kernels of seir and node)
    r = {"bound": bound, "kernel": roofline_kernel_name(a), "achieved": achieved_tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
         "frac": achieved_tflops / FP64_PEAK_TFLOPS, "traffic": pmc_traffic(a), "traffic_source": TRAFFIC_SOURCE,
         "note": "bound: valu = FP64 vector unit, mfma = FP64 matrix cores (78.6 TF each, and on gfx950 ONE issue port: DESIGN.md 2b), hbm = 8 TB/s; algorithmic %g flop per "
                 "adjoint eval; traffic = FETCH_SIZE + WRITE_SIZE bytes per launch of the kernel from the separate rocprofv3 --pmc passes of this "
                 "command (profiles/r06_pmc_<workload>.md, tools/prof_r06.sh), null for non-default commands" % FLOPS[fkey][1]}
    if bound == "hbm":
        gbps = mu_stream_bytes(stats, n_param) / bwd_s / 1e9
        r.update({"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                  "flop_frac": achieved_tflops / FP64_PEAK_TFLOPS, "achieved_tflops": achieved_tflops})
    return r


SENSE_NAME = {"adjoint": "InterpolatingAdjoint", "discrete": "discretise-then-optimise (ForwardDiffSensitivity-equivalent)",
              "fast": "InterpolatingAdjoint with lambda-only error control (SURVEY 8(b) fast mode; not the reference's step sequence)"}
# which unit dominates each backward kernel: the LV / SEIR kernels run on the FP64 VALU, Fisher-KPP on the FP64 matrix cores
# (both peaks are 78.6 TF; the schema's "bound" offers hbm | mfma)
BWD_KERNEL = {"adjoint": "adj_kernel (interpolating adjoint)", "discrete": "dadj_kernel (frozen-step reverse sweep)",
              "fast": "adj_kernel, fast mode (lambda-only error control)"}


def roofline_kernel_name(a):
    """the dominant kernel of the command, as rocprofv3 names it, and the unit it runs on"""
    ls = a.sensealg == "adjoint" and a.lanes in (0, 16)
    if a.workload == "node" and a.sensealg == "fast" and a.lanes in (0, 16):
        return ("nodelf::node_lsf_adj_kernel (fast mode: lambda-only error control, 16 trajectories per block in lock-step, network AND "
                "parameter cotangent on the FP64 matrix cores -- block-resident accumulators, no mu in HBM)")
    if a.workload == "seir" and a.sensealg == "fast" and a.lanes in (0, 16):
        return ("seirlf::seir_lsf_adj_kernel (fast mode: lambda-only error control, 16 trajectories per block in lock-step, network AND "
                "parameter cotangent on the FP64 matrix cores -- block-resident accumulators, no mu in HBM)")
    if a.workload == "seir" and ls:
        return ("seirls2::seir_ls2_adj_kernel (interpolating adjoint, 16 trajectories per block in lock-step), network on the FP64 matrix cores, "
                "parameter-slot sums / controller on the FP64 vector unit")
    if a.workload == "node" and ls:
        return ("nodels2::node_ls2_adj_kernel (interpolating adjoint, 16 trajectories per block in lock-step), network on the FP64 matrix cores, "
                "parameter-slot sums / controller on the FP64 vector unit")
    if a.workload == "kpp":
        return BWD_KERNEL[a.sensealg] + (", FP64 vector unit (pointwise network, weights broadcast out of registers by DPP) + FP64 matrix cores (parameter "
                                         "contraction): ONE issue port on gfx950, one 78.6 TF roof (DESIGN.md 2b)")
    return BWD_KERNEL[a.sensealg] + ", FP64 VALU (no MFMA: 5- and 64-wide layers stay on the vector unit)"


def SENSE_OBJ(U, name):
    return U.ForwardDiffSensitivity() if name == "discrete" else U.FastInterpolatingAdjoint() if name == "fast" else None


def synth_inputs_other(workload, N, rank, device):
    """SEIR (BASELINE configs[2] per-GPU share) and Fisher-KPP (configs[3]) synthetic inputs, SURVEY.md 8(d)."""
    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    rng = np.random.default_rng(1234 + rank)
    if workload in ("seir", "node"):
        S0 = 14e6
        u0 = np.zeros((N, 7))
        u0[:, 0] = rng.uniform(0.8, 0.95, N) * S0
        u0[:, 4] = S0
        t = np.arange(22.0)
        tspan, f_true, f_ude = (0.0, 21.0), models.corona(), models.dudt_() if workload == "seir" else models.dudt_node()
        theta = (models.seir_chain() if workload == "seir" else models.seir_node_chain()).glorot_uniform(rng)
        if workload == "node":
            u0[:, 1] = rng.uniform(5.0, 20.0, N)      # a few exposed, so that the data rows 2:4 are not identically zero
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


def reference_julia():
    """the reference's own DiffEqFlux path is timed "only if the GPU box provides it".  Observed, not assumed: is there
    a `julia` on this box (PATH and the usual install prefixes)?  "absent" -> `cpu_baseline.kind` stays "port" (the oracle); a path ->
    reported as found (its package depot is not in this image either: no network -- the scripts' Manifest cannot be instantiated)."""
    import glob
    import shutil
    exe = shutil.which("julia")
    if exe is None:
        cand = sorted(glob.glob("/opt/julia*/bin/julia") + glob.glob("/usr/local/julia*/bin/julia") + glob.glob(os.path.expanduser("~/.juliaup/bin/julia")))
        exe = cand[0] if cand else None
    return "absent" if exe is None else "found at %s (not run: the scripts' Manifest.toml cannot be instantiated without a package server)" % exe


def cpu_baseline(theta, u0, t, data, seconds_target=15.0, workload="lv", mask=None):
    """The CPU restatement (oracle, kind "port") on this box's host cores, bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    cores = os.cpu_count() or 1
    if workload == "lv":
        m, o = O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6)
    elif workload in ("seir", "node"):
        m, o = (O.seir_ude() if workload == "seir" else O.seir_node()), O.opts(O.VERN7, 1e-6, 1e-6)
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
    return {"value": evals / dt, "unit": "RHS-evals/s", "cores": cores, "kind": "port", "reference_julia": reference_julia(),
            "sample": "%d of the %d trajectories, one loss+adjoint-gradient pass, OpenMP over trajectories (%.1f s)" % (n, len(u0), dt)}


def pmc_traffic(a):
    """HBM bytes per launch of the dominant backward kernel of the default command of each workload, as collected by
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/prof_r02.sh; counters in KB) and committed under profiles/ --
    PMC passes cannot run inside the timed bench itself.  None when the run is not that command or the summary is absent."""
    if a.workload == "seir" and a.sensealg == "fast" and not (a.lanes or a.waves or a.traj):
        return pmc_any("seir_fast", "seirlf::seir_lsf_adj_kernel<")
    if a.workload == "node" and a.sensealg == "fast" and not (a.lanes or a.waves or a.traj):
        return pmc_any("node_fast", "nodelf::node_lsf_adj_kernel<")
    if a.net != "s1" or a.alg != "tsit5" or a.lanes or a.waves or a.traj or a.sensealg == "fast":
        return None
    kern = "adj_kernel<" if a.sensealg == "adjoint" else "dadj_kernel<"
    if a.workload == "seir" and a.sensealg == "adjoint" and a.lanes in (0, 16):
        kern = "seirls2::seir_ls2_adj_kernel<"     # the lock-step matrix-core backward kernel (the default)
    if a.workload == "node" and a.sensealg == "adjoint" and a.lanes in (0, 16):
        kern = "nodels2::node_ls2_adj_kernel<"
    return pmc_any(a.workload, "`void " + kern)


def run_hjb(a, rank, world, local, device, dist):
    """SURVEY.md 8(f) N1 / BASELINE configs[4] per-GPU share: one loss + gradient evaluation of the deep-BSDE training
    step of highdim_pde/lambaem.jl (d = 100, hls = 110, lambda = 1, x0 = 0, tspan (0, 1), adaptive LambaEM) for
    `--traj` trajectories per GPU.  Tolerances: the script's 1e-4 take 1.4e4 .. 4.8e4 steps per trajectory under this
    restatement of Lamba's estimator (oracle/sde_oracle.h; examples/highdim_pde_lambaem.py runs that call); the bench uses
    abstol = reltol = --tol (default 0.1, ~220 steps; 1e-2: ~700 steps; the accepted-step store grows by itself)."""
    from universal_differential_equations_amd import pde
    M = a.traj or 16384     # 64 slots per CU: every slot serves two trajectories on average through the queue
    alg = pde.NNPDENS(100, 110, opt=pde.ADAM(0.03))
    theta_h = alg.init_params(np.random.default_rng(0))
    prob = pde.TerminalPDEProblem(pde.hjb(1.0), np.zeros(100), (0.0, 1.0))
    bs = pde.DeviceBSDE(prob, alg, pde.LambaEM(), M, device=device, abstol=a.tol, reltol=a.tol, seed=1234 + rank, max_steps=a.max_steps)
    theta = torch.tensor(theta_h, device=device)
    # mean over all ranks' trajectories: ONE all-reduce of [grad; loss].  --allreduce udecore: through libudecore's own RCCL binding
    # (ude_allreduce_grad on double[np + 1], what a non-Python host calls), bootstrapped over the torch.distributed group
    comm = None
    if dist is not None and a.allreduce == "udecore":
        from universal_differential_equations_amd.parallel import Comm
        comm = Comm.from_torch_dist(bs.eng, dist)
    buf = torch.zeros(bs.np + 1, dtype=torch.float64 if comm is not None else torch.float32, device=device)

    def step(it=0):
        loss, g = bs.loss_grad(theta, it=it)
        if dist is not None:
            buf[:-1] = g
            buf[-1] = loss[0].to(buf.dtype)
            if comm is not None:
                bs.eng.set_stream(torch.cuda.current_stream().cuda_stream)
                comm.allreduce(buf)
            else:
                dist.all_reduce(buf)
            buf.div_(world)
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    fwd_ms, bwd_ms = [], []
    for i in range(3):
        step(i)
        torch.cuda.synchronize()
        f, b = bs.kernel_ms()
        fwd_ms.append(f)
        bwd_ms.append(b)
    nf = int(bs.stats[:, 0].sum().item())
    nacc = int(bs.stats[:, 1].sum().item())
    nfail = int((bs.retcode != 0).sum().item())
    ev = torch.tensor([nf + nacc, nfail], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(ev)
    ms_per_step = elapsed / a.steps * 1e3
    if rank == 0:
        fwd = float(np.mean(fwd_ms)) * 1e-3
        achieved = nf * HJB_FLOP_PER_EVAL / fwd / 1e12
        out = {
            "metric": "ODE RHS-evals/s (fwd+adjoint)", "value": float(ev[0].item()) * a.steps / elapsed, "unit": "RHS-evals/s",
            "n_gpus": world, "rccl_ranks": rccl_ranks(a, dist), "allreduce": None if dist is None else a.allreduce,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4] per-GPU share: highdim_pde/lambaem.jl deep-BSDE step (100-dim HJB, NNPDENS chains "
                                   "100-110-110-1 and 101-110-110-110-100 relu, 70171 params), %d trajectories per GPU, adaptive LambaEM "
                                   "abstol=reltol=%g, loss + gradient through the stepper" % (M, a.tol),
                       "trajectories_per_gpu": M, "net_evals_per_step_fwd": nf, "bwd_columns_per_step": nacc,
                       "steps_per_trajectory_mean": nacc / M, "failed_trajectories": int(ev[1].item()),
                       "fwd_kernel_ms": float(np.mean(fwd_ms)), "bwd_kernel_ms": float(np.mean(bwd_ms)),
                       "bwd_achieved_tflops": nacc * HJB_FLOP_PER_BWD_COL / (float(np.mean(bwd_ms)) * 1e-3) / 1e12},
            "roofline": {"bound": "mfma", "kernel": "hjb_fwd_kernel (three batched network evaluations per step attempt on v_mfma_f32_32x32x2_f32)",
                         "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": pmc_any("hjb", "hjb_fwd_kernel") if not a.traj and a.tol == 0.1 else None, "traffic_source": TRAFFIC_SOURCE,
                         "note": "FP32 matrix peak 157.3 TF; algorithmic %g flop per network evaluation x evaluations of live "
                                 "trajectories / forward kernel time (HIP events)" % HJB_FLOP_PER_EVAL},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_hjb(theta_h, a.tol)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_hjb(theta_h, tol, seconds_target=15.0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _sde_oracle as S
    cores = os.cpu_count() or 1
    D = S.desc(abstol=tol, reltol=tol, seed=1234)
    n = max(cores, 8)
    t0 = time.perf_counter()
    r = S.loss_grad(D, n, np.zeros(100), theta_h, nthreads=cores)
    dt = time.perf_counter() - t0
    n2 = int(max(n, min(8192, seconds_target / max(dt / n, 1e-9))))
    if n2 > n:
        t0 = time.perf_counter()
        r = S.loss_grad(D, n2, np.zeros(100), theta_h, nthreads=cores)
        dt = time.perf_counter() - t0
        n = n2
    evals = int(r["stats"][:, 0].sum() + r["stats"][:, 1].sum())
    return {"value": evals / dt, "unit": "RHS-evals/s", "cores": cores, "kind": "port", "reference_julia": reference_julia(),
            "sample": "%d trajectories, one loss+gradient pass of the CPU restatement, OpenMP over trajectories (%.1f s)" % (n, dt)}


def pmc_any(stem, kernel_prefix):
    """this round's PMC summary of a workload if it has been collected, else last round's"""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        v = pmc_traffic_file("%s_pmc_%s.md" % (rnd, stem), kernel_prefix)
        if v:
            return v
    return None


def pmc_traffic_file(fname, kernel_prefix):
    """FETCH_SIZE + WRITE_SIZE (KB -> bytes) per launch of a kernel from a committed rocprofv3 --pmc summary, or None"""
    try:
        txt = open(os.path.join(ROOT, "profiles", fname)).read()
    except OSError:
        return None
    for blk in txt.split("### "):
        if kernel_prefix in blk.split("\n", 1)[0]:
            vals = {}
            for line in blk.splitlines():
                c = [x.strip() for x in line.split("|")]
                if len(c) >= 5 and c[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[c[1]] = float(c[4])
            if len(vals) == 2:
                return (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return None


def timed_steps(fn, steps, every=False):
    """median wall-clock of `steps` individually synchronised calls, in ms (a few steps only: one host hiccup -- a page fault, a
    late code-object load -- must not become the number); every = True: all of them"""
    ts = []
    for i in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return ts if every else float(np.median(ts))


def edited_net(net):
    """the EDITED networks of the run-time-shape lines: (model, theta, description)"""
    from universal_differential_equations_amd import models
    if net == "tanh5":
        ch = models.Chain(models.Dense(2, 5, "tanh"), models.Dense(5, 5, "tanh"), models.Dense(5, 5, "tanh"), models.Dense(5, 2, "identity"))
        return models.ude_dynamics(ch), 0.3 * ch.glorot_uniform(np.random.default_rng(7)), "2-5-5-5-2 tanh (activations edited), 87 params, 0.3 x glorot"
    if net == "shape8":
        ch = models.Chain(models.Dense(2, 8, "tanh"), models.Dense(8, 8, "tanh"), models.Dense(8, 8, "tanh"), models.Dense(8, 2, "identity"))
        return models.ude_dynamics(ch), 0.1 * ch.glorot_uniform(np.random.default_rng(7)), "2-8-8-8-2 tanh (widths edited), 186 params, 0.1 x glorot"
    if net == "shape63":
        ch = models.Chain(models.Dense(3, 64, "tanh"), models.Dense(64, 63, "tanh"), models.Dense(63, 1, "identity"))
        return models.dudt_(ch), ch.glorot_uniform(np.random.default_rng(0)), "3-64-63-1 tanh (one neuron edited away), 4415 params"
    raise ValueError(net)


def quick_measure(name, device, steps=5, warmup=1):
    """One of the OTHER workloads, measured in the same process as the headline (driver-observed): the median of `steps`
    individually synchronised steps after `warmup`, kernel times from the library's HIP events, algorithmic roofline fraction of the
    dominant kernel as in the stand-alone `--workload` runs.  Returns a small dict for config.other_workloads."""
    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    t_setup = time.perf_counter()
    if name in ("hjb", "hjb_script_tol"):
        from universal_differential_equations_amd import pde
        # hjb: the throughput configuration (16 384 trajectories per GPU, tol 0.1); hjb_script_tol: the SCRIPT's own call
        # (lambaem.jl:27-34: trajectories = 100, abstol = reltol = 1e-4) -- 1.4e4 .. 4.8e4 steps per trajectory, a latency measurement
        M, tol = (16384, 0.1) if name == "hjb" else (100, 1e-4)
        alg = pde.NNPDENS(100, 110, opt=pde.ADAM(0.03))
        theta = torch.tensor(alg.init_params(np.random.default_rng(0)), device=device)
        prob = pde.TerminalPDEProblem(pde.hjb(1.0), np.zeros(100), (0.0, 1.0))
        bs = pde.DeviceBSDE(prob, alg, pde.LambaEM(), M, device=device, abstol=tol, reltol=tol, seed=1234)
        if name == "hjb_script_tol":
            steps = min(steps, 2)
        for i in range(warmup):
            bs.loss_grad(theta, it=i)
        # (step and kernel time from the SAME calls -- the Philox iteration `it` changes the step count of the adaptive solve from call
        #  to call: the median step of some calls next to the kernel time of the last one once reported a kernel longer than its step)
        kms = []

        def one_bsde(i):
            bs.loss_grad(theta, it=i, check_store=False)
            torch.cuda.synchronize()
            kms.append(bs.kernel_ms())
        ts = timed_steps(one_bsde, steps, every=True)
        pick = int(np.argsort(ts)[len(ts) // 2])        # the median step, and ITS kernel times
        ms, (f, b) = ts[pick], kms[pick]
        nf, nacc = int(bs.stats[:, 0].sum().item()), int(bs.stats[:, 1].sum().item())
        ach = nf * HJB_FLOP_PER_EVAL / (f * 1e-3) / 1e12
        return {"workload": "configs[4] per-GPU share: deep-BSDE step, 16384 trajectories, LambaEM tol 0.1" if name == "hjb" else
                "highdim_pde/lambaem.jl's own call: one deep-BSDE step, 100 trajectories, LambaEM abstol = reltol = 1e-4", "ms_per_step": ms,
                "evals_per_s": (nf + nacc) / (ms * 1e-3), "dominant_kernel": "hjb_fwd_kernel", "kernel_ms": f, "bwd_kernel_ms": b,
                "bound": "mfma", "achieved_tflops": ach, "peak_tflops": FP32_MFMA_PEAK_TFLOPS, "frac": ach / FP32_MFMA_PEAK_TFLOPS, "unit": "mfma-f32",
                "failed_trajectories": int((bs.retcode != 0).sum().item()), "traffic": pmc_any("hjb", "hjb_fwd_kernel")}
    sense, wl, net = "adjoint", name, "s1"
    lanes, n_lv = 0, 10000
    if name == "lv_tanh32":
        wl, net = "lv", "tanh32"
    elif name == "lv_tanh5":
        wl, net = "lv", "tanh5"    # configs[1] with the ACTIVATIONS edited (rbf -> tanh), the scripts' widths: the five-lane run-time-shape instance
    elif name == "lv_shape8":
        wl, net = "lv", "shape8"   # configs[1] with an EDITED network, 2-8-8-8-2 tanh: no compiled instance, the run-time-shape lane-group instance
    elif name == "lv_discrete":
        wl, sense = "lv", "discrete"
    elif name == "lv_trained":
        wl, net = "lv", "trained"   # "theta_init and theta_trained (two runs)" -- the headline command at the fixture's TRAINED parameters
    elif name == "seir_fast":
        wl, sense = "seir", "fast"
    elif name == "node_fast":
        wl, sense = "node", "fast"
    elif name == "seir_shape63":
        wl = "seir"    # the exposure UDE with an EDITED network, 3-64-63-1: no compiled instance, the runtime-shape lock-step instances
    elif name in LV_VARIANTS:
        # the headline command with the chip FILLED (10 000 trajectories are 834 wavefronts on 1024 SIMDs: one partial round), and
        # with north_star's literal "one wavefront per trajectory" layout (64 lanes, lane j = neuron j: the runtime-shape kernel)
        wl, (n_lv, lanes) = "lv", LV_VARIANTS[name]
    mask = None
    if wl == "lv":
        N = n_lv
        theta_h, u0_d, t, data = synth_inputs(N, 0, device)
        f_lv = models.ude_dynamics()
        if net == "tanh32":
            f_lv = models.ude_dynamics(models.tanh32_chain())
            theta_h = 0.1 * models.tanh32_chain().glorot_uniform(np.random.default_rng(7))
        if net in ("tanh5", "shape8"):
            f_lv, theta_h, _ = edited_net(net)
        if net == "trained":   # scenario_1.jl:113-126: the parameters ADAM + BFGS end at (the stored artifact's `trained_parameters`)
            theta_h = np.array(json.load(open(os.path.join(ROOT, "tests", "golden", "Scenario_1_recovery_0.005.json")))["trained_parameters"])
        ens = U.DeviceEnsemble(f_lv, U.Tsit5(), (0.0, 3.0), t, u0_d, data=data, abstol=1e-6, reltol=1e-6, sensealg=SENSE_OBJ(U, sense), lanes_per_traj=lanes)
        desc = "configs[1] with %s" % ("the 2-32-2 tanh net (BASELINE's literal '2-layer tanh MLP')" if net == "tanh32" else
                                      "theta_trained of the reference's stored run (scenario_1.jl:113-126) instead of theta_init: the second run of SURVEY 8(d) C2" if net == "trained" else
                                      "the network edited to 2-8-8-8-2 tanh (no compiled instance: the run-time-shape instance of the lane-group kernels, 8 lanes per trajectory)" if net == "shape8" else
                                      "the activations edited to tanh, 2-5-5-5-2 (no compiled instance: the run-time-shape instance on 5 lanes per trajectory, the headline layout)" if net == "tanh5" else
                                      "the discretise-then-optimise gradient"
                                      if sense == "discrete" else "%d trajectories per GPU, %s" % (N, "64 lanes per trajectory (one wavefront per trajectory)" if lanes == 64
                                                                                                    else "5 lanes per trajectory (the headline kernel)"))
    else:
        N = {"seir": 6250, "kpp": 256, "node": 6250}[wl]
        w = synth_inputs_other(wl, N, 0, device)
        theta_h, u0_d, t, data, mask = w["theta"], w["u0"], w["t"], w["data"], w["mask"]
        if name == "seir_shape63":
            w["f"], theta_h, _ = edited_net("shape63")
        ens = U.DeviceEnsemble(w["f"], w["alg"], w["tspan"], t, u0_d, data=data, row_mask=mask, sensealg=SENSE_OBJ(U, sense), **w["tol"])
        desc = {"seir": "configs[2] per-GPU share: SEIR exposure UDE, 6250 trajectories, Vern7 1e-6" +
                        (", fast mode (lambda-only error control; parameter cotangent = block-level matrix-core accumulation, no mu in HBM)" if sense == "fast" else ""),
                "node": "SEIR neural ODE 7-64-64-64-7 on the configs[2] ensemble, 6250 trajectories, Vern7 1e-6" +
                        (", fast mode (block-level matrix-core accumulation of the parameter cotangent)" if sense == "fast" else ""),
                "kpp": "configs[3]: Fisher-KPP UDE, 1024 points x 256 PDEs, Tsit5"}[wl]
        if name == "seir_shape63":
            desc = "configs[2] per-GPU share with the exposure network edited to 3-64-63-1 (no compiled instance: runtime-shape instances of the lock-step matrix-core kernels), 6250 trajectories, Vern7 1e-6"
    theta = torch.tensor(theta_h, dtype=torch.float64, device=device)
    for _ in range(warmup):
        ens.loss_grad(theta)
    # (kernel times: the MEDIAN over the same steps, read after each step's synchronisation -- the last step alone is one sample)
    kms = []

    def one(i):
        ens.loss_grad(theta)
        torch.cuda.synchronize()
        kms.append(ens.kernel_ms())

    ms = timed_steps(one, steps)
    f, b = float(np.median([k[0] for k in kms])), float(np.median([k[1] for k in kms]))
    nf_fwd, nf_bwd = int(ens.stats[:, 0].sum().item()), int(ens.stats[:, 4].sum().item())
    flop_key = "lv_tanh32" if name == "lv_tanh32" else "lv_shape8" if name == "lv_shape8" else wl   # (lv_tanh5, lv_trained: the headline's 2-5-5-5-2 count)
    ach = nf_bwd * FLOPS[flop_key][1] / (b * 1e-3) / 1e12
    kern = "dadj_kernel" if sense == "discrete" else "seirlf::seir_lsf_adj_kernel" if name == "seir_fast" else "nodelf::node_lsf_adj_kernel" if name == "node_fast" else "seirls2::seir_ls2_adj_kernel" if wl == "seir" else "nodels2::node_ls2_adj_kernel" if wl == "node" else "adj_kernel"
    pm = name if name in ("lv_tanh32", "lv_discrete", "seir_fast", "node_fast") else wl
    if name in LV_VARIANTS or name in ("seir_shape63", "lv_shape8", "lv_tanh5", "lv_trained"):
        pm = "none"    # (no committed counter pass for these commands)
    out = {"workload": desc, "ms_per_step": ms, "evals_per_s": (nf_fwd + nf_bwd) / (ms * 1e-3), "dominant_kernel": kern, "kernel_ms": b,
           "fwd_kernel_ms": f, "bound": BOUND[name], "achieved_tflops": ach, "peak_tflops": FP64_PEAK_TFLOPS, "frac": ach / FP64_PEAK_TFLOPS,
           "unit": "valu-f64 + mfma-f64 (one issue port)" if wl == "kpp" else ("mfma-f64 + valu-f64" if wl in ("seir", "node") else "valu-f64"), "failed_trajectories": int((ens.retcode != 0).sum().item()),
           "traffic": pmc_any(pm, "`void " + kern + "<"), "setup_s": time.perf_counter() - t_setup}
    if name in LV_VARIANTS:
        per_wave = 1 if lanes == 64 else 64 // (lanes or 5)
        waves = -(-N // per_wave)
        out.update({"trajectories": N, "lanes_per_trajectory": lanes or 5, "wavefronts": waves, "wavefronts_per_simd": waves / 1024.0,
                    "bwd_evals_per_s": nf_bwd / (b * 1e-3)})
    if BOUND[name] == "hbm":  # frac = against the roof that bounds the kernel; the flop fraction stays beside it
        gbps = mu_stream_bytes(ens.stats, len(theta_h)) / (b * 1e-3) / 1e9
        out.update({"flop_frac": out["frac"], "achieved_gbps": gbps, "peak_gbps": HBM_PEAK_GBPS, "frac": gbps / HBM_PEAK_GBPS})
    return out


def rccl_ranks(a, dist):
    """how many ranks an RCCL communicator of this run spans: the torch.distributed nccl group (= RCCL on ROCm) and / or libudecore's
    own RCCL binding; 0 for a single process, for the gloo rehearsals and for `--allreduce p2p` over a gloo bootstrap"""
    if dist is None:
        return 0
    if getattr(a, "dist_backend", "nccl") == "nccl" or a.allreduce == "udecore":
        return dist.get_world_size()
    return 0


def self_launch(n):
    """re-exec this command line under torch.distributed.run with n ranks on 127.0.0.1 (a free port unless MASTER_PORT is given)"""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: the launcher's WORLD_SIZE, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--traj", type=int, default=0, help="trajectories per GPU (0 = workload default)")
    ap.add_argument("--workload", default="lv", choices=["lv", "seir", "kpp", "hjb", "node"],
                    help="lv = BASELINE configs[1] (the headline); seir / kpp / hjb = configs[2] per-GPU share / configs[3] / configs[4] per-GPU share; "
                         "node = the script's pure neural ODE 7-64-64-64-7 (seir_exposure.jl:53-83) on the configs[2] ensemble")
    ap.add_argument("--tol", type=float, default=0.1, help="hjb workload: abstol = reltol of the adaptive LambaEM solves")
    ap.add_argument("--max-steps", type=int, default=0, help="hjb workload: accepted-step store per trajectory (0 = library default 512)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per trajectory (0 = library default)")
    ap.add_argument("--waves", type=int, default=0, help="adjoint kernel variant: waves per SIMD (0 = default)")
    ap.add_argument("--alg", default="tsit5")
    ap.add_argument("--sensealg", default="adjoint", choices=["adjoint", "discrete", "fast"],
                    help="adjoint = InterpolatingAdjoint (the north-star path); discrete = frozen-step reverse sweep (a9); "
                         "fast = interpolating adjoint with lambda-only error control (opt-in, not the reference's step sequence)")
    ap.add_argument("--net", default="s1", choices=["s1", "tanh32", "tanh5", "shape8", "shape63"],
                    help="lv workload: s1 = the reference 2-5-5-5-2 rbf chain (headline), tanh32 = BASELINE's '2-layer tanh' 2-32-2, tanh5 / shape8 = "
                         "the chain with edited activations / widths (run-time-shape instances); seir workload: shape63 = the exposure network edited to 3-64-63-1")
    ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per step (memset + forward + adjoint + reductions) "
                                                         "instead of launching the six operations individually")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip config.other_workloads (the headline run also measures seir, kpp, hjb, "
                                                             "node, lv_tanh32 and lv_discrete: median of 5 steps each)")
    ap.add_argument("--allreduce", default="torch", choices=["torch", "udecore", "p2p"],
                    help="N > 1: transport of the one all-reduce per gradient: torch.distributed nccl, libudecore's RCCL binding, or libudecore's "
                         "one-shot cross-process P2P reducer (IPC windows, one kernel per rank, rank-ordered deterministic sum; no RCCL)")
    a = ap.parse_args()
    if a.gpus is None:   # (advisor, round 5: `torchrun --nproc-per-node N bench.py` without --gpus runs as N ranks, as it did before round 5)
        a.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    # `python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks ourselves -- the same
    # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` the driver uses, one process per GPU -- and pass its exit
    # code on; rank 0 of the children prints the JSON line.  Under a launcher the world it set up must be the one asked for.
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and not os.environ.get("UDE_BENCH_FORCE_DIST"):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    # (rehearsal of the N > 1 path on a 1-GPU box: UDE_BENCH_DEVICE=0 puts every rank on that device, UDE_BENCH_BACKEND=gloo
    #  replaces RCCL, which refuses two ranks on one device; the driver's runs use neither)
    if os.environ.get("UDE_BENCH_DEVICE"):
        local = int(os.environ["UDE_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("UDE_BENCH_FORCE_DIST"):   # (FORCE_DIST: a one-rank group, so that a 1-GPU box exercises the collective code)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("UDE_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()      # n_gpus on the JSON line = the ranks the process group really has
        a.dist_backend = backend

    if a.workload == "hjb":
        return run_hjb(a, rank, world, local, device, dist)
    import universal_differential_equations_amd as U
    from universal_differential_equations_amd import models
    from universal_differential_equations_amd.parallel import Comm, allreduce_payload, pack_payload

    N = a.traj or {"lv": 10000, "seir": 6250, "kpp": 256, "node": 6250}[a.workload]
    mask = None
    if a.workload == "lv":
        theta_h, u0_d, t, data = synth_inputs(N, rank, device)
        alg = U.Tsit5() if a.alg == "tsit5" else U.Vern7()
        f_lv = models.ude_dynamics()
        net_desc = "2-5-5-5-2 rbf, 87 params, theta_init of scenario_1" if a.net == "s1" else "2-32-2 tanh, 162 params, 0.1 x glorot"
        if a.net == "tanh32":
            f_lv = models.ude_dynamics(models.tanh32_chain())
            theta_h = 0.1 * models.tanh32_chain().glorot_uniform(np.random.default_rng(7))
        elif a.net in ("tanh5", "shape8"):
            f_lv, theta_h, net_desc = edited_net(a.net)
        ens = U.DeviceEnsemble(f_lv, alg, (0.0, 3.0), t, u0_d, data=data, lanes_per_traj=a.lanes,
                               waves_per_simd=a.waves, abstol=1e-6, reltol=1e-6,
                               sensealg=SENSE_OBJ(U, a.sensealg))
        wl_name = ("BASELINE configs[1]: LV UDE (%s), %d trajectories per GPU, "
                   "%s abstol=reltol=1e-6, 31 save points, loss + %s gradient"
                   % (net_desc, N, a.alg, SENSE_NAME[a.sensealg]))
    else:
        w = synth_inputs_other(a.workload, N, rank, device)
        theta_h, u0_d, t, data, mask = w["theta"], w["u0"], w["t"], w["data"], w["mask"]
        if a.workload == "seir" and a.net == "shape63":
            w["f"], theta_h, w["desc_net"] = edited_net("shape63")
        ens = U.DeviceEnsemble(w["f"], w["alg"], w["tspan"], t, u0_d, data=data, row_mask=mask, lanes_per_traj=a.lanes,
                               sensealg=SENSE_OBJ(U, a.sensealg), **w["tol"])
        wl_name = {"seir": "BASELINE configs[2] per-GPU share: SEIR exposure UDE (7 states, NN 3-64-64-1 tanh, 4481 params), %d trajectories "
                           "per GPU, Vern7 abstol=reltol=1e-6, 22 save points, loss rows 2:4 + %s gradient",
                   "node": "SEIR neural ODE (seir_exposure.jl:53-83: 7 states, FastChain 7-64-64-64-7 tanh, 9287 params) on the configs[2] ensemble, %d "
                           "trajectories per GPU, Vern7 abstol=reltol=1e-6, 22 save points, loss rows 2:4 + %s gradient",
                   "kpp": "BASELINE configs[3]: Fisher-KPP UDE, 1024 points (dx = 0.04), NN 1-10-20-10-1 tanh + 3-tap stencil (466 params), "
                          "%d PDEs per GPU, Tsit5 default tol, 11 save points, loss + %s gradient"}[a.workload] % (N, SENSE_NAME[a.sensealg])
        if "desc_net" in w:   # (--net shape63: the exposure network edited, no compiled instance)
            wl_name = wl_name.replace("NN 3-64-64-1 tanh, 4481 params", "NN " + w["desc_net"] + ": run-time-shape instances of the lock-step kernels")
    theta = torch.tensor(theta_h, dtype=torch.float64, device=device)

    # ONE collective per gradient: double[np + 4] = [grad; loss; sum nf; sum naccept; sum nreject].
    # --allreduce udecore: libudecore's own RCCL binding (ude_allreduce_grad, what a non-Python host uses), bootstrapped
    # over the torch.distributed group; default: torch.distributed (backend "nccl" = the same RCCL)
    comm = None
    if dist is not None and a.allreduce == "udecore":
        comm = Comm.from_torch_dist(ens.eng, dist)
    elif dist is not None and a.allreduce == "p2p":
        comm = Comm.p2p_from_torch_dist(ens.eng, dist, len(theta_h) + 4)

    replay = ens.graph(theta) if a.graph else None

    def step():
        g = replay() if replay is not None else ens.loss_grad(theta)
        if dist is None:
            return g
        buf = pack_payload(g, ens.stats)
        if comm is not None:
            ens.eng.set_stream(torch.cuda.current_stream().cuda_stream)
            if getattr(comm, "mp", False):
                comm.allreduce_mp(buf)
            else:
                comm.allreduce(buf)
        else:
            allreduce_payload(buf, dist)
        return buf

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

    # a second, >= 1 s sustained loop (the timed region above is only tens of ms): corroborates ms_per_step
    n_sus = max(a.steps, int(1.2 / max(elapsed / a.steps, 1e-6)))
    barrier()
    ts0 = time.perf_counter()
    for _ in range(n_sus):
        step()
    barrier()
    sustained_ms = (time.perf_counter() - ts0) / n_sus * 1e3

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
    # lock-step utilisation of the backward kernel: trajectories that share a wavefront (or a block of lock-step slots) make their
    # step attempts together, so a group runs as long as its slowest member, and a one-round launch as long as its slowest group
    att = (ens.stats[:, 5] + ens.stats[:, 6]).to(torch.float64).cpu().numpy()
    grp = {"lv": 12 if not a.lanes else max(1, 64 // a.lanes), "seir": 16, "node": 1, "kpp": 1}[a.workload]
    pad = (-len(att)) % grp
    gmax = np.concatenate([att, np.zeros(pad)]).reshape(-1, grp).max(axis=1)
    lane_step_util = float(att.sum() / max((gmax * grp).sum(), 1.0))
    critical_path = float(att.max() / max(att.mean(), 1e-9))
    evals = torch.tensor([nf_fwd + nf_bwd, nfail], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(evals)
    total_evals = float(evals[0].item())
    ms_per_step = elapsed / a.steps * 1e3
    value = total_evals * a.steps / elapsed

    if rank == 0:
        bwd = float(np.mean(bwd_ms)) * 1e-3
        fkey = "lv_tanh32" if (a.workload == "lv" and a.net == "tanh32") else "lv_shape8" if (a.workload == "lv" and a.net == "shape8") else a.workload
        flops_bwd = nf_bwd * FLOPS[fkey][1]
        achieved = flops_bwd / bwd / 1e12
        out = {
            "metric": "ODE RHS-evals/s (fwd+adjoint)", "value": value, "unit": "RHS-evals/s", "n_gpus": world,
            "rccl_ranks": rccl_ranks(a, dist), "allreduce": None if dist is None else a.allreduce,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name,
                       "trajectories_per_gpu": N, "sensealg": a.sensealg, "hip_graph": bool(a.graph), "lanes_per_trajectory": a.lanes or "default",
                       "evals_per_step_fwd": nf_fwd, "evals_per_step_bwd": nf_bwd, "failed_trajectories": int(evals[1].item()),
                       "adjoint_grad_wallclock_ms": ms_per_step, "sustained_loop": {"steps": n_sus, "ms_per_step": sustained_ms},
                       "lane_step_util": lane_step_util, "bwd_attempts_max_over_mean": critical_path,
                       "fwd_kernel_ms": float(np.mean(fwd_ms)),
                       "bwd_kernel_ms": float(np.mean(bwd_ms))},
            "roofline": headline_roofline(a, fkey, achieved, bwd, ens.stats, len(theta_h)),
        }
        if not a.no_cpu_baseline and world == 1:  # (the CPU leg is a rank-0, N=1 measurement)
            out["cpu_baseline"] = cpu_baseline(theta_h, u0_d.cpu().numpy(), t, data.cpu().numpy(), workload=a.workload, mask=mask)
        default_cmd = a.workload == "lv" and a.net == "s1" and a.sensealg == "adjoint" and not (a.traj or a.lanes or a.waves or a.graph) and a.alg == "tsit5"
        if world == 1 and default_cmd and not a.no_others:
            # every other workload in the same, driver-timed process (outside the headline's timed region; headline fields untouched)
            del ens
            torch.cuda.empty_cache()
            others = {}
            for name in OTHER_WORKLOADS:
                try:
                    others[name] = quick_measure(name, device)
                    others[name]["kernel_within_step"] = kernel_within_step(others[name])
                except Exception as e:  # a failing secondary workload must not take the headline line with it
                    others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.empty_cache()
            out["config"]["other_workloads"] = others
        print(json.dumps(out))
    if comm is not None and getattr(comm, "mp", False):
        # a peer that arrived late ends a call in NaN + a counted timeout, and the communicator refuses every later call: a run
        # with a timeout is not a measurement
        nt = comm.p2p_timeouts()
        if nt:
            raise SystemExit("bench.py: rank %d: %d cross-process all-reduce call(s) timed out -- gradients were NaN" % (rank, nt))
    if dist is not None:
        dist.barrier()
        if comm is not None:
            comm.close(dist)
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
