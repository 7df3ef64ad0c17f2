"""Rehearsal of the driver's multi-GPU bench launch on a one-GPU box: `python -m torch.distributed.run --nproc-per-node 2 bench.py
--gpus 2` with both ranks on device 0 (UDE_BENCH_DEVICE=0) and gloo in place of RCCL (which refuses two ranks on one device;
UDE_BENCH_BACKEND=gloo).  Everything else is the real N > 1 path: one process per rank, the ONE all-reduce of
double[np + 4] per gradient, barrier + synchronize around the timed region, MAX over ranks, rank 0 prints the JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("workload,traj,port,transport", [("lv", "2000", "29541", "torch"), ("seir", "300", "29545", "torch"),
                                                         ("lv", "2000", "29547", "p2p")])
def test_bench_two_ranks_on_one_gpu(workload, traj, port, transport):
    """transport "p2p": `bench.py --gpus 2 --allreduce p2p` -- the one all-reduce per gradient through libudecore's cross-process
    one-shot reducer (IPC windows, rank-ordered sum), the comparison with RCCL that SURVEY.md 8(e) asks for, from the very layout
    the driver launches (one process per GPU); here both ranks share device 0"""
    env = dict(os.environ, UDE_BENCH_DEVICE="0", UDE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--traj", traj,
           "--workload", workload, "--no-cpu-baseline", "--no-others", "--allreduce", transport]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["failed_trajectories"] == 0
    # whole-job value: both ranks' evaluations over the slower rank's time
    per_rank = d["config"]["evals_per_step_fwd"] + d["config"]["evals_per_step_bwd"]
    assert d["value"] > per_rank * 3 / (d["ms_per_step"] * 3e-3) * 1.5


@pytest.mark.parametrize("transport", ["torch", "p2p"])
def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks(transport):
    """round 5: `python bench.py --gpus 2` with NO launcher around it (WORLD_SIZE unset -- the shape of the driver's N = 1 command with
    another N): bench.py re-executes itself under torch.distributed.run with two ranks; the line says n_gpus = 2 = the size of the
    process group, and names the transport; under a launcher whose world differs from --gpus it refuses"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(UDE_BENCH_DEVICE="0", UDE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--traj", "2000",
           "--no-cpu-baseline", "--no-others", "--allreduce", transport]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["allreduce"] == transport and d["rccl_ranks"] == 0    # (gloo rehearsal: no RCCL communicator)
    assert d["config"]["failed_trajectories"] == 0
    bad = subprocess.run(cmd[:3] + ["4"] + cmd[4:], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert bad.returncode != 0 and "--gpus 4" in bad.stderr


@pytest.mark.parametrize("workload", ["hjb", "lv"])
def test_bench_collective_through_libudecore_single_rank(workload):
    """`bench.py --allreduce udecore`: the one all-reduce per gradient through libudecore's own RCCL binding (ude_comm_create from a
    unique id + ude_allreduce_grad on the context's stream) instead of torch.distributed -- what a non-Python host calls.  RCCL
    wants one device per rank, so on a one-GPU box this is a ONE-rank communicator (UDE_BENCH_FORCE_DIST=1): the binding, the
    payload packing and the mean over ranks run, the numbers must equal the plain single-process run."""
    env = dict(os.environ, UDE_BENCH_FORCE_DIST="1", UDE_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    extra = ["--workload", "hjb", "--traj", "2048"] if workload == "hjb" else ["--traj", "2000"]
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-others",
           "--allreduce", "udecore"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["failed_trajectories"] == 0


@pytest.mark.parametrize("workload,traj", [("lv", "600"), ("seir", "96")])
def test_bench_gpus_8_plain_command_through_the_cross_process_reducer(workload, traj):
    """round 6: what the driver will run on a full node -- `python bench.py --gpus 8` as a plain command -- rehearsed with all eight
    ranks on device 0 (gloo bootstrap, `--allreduce p2p`: eight IPC windows): the line says n_gpus = 8, nothing timed out, no
    trajectory failed; `--workload seir` is the configuration north_star shards (configs[2])"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(UDE_BENCH_DEVICE="0", UDE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", UDE_P2P_TIMEOUT_MS="20000")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--traj", traj, "--workload", workload,
           "--no-cpu-baseline", "--no-others", "--allreduce", "p2p"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]          # (a counted timeout makes bench.py exit non-zero)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["allreduce"] == "p2p" and d["scaling"] == "weak"
    assert d["config"]["failed_trajectories"] == 0
    per_rank = d["config"]["evals_per_step_fwd"] + d["config"]["evals_per_step_bwd"]
    assert d["value"] > per_rank * 4 / (d["ms_per_step"] * 1e-3)     # whole-job value: eight ranks' evaluations over the slowest rank's time
