"""SURVEY.md 8(e) behind the C ABI: ude_comm_* / ude_allreduce_grad* (RCCL bound inside libudecore, one-shot P2P reducer).
A 1-GPU box exercises the real RCCL calls with a single-rank communicator; the 2-device tests run wherever a second
MI355X is visible (the driver's multi-GPU node) and are skipped otherwise."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import universal_differential_equations_amd as U
from universal_differential_equations_amd import models
from universal_differential_equations_amd.parallel import Comm, pack_payload, shard_bounds

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_rccl_communicator_from_unique_id():
    eng = U.Engine.get(0)
    ident = (C.c_char * 128)()
    assert eng.L.ude_comm_unique_id(ident) == 0 and any(bytes(ident))
    h = C.c_void_p()
    eng.check(eng.L.ude_comm_create(eng.h, 1, 0, bytes(ident), C.byref(h)))
    buf = torch.arange(91, dtype=torch.float64, device="cuda:0") * 0.25 - 3.0
    ref = buf.clone()
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.check(eng.L.ude_allreduce_grad(h, C.c_void_p(buf.data_ptr()), buf.numel()))
    torch.cuda.synchronize()
    assert torch.equal(buf, ref)          # sum over one rank
    eng.L.ude_comm_destroy(h)


def test_local_communicator_rccl_and_p2p_one_device():
    eng = U.Engine.get(0)
    comm = Comm.local([eng])
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for p2p in (False, True):
        buf = torch.linspace(-1, 1, 4485, dtype=torch.float64, device="cuda:0")
        ref = buf.clone()
        comm.allreduce([buf], p2p=p2p)
        torch.cuda.synchronize()
        assert torch.equal(buf, ref)
    comm.close()


def _lv_inputs(n):
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "Scenario_1_recovery_0.005.json")))
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    rng = np.random.default_rng(1234)
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (n, 2)))
    return th, u0, t, np.repeat(X[None], n, axis=0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X in one node")
def test_two_gpu_gradient_equals_one_gpu_and_rccl_equals_p2p():
    n = 1000
    th, u0, t, data = _lv_inputs(n)

    def ens_on(dev, lo, hi):
        d = torch.device("cuda", dev)
        with torch.cuda.device(d):
            e = U.DeviceEnsemble(models.ude_dynamics(), U.Tsit5(), (t[0], t[-1]), t, torch.tensor(u0[lo:hi], device=d),
                                 data=torch.tensor(data[lo:hi], device=d), abstol=1e-6, reltol=1e-6)
            g = e.loss_grad(torch.tensor(th, device=d))
            return e, pack_payload(g, e.stats)

    e_all, full = ens_on(0, 0, n)
    parts = [ens_on(r, *shard_bounds(n, 2, r)) for r in range(2)]
    comm = Comm.local([p[0].eng for p in parts])
    a = [p[1].clone() for p in parts]
    b = [p[1].clone() for p in parts]
    comm.allreduce(a, p2p=False)
    comm.allreduce(b, p2p=True)
    for d in range(2):
        torch.cuda.synchronize(d)
    assert torch.equal(a[0].cpu(), a[1].cpu()) and torch.equal(b[0].cpu(), b[1].cpu())      # every rank holds the same bits
    assert torch.equal(a[0].cpu(), b[0].cpu())                                                # RCCL == fixed-order P2P (two addends)
    ref = full.cpu().numpy()
    got = b[0].cpu().numpy()
    assert np.abs(got[:-3] - ref[:-3]).max() <= 1e-12 * np.abs(ref[:-3]).max()
    assert np.array_equal(got[-3:], ref[-3:])                                                  # counters: exact
    comm.close()


P2P_MP_CHILD = r"""
import os, sys, ctypes as C
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import universal_differential_equations_amd as U
from universal_differential_equations_amd.parallel import Comm
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["UDE_TEST_PORT"], rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
eng = U.Engine.get(0)                                   # both ranks on device 0: the IPC window is opened across PROCESSES
eng.set_stream(torch.cuda.current_stream().cuda_stream)
comm = Comm.p2p_from_torch_dist(eng, dist, 4485)
ok = True
for call in range(40):                                  # repeated calls: both slot parities are reused many times
    n = [91, 4485, 1, 1024][call %% 4]
    rng = np.random.default_rng(1000 * call)
    parts = [rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8) for _ in range(world)]
    want = parts[0].copy()
    for r in range(1, world):
        want = want + parts[r]                          # rank order, left to right: the reducer's association
    buf = torch.tensor(parts[rank], dtype=torch.float64, device="cuda:0")
    comm.allreduce_mp(buf)
    torch.cuda.synchronize()
    ok = ok and np.array_equal(buf.cpu().numpy(), want)
assert comm.p2p_timeouts() == 0
# a peer that never arrives: rank 1 skips a call; rank 0's result is NaN after the timeout, the GPU is not hung, the counter says so
if rank == 0:
    buf = torch.ones(8, dtype=torch.float64, device="cuda:0")
    comm.allreduce_mp(buf)
    torch.cuda.synchronize()
    lost = bool(torch.isnan(buf).all().item()) and comm.p2p_timeouts() == 1
else:
    lost = True
dist.barrier()
print("RESULT rank %%d ok %%s lost %%s" %% (rank, ok, lost), flush=True)
dist.destroy_process_group()
os._exit(0)                                              # (the windows stay mapped in the peer: skip the destructors' ordering)
"""


def test_cross_process_p2p_reducer_two_ranks_on_one_gpu(tmp_path):
    """round 4: `ude_allreduce_grad_p2p` with ONE PROCESS PER GPU (what `bench.py --gpus N` launches): IPC windows, one kernel per
    rank and call, sums in rank order -- rehearsed with two processes on one device (the IPC path is the same; the peer reads then
    stay inside one HBM).  40 calls of four payload sizes, bit-identical to the left-to-right sum on both ranks; a missing peer ends
    in NaN + a counted timeout, not in a hung GPU."""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", UDE_TEST_PORT=str(port), UDE_P2P_TIMEOUT_MS="1500", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", P2P_MP_CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (rank, se[-2000:])
        assert "RESULT rank %d ok True lost True" % rank in so, (so[-500:], se[-1500:])
