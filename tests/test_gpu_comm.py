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


def test_pack_counters_dev_matches_pack_payload():
    """ude_pack_counters_dev (the Julia shim's way to the double[np + 4] payload) == parallel.pack_payload's torch reduction"""
    eng = U.Engine.get(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for N in (1, 63, 6250):
        stats = torch.randint(0, 100000, (N, 8), dtype=torch.int64, device="cuda:0")
        grad_loss = torch.linspace(-2, 2, 88, dtype=torch.float64, device="cuda:0")
        want = pack_payload(grad_loss, stats)
        got = torch.cat([grad_loss, torch.full((3,), -1.0, dtype=torch.float64, device="cuda:0")])
        eng.check(eng.L.ude_pack_counters_dev(eng.h, N, C.c_void_p(stats.data_ptr()), C.c_void_p(got.data_ptr()), 87))
        torch.cuda.synchronize()
        assert torch.equal(got, want)


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
# a float32 / strided buffer is refused on the Python side (the C side would read numel() doubles from it)
for bad in (torch.ones(8, dtype=torch.float32, device="cuda:0"), torch.ones(16, dtype=torch.float64, device="cuda:0")[::2]):
    try:
        comm.allreduce_mp(bad)
        ok = False
    except AssertionError:
        pass
# a peer that arrives too late: rank 1 sits out rank 0's call 41 until rank 0 has given up.
#   rank 0: NaN after the timeout, the GPU is not hung, the counter says so; the timeout is STICKY -- the next call is NaN at once
#           (nothing published: slot reuse is no longer safe), and once the host has read the count the call itself fails (-6)
#   rank 1: its own call 41 still finds rank 0's payload OF CALL 41 (rank 0 never overwrote it) and sums it correctly; its call 42
#           fails FAST (rank 0 has given up and says so in its window) instead of summing a slot of another call
import time
from universal_differential_equations_amd import _lib
if rank == 0:
    buf = torch.ones(8, dtype=torch.float64, device="cuda:0")
    comm.allreduce_mp(buf)
    torch.cuda.synchronize()
    lost = bool(torch.isnan(buf).all().item())
    buf2 = torch.ones(8, dtype=torch.float64, device="cuda:0")
    t0 = time.perf_counter()
    comm.allreduce_mp(buf2)                               # host does not know yet: the kernel refuses by itself
    torch.cuda.synchronize()
    lost = lost and bool(torch.isnan(buf2).all().item()) and time.perf_counter() - t0 < 1.0
    lost = lost and comm.p2p_timeouts() == 1
    try:
        comm.allreduce_mp(buf2)
        lost = False
    except _lib.UdeError as e:
        lost = lost and e.code == _lib.UDE_ERR_TIMEOUT
    dist.barrier()
    dist.barrier()
else:
    dist.barrier()                                        # rank 0 has timed out
    buf = torch.full((8,), 2.5, dtype=torch.float64, device="cuda:0")
    comm.allreduce_mp(buf)                                # call 41 of this rank: rank 0's call-41 payload is still in its slot
    torch.cuda.synchronize()
    lost = bool((buf == 3.5).all().item())
    t0 = time.perf_counter()
    comm.allreduce_mp(buf)                                # call 42: rank 0 will never publish it
    torch.cuda.synchronize()
    lost = lost and bool(torch.isnan(buf).all().item()) and time.perf_counter() - t0 < 1.0 and comm.p2p_timeouts() == 1
    dist.barrier()
# REAL teardown (no os._exit): disconnect handshake on both ranks, the host group's barrier, then the windows are freed
comm.close(dist)
torch.cuda.synchronize()
print("RESULT rank %%d ok %%s lost %%s" %% (rank, ok, lost), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


P2P_TEARDOWN_CHILD = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import universal_differential_equations_amd as U
from universal_differential_equations_amd.parallel import Comm
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["UDE_TEST_PORT"], rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
eng = U.Engine.get(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
ok = True
for rnd in range(3):                                    # create / use / destroy, three times: windows are really freed and re-made
    comm = Comm.p2p_from_torch_dist(eng, dist, 128)
    for call in range(5):
        buf = torch.full((100,), float(rank + 1 + call), dtype=torch.float64, device="cuda:0")
        comm.allreduce_mp(buf)
    # NO synchronize: rank 1 goes straight into the teardown while rank 0 may still be inside its last call -- the handshake of
    # ude_comm_destroy (no host barrier here: close() without the group) must keep rank 1's window alive until rank 0 is through
    if rank == 0:
        import time; time.sleep(0.2)
    comm.close()
    torch.cuda.synchronize()
    want = sum(r + 1 + 4 for r in range(world))
    ok = ok and bool((buf == want).all().item())
print("RESULT rank %%d ok %%s" %% (rank, ok), flush=True)
dist.barrier()
dist.destroy_process_group()
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


def test_cross_process_p2p_teardown_without_host_barrier():
    """ude_comm_destroy alone (no host barrier, ranks skewed, a call possibly still in flight on the peer): the device-side "closed"
    handshake keeps every window alive until no peer reads it; three create / use / destroy rounds in two real processes that exit
    through their normal destructors"""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", UDE_TEST_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", P2P_TEARDOWN_CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (rank, se[-2000:])
        assert "RESULT rank %d ok True" % rank in so, (so[-500:], se[-1500:])


P2P_EIGHT_CHILD = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import universal_differential_equations_amd as U
from universal_differential_equations_amd.parallel import Comm
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["UDE_TEST_PORT"], rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
eng = U.Engine.get(0)                                   # all ranks on device 0: eight IPC windows, eight flag / closed words per window
eng.set_stream(torch.cuda.current_stream().cuda_stream)
comm = Comm.p2p_from_torch_dist(eng, dist, 9291)
ok = True
for call in range(24):                                  # the four payloads of the workloads: np + 4 for LV, Fisher-KPP, SEIR, the neural ODE
    n = [91, 470, 4485, 9291][call %% 4]
    rng = np.random.default_rng(7000 + call)
    parts = [rng.standard_normal(n) * 10.0 ** rng.integers(-8, 8) for _ in range(world)]
    want = parts[0].copy()
    for r in range(1, world):
        want = want + parts[r]                          # rank order, left to right: the reducer's association
    buf = torch.tensor(parts[rank], dtype=torch.float64, device="cuda:0")
    comm.allreduce_mp(buf)
    torch.cuda.synchronize()
    ok = ok and np.array_equal(buf.cpu().numpy(), want)
nt = comm.p2p_timeouts()
comm.close(dist)                                        # the device-side "closed" handshake with seven peers + the host barrier
torch.cuda.synchronize()
print("RESULT rank %%d ok %%s timeouts %%d" %% (rank, ok, nt), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_cross_process_p2p_reducer_eight_ranks_on_one_gpu():
    """round 6: the cross-process reducer with the EIGHT windows / flag words a full node needs (two ranks were all that had ever run):
    eight processes on one device, 24 calls of the four payload sizes, every rank's result bit-identical to the left-to-right sum in
    rank order, no timeout, real teardown"""
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", UDE_TEST_PORT=str(port), UDE_P2P_TIMEOUT_MS="20000", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", P2P_EIGHT_CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (rank, se[-2000:])
        assert "RESULT rank %d ok True timeouts 0" % rank in so, (so[-500:], se[-1500:])
