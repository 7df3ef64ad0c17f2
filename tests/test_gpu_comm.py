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
