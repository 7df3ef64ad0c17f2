"""N>1 path on CPU: trajectories shard by contiguous blocks, ONE all-reduce(sum) of double[np + 4] =
[grad; loss; sum nf; sum naccept; sum nreject] per gradient.  world_size 2, gloo backend; the
per-rank compute is the CPU oracle (tests only: no GPU here; the device path of the same payload is tests/test_gpu_comm.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(n_total):
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "Scenario_1_recovery_0.005.json")))
    X = np.array(g["X"]["data_colmajor"]).reshape(31, 2)
    t = np.array(g["solution"]["t"])
    th = np.array(g["initial_parameters"])
    rng = np.random.default_rng(1234)
    u0 = np.array([0.44249296, 4.6280594]) * (1 + 0.2 * rng.uniform(-1, 1, (n_total, 2)))
    data = np.repeat(X[None], n_total, axis=0)
    return th, u0, t, data


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import _oracle as O
    from universal_differential_equations_amd.parallel import allreduce_payload, pack_payload, shard_bounds

    dist.init_process_group("gloo", rank=rank, world_size=world)
    th, u0, t, data = _inputs(n_total)
    lo, hi = shard_bounds(n_total, world, rank)
    r = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0[lo:hi], [t[0], t[-1]], th, t, data[lo:hi])
    buf = torch.tensor(pack_payload(np.concatenate([r["grad_theta"], [r["loss"]]]), r["stats"]))
    allreduce_payload(buf, dist)
    if rank == 0:
        q.put(buf.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_partition():
    from universal_differential_equations_amd.parallel import shard_bounds
    for n, w in ((10, 2), (10, 3), (50000, 8), (7, 8), (10000, 1)):
        b = [shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(50000, 8, 3) == (18750, 25000)      # BASELINE C3: 6250 per GPU


@pytest.mark.timeout(300)
def test_two_rank_gradient_equals_single_process():
    import torch.multiprocessing as mp

    import _oracle as O
    n_total, world = 24, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    th, u0, t, data = _inputs(n_total)
    ref = O.loss_grad_ensemble(O.lv_ude_s1(), O.opts(O.TSIT5, 1e-6, 1e-6), u0, [t[0], t[-1]], th, t, data)
    full = np.concatenate([ref["grad_theta"], [ref["loss"]]])
    assert got.shape == (full.size + 3,)
    assert np.abs(got[:-3] - full).max() <= 1e-12 * np.abs(full).max()
    st = ref["stats"].sum(0)
    assert got[-3:].tolist() == [st[0] + st[4], st[1] + st[5], st[2] + st[6]]      # counters ride in the same payload, exactly
