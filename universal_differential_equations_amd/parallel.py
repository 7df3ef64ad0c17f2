"""Multi-GPU: trajectories are independent units sharing theta, so an ensemble shards by contiguous
blocks of trajectories and the only exchange per gradient is ONE all-reduce(sum) of double[np + 4] =
[grad(np); loss; sum nf; sum naccept; sum nreject].  The forward-only path has no collective.

Two transports behind the same payload:
  * `Comm` -- libudecore's own RCCL binding / one-shot P2P reducer (include/udecore.h ude_comm_*, ude_allreduce_grad*):
    what a non-Python host (the Julia shim) uses; `Comm.from_torch_dist` bootstraps the RCCL unique id over an existing
    torch.distributed group, `Comm.local` builds the communicators of all devices of one process.
  * torch.distributed (backend "nccl" = RCCL; "gloo" for the CPU tests) through `allreduce_payload`.
"""
import ctypes as C

import numpy as np

from . import _lib


def shard_bounds(n_total, world, rank):
    """contiguous block partition: first (n_total % world) ranks get one extra trajectory"""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_payload(grad_loss, stats):
    """[grad(np); loss] (np + 1 doubles) and the per-trajectory stats (N, 8) -> double[np + 4] =
    [grad; loss; sum nf (fwd + bwd); sum naccept; sum nreject]: ONE buffer, ONE collective per gradient.
    Works on torch tensors (device-resident) and numpy arrays alike."""
    s = stats.sum(0)
    if hasattr(s, "to"):      # torch
        import torch
        # the payload is ALWAYS float64 (a Float32 problem's counters exceed 2^24 on a real ensemble)
        c = torch.stack([s[0] + s[4], s[1] + s[5], s[2] + s[6]]).to(torch.float64)
        return torch.cat([grad_loss.to(torch.float64), c])
    return np.concatenate([np.asarray(grad_loss, dtype=np.float64), np.array([s[0] + s[4], s[1] + s[5], s[2] + s[6]], dtype=np.float64)])


def allreduce_payload(buf, dist=None):
    """buf: tensor double[np + 4] on this rank's device (or CPU for gloo).  In place; sum over ranks."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


allreduce_grad = allreduce_payload   # (name used by bench.py / examples)


class Comm:
    """ude_comm handles of this process: one per device it drives"""

    def __init__(self, engines, handles):
        self.engines, self.handles = engines, handles
        self.L = _lib.load()

    @classmethod
    def from_torch_dist(cls, engine, dist):
        """one process per GPU: rank 0 creates the RCCL unique id, the existing process group carries it to the others"""
        L = _lib.load()
        ident = (C.c_char * 128)()
        if dist.get_rank() == 0:
            rc = L.ude_comm_unique_id(ident)
            if rc:
                raise _lib.UdeError(rc, "ude_comm_unique_id failed (librccl not loadable?)")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        h = C.c_void_p()
        engine.check(L.ude_comm_create(engine.h, dist.get_world_size(), dist.get_rank(), box[0], C.byref(h)))
        return cls([engine], [h])

    @classmethod
    def p2p_from_torch_dist(cls, engine, dist, n_max):
        """one process per GPU, NO RCCL: every rank exports the IPC handle of its exchange window, the existing process group
        all-gathers the 64-byte handles, every rank opens its peers' windows (ude_comm_create_p2p / ude_comm_p2p_connect)"""
        L = _lib.load()
        handle = (C.c_char * 64)()
        h = C.c_void_p()
        engine.check(L.ude_comm_create_p2p(engine.h, dist.get_world_size(), dist.get_rank(), int(n_max), handle, C.byref(h)))
        box = [None] * dist.get_world_size()
        dist.all_gather_object(box, bytes(handle))
        engine.check(L.ude_comm_p2p_connect(h, b"".join(box)))
        dist.barrier()                       # every rank has opened every window before the first call publishes into one
        comm = cls([engine], [h])
        comm.mp = True
        return comm

    def allreduce_mp(self, buf):
        """the cross-process one-shot reducer (p2p_from_torch_dist): in place, on the context's stream.  `buf`: ONE contiguous
        float64 device tensor (the C side reads numel() doubles from data_ptr(): any other dtype or a strided view is out of bounds)"""
        assert buf.dtype.is_floating_point and buf.element_size() == 8 and buf.is_contiguous() and buf.is_cuda, \
            "allreduce_mp needs a contiguous float64 CUDA tensor"
        self.engines[0].check(self.L.ude_allreduce_grad_p2p_mp(self.handles[0], C.c_void_p(buf.data_ptr()), buf.numel()))
        return buf

    def p2p_timeouts(self):
        n = C.c_int32(0)
        self.engines[0].check(self.L.ude_comm_p2p_status(self.handles[0], C.byref(n)))
        return n.value

    @classmethod
    def local(cls, engines):
        """all devices of one process (what a single-threaded Julia host does)"""
        L = _lib.load()
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e.h for e in engines])
        out = (C.c_void_p * n)()
        engines[0].check(L.ude_comm_create_local(n, ctxs, out))
        return cls(list(engines), [C.c_void_p(out[i]) for i in range(n)])

    def allreduce(self, bufs, p2p=False):
        """bufs: one torch float64 CUDA tensor per device of this Comm (same length).  In place, on the contexts' streams."""
        n = len(self.handles)
        if not isinstance(bufs, (list, tuple)):
            bufs = [bufs]
        assert len(bufs) == n and all(b.dtype.is_floating_point and b.element_size() == 8 for b in bufs)
        cnt = bufs[0].numel()
        if n == 1 and not p2p:
            self.engines[0].check(self.L.ude_allreduce_grad(self.handles[0], C.c_void_p(bufs[0].data_ptr()), cnt))
            return bufs
        comms = (C.c_void_p * n)(*[h.value for h in self.handles])
        ptrs = (C.c_void_p * n)(*[b.data_ptr() for b in bufs])
        fn = self.L.ude_allreduce_grad_p2p if p2p else self.L.ude_allreduce_grad_local
        self.engines[0].check(fn(n, comms, ptrs, cnt))
        return bufs

    def close(self, dist=None):
        """Teardown.  A cross-process P2P communicator is torn down in two steps with the host group's barrier between them:
        every rank disconnects (device handshake: no peer is still inside a call on this rank's window; the peers' windows are
        unmapped), barrier, then the windows are freed -- every importer has unmapped before any exporter frees."""
        if getattr(self, "mp", False):
            for h in self.handles:
                self.L.ude_comm_p2p_disconnect(h)    # (a peer that died: reported, the local teardown goes on)
            if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
                dist.barrier()
        for h in self.handles:
            self.L.ude_comm_destroy(h)
        self.handles = []
