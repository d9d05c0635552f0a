"""GOP-per-GPU sharding of an open-GOP sequence.

Rank r of N owns pictures [r*m*G, (r+1)*m*G] inclusive (m GOPs of G pictures; both boundary pictures are
resident on both neighbours).  Per temporal level everything is local except the update of the two boundary
even pictures, which needs the neighbour's boundary B picture (its high_t picture, filtered motion field and
frame type) -- update.cpp:605-614 applies the NEXT update of the left B picture before the PREV update of
the right one.  Both neighbours then compute the identical boundary picture redundantly; there is no other
data-path communication (the JPEG 2000 stage is embarrassingly parallel).

Two transports for the halo records:
  RcclExchange ...... device-to-device over xGMI through the C ABI (qsvc_comm_init / qsvc_halo_exchange)
  HostExchange ...... host-staged over any torch.distributed backend (gloo on CPU-only test machines)
A rank that keeps several sequences in flight on its GPU (one engine each) sends their records as ONE message per neighbour:
exchange_batch(engines, t, block_size) -- one communicator per rank (held by engines[0]) however many sequences it codes at once.
"""
from __future__ import annotations

import numpy as np


def shard_range(rank: int, world: int, gops_per_rank: int, gop_size: int):
    """(first, last) picture indices (inclusive) of rank's shard."""
    first = rank * gops_per_rank * gop_size
    return first, first + gops_per_rank * gop_size


class RcclExchange:
    def __init__(self, engine, rank, world, uid: bytes):
        engine.comm_init(rank, world, uid)

    def exchange(self, engine, t, block_size):
        engine.halo_exchange(t)

    def exchange_batch(self, engines, t, block_size):
        type(engines[0]).halo_exchange_batch(engines, t)


class HostExchange:
    """Host-staged halo exchange over torch.distributed point-to-point ops (line topology, no wrap)."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def exchange(self, engine, t, block_size):
        import torch
        n = engine.halo_bytes(block_size)
        ops, rx = [], {}
        for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):
            if peer < 0 or peer >= self.world:
                engine.halo_import(t, side, None)
                continue
            tx = torch.from_numpy(np.ascontiguousarray(engine.halo_export(t, side, block_size)))
            rx[side] = torch.empty(n, dtype=torch.uint8)
            ops.append(self.dist.P2POp(self.dist.isend, tx, peer))
            ops.append(self.dist.P2POp(self.dist.irecv, rx[side], peer))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for side, buf in rx.items():
            engine.halo_import(t, side, buf.numpy())

    def exchange_batch(self, engines, t, block_size):
        """The records of all the rank's engines as one message per neighbour (the layout the RCCL batch sends)."""
        import torch
        E = type(engines[0])
        ops, rx = [], {}
        for side, peer in ((0, self.rank - 1), (1, self.rank + 1)):
            if peer < 0 or peer >= self.world:
                E.halo_batch_import(engines, t, side, None)
                continue
            tx = torch.from_numpy(np.ascontiguousarray(E.halo_batch_export(engines, t, side)))
            rx[side] = torch.empty(tx.numel(), dtype=torch.uint8)
            ops.append(self.dist.P2POp(self.dist.isend, tx, peer))
            ops.append(self.dist.P2POp(self.dist.irecv, rx[side], peer))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for side, buf in rx.items():
            E.halo_batch_import(engines, t, side, buf.numpy())


class BatchExchange:
    """The sequences one rank keeps in flight (one engine and host thread each) reach a level's exchange point together; the LAST
    one to arrive runs ONE batched exchange for all of them (transport.exchange_batch: one message per neighbour), so every rank
    issues exactly one exchange per level whatever its number of sequences, in the same order on every rank.  A sequence that fails
    breaks the barrier for the others instead of leaving them waiting."""

    def __init__(self, transport, engines, timeout=300):
        import threading
        self.transport, self.engines = transport, list(engines)
        self.args = [None] * len(self.engines)
        self.error = None
        self.bar = threading.Barrier(len(self.engines), action=self._run, timeout=timeout)

    def _run(self):
        try:
            ts = {a[0] for a in self.args}
            bss = {a[1] for a in self.args}
            if len(ts) != 1 or len(bss) != 1:
                raise RuntimeError(f"sequences of one rank reached different exchange points: levels {ts}, block sizes {bss}")
            self.transport.exchange_batch(self.engines, ts.pop(), bss.pop())
        except Exception as ex:  # noqa: BLE001
            self.error = ex

    def abort(self):
        self.bar.abort()

    def member(self, k):
        outer = self

        class Member:
            def exchange(self, engine, t, block_size):
                outer.args[k] = (t, block_size)
                outer.bar.wait()
                if outer.error is not None:
                    raise outer.error
        return Member()
