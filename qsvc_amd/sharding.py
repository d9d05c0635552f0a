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
