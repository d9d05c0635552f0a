"""TEST DOUBLE: an Engine look-alike computed by the oracle (CPU), used only to exercise the host-side
orchestration (pipeline.Compressor, sharding.HostExchange) in multi-process gloo tests on machines without
a GPU.  The product never constructs this."""
import numpy as np

from emul import emul as E
from oracle import oracle as O

LOW, HIGH = 0, 1


class FakeEngine:
    def __init__(self):
        self.seq = {}
        self.mv_raw, self.mv_filt, self.types, self.halo = {}, {}, {}, {}

    def seq_config(self, W, H, keep_prediction=False):
        self.W, self.H, self.F = W, H, O.frame_bytes(W, H)

    def set_mctf_options(self, block_overlaping=0, border_size=0, subpixel_accuracy=0):
        assert (block_overlaping, border_size, subpixel_accuracy) == (0, 0, 0)

    def put_frames(self, kind, t, frames, first=0, total=None):
        self.seq[(kind, t)] = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, self.F).copy()

    def get_frames(self, kind, t, first, count):
        return self.seq[(kind, t)][first:first + count].copy()

    def get_motion(self, kind, t, nB, bs):
        return (self.mv_raw if kind == 0 else self.mv_filt)[t][:nB]

    def get_frame_types(self, t, nB):
        return self.types[t][:nB]

    def motion_estimate(self, t, P, bs, S):
        self.mv_raw[t] = O.motion_estimate(self.seq[(LOW, t - 1)][:P], self.W, self.H, bs, S)

    def decorrelate(self, t, P, bs, S, always_B=0):
        d = O.decorrelate(self.seq[(LOW, t - 1)][:P], self.W, self.H, bs, S, self.mv_raw[t], always_B)
        self.seq[(HIGH, t)], self.mv_filt[t], self.types[t] = d["high"], d["mv"], d["types"]
        self.halo[t] = [None, None]

    def halo_bytes(self, bs):
        return self.F + 4 * (self.H // bs) * (self.W // bs) * 2 + 16

    def halo_export(self, t, side, bs):
        i = 0 if side == 0 else len(self.types[t]) - 1
        tb = np.zeros(16, dtype=np.uint8)
        tb[0] = self.types[t][i]
        return np.concatenate([self.seq[(HIGH, t)][i], self.mv_filt[t][i].reshape(-1).view(np.uint8), tb])

    def halo_import(self, t, side, rec):
        self.halo[t][side] = None if rec is None else np.ascontiguousarray(rec, dtype=np.uint8).copy()

    # the batch entry points of hip.Engine (records of several engines back to back)
    @staticmethod
    def halo_batch_export(engines, t, side):
        bs = engines[0].H // engines[0].mv_filt[t].shape[2]
        return np.concatenate([e.halo_export(t, side, bs) for e in engines])

    @staticmethod
    def halo_batch_import(engines, t, side, data):
        if data is None:
            for e in engines:
                e.halo_import(t, side, None)
            return
        parts = np.split(np.ascontiguousarray(data, dtype=np.uint8), len(engines))
        for e, rec in zip(engines, parts):
            e.halo_import(t, side, rec)

    def update(self, t, P, bs, uf=0.25):
        def h(side):
            rec = self.halo[t][side]
            if rec is None:
                return None
            mvb = 4 * (self.H // bs) * (self.W // bs) * 2
            mv = rec[self.F:self.F + mvb].view(np.int16)
            return (rec, int(rec[self.F + mvb]), int(np.abs(mv).max()) if mv.size else 0)
        r, low = E.update(self.seq[(LOW, t - 1)][:P], self.W, self.H, bs, self.seq[(HIGH, t)], self.mv_filt[t], self.types[t], uf,
                          halo_l=h(0), halo_r=h(1))
        assert r == 0
        self.seq[(LOW, t)] = low
