"""Device-resident `mctf compress` pipeline: the orchestration of compress.py / analyze.py /
analyze_step.py / texture_compress.py / texture_compress_fb_j2k.py with every stage running on the GPU
through the C ABI and nothing but code-streams crossing PCIe.

The compute backend is an `Engine` (qsvc_amd/hip.py).  `exchange` (optional) implements the open-GOP
boundary exchange between neighbouring shards (qsvc_amd/sharding.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field

from . import schedule
from .hip import HIGH, LOW, j2k_params


@dataclass
class CompressParams:  # defaults of compress.py:57-106
    pixels_in_x: int = 352
    pixels_in_y: int = 288
    always_B: int = 0
    block_overlaping: int = 0      # decorrelate.cpp:84-172 (0, 1 or a power of two)
    border_size: int = 0           # motion_estimate.cpp:213-217
    block_size: int | None = None
    block_size_min: int | None = None
    GOPs: int = 1
    quantization_texture: int = 45000
    quantization_step: int = 0
    search_range: int = 4
    TRLs: int = 4
    SRLs: int = 5
    nLayers: int = 5
    update_factor: float = 0.25
    reversible: bool = False       # build extension: the reference hard-codes Creversible=no
    lossless: bool = False         # build extension: one layer, every coding pass (ignores the slopes)
    base_step: float = 1.0 / 256
    cblk: int = 64                 # code-block size (Kakadu default Cblk={64,64}); 32 is a valid, faster alternative

    def resolve(self):
        d = schedule.default_block_size(self.pixels_in_x, self.pixels_in_y)
        if self.block_size is None:
            self.block_size = d
        if self.block_size_min is None:
            self.block_size_min = d
        return self


class Compressor:
    def __init__(self, engine, params: CompressParams, exchange=None, motion_engine=None):
        self.e = engine
        # a second context of the same GPU for the motion-field coding: its (short, serial) Tier-1 chain then runs beside the texture
        # coding of the same sequence instead of in front of it (motion_async)
        self.me = motion_engine
        self._pool = None
        self.p = params.resolve()
        self.exchange = exchange
        self.levels = schedule.level_schedule(self.p.GOPs, self.p.TRLs, self.p.block_size, self.p.block_size_min, self.p.search_range)
        self.pictures = schedule.pictures(self.p.GOPs, self.p.TRLs)
        self.slopes = schedule.slope_table(self.p.TRLs, self.p.quantization_texture, self.p.nLayers, self.p.quantization_step)

    def configure(self, keep_prediction=False):
        self.e.seq_config(self.p.pixels_in_x, self.p.pixels_in_y, keep_prediction)
        self.e.set_mctf_options(self.p.block_overlaping, self.p.border_size, 0)

    def load(self, low0):
        self.e.put_frames(LOW, 0, low0)

    # analyze.py:121-153 -> analyze_step.py:117-232
    def analyze(self):
        for (t, P, S, bs) in self.levels:
            self.e.motion_estimate(t, P, bs, S)
            self.e.decorrelate(t, P, bs, S, self.p.always_B)
            if self.exchange is not None:
                self.exchange.exchange(self.e, t, bs)
            self.e.update(t, P, bs, self.p.update_factor)

    def j2k_params_for(self, subband_index):
        """subband_index into the slope table: 0 = L, k = H_{TRLs-k} (texture_compress.py:183-217)."""
        if self.p.lossless:
            return j2k_params(reversible=True, levels=max(self.p.SRLs - 1, 0), layers=1, cblk=self.p.cblk)
        return j2k_params(reversible=self.p.reversible, levels=max(self.p.SRLs - 1, 0), layers=self.p.nLayers,
                          slopes=self.slopes[subband_index], base_step=self.p.base_step, cblk=self.p.cblk)

    # texture_compress.py:183-217: high_t (pictures-1 each) then low_{TRLs-1}
    def texture_items(self, low_first=0):
        items = []
        for (t, P, S, bs) in self.levels:
            items.append((HIGH, t, 0, P // 2, self.j2k_params_for(self.p.TRLs - t)))
        T = self.p.TRLs - 1
        nlow = self.levels[-1][1] // 2 + 1 if self.levels else self.pictures
        items.append((LOW, T, low_first, nlow - low_first, self.j2k_params_for(0)))
        return items

    def texture(self, low_first=0, copy=True):
        """-> {(kind, t): [code-stream per (picture, component)]}; copy=False: views valid until the next call."""
        items = [it for it in self.texture_items(low_first) if it[3] > 0]
        streams = self.e.j2k_encode_multi(items, copy=copy)  # one batch: every subband picture of the sequence
        return {(it[0], it[1]): s for it, s in zip(items, streams)}

    # compress.py:197-212 -> motion_compress.py:146-186 -> motion_compress_j2k.py:98-142
    def motion(self, want_residues=False, copy=True):
        """The motion-field coding chain on the device-resident filtered fields, one native call (qsvc_motion_encode): inter-level /
        bidirectional decorrelation on the host side of the library (the fields are kilobytes), the reference's demux cut, and the
        lossless coding of every 16-bit component plane of every level in ONE batch on the GPU.
        -> ({t: motion_residue_t} or None, {t: [code-stream per (field, component)]})."""
        if not self.levels:
            return ({} if want_residues else None), {}
        if self.p.block_size > self.p.block_size_min and len(self.levels) > 1:
            raise NotImplementedError("motion_compress with a shrinking block size (motion_compress.py:159-166 quirk) is not reproduced")
        streams, res = self.e.motion_encode([(t, P // 2) for (t, P, S, bs) in self.levels], self.p.block_size, want_residues, copy)
        return res, streams

    def code(self, low_first=0, copy=True):
        """compress.py:197-228 behind the analysis, in ONE batch (qsvc_compress_code): the motion fields' component planes ride in the
        textures' Tier-1 launches.  -> ({(kind, t): texture streams}, {t: motion streams})."""
        items = [it for it in self.texture_items(low_first) if it[3] > 0]
        if self.p.block_size > self.p.block_size_min and len(self.levels) > 1:
            raise NotImplementedError("motion_compress with a shrinking block size (motion_compress.py:159-166 quirk) is not reproduced")
        tex, mot = self.e.compress_code(items, [(t, P // 2) for (t, P, S, bs) in self.levels], self.p.block_size, copy)
        return {(it[0], it[1]): s for it, s in zip(items, tex)}, mot

    def motion_async(self):
        """compress.py:197-212 codes the motion between the analysis and the textures; the two are independent.  With a motion engine:
        the filtered fields come to the host (kilobytes) and a worker thread codes them on the second context while the caller goes on
        to texture(); returns a future of motion()'s result.  Without one: motion() now, wrapped in a finished future."""
        from concurrent.futures import Future, ThreadPoolExecutor
        if self.me is None or not self.levels:
            f = Future()
            f.set_result(self.motion(copy=False))
            return f
        from .hip import MOTION_FILTERED
        lv = [(t, P // 2) for (t, P, S, bs) in self.levels]
        filt = {t: self.e.get_motion(MOTION_FILTERED, t, n, self.p.block_size) for (t, n) in lv}
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=1)
            self.me.seq_config(self.p.pixels_in_x, self.p.pixels_in_y)

        def work():
            streams, _ = self.me.motion_encode(lv, self.p.block_size, False, False, filtered=filt)
            return None, streams
        return self._pool.submit(work)

    def compress(self, with_motion=False):
        if self.p.TRLs > 1:
            self.analyze()
            if with_motion:
                self.motion()
        return self.texture()
