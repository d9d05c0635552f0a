"""GPU box: where the motion half of the decoder chain spends its time (motion streams -> planes -> the two correlators)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip, motion as motion_chain
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
W, H, TRLS = 1920, 1080, 4
e = hip.Engine(0)
comp = Compressor(e, CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=TRLS, GOPs=2, block_size=8, block_size_min=8, SRLs=5, lossless=True))
comp.configure(); comp.load(make_clip("pan_noise", comp.pictures, W, H, 43))
comp.analyze()
_, mcs = comp.motion()
mcs = {t: [bytes(c) for c in v] for t, v in mcs.items()}

streams = [cs for (t, Pt, St, bst) in comp.levels for cs in mcs[t]]
print(len(streams), sum(len(s) for s in streams))
d = hip.Engine(0); d.seq_config(W, H)
d.j2k_decode_planes(streams)
d.set_profiling(True)
for rep in range(3):
    t0 = time.perf_counter()
    mplanes = d.j2k_decode_planes(streams)
    t1 = time.perf_counter()
    res, k = {}, 0
    for (t, Pt, St, bst) in comp.levels:
        nf, BY, BX = Pt // 2, H // bst, W // bst
        comps4 = np.stack(mplanes[k:k + 4 * nf]).reshape(nf, 4, BY, BX).transpose(1, 0, 2, 3)
        res[t] = motion_chain.mux_components(np.ascontiguousarray(comps4))
        k += 4 * nf
    t2 = time.perf_counter()
    fields = motion_chain.field_chain(res, TRLS)
    t3 = time.perf_counter()
    print("decode %.2f ms, mux %.2f ms, chain %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3))
for k, (ms, n) in d.timings().items():
    print(f"  {k:20s} {ms / 3:8.3f} ms")
