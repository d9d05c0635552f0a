"""Decoder timing (GPU box): python tools/time_decode.py -- JPEG 2000 decode of the 51 subband planes of one config-3 sequence:
wall clock per call, the kernel groups' HIP-event times, and the host parser alone (max_layers irrelevant: lossless, one layer)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
W, H = 1920, 1080
e = hip.Engine(0)
comp = Compressor(e, CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=8, block_size_min=8, SRLs=5, lossless=True))
comp.configure(); comp.load(make_clip("pan_noise", comp.pictures, W, H, 43))
comp.analyze()
out = comp.texture()
streams = [cs for k in out for cs in out[k]]
d = hip.Engine(0)
d.seq_config(W, H)
d.j2k_decode_planes(streams)
d.set_profiling(True)
N = 3
t0 = time.perf_counter()
for _ in range(N):
    planes = d.j2k_decode_planes(streams)
dt = (time.perf_counter() - t0) / N
print(f"decode of {len(streams)} planes: {dt * 1e3:.1f} ms per call")
for k, (ms, n) in d.timings().items():
    print(f"  {k:20s} {ms / N:8.3f} ms")
