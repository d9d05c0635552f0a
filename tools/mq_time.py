#!/usr/bin/env python3
"""Tier-1 kernel times of ONE config-3 sequence for the library named by QSVC_HIP_LIB (A/B of kernel variants):
python3 tools/mq_time.py [steps]  ->  j2k_t1_cm / j2k_t1_mq ms per step (HIP events), ms per whole encode, SHA-256 of the streams."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qsvc_amd import hip
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W, H = 1920, 1080
os.environ.setdefault("QSVC_T2_THREADS", "8")
prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=8, block_size_min=8, search_range=4, SRLs=5, lossless=True)
e = hip.Engine(0)
c = Compressor(e, prm, None)
c.configure()
c.load(make_clip(os.environ.get("CLIP", "pan_noise"), c.pictures, W, H, 43))
def step():
    c.analyze()
    return c.texture(0, copy=False)
out = step(); step()
h = hashlib.sha256()
for k in sorted(out):
    for s in out[k]:
        h.update(bytes(s))
e.sync(); e.set_profiling(True)
t0 = time.perf_counter()
for _ in range(steps):
    step()
e.sync()
ms = (time.perf_counter() - t0) / steps * 1e3
tm = e.timings()
print(os.environ.get("QSVC_HIP_LIB", "default"), "lanes", os.environ.get("QSVC_MQ_LANES", "64"),
      " ".join(f"{k} {tm[k][0] / steps:.3f}" for k in ("j2k_t1_cm", "j2k_t1_mq") if k in tm), f"encode {ms:.3f} ms", h.hexdigest()[:16], flush=True)
if os.environ.get("MQ_TIME_ALL"):
    print("  " + " ".join(f"{k}={v[0] / steps:.3f}" for k, v in tm.items()), "sum=%.3f" % (sum(v[0] for v in tm.values()) / steps), flush=True)
