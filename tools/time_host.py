"""Host-side cost of the MCTF stage calls of one sequence (tools, GPU box): how long each C-ABI call takes to RETURN (its launches
queued, nothing awaited) and how long the GPU then needs.  python tools/time_host.py [cfg3|cfg4] [iters]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H, bs, P, TRLs = (1920, 1080, 8, 17, 4) if cfg == "cfg3" else (3840, 2160, 16, 33, 5)
clip = make_clip("pan_noise", P, W, H, 43)
e = hip.Engine(0)
e.seq_config(W, H)
e.put_frames(hip.LOW, 0, clip)
acc = {}
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
def run(measure):
    p, S = P, 4
    for t in range(1, TRLs):
        t0 = time.perf_counter(); e.motion_estimate(t, p, bs, S)
        if measure: tick("motion_estimate (return)", t0)
        t0 = time.perf_counter(); e.decorrelate(t, p, bs, S, 0)
        if measure: tick("decorrelate (return)", t0)
        t0 = time.perf_counter(); e.update(t, p, bs, 0.25)
        if measure: tick("update (return)", t0)
        p = (p + 1) // 2; S *= 2
    t0 = time.perf_counter(); e.sync()
    if measure: tick("sync (GPU drains, decisions resolved)", t0)
for _ in range(3): run(False)
t0 = time.perf_counter()
for _ in range(iters): run(True)
dt = (time.perf_counter() - t0) / iters
print("%s MCTF wall %.3f ms per sequence (profiling off)" % (cfg, dt * 1e3))
for k, v in acc.items(): print(f"  {k:40s} {v / iters * 1e3:8.3f} ms")
