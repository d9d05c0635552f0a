"""Per-stage HIP-event times of the MCTF levels of one sequence (tools, GPU box):  python tools/time_mctf.py [cfg3|cfg4] [iters]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, H, bs, P, TRLs = (1920, 1080, 8, 17, 4) if cfg == "cfg3" else (3840, 2160, 16, 33, 5)
kind = sys.argv[3] if len(sys.argv) > 3 else "pan_noise"
if kind == "static":   # one picture + fresh noise per picture: zero motion, no displaced run leaves the picture
    base = make_clip("pan_noise", 1, W, H, 43)[0].astype(np.int16)
    rng = np.random.default_rng(1)
    clip = np.stack([np.clip(base + rng.integers(-2, 3, size=base.shape), 0, 255).astype(np.uint8) for _ in range(P)])
else:
    clip = make_clip(kind, P, W, H, 43)
e = hip.Engine(0)
e.seq_config(W, H)
e.put_frames(hip.LOW, 0, clip)
def run():
    p, S = P, 4
    for t in range(1, TRLs):
        e.analyze_step(t, p, bs, S)
        p = (p + 1) // 2; S *= 2
run(); e.sync()
e.set_profiling(True)
t0 = time.time()
for _ in range(iters): run()
e.sync(); dt = (time.time() - t0) / iters
print("%s MCTF %d levels %d pictures: %.3f ms wall" % (cfg, TRLs - 1, P, dt * 1e3))
tot = 0.0
for k, (ms, n) in e.timings().items():
    print(f"  {k:20s} {ms/iters:8.3f} ms/iter  ({n // iters} launches)"); tot += ms / iters
print(f"  {'sum of kernels':20s} {tot:8.3f} ms/iter")
