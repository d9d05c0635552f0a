"""Throughput with several encoder contexts in flight on one GPU (one host thread per context)."""
import sys, time, threading
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
W, H = 1920, 1080
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prm = dict(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=8, block_size_min=8, search_range=4, SRLs=5, lossless=True)
clip = make_clip("pan_noise", 17, W, H, 43)
comps = []
for k in range(nctx):
    e = hip.Engine(0)
    c = Compressor(e, CompressParams(**prm)); c.configure(); c.load(clip)
    c.analyze(); c.texture(); e.sync()
    comps.append(c)
def work(c, n):
    for _ in range(n):
        c.analyze(); c.texture(copy=False)
    c.e.sync()
t0 = time.time()
th = [threading.Thread(target=work, args=(c, steps)) for c in comps]
[t.start() for t in th]; [t.join() for t in th]
dt = time.time() - t0
print(f"contexts={nctx} steps/ctx={steps}: {dt*1e3/(nctx*steps):.1f} ms per step amortised -> {17*nctx*steps/dt:.1f} fps")
