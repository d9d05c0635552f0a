import sys, time
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
W, H = 1920, 1080
e = hip.Engine(0)
c = Compressor(e, CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=2, block_size=8, block_size_min=8, search_range=4, SRLs=5, lossless=True))
c.configure(); c.load(make_clip("pan_noise", 17, W, H, 43))
c.analyze(); c.texture(); e.sync()
e.set_profiling(True)
N = 5
ta = tt = 0
for _ in range(N):
    t0 = time.perf_counter(); c.analyze(); e.sync(); t1 = time.perf_counter(); c.texture(copy=False); t2 = time.perf_counter()
    ta += t1 - t0; tt += t2 - t1
tm = e.timings()
mctf = sum(ms for k, (ms, n) in tm.items() if not k.startswith("j2k")) / N
j2k = sum(ms for k, (ms, n) in tm.items() if k.startswith("j2k")) / N
print(f"analyze wall {ta/N*1e3:.2f} ms (kernels {mctf:.2f});  texture wall {tt/N*1e3:.2f} ms (kernels {j2k:.2f})")
