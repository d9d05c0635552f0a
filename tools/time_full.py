import sys, time, os, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.pipeline import CompressParams, Compressor
from qsvc_amd.synth import make_clip
W,H=1920,1080
kind = sys.argv[1] if len(sys.argv)>1 else "pan_noise"
mode = sys.argv[2] if len(sys.argv)>2 else "53"
gops = int(sys.argv[3]) if len(sys.argv)>3 else 2
cblk = int(sys.argv[4]) if len(sys.argv)>4 else 64
e = hip.Engine(0)
prm = CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=4, GOPs=gops, block_size=8, block_size_min=8, search_range=4, SRLs=5,
                     lossless=(mode=="53"), reversible=False, cblk=cblk)
comp = Compressor(e, prm)
clip = make_clip(kind, comp.pictures, W, H, 43)
comp.configure(); comp.load(clip)
def run():
    comp.analyze()
    out = comp.texture()
    return sum(len(c) for v in out.values() for c in v)
sz=run(); e.sync()
e.set_profiling(True)
N=3
t0=time.time()
for _ in range(N): run()
e.sync(); dt=(time.time()-t0)/N
P=comp.pictures
print(f"LANES={os.environ.get('QSVC_T1_LANES','64')} {kind} mode={mode} pictures={P}: {dt*1e3:.1f} ms -> {P/dt:.1f} fps; bytes {sz} ({sz*8/(P*W*H):.2f} bpp)")
for k,(ms,n) in e.timings().items(): print(f"  {k:20s} {ms/N:8.3f} ms/iter  ({n//N} launches/iter)")
print(" ", e.j2k_stats())
