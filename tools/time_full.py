import sys, time, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
W,H,bs=1920,1080,8
P=17
kind = sys.argv[1] if len(sys.argv)>1 else "pan_noise"
rev = (sys.argv[2] != "97") if len(sys.argv)>2 else True
clip = make_clip(kind, P, W, H, 43)
e = hip.Engine(0)
e.seq_config(W,H)
e.put_frames(hip.LOW,0,clip)
prm = hip.j2k_params(reversible=rev, levels=4)
def run():
    p, S = P, 4
    sizes=0
    for t in range(1,4):
        e.analyze_step(t,p,bs,S)
        nB=p//2
        cs = e.j2k_encode_frames(hip.HIGH,t,0,nB,prm); sizes+=sum(map(len,cs))
        p=(p+1)//2; S*=2
    cs = e.j2k_encode_frames(hip.LOW,3,0,p,prm); sizes+=sum(map(len,cs))
    return sizes
sz=run(); e.sync()
e.set_profiling(True)
N=3
t0=time.time()
for _ in range(N): run()
e.sync(); dt=(time.time()-t0)/N
print(f"{kind} rev={rev}: full encode 17 frames: {dt*1e3:.1f} ms -> {16/dt:.1f} fps; bytes {sz} ({sz*8/(17*W*H):.2f} bpp)")
for k,(ms,n) in e.timings().items(): print(f"  {k:20s} {ms/N:8.3f} ms/iter  ({n//N} launches/iter)")
print(e.j2k_stats())
