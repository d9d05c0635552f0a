import sys, time, numpy as np
sys.path.insert(0, '.')
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
W,H,bs=1920,1080,8
P=17
clip = make_clip("pan_noise", P, W, H, 43)
e = hip.Engine(0)
e.seq_config(W,H)
e.put_frames(hip.LOW,0,clip)
def run():
    p, S = P, 4
    for t in range(1,4):
        e.analyze_step(t,p,bs,S)
        p=(p+1)//2; S*=2
run(); e.sync()
e.set_profiling(True)
t0=time.time()
for _ in range(3): run()
e.sync(); dt=(time.time()-t0)/3
print("MCTF 3 levels 17 frames: %.2f ms -> %.1f fps"%(dt*1e3, 16/dt))
for k,(ms,n) in e.timings().items(): print(f"  {k:20s} {ms/3:8.3f} ms/iter  ({n} launches)")
