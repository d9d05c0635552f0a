#!/usr/bin/env python3
"""GPU box: the JPEG 2000 stage of N pictures alone (for rocprofv3 kernel traces / PMC passes of the DWT kernels, level by level).
usage: dwt_prof.py [97|53] [W H N levels reps]   default: 97 3840 2160 33 5 4 (config 4's texture batch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
a = sys.argv[1:]
rev = (a[0] if a else "97") == "53"
W, H, N, L, reps = (int(x) for x in (a[1:6] if len(a) >= 6 else (3840, 2160, 33, 5, 4)))
e = hip.Engine(0)
e.seq_config(W, H)
e.put_frames(hip.LOW, 0, make_clip("pan_noise", N, W, H, 43))
prm = hip.j2k_params(reversible=rev, levels=L, layers=1)
def run():
    try:
        e.j2k_encode_multi([(hip.LOW, 0, 0, N, prm)], copy=False)
    except hip.QsvcError as ex:   # (ablation builds of the kernels produce planes the later stages may refuse)
        print("refused:", ex)
for it in range(reps):
    run()
e.set_profiling(True)
for it in range(3):
    run()
print({k: round(v[0] / 3, 4) for k, v in e.timings().items()})
