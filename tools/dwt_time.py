#!/usr/bin/env python3
"""GPU box: time the JPEG 2000 DWT stage alone (HIP events of the library) on the 17-picture 1080p step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
W, H = 1920, 1080
e = hip.Engine(0)
e.seq_config(W, H)
clip = make_clip("pan_noise", 17, W, H, 43)
e.put_frames(hip.LOW, 0, clip)
rev = (sys.argv[1] if len(sys.argv) > 1 else "53") == "53"
prm = hip.j2k_params(reversible=rev, levels=4, layers=1)
for it in range(2):
    e.j2k_encode_multi([(hip.LOW, 0, 0, 17, prm)], copy=False)
e.set_profiling(True)
n = 5
for it in range(n):
    e.j2k_encode_multi([(hip.LOW, 0, 0, 17, prm)], copy=False)
t = e.timings()
print({k: round(v[0] / n, 4) for k, v in t.items()})
