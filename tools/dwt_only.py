#!/usr/bin/env python3
"""GPU box: run ONLY the forward DWT of the 17-picture 1080p step (unit-test entry point on one luma plane x N) -- for PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qsvc_amd import hip
from qsvc_amd.synth import make_clip
W, H = 1920, 1080
e = hip.Engine(0)
e.seq_config(W, H)
clip = make_clip("pan_noise", 3, W, H, 43)
e.put_frames(hip.LOW, 0, clip)
prm = hip.j2k_params(reversible=True, levels=4, layers=1)
try:
    e.j2k_encode_multi([(hip.LOW, 0, 0, 3, prm)], copy=False)
except Exception as ex:
    print("expected with stores removed:", ex)
