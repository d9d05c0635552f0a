"""Single-GPU sanity of the RCCL plumbing: unique id creation and a 1-rank communicator (no peers, no traffic)."""
import sys
sys.path.insert(0, '.')
import torch  # noqa: F401  (as in bench.py: torch is imported first, its own ROCm libraries are in the process)
from qsvc_amd import hip
uid = hip.Engine.comm_unique_id()
print("unique id ok", len(uid))
e = hip.Engine(0)
e.comm_init(0, 1, uid)
print("comm_init(1 rank) ok")
e.seq_config(64, 64)
import numpy as np
from qsvc_amd.synth import make_clip
clip = make_clip("pan", 5, 64, 64, 1)
e.put_frames(hip.LOW, 0, clip)
e.motion_estimate(1, 5, 16, 4); e.decorrelate(1, 5, 16, 4)
e.halo_exchange(1)   # no neighbours: packs nothing, clears both halos
e.update(1, 5, 16, 0.25)
print("halo_exchange(no neighbours) ok")
e.close()
