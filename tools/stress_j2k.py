#!/usr/bin/env python3
"""Randomised parity stress of the JPEG 2000 stage (GPU box): HIP code-streams against the T.800 restatement (oracle/, test
infrastructure) for random plane sizes, contents, levels, code-block sizes, reversible and irreversible, 8-bit planes and signed
16-bit planes.   python3 tools/stress_j2k.py [cases=300] [seed=1] [max_size=260]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import j2k as J  # noqa: E402
from qsvc_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAXS = int(sys.argv[3]) if len(sys.argv) > 3 else 260
e = hip.Engine(0)
bad = 0
for k in range(n):
    W, H = int(rng.integers(1, MAXS)), int(rng.integers(1, MAXS))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        p = rng.integers(0, 256, size=(H, W))
    elif kind == 1:
        p = np.full((H, W), int(rng.integers(0, 256)))
        for _ in range(int(rng.integers(0, 6))):
            p[rng.integers(0, H), rng.integers(0, W)] = int(rng.integers(0, 256))
    elif kind == 2:
        yy, xx = np.mgrid[0:H, 0:W]
        p = (128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.int64)
    elif kind == 3:
        p = np.clip(128 + np.rint(rng.laplace(0, 6, size=(H, W))), 0, 255)
    elif kind == 4:
        p = np.where(rng.random((H, W)) < 0.03, 255, 0)
    else:
        p = np.clip(np.cumsum(rng.integers(-3, 4, size=(H, W)), axis=1) + 128, 0, 255)
    p = np.ascontiguousarray(p, dtype=np.uint8)
    rev = bool(rng.integers(0, 2))
    levels = int(rng.integers(0, 6))
    cblk = int(rng.choice([64, 64, 32, 16]))
    want = J.encode_plane(p, reversible=rev, levels=levels, cblk=cblk)
    got = e.j2k_encode_plane(p, hip.j2k_params(reversible=rev, levels=levels, cblk=cblk))
    if got != want:
        bad += 1
        print("MISMATCH", k, W, H, kind, rev, levels, cblk, len(got), len(want), flush=True)
    if k % 50 == 49:
        print(k + 1, "cases,", bad, "mismatches", flush=True)
for k in range(n // 6):  # signed 16-bit planes (motion fields): Clevels = 0, reversible
    nn, H, W = int(rng.integers(1, 5)), int(rng.integers(1, 140)), int(rng.integers(1, 250))
    amp = int(rng.choice([3, 40, 300, 16000]))
    pl = rng.integers(-amp, amp + 1, size=(nn, H, W)).astype(np.int16)
    if rng.random() < 0.3:
        pl[rng.random(pl.shape) < 0.9] = 0
    got = e.j2k_encode_planes_s16(pl, hip.j2k_params(reversible=True, levels=0))
    for i in range(nn):
        if got[i] != J.encode_plane_s16(pl[i]):
            bad += 1
            print("MISMATCH s16", k, i, H, W, amp, flush=True)
print("done:", bad, "mismatches")
sys.exit(1 if bad else 0)
