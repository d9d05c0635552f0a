#!/usr/bin/env python3
"""Randomised parity stress of the layered (PCRD) path on the GPU box: several quality layers with random slopes, HIP streams
against the T.800 restatement (oracle/, test infrastructure): per-pass lengths and distortion sums must agree exactly.
   python3 tools/stress_layers.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import j2k as J
from qsvc_amd import hip
rng = np.random.default_rng(5)
e = hip.Engine(0)
bad = 0
for k in range(60):
    W, H = int(rng.integers(8, 300)), int(rng.integers(8, 300))
    kind = k % 3
    if kind == 0: p = rng.integers(0, 256, size=(H, W))
    elif kind == 1:
        yy, xx = np.mgrid[0:H, 0:W]; p = 128 + 90 * np.sin(xx / 11.0) * np.cos(yy / 5.0) + rng.integers(-4, 5, size=(H, W))
    else:
        p = np.full((H, W), 100); p[rng.random((H, W)) < 0.02] = 250
    p = np.ascontiguousarray(np.clip(p, 0, 255), dtype=np.uint8)
    rev = bool(rng.integers(0, 2)); levels = int(rng.integers(1, 5)); layers = int(rng.integers(2, 6))
    slopes = tuple(sorted((int(x) for x in rng.integers(43000, 46500, size=layers)), reverse=True))
    want = J.encode_plane(p, reversible=rev, levels=levels, layers=layers, slopes=slopes)
    got = e.j2k_encode_plane(p, hip.j2k_params(reversible=rev, levels=levels, layers=layers, slopes=slopes))
    if got != want:
        bad += 1; print("MISMATCH", k, W, H, kind, rev, levels, layers, slopes, len(got), len(want), flush=True)
print("layers stress done:", bad, "mismatches")
