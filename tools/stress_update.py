#!/usr/bin/env python3
"""Randomised parity stress of the update step (GPU box): the scatter form and the gather form of the HIP path against the
restatement (oracle/, test infrastructure) on random geometries, block sizes, reaches, picture counts, frame types ('I' pictures
contribute nothing), update factors and ARBITRARY motion fields (footprints overlap, sources fold onto the picture's border).
   python3 tools/stress_update.py [cases=60] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from qsvc_amd import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
e = hip.Engine(0)
bad = 0
for case in range(n):
    bs = int(rng.choice([8, 16, 32]))
    W = bs * int(rng.integers(2, 21))
    H = bs * int(rng.integers(1, 13))
    if W % 8 or H % 2 or W < 16 or H < 16:
        continue
    R = int(rng.choice([1, 2, 4, 7, 16, 33, 60, 100]))
    nB = int(rng.integers(1, 4))
    P = 2 * nB + 1
    F = W * H * 3 // 2
    low = rng.integers(0, 256, size=(P, F), dtype=np.uint8)
    high = rng.integers(100, 156, size=(nB, F), dtype=np.uint8)
    if rng.random() < 0.5:
        high[:, ::5] = rng.integers(0, 256, size=high[:, ::5].shape, dtype=np.uint8)
    mv = rng.integers(-R, R + 1, size=(nB, 4, H // bs, W // bs), dtype=np.int16)
    if rng.random() < 0.5:
        mv[:, :, ::2, :] = mv[:, :, :1, :1]          # stretches of equal vectors
    types = bytes(rng.choice([ord("B"), ord("B"), ord("I")], size=nB).astype(np.uint8))
    uf = float(rng.choice([0.25, 0.25, 0.5, 0.3]))
    want = O.update(low, W, H, bs, high, mv, types, uf)
    for form in (1, 0):
        e.set_tuning("upd_scatter", form)
        e.seq_config(W, H, True)
        e.put_frames(hip.LOW, 0, low)
        e.put_frames(hip.HIGH, 1, high)
        e.put_motion(hip.MOTION_FILTERED, 1, mv, bs)
        e.put_frame_types(1, types)
        e.update(1, P, bs, uf)
        got = e.get_frames(hip.LOW, 1, 0, nB + 1)
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH case {case}: {W}x{H} bs {bs} R {R} nB {nB} types {types} uf {uf} form {form}: {int((got != want).sum())} samples differ")
    if (case + 1) % 20 == 0:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
e.set_tuning("upd_scatter", 1)
print(f"update stress done: {bad} mismatches")
sys.exit(1 if bad else 0)
