#!/usr/bin/env python3
"""Randomised parity stress of the predict step (GPU box): decorrelate on ARBITRARY motion fields (every block its own random pair
of vectors up to the search range, so predictions leave the picture on every side) against the restatement (oracle/, test
infrastructure): residues, frame types, filtered motion and the entropy histograms.
   python3 tools/stress_predict.py [cases=60] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from qsvc_amd import hip  # noqa: E402
from qsvc_amd.synth import make_clip  # noqa: E402

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
    S = int(rng.choice([4, 8, 16]))
    nB = int(rng.integers(1, 4))
    P = 2 * nB + 1
    kind = str(rng.choice(["pan_noise", "noise", "edge", "binary"]))
    low = make_clip(kind, P, W, H, int(rng.integers(1, 1000)))
    mv = rng.integers(-S, S + 1, size=(nB, 4, H // bs, W // bs), dtype=np.int16)
    if rng.random() < 0.5:
        mv[:, :, 1::2, :] = 0
    always_B = int(rng.random() < 0.5)
    want = O.decorrelate(low, W, H, bs, S, mv, always_B, want_hists=True)
    e.seq_config(W, H, True)
    e.put_frames(hip.LOW, 0, low)
    e.put_motion(hip.MOTION_RAW, 1, mv, bs)
    e.decorrelate(1, P, bs, S, always_B)
    got_types = bytes(e.get_frame_types(1, nB))
    got_high = e.get_frames(hip.HIGH, 1, 0, nB)
    got_mv = e.get_motion(hip.MOTION_FILTERED, 1, nB, bs)
    got_h = e.get_histograms(1, nB).astype(np.int64)
    ok = got_types == want["types"] and np.array_equal(got_high, want["high"]) and np.array_equal(got_mv, want["mv"]) and (always_B or np.array_equal(got_h, np.asarray(want["hists"]).astype(np.int64)))   # (with always_B the reference computes no entropies: decorrelate.cpp:945-979)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {kind} {W}x{H} bs {bs} S {S} nB {nB} always_B {always_B}: types {got_types} / {want['types']}, "
              f"high {int((got_high != want['high']).sum())} samples, mv {int((got_mv != want['mv']).sum())}, hist {int((got_h != np.asarray(want['hists'])).sum())}")
    if (case + 1) % 20 == 0:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"predict stress done: {bad} mismatches")
sys.exit(1 if bad else 0)
