#!/usr/bin/env python3
"""Goldens for the reference's optional MCTF flags, from the REFERENCE's own tools (build container
only): --block_overlaping (decorrelate / correlate, decorrelate.cpp:84-172) and --border_size (motion_estimate,
motion_estimate.cpp:213-217,714).  Same recipe as make_goldens.py: oracle/_ref binaries on seeded synthetic clips, default glibc
malloc (with --border_size the reference reads the never-filled border of its predicted plane: zeros of a fresh heap), data only.
--subpixel_accuracy (motion_estimate.cpp:361-412,716-735,779-782; decorrelate.cpp:534-566,656-686,828-861): the search reads plane
borders nothing ever wrote and, through texture::alloc's row-pointer bug at the up-sampled size, glibc's chunk headers -- repeatable
under default malloc (tests/golden/subpel_probe.json), which is what these goldens are made with.

    python tests/golden/make_n4_goldens.py            # rewrites tests/golden/n4_*.npz
    python tests/golden/make_n4_goldens.py <case> …   # only the named n4_<case>.npz
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from qsvc_amd.synth import make_clip  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name, kind, seed, W, H, TRLs, block_size, search_range, block_overlaping, border_size, full_dump[, subpixel_accuracy[, pictures]]
CASES = [
    ("ov2_tiny", "pan_noise", 3, 128, 96, 3, 16, 4, 2, 0, True),
    ("ov4_tiny_edge", "edge", 47, 96, 64, 3, 16, 4, 4, 0, True),
    ("ov1_sat", "saturating", 45, 64, 48, 3, 8, 4, 1, 0, True),
    ("bord2_tiny", "edge", 47, 96, 64, 3, 16, 8, 0, 2, True),
    ("ov2_bord2_cif", "pan_noise", 43, 352, 288, 3, 32, 4, 2, 2, False),
    ("ov2_bord2_q1_360x200", "edge", 49, 360, 200, 4, 8, 4, 2, 2, False),
    ("ov4_bord1_360x200", "edge", 49, 360, 200, 3, 8, 4, 4, 1, False),
    ("ov8_binary", "binary", 48, 64, 64, 2, 16, 4, 8, 0, False),
    ("bord8_cif", "pan_noise", 43, 352, 288, 2, 32, 4, 0, 8, False),
    # --subpixel_accuracy = 1, 2, 3 (105 of the 217 call sites in trunk/tests use 2 or 3)
    ("sub1_tiny", "pan_noise", 3, 128, 96, 3, 16, 4, 0, 0, True, 1),
    ("sub2_tiny_edge", "edge", 47, 96, 64, 3, 16, 4, 0, 0, True, 2),
    ("sub1_q1_360x200", "edge", 49, 360, 200, 4, 8, 4, 0, 0, False, 1),       # S reaches 16: the Q1 row at the up-sampled size
    ("sub2_sat_cif", "saturating", 45, 352, 288, 3, 32, 4, 0, 0, False, 2),
    ("sub1_ov2_bord2", "pan_noise", 43, 128, 96, 3, 16, 4, 2, 2, False, 1),   # with --block_overlaping and --border_size
    ("sub3_q7_200x120", "pan", 42, 200, 120, 2, 32, 2, 0, 0, False, 3),       # ragged block grid (Q7), three sub-pixel levels
    ("sub1_binary", "binary", 48, 64, 64, 2, 16, 4, 0, 0, False, 1),          # every picture 'I'
    ("sub1_q5_350x286", "pan_noise", 43, 350, 286, 3, 32, 8, 0, 0, False, 1), # odd pyramid sizes (Q5) under the sub-pixel search
    ("sub2_sat_360x200", "saturating", 45, 360, 200, 4, 8, 4, 0, 0, False, 2),  # 'I' pictures among 'B' ones
    ("sub1_q5_360x200_s32", "edge", 49, 360, 200, 2, 8, 32, 0, 0, False, 1, 5),  # Q5 (200 % 16 = 8) on a tiled grid; 5 pictures, one level
]


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle` in the build container"
    only = set(sys.argv[1:])
    for case in CASES:
        name, kind, seed, W, H, TRLs, bs, S, ov, bord, full = case[:11]
        sub = case[11] if len(case) > 11 else 0
        if only and name not in only:
            continue
        frames = case[12] if len(case) > 12 else 2 ** (TRLs - 1) + 1
        clip = make_clip(kind, frames, W, H, seed)
        levels = O.ref_analyze(clip, W, H, TRLs, bs, S, block_overlaping=ov, border_size=bord, subpixel_accuracy=sub)
        out = dict(meta=np.array([frames, W, H, TRLs, bs, S, seed, ov, bord] + ([sub] if sub else []), dtype=np.int64), kind=np.array(kind),
                   input_sha=np.array(sha(clip)))
        if full:
            out["input"] = clip
        for t, lv in enumerate(levels, 1):
            out[f"motion_{t}"] = lv["motion"]
            out[f"motion_filtered_{t}"] = lv["motion_filtered"]
            out[f"frame_types_{t}"] = np.frombuffer(lv["frame_types"], dtype=np.uint8)
            for k in ("high", "low", "prediction"):
                out[f"{k}_{t}_sha"] = np.array(sha(lv[k]))
                if full:
                    out[f"{k}_{t}"] = lv[k]
        low = levels[-1]["low"]
        for t in range(len(levels), 0, -1):  # the inverse chain with the reference's correlate --block_overlaping
            lv = levels[t - 1]
            r = O.ref_synthesize_step(low, lv["high"], lv["motion_filtered"], lv["frame_types"], W, H, bs, min(S * 2 ** (t - 1), 128), t=t,
                                      block_overlaping=ov, subpixel_accuracy=sub)
            low = r["low"]
            out[f"synth_low_{t - 1}_sha"] = np.array(sha(low))
        np.savez_compressed(os.path.join(HERE, f"n4_{name}.npz"), **out)
        print(name, [lv["frame_types"] for lv in levels])


if __name__ == "__main__":
    main()
