#!/usr/bin/env python3
"""Full-size pins from the REFERENCE's own tools (build container only): SHA-256 digests, nothing else.

BASELINE.json config 3 (1920x1080, 17 pictures, TRLs=4, block 8, search 4->8->16) on two seeded clips and the first two
temporal levels of config 4 (3840x2160, block 16, search 4->8; 9 pictures are enough for two levels) are run through
oracle/_ref -- split / motion_estimate / decorrelate / update compiled from /root/reference by oracle/Makefile -- chained
as analyze.py:110-153 / analyze_step.py:117-232 chain them.  Per level the digest of every output file is committed
(whole file and per picture, so that a mismatch names the picture): motion_t, motion_filtered_t, frame_types_t (verbatim,
a few bytes), high_t, low_t.  The GPU tests (tests/test_gpu_fullsize.py) compare the HIP path with these at zero oracle
cost on the GPU box.

    python tests/golden/make_fullsize_goldens.py      # rewrites tests/golden/fullsize_digests.json  (~5 min, 1 core)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from qsvc_amd.synth import make_clip  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name, kind, seed, frames, W, H, TRLs, block_size, search_range
CASES = [
    ("cfg3_pan_noise", "pan_noise", 43, 17, 1920, 1080, 4, 8, 4),
    ("cfg3_edge", "edge", 47, 17, 1920, 1080, 4, 8, 4),
    ("cfg4_pan_noise_2levels", "pan_noise", 43, 9, 3840, 2160, 3, 16, 4),
    # BASELINE config 4 whole: 33 pictures, TRLs=5, search 4 -> 8 -> 16 -> 32 (L = 4 at the last level)
    ("cfg4_pan_noise", "pan_noise", 43, 33, 3840, 2160, 5, 16, 4),
    # Q5 at full size: BASELINE's 1080p geometry one temporal level deeper (level 4: search 32, L = 4, 1080 % 16 = 8 -- the
    # analyse / synthesise walk of motion_estimate.cpp:283-309 is not perfect reconstruction there), and the same search range
    # with two B pictures (the PREV reference of the second is the state the first walk left)
    ("q5_1080p_trls5", "edge", 47, 17, 1920, 1080, 5, 8, 4),
    ("q5_1080p_s32", "pan_noise", 43, 5, 1920, 1080, 2, 8, 32),
]


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle` in the build container"
    only = set(sys.argv[1:])
    path = os.path.join(HERE, "fullsize_digests.json")
    out = json.load(open(path)) if os.path.exists(path) and only else {}
    for name, kind, seed, frames, W, H, TRLs, bs, S in CASES:
        if only and name not in only:
            continue
        t0 = time.perf_counter()
        clip = make_clip(kind, frames, W, H, seed)
        timing = {}
        levels = O.ref_analyze(clip, W, H, TRLs, bs, S, timing=timing)
        case = dict(kind=kind, seed=seed, frames=frames, W=W, H=H, TRLs=TRLs, block_size=bs, search_range=S,
                    input_sha=sha(clip), levels=[], ref_seconds={k: round(v, 2) for k, v in timing.items()})
        for t, lv in enumerate(levels, 1):
            d = dict(t=t, pictures=int(lv["pictures"]), search_range=int(lv["search_range"]), block_size=int(lv["block_size"]),
                     frame_types=lv["frame_types"].decode("ascii"))
            for k in ("motion", "motion_filtered", "high", "low"):
                d[k] = sha(lv[k])
                d[k + "_per_picture"] = [sha(x)[:16] for x in lv[k]]
            case["levels"].append(d)
        out[name] = case
        print(name, [lv["frame_types"] for lv in levels], f"{time.perf_counter() - t0:.1f} s", timing, flush=True)
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
