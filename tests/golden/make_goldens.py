#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own tools (build container only).

The reference (claudio382/QSVC) ships no golden vectors, so the pins are made here by running the
reference's MCTF stage tools -- compiled from /root/reference by oracle/Makefile into oracle/_ref,
sources never copied -- on seeded synthetic clips (qsvc_amd/synth.py), exactly as
analyze_step.py:117-232 chains them, with default glibc malloc.  What is committed is DATA only:
tiny outputs verbatim (motion fields, frame types, whole small-geometry pictures) and SHA-256
digests of the frame-sized outputs of larger cases.

    python tests/golden/make_goldens.py        # rewrites tests/golden/mctf_*.npz, dwt_kat.npz
    python tests/golden/make_goldens.py <case> …   # only the named mctf_<case>.npz
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

# name, kind, seed, frames, W, H, TRLs, block_size, search_range, full_dump
CASES = [
    ("tiny_pan", "pan", 42, 5, 64, 64, 3, 16, 4, True),
    ("tiny_edge", "edge", 47, 5, 96, 64, 3, 16, 4, True),
    ("tiny_sat", "saturating", 45, 5, 64, 48, 3, 8, 4, True),
    ("tiny_binary", "binary", 48, 5, 64, 64, 2, 16, 4, True),
    ("cif_pan", "pan", 42, 5, 352, 288, 3, 32, 4, False),
    ("cif_scene_cut", "scene_cut", 44, 9, 352, 288, 3, 32, 4, False),
    ("cif_saturating", "saturating", 45, 5, 352, 288, 3, 32, 4, False),
    ("cif_noise", "noise", 46, 5, 352, 288, 2, 32, 4, False),
    ("cif_binary", "binary", 48, 5, 352, 288, 2, 32, 4, False),
    ("cif_edge", "edge", 47, 9, 352, 288, 3, 32, 4, False),
    ("q1_edge_360x200", "edge", 49, 9, 360, 200, 4, 8, 4, False),   # S reaches 16: border quirk Q1
    ("sat_360x200", "saturating", 45, 9, 360, 200, 4, 8, 4, False),  # has 'I' pictures
    ("odd_350x286", "pan_noise", 43, 5, 350, 286, 3, 32, 8, False),  # non-PR sizes (Q5), Q7
    ("q7_200x120", "pan", 42, 5, 200, 120, 2, 32, 4, False),         # rows/cols beyond BY*bs (Q7)
    # deep ME pyramids (motion_estimate.cpp:277: L = rint(log2 S) - 1 up to 6) and the S <- min(2S, 128) clamp of
    # analyze.py:144-147: at the coarse levels one 16x16 block window covers far more than the LL quadrant (Q6/H3)
    ("deep_pan_256x192", "pan_noise", 43, 129, 256, 192, 8, 16, 4, False),    # S = 4, 8, 16, 32, 64, 128, 128 (TRLs 8)
    ("deep_edge_s32_256x192", "edge", 47, 9, 256, 192, 4, 16, 32, False),     # started at --search_range=32: 32, 64, 128
    ("deep_sat_384x256", "saturating", 45, 17, 384, 256, 5, 32, 16, False),   # block 32: 16, 32, 64, 128
    # search ranges 1 and 2 (79 call sites of --search_range=1 in trunk/tests): dwt_levels = -1 / 0 (motion_estimate.cpp:277),
    # one search on the pictures themselves, prediction border 4 S = 4 / 8
    ("s1_cif_pan", "pan_noise", 43, 9, 352, 288, 4, 32, 1, False),            # S = 1, 2, 4
    ("s1_edge_360x200", "edge", 49, 9, 360, 200, 3, 8, 1, False),             # S = 1, 2; block 8, vectors leave the picture
    ("s2_tiny_edge", "edge", 47, 5, 96, 64, 3, 16, 2, True),                  # S = 2, 4
    ("s1_sat_q7_200x120", "saturating", 45, 5, 200, 120, 2, 32, 1, False),    # S = 1 on a picture the block grid does not tile
    # Q5: an odd size along the search pyramid -- the analyse (floor) / synthesise (ceil) walk is not perfect reconstruction,
    # the searched state is what the replay left, and a reference plane keeps it from one B picture to the next
    ("q5_edge_360x200_s32", "edge", 49, 9, 360, 200, 3, 8, 32, False),        # S = 32 (200 % 16 = 8), 64; 4 B pictures
    ("q5_pan_350x286", "pan_noise", 51, 9, 350, 286, 4, 16, 8, False),        # S = 8, 16, 32; ragged block grid as well (Q7)
    ("q5_sat_1080ish_480x270", "saturating", 45, 5, 480, 270, 3, 8, 16, False),  # 1080p / 4: 270 % 8 = 6 at S = 16, then 32
]


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle` in the build container"
    only = set(sys.argv[1:])
    for name, kind, seed, frames, W, H, TRLs, bs, S, full in CASES:
        if only and name not in only:
            continue
        clip = make_clip(kind, frames, W, H, seed)
        levels = O.ref_analyze(clip, W, H, TRLs, bs, S)
        out = dict(meta=np.array([frames, W, H, TRLs, bs, S, seed], dtype=np.int64), kind=np.array(kind),
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
        # motion-field residues (motion_compress.py:146-186) by the reference's own decorrelators; the cases all keep one
        # block size over the levels, which is the configuration `compress` drives them in
        filt = {t: lv["motion_filtered"] for t, lv in enumerate(levels, 1)}
        for t, r in O.ref_motion_residues(filt, H // bs, W // bs).items():
            out[f"motion_residue_{t}"] = r
        # inverse chain (synthesize.py): the reference's un_update / correlate / merge from the coarsest level down, every
        # level fed with the level above's RECONSTRUCTED low pictures
        low = levels[-1]["low"]
        S_t = [S * 2 ** k for k in range(len(levels))]
        for t in range(len(levels), 0, -1):
            lv = levels[t - 1]
            r = O.ref_synthesize_step(low, lv["high"], lv["motion_filtered"], lv["frame_types"], W, H, bs, min(S_t[t - 1], 128), t=t)
            low = r["low"]
            out[f"synth_low_{t - 1}_sha"] = np.array(sha(low))
            if full:
                out[f"synth_low_{t - 1}"] = low
        np.savez_compressed(os.path.join(HERE, f"mctf_{name}.npz"), **out)
        print(name, [lv["frame_types"] for lv in levels])

    if only:
        return
    # known answers of the in-tree transforms (dwt2d.cpp / 5_3.cpp / Haar.cpp) via oracle/_ref/dwt_probe
    rng = np.random.default_rng(7)
    kat = {}
    for (h, w, lv) in [(2, 2, 1), (3, 3, 1), (4, 5, 1), (5, 4, 1), (8, 9, 2), (9, 8, 2), (16, 17, 3), (17, 16, 2),
                       (32, 48, 3), (30, 50, 2), (36, 44, 2)]:
        x = rng.integers(0, 256, size=(h, w)).astype(np.int16)
        kat[f"in_{h}x{w}_{lv}"] = x
        a = O.ref_dwt(x, lv, inverse=False, filt="53")
        kat[f"a53_{h}x{w}_{lv}"] = a
        kat[f"s53_{h}x{w}_{lv}"] = O.ref_dwt(x, lv, inverse=True, filt="53")
        kat[f"shaar_{h}x{w}_{lv}"] = O.ref_dwt(x, lv, inverse=True, filt="haar")
    np.savez_compressed(os.path.join(HERE, "dwt_kat.npz"), **kat)
    print("dwt_kat", len(kat))

    make_motion_kat()
    make_header_size_goldens()


def make_motion_kat():
    """The two motion decorrelators AND their inverses (the same sources without -D ANALYZE) on adversarial fields: odd negatives
    for the C division, int16 wrap, a reference file shorter than the predicted one (the tools keep the last field read)."""
    rng = np.random.default_rng(13)
    mk = {}
    for i, (f1, f2, BY, BX) in enumerate([(4, 2, 3, 5), (5, 2, 2, 2), (2, 1, 4, 3), (1, 0, 2, 3)]):
        a = rng.integers(-32768, 32768, size=(f1, 4, BY, BX)).astype(np.int16)
        b = rng.integers(-32768, 32768, size=(f2, 4, BY, BX)).astype(np.int16)
        mk[f"pred_{i}"], mk[f"ref_{i}"] = a, b
        r = O.ref_motion_residues({1: a, 2: b}, BY, BX) if f2 else None
        if r is not None:
            mk[f"inter_{i}"], mk[f"bidir_{i}"] = r[1], r[2]
            # read the same arrays as residues: motion_2 = bidirectional correlate of b, motion_1 = a + motion_2 / 2
            c = O.ref_motion_fields({1: a, 2: b}, BY, BX)
            mk[f"corr_inter_{i}"], mk[f"corr_bidir_{i}"] = c[1], c[2]
        else:
            mk[f"bidir_{i}"] = O.ref_motion_residues({1: a}, BY, BX)[1]
            mk[f"corr_bidir_{i}"] = O.ref_motion_fields({1: a}, BY, BX)[1]
    np.savez_compressed(os.path.join(HERE, "motion_kat.npz"), **mk)
    print("motion_kat", len(mk))


def make_header_size_goldens():
    """header_size (trunk/src/header_size.c compiled into oracle/_ref) on deterministic marker streams + adversarial byte
    patterns for its consuming scanner (FF FF 91 00 04 does not match; files ending inside a marker)."""
    import json, subprocess, tempfile
    hs = []
    rng = np.random.default_rng(11)
    bodies = []
    for case in range(8):
        body = bytearray(b"\xff\x4f" + bytes(rng.integers(0, 255, size=int(rng.integers(20, 90)), dtype=np.uint8)))
        for k in range(int(rng.integers(1, 7))):
            body += b"\xff\x91\x00\x04" + bytes([0, k])
            n = int(rng.choice([1, 1, 5, 40]))
            body += bytes(rng.integers(0, 255, size=n, dtype=np.uint8))
        body += b"\xff\xd9"
        bodies.append(bytes(body))
    sop = b"\xff\x91\x00\x04"
    bodies += [b"", b"\xff", b"\xff\x91", b"\xff\x91\x00", sop, sop + b"\x00\x00\xff\xd9", b"\xff\xff\x91\x00\x04\x00\x01ab" + sop + b"\x00",
               b"\xff\x4fxx\xff\x91\xff\x91\x00\x04\x00\x00" + sop + b"\x00\x01\x00" + sop + b"\x00\x02\xff\xd9",
               b"\xff\x4f" + sop + b"\x00\x00\xff" + sop + b"\x00\x01\x80\xff\x91\x00\xff\x91\x00\x04\xff",
               bytes(rng.integers(0, 256, size=3000, dtype=np.uint8)) + sop + b"\x00\x00q" + sop]
    for body in bodies:
        with tempfile.NamedTemporaryFile(suffix=".j2c", delete=False) as f:
            f.write(body)
        out = subprocess.run([os.path.join(O.REF_DIR, "header_size"), f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        hs.append({"hex": body.hex(), "out": int([l for l in out.splitlines() if l.startswith("OUT")][0][4:])})
    json.dump(hs, open(os.path.join(HERE, "header_size.json"), "w"))
    print("header_size", [h["out"] for h in hs])


if __name__ == "__main__":
    if sys.argv[1:] == ["header_size"]:
        make_header_size_goldens()
    elif sys.argv[1:] == ["motion_kat"]:
        make_motion_kat()
    else:
        main()
