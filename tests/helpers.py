"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_cases():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "mctf_*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, f"mctf_{name}.npz"), allow_pickle=False)
    frames, W, H, TRLs, bs, S, seed = (int(v) for v in z["meta"])
    return z, dict(frames=frames, W=W, H=H, TRLs=TRLs, bs=bs, S=S, seed=seed, kind=str(z["kind"]))


def golden_input(z, m):
    from qsvc_amd.synth import make_clip
    clip = z["input"] if "input" in z.files else make_clip(m["kind"], m["frames"], m["W"], m["H"], m["seed"])
    assert sha(clip) == str(z["input_sha"]), "synthetic clip generator drifted from the committed goldens"
    return clip


def check_levels_against_golden(z, levels, what=("motion", "motion_filtered", "frame_types", "high", "low")):
    """levels: list of dicts with motion/motion_filtered/frame_types/high/low[/prediction] per temporal level."""
    for t, lv in enumerate(levels, 1):
        if "motion" in what:
            np.testing.assert_array_equal(lv["motion"], z[f"motion_{t}"], err_msg=f"motion_{t}")
        if "motion_filtered" in what:
            np.testing.assert_array_equal(lv["motion_filtered"], z[f"motion_filtered_{t}"], err_msg=f"motion_filtered_{t}")
        if "frame_types" in what:
            assert bytes(lv["frame_types"]) == z[f"frame_types_{t}"].tobytes(), f"frame_types_{t}"
        for k in ("high", "low", "prediction"):
            if k in what and lv.get(k) is not None:
                assert sha(lv[k]) == str(z[f"{k}_{t}_sha"]), f"{k}_{t} digest"
                if f"{k}_{t}" in z.files:
                    np.testing.assert_array_equal(lv[k], z[f"{k}_{t}"], err_msg=f"{k}_{t}")


def n4_cases():
    return sorted(os.path.basename(p)[3:-4] for p in glob.glob(os.path.join(GOLDEN, "n4_*.npz")))


def load_n4(name):
    """Goldens of the optional MCTF flags (tests/golden/make_n4_goldens.py): meta also holds block_overlaping, border_size and (when
    not zero) subpixel_accuracy."""
    z = np.load(os.path.join(GOLDEN, f"n4_{name}.npz"), allow_pickle=False)
    frames, W, H, TRLs, bs, S, seed, ov, bord = (int(v) for v in z["meta"][:9])
    sub = int(z["meta"][9]) if len(z["meta"]) > 9 else 0
    return z, dict(frames=frames, W=W, H=H, TRLs=TRLs, bs=bs, S=S, seed=seed, ov=ov, bord=bord, sub=sub, kind=str(z["kind"]))
