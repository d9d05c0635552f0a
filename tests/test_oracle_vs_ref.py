"""The CPU restatement against the reference's own tools compiled into oracle/_ref, on fresh seeds
and geometries beyond the committed goldens.  Runs wherever oracle/_ref exists (the build container,
and the GPU box because the built binaries travel with the snapshot); skipped otherwise."""
import numpy as np
import pytest

from qsvc_amd.synth import make_clip

CASES = [
    ("pan", 11, 5, 352, 288, 32, 4),
    ("pan_noise", 12, 5, 128, 96, 16, 8),
    ("edge", 13, 5, 160, 96, 8, 16),
    ("saturating", 14, 5, 128, 128, 8, 4),
    ("noise", 15, 3, 96, 64, 16, 4),
    ("pan", 16, 3, 98, 70, 16, 8),     # odd intermediate sizes: literal analyse/synthesise replay (Q5)
    ("edge", 17, 5, 360, 200, 8, 16),  # Q1 row aliasing reachable
    ("pan_noise", 18, 9, 72, 56, 16, 4),   # Q7: both margins unpredicted, four B pictures deep in the stale-state recurrence
    ("edge", 19, 9, 120, 68, 32, 4),       # Q7: margins wider than the chroma/2 filter support
    ("scene_cut", 20, 7, 1920 // 8, 1080 // 8 + 1, 64 // 8 * 2, 4),  # 240x136 bs 16: bottom margin only (1080p/bs 64 in small)
]


@pytest.mark.parametrize("kind,seed,frames,W,H,bs,S", CASES)
def test_restatement_equals_compiled_reference(oracle, kind, seed, frames, W, H, bs, S):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    clip = make_clip(kind, frames, W, H, seed)
    a = oracle.analyze_step(clip, W, H, bs, S, want_prediction=True)
    b = oracle.ref_analyze_step(clip, W, H, bs, S)
    for k in ("motion", "motion_filtered", "high", "low", "prediction"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["frame_types"] == b["frame_types"]


@pytest.mark.parametrize("kind,seed,frames,W,H,bs,S", [c for c in CASES if c[0] != "pan" or c[2] == 5] + [("binary", 48, 5, 64, 64, 16, 4)])
def test_inverse_restatement_equals_compiled_reference(oracle, kind, seed, frames, W, H, bs, S):
    """un_update / correlate / merge of oracle/_ref against qo_un_update / qo_correlate on the analysis outputs."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    clip = make_clip(kind, frames, W, H, seed)
    a = oracle.analyze_step(clip, W, H, bs, S)
    got = oracle.synthesize_step(a["low"], a["high"], a["motion_filtered"], a["frame_types"], W, H, bs, S)
    want = oracle.ref_synthesize_step(a["low"], a["high"], a["motion_filtered"], a["frame_types"], W, H, bs, S)
    for k in ("even", "odd", "low"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_dwt_restatement_equals_reference_templates(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    for (h, w, lv) in [(6, 10, 1), (7, 11, 2), (20, 28, 2), (33, 17, 1)]:
        x = rng.integers(-200, 400, size=(h, w)).astype(np.int16)
        np.testing.assert_array_equal(oracle.dwt53t(x, lv), oracle.ref_dwt(x, lv))
        np.testing.assert_array_equal(oracle.dwt53t(x, lv, inverse=True), oracle.ref_dwt(x, lv, inverse=True))


# --subpixel_accuracy: the heap-image restatement (oracle/mctf_subpel_oracle.c) against the compiled tools
SUBPEL = [
    # kind, seed, frames, W, H, TRLs, bs, S, border_size, block_overlaping, subpixel_accuracy
    ("pan_noise", 61, 5, 128, 96, 3, 16, 4, 0, 0, 1),
    ("edge", 62, 9, 160, 96, 3, 8, 8, 0, 0, 2),
    ("saturating", 63, 5, 128, 128, 2, 8, 4, 1, 0, 1),
    ("binary", 64, 5, 64, 64, 2, 16, 4, 0, 0, 2),
    ("pan", 65, 5, 98, 70, 2, 16, 8, 0, 0, 1),        # odd pyramid sizes (Q5) under the sub-pixel levels
    ("edge", 66, 5, 120, 68, 2, 32, 4, 0, 2, 1),      # ragged block grid + overlap
    ("noise", 67, 3, 96, 64, 2, 16, 2, 0, 0, 3),
]


@pytest.mark.parametrize("kind,seed,frames,W,H,TRLs,bs,S,bord,ov,sub", SUBPEL)
def test_subpixel_restatement_equals_compiled_reference(oracle, kind, seed, frames, W, H, TRLs, bs, S, bord, ov, sub):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    clip = make_clip(kind, frames, W, H, seed)
    got = oracle.analyze(clip, W, H, TRLs, bs, S, border_size=bord, block_overlaping=ov, subpixel_accuracy=sub)
    want = oracle.ref_analyze(clip, W, H, TRLs, bs, S, border_size=bord, block_overlaping=ov, subpixel_accuracy=sub)
    for t, (g, w) in enumerate(zip(got, want), 1):
        for k in ("motion", "motion_filtered", "high", "low"):
            np.testing.assert_array_equal(g[k], w[k], err_msg=f"{k}_{t}")
        assert g["frame_types"] == w["frame_types"], f"frame_types_{t}"
    low = want[-1]["low"]
    for t in range(len(want), 0, -1):  # the inverse chain: correlate --subpixel_accuracy (un_update ignores the flag)
        lv, St = want[t - 1], min(S * 2 ** (t - 1), 128)
        a = oracle.synthesize_step(low, lv["high"], lv["motion_filtered"], lv["frame_types"], W, H, bs, St, block_overlaping=ov, subpixel_accuracy=sub)
        b = oracle.ref_synthesize_step(low, lv["high"], lv["motion_filtered"], lv["frame_types"], W, H, bs, St, t=t, block_overlaping=ov,
                                       subpixel_accuracy=sub)
        np.testing.assert_array_equal(a["low"], b["low"], err_msg=f"synth_low_{t - 1}")
        low = b["low"]
