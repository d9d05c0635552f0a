"""The CPU restatement (oracle/) against the committed fixtures produced by the reference's own
compiled tools (tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from helpers import GOLDEN, check_levels_against_golden, golden_cases, golden_input, load_golden, sha


@pytest.mark.parametrize("name", golden_cases())
def test_mctf_restatement_matches_reference_goldens(oracle, name):
    z, m = load_golden(name)
    clip = golden_input(z, m)
    levels = []
    low, bs, S = clip, m["bs"], m["S"]
    for t in range(1, m["TRLs"]):
        lv = oracle.analyze_step(low, m["W"], m["H"], bs, S, want_prediction=True)
        levels.append(lv)
        low, S = lv["low"], min(2 * S, 128)
    check_levels_against_golden(z, levels, what=("motion", "motion_filtered", "frame_types", "high", "low", "prediction"))
    # inverse chain (un_update, correlate, merge of the reference; synthesize.py), each level fed with the reconstruction
    S_t = [min(m["S"] * 2 ** k, 128) for k in range(len(levels))]
    for t in range(len(levels), 0, -1):
        lv = levels[t - 1]
        low = oracle.synthesize_step(low, lv["high"], lv["motion_filtered"], lv["frame_types"], m["W"], m["H"], bs, S_t[t - 1])["low"]
        assert sha(low) == str(z[f"synth_low_{t - 1}_sha"]), f"synth_low_{t - 1}"
        if f"synth_low_{t - 1}" in z.files:
            np.testing.assert_array_equal(low, z[f"synth_low_{t - 1}"])


def test_dwt_known_answers(oracle):
    import os
    z = np.load(os.path.join(GOLDEN, "dwt_kat.npz"))
    n = 0
    for k in z.files:
        if not k.startswith("in_"):
            continue
        tag = k[3:]
        lv = int(tag.split("_")[1])
        x = z[k]
        np.testing.assert_array_equal(oracle.dwt53t(x, lv), z[f"a53_{tag}"], err_msg=f"analyze {tag}")
        np.testing.assert_array_equal(oracle.dwt53t(x, lv, inverse=True), z[f"s53_{tag}"], err_msg=f"synthesize {tag}")
        np.testing.assert_array_equal(oracle.haar(x, lv, inverse=True), z[f"shaar_{tag}"], err_msg=f"haar {tag}")
        n += 1
    assert n >= 10


def test_dwt53t_perfect_reconstruction(oracle):
    rng = np.random.default_rng(3)
    for (h, w, lv) in [(16, 16, 2), (24, 40, 3), (64, 48, 3)]:
        x = rng.integers(-300, 300, size=(h, w)).astype(np.int16)
        np.testing.assert_array_equal(oracle.dwt53t(oracle.dwt53t(x, lv), lv, inverse=True), x)


def test_entropy_and_frame_type_rule(oracle):
    import ctypes as C
    L = oracle.lib()
    h = np.zeros(256, dtype=np.int32)
    h[10] = 5
    assert L.qo_entropy(h, 256) == 0.0
    h[20] = 5
    assert L.qo_entropy(h, 256) == 1.0
    # predicted_size <= residue_size + motion_size  ->  'I' (decorrelate.cpp:979)
    flat = np.full(256, 4, dtype=np.int32)
    one = np.zeros(256, dtype=np.int32); one[128] = 1024
    assert L.qo_frame_type(flat, one, one, 32, 32, 2, 2, 0, None) == ord("B")
    assert L.qo_frame_type(one, flat, one, 32, 32, 2, 2, 0, None) == ord("I")
    assert L.qo_frame_type(one, one, one, 32, 32, 2, 2, 0, None) == ord("I")   # tie -> I
    assert L.qo_frame_type(one, flat, one, 32, 32, 2, 2, 1, None) == ord("B")  # always_B


def test_restatement_matches_reference_digests_at_1080p(oracle):
    """BASELINE config 3 geometry (1920x1080, block 8, S = 4 -> 8 -> 16): all three levels of the restatement against the digests
    the compiled reference tools produced (tests/golden/fullsize_digests.json).  ~1 min of CPU."""
    import json, os
    from helpers import GOLDEN, sha
    from qsvc_amd.synth import make_clip
    c = json.load(open(os.path.join(GOLDEN, "fullsize_digests.json")))["cfg3_pan_noise"]
    clip = make_clip(c["kind"], c["frames"], c["W"], c["H"], c["seed"])
    assert sha(clip) == c["input_sha"]
    levels = oracle.analyze(clip, c["W"], c["H"], c["TRLs"], c["block_size"], c["search_range"])
    for lv, want in zip(levels, c["levels"]):
        assert lv["frame_types"].decode() == want["frame_types"]
        for k in ("motion", "motion_filtered", "high", "low"):
            assert sha(lv[k]) == want[k], f"{k}_{want['t']}"
