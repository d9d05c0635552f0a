"""SURVEY.md 8f N4 -- the reference's optional MCTF flags, pinned by goldens the compiled reference tools produced
(tests/golden/n4_*.npz): --block_overlaping (decorrelate.cpp:84-172), --border_size (motion_estimate.cpp:213-217,714) and
--subpixel_accuracy (motion_estimate.cpp:361-412, decorrelate.cpp:656-686,828-861).
CPU: the restatement; GPU: the HIP path through the C ABI (forward levels and the inverse chain)."""
import numpy as np
import pytest

from helpers import check_levels_against_golden, golden_input, load_n4, n4_cases, sha


@pytest.mark.parametrize("name", n4_cases())
def test_restatement_matches_reference_goldens(oracle, name):
    z, m = load_n4(name)
    clip = golden_input(z, m)
    levels = oracle.analyze(clip, m["W"], m["H"], m["TRLs"], m["bs"], m["S"], block_overlaping=m["ov"], border_size=m["bord"],
                            subpixel_accuracy=m["sub"])
    for t, lv in enumerate(levels, 1):
        d = oracle.decorrelate(clip if t == 1 else levels[t - 2]["low"], m["W"], m["H"], m["bs"], min(m["S"] * 2 ** (t - 1), 128), lv["motion"],
                               want_prediction=True, block_overlaping=m["ov"], subpixel_accuracy=m["sub"])
        lv["prediction"] = d["prediction"]
    check_levels_against_golden(z, levels, what=("motion", "motion_filtered", "frame_types", "high", "low", "prediction"))
    low = levels[-1]["low"]
    for t in range(len(levels), 0, -1):
        lv = levels[t - 1]
        low = oracle.synthesize_step(low, lv["high"], lv["motion_filtered"], lv["frame_types"], m["W"], m["H"], m["bs"],
                                     min(m["S"] * 2 ** (t - 1), 128), block_overlaping=m["ov"], subpixel_accuracy=m["sub"])["low"]
        assert sha(low) == str(z[f"synth_low_{t - 1}_sha"]), f"synth_low_{t - 1}"


GPU_REFUSED = set()   # (every golden chain runs on the GPU path, incl. sub-pixel accuracy on pictures the block grid does not tile)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in n4_cases() if n not in GPU_REFUSED])
def test_hip_matches_reference_goldens(name):
    from qsvc_amd import hip
    z, m = load_n4(name)
    clip = golden_input(z, m)
    W, H, bs = m["W"], m["H"], m["bs"]
    e = hip.Engine(0)
    try:
        e.seq_config(W, H, keep_prediction=True)
        e.set_mctf_options(block_overlaping=m["ov"], border_size=m["bord"], subpixel_accuracy=m["sub"])
        e.put_frames(hip.LOW, 0, clip)
        P, S, levels = m["frames"], m["S"], []
        for t in range(1, m["TRLs"]):
            e.analyze_step(t, P, bs, S)
            nB = P // 2
            levels.append(dict(motion=e.get_motion(hip.MOTION_RAW, t, nB, bs), motion_filtered=e.get_motion(hip.MOTION_FILTERED, t, nB, bs),
                               frame_types=e.get_frame_types(t, nB), high=e.get_frames(hip.HIGH, t, 0, nB), low=e.get_frames(hip.LOW, t, 0, nB + 1),
                               prediction=e.get_frames(hip.PRED, t, 0, nB), P=P, S=S))
            P, S = (P + 1) // 2, min(2 * S, 128)
        check_levels_against_golden(z, levels, what=("motion", "motion_filtered", "frame_types", "high", "low", "prediction"))
        # inverse chain on the device (un_update + correlate with the same --block_overlaping)
        for t in range(len(levels), 0, -1):
            lv = levels[t - 1]
            e.synthesize_step(t, 2 * (lv["P"] // 2) + 1, bs, lv["S"])
            assert sha(e.get_frames(hip.LOW, t - 1, 0, 2 * (lv["P"] // 2) + 1)) == str(z[f"synth_low_{t - 1}_sha"]), f"synth_low_{t - 1}"
    finally:
        e.close()


@pytest.mark.gpu
def test_unsupported_options_are_refused():
    from qsvc_amd import hip
    e = hip.Engine(0)
    try:
        e.seq_config(128, 96)
        with pytest.raises(hip.QsvcError) as ei:
            e.set_mctf_options(subpixel_accuracy=4)
        assert ei.value.code == -2
        with pytest.raises(hip.QsvcError):
            e.set_mctf_options(block_overlaping=3)
        # sub-pixel accuracy: search range 1 (the reference's own motion_estimate crashes there) and vectors that leave the 256-bin
        # motion histogram
        from qsvc_amd.synth import make_clip
        e.set_mctf_options(subpixel_accuracy=2)
        e.put_frames(hip.LOW, 0, make_clip("pan", 3, 128, 96, 1))
        for S in (1, 32):
            with pytest.raises(hip.QsvcError) as ei:
                e.motion_estimate(1, 3, 16, S)
            assert ei.value.code == -2
    finally:
        e.close()


def test_subpixel_probe_record():
    """Why --subpixel_accuracy needs a model of the reference's heap.  The committed record of what the reference's own
    tools do with it (tests/golden/make_subpel_probe.py -> subpel_probe.json): repeatable under default malloc, but motion_t itself
    changes with the heap's fill pattern (MALLOC_PERTURB_) on every clip -- the search reads plane borders nothing ever wrote."""
    import json, os
    from helpers import GOLDEN
    rec = json.load(open(os.path.join(GOLDEN, "subpel_probe.json")))
    assert len(rec) >= 8
    for r in rec:
        assert r["exit_codes"] == [0, 0, 0, 0] and r["repeat_identical"]
        assert "motion_1" in r["files_depending_on_heap_fill"]
        assert r["malloc_perturb_165"]["motion_1"] != r["default_malloc"]["motion_1"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind,seed,frames,W,H,TRLs,bs,S,sub", [("pan_noise", 71, 9, 200, 120, 3, 32, 4, 1), ("edge", 72, 5, 120, 68, 2, 16, 4, 2),
                                                                ("saturating", 73, 9, 72, 56, 3, 16, 2, 1)])
def test_hip_subpixel_on_ragged_pictures_equals_restatement(oracle, kind, seed, frames, W, H, TRLs, bs, S, sub):
    """Sub-pixel accuracy on pictures the block grid does not tile (the reference's default block size at 1080p is such a case):
    the up-sampled prediction planes carry their in-place analysis state from picture to picture (Q7), several B pictures deep,
    forward and back."""
    from qsvc_amd import hip
    from qsvc_amd.synth import make_clip
    clip = make_clip(kind, frames, W, H, seed)
    want = oracle.analyze(clip, W, H, TRLs, bs, S, subpixel_accuracy=sub)
    e = hip.Engine(0)
    try:
        e.seq_config(W, H, keep_prediction=True)
        e.set_mctf_options(subpixel_accuracy=sub)
        e.put_frames(hip.LOW, 0, clip)
        P, St = frames, S
        for t, w in enumerate(want, 1):
            e.analyze_step(t, P, bs, St)
            nB = P // 2
            np.testing.assert_array_equal(e.get_motion(hip.MOTION_RAW, t, nB, bs), w["motion"], err_msg=f"motion_{t}")
            assert e.get_frame_types(t, nB) == w["frame_types"]
            np.testing.assert_array_equal(e.get_frames(hip.HIGH, t, 0, nB), w["high"], err_msg=f"high_{t}")
            np.testing.assert_array_equal(e.get_frames(hip.LOW, t, 0, nB + 1), w["low"], err_msg=f"low_{t}")
            P, St = (P + 1) // 2, min(2 * St, 128)
        low = want[-1]["low"]
        for t in range(len(want), 0, -1):
            w, Pt = want[t - 1], want[t - 1]["pictures"]
            low = oracle.synthesize_step(low, w["high"], w["motion_filtered"], w["frame_types"], W, H, bs, w["search_range"], subpixel_accuracy=sub)["low"]
            e.synthesize_step(t, 2 * (Pt // 2) + 1, bs, w["search_range"])
            np.testing.assert_array_equal(e.get_frames(hip.LOW, t - 1, 0, 2 * (Pt // 2) + 1), low, err_msg=f"synth_low_{t - 1}")
    finally:
        e.close()
