"""The kernels' per-thread code (qsvc_amd/csrc/mctf_threads.h), driven by host loops (tests/emul), against the goldens the
reference's own tools produced -- indexing and arithmetic of the HIP path checked where there is no GPU.  Includes the
pictures the block grid does not tile (Q7: stale prediction margins) which the GPU tests repeat on the device."""
import numpy as np
import pytest

from helpers import golden_input, load_golden, sha

CASES = ["tiny_pan", "tiny_sat", "tiny_binary", "q7_200x120", "s2_tiny_edge", "s1_sat_q7_200x120", "odd_350x286"]


@pytest.fixture(scope="module")
def E():
    from emul import emul
    return emul


@pytest.mark.parametrize("name", CASES)
def test_thread_code_reproduces_golden_chain(E, name):
    z, m = load_golden(name)
    low = golden_input(z, m)
    W, H, bs, S = m["W"], m["H"], m["bs"], m["S"]
    for t in range(1, m["TRLs"]):
        r, mv = E.motion_estimate(low, W, H, bs, S)
        assert r == 0
        np.testing.assert_array_equal(mv, z[f"motion_{t}"], err_msg=f"motion_{t}")
        r, high, pred, hist = E.predict(low, W, H, bs, S, mv)
        assert r == 0
        types = z[f"frame_types_{t}"].tobytes()
        filt = z[f"motion_filtered_{t}"]
        for i, ty in enumerate(types):
            if ty == ord("I"):
                high[i] = low[2 * i + 1]
        assert sha(pred) == str(z[f"prediction_{t}_sha"]), f"prediction_{t}"
        assert sha(high) == str(z[f"high_{t}_sha"]), f"high_{t}"
        r, lo = E.update(low, W, H, bs, high, filt, types)
        assert r == 0
        assert sha(lo) == str(z[f"low_{t}_sha"]), f"low_{t}"
        low, S = lo, min(2 * S, 128)


RAGGED = [("pan_noise", 18, 9, 72, 56, 16, 4), ("edge", 19, 9, 120, 68, 32, 4), ("scene_cut", 20, 7, 240, 136, 16, 4),
          ("saturating", 21, 5, 80, 48, 32, 8)]


@pytest.mark.parametrize("kind,seed,frames,W,H,bs,S", RAGGED)
def test_ragged_pictures_equal_restatement(E, oracle, kind, seed, frames, W, H, bs, S):
    """Q7 beyond the golden: more pictures (deeper margin recurrence), both margins, margins wider than a block."""
    from qsvc_amd.synth import make_clip
    low = make_clip(kind, frames, W, H, seed)
    want = oracle.analyze_step(low, W, H, bs, S, want_prediction=True)
    r, mv = E.motion_estimate(low, W, H, bs, S)
    assert r == 0
    np.testing.assert_array_equal(mv, want["motion"])
    r, high, pred, hist = E.predict(low, W, H, bs, S, mv)
    assert r == 0
    np.testing.assert_array_equal(pred, want["prediction"])
    for i, ty in enumerate(want["frame_types"]):
        if ty == ord("I"):
            high[i] = low[2 * i + 1]
    np.testing.assert_array_equal(high, want["high"])
    r, lo = E.update(low, W, H, bs, high, want["motion_filtered"], want["frame_types"])
    assert r == 0
    np.testing.assert_array_equal(lo, want["low"])


SYNTH = [("pan_noise", 12, 5, 128, 96, 16, 4), ("saturating", 14, 5, 128, 128, 8, 4), ("binary", 48, 5, 64, 64, 16, 4),
         ("pan_noise", 18, 9, 72, 56, 16, 4), ("edge", 17, 5, 120, 72, 8, 16)]


@pytest.mark.parametrize("kind,seed,frames,W,H,bs,S", SYNTH)
def test_inverse_level_thread_code_equals_restatement(E, oracle, kind, seed, frames, W, H, bs, S):
    """un_update + correlate + merge (synthesize_step.py) through the kernels' thread code, incl. 'I' pictures and Q7."""
    from qsvc_amd.synth import make_clip
    clip = make_clip(kind, frames, W, H, seed)
    a = oracle.analyze_step(clip, W, H, bs, S)
    want = oracle.synthesize_step(a["low"], a["high"], a["motion_filtered"], a["frame_types"], W, H, bs, S)
    r, got = E.synthesize_step(a["low"], W, H, bs, S, a["high"], a["motion_filtered"], a["frame_types"])
    assert r == 0
    np.testing.assert_array_equal(got[0::2], want["even"])
    np.testing.assert_array_equal(got[1::2], want["odd"])
