"""GPU parity: the HIP path (through the C ABI) against the oracle and the committed reference goldens."""
import numpy as np
import pytest

from helpers import check_levels_against_golden, golden_cases, golden_input, load_golden
from qsvc_amd.synth import make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from qsvc_amd import hip
    e = hip.Engine(0)
    yield e
    e.close()


def run_levels(eng, clip, W, H, TRLs, bs, S, always_B=0, uf=0.25, keep_prediction=True):
    from qsvc_amd import hip
    eng.seq_config(W, H, keep_prediction)
    eng.put_frames(hip.LOW, 0, clip)
    P = clip.shape[0]
    out = []
    for t in range(1, TRLs):
        nB = P // 2
        eng.motion_estimate(t, P, bs, S)
        eng.decorrelate(t, P, bs, S, always_B)
        eng.update(t, P, bs, uf)
        out.append(dict(motion=eng.get_motion(hip.MOTION_RAW, t, nB, bs), motion_filtered=eng.get_motion(hip.MOTION_FILTERED, t, nB, bs),
                        frame_types=eng.get_frame_types(t, nB), high=eng.get_frames(hip.HIGH, t, 0, nB),
                        low=eng.get_frames(hip.LOW, t, 0, nB + 1),
                        prediction=eng.get_frames(hip.PRED, t, 0, nB) if keep_prediction else None,
                        hists=eng.get_histograms(t, nB)))
        P = (P + 1) // 2
        S = min(2 * S, 128)
    return out


# every chain the reference's tools produced: incl. q7_200x120 (pictures the block grid does not tile), s1_* / s2_* (search
# ranges 1 and 2: no pyramid) and odd_350x286 / q5_* (an odd size along the search pyramid: the literal replay, mctf_replay.hip)
SUPPORTED = golden_cases()


@pytest.mark.parametrize("name", SUPPORTED)
def test_hip_matches_reference_goldens(eng, name):
    z, m = load_golden(name)
    clip = golden_input(z, m)
    levels = run_levels(eng, clip, m["W"], m["H"], m["TRLs"], m["bs"], m["S"])
    check_levels_against_golden(z, levels, what=("motion", "motion_filtered", "frame_types", "high", "low", "prediction"))


@pytest.mark.parametrize("name", ["sat_360x200", "deep_pan_256x192", "cif_binary", "cif_pan", "s1_cif_pan", "s2_tiny_edge",
                                  "q5_edge_360x200_s32"])
def test_chained_levels_with_deferred_decisions(eng, name):
    """The whole level chain queued without reading anything in between (what `mctf compress` does): decorrelate defers its I/B
    decisions and the chain runs on as if every picture were 'B'; the first read takes the decisions and, where a picture is 'I'
    (sat_360x200: level 3; deep_pan: levels 5, 6 and 7 one after the other), replays what consumed the speculative state.
    Everything must equal the reference's files as in the level-by-level test."""
    from qsvc_amd import hip
    z, m = load_golden(name)
    clip = golden_input(z, m)
    W, H, bs = m["W"], m["H"], m["bs"]
    eng.seq_config(W, H, True)
    eng.put_frames(hip.LOW, 0, clip)
    P, S, sched = clip.shape[0], m["S"], []
    for t in range(1, m["TRLs"]):
        eng.motion_estimate(t, P, bs, S)
        eng.decorrelate(t, P, bs, S, 0)
        eng.update(t, P, bs, 0.25)
        sched.append((t, P // 2))
        P, S = (P + 1) // 2, min(2 * S, 128)
    levels = [dict(motion=eng.get_motion(hip.MOTION_RAW, t, nB, bs), motion_filtered=eng.get_motion(hip.MOTION_FILTERED, t, nB, bs),
                   frame_types=eng.get_frame_types(t, nB), high=eng.get_frames(hip.HIGH, t, 0, nB), low=eng.get_frames(hip.LOW, t, 0, nB + 1),
                   prediction=eng.get_frames(hip.PRED, t, 0, nB)) for (t, nB) in sched]
    check_levels_against_golden(z, levels, what=("motion", "motion_filtered", "frame_types", "high", "low", "prediction"))


@pytest.mark.parametrize("name", SUPPORTED)
def test_hip_inverse_chain_matches_reference_goldens(eng, name):
    """un_update + correlate (+ merge = the layout) down from the coarsest level, on the device-resident analysis outputs:
    every reconstructed low_{t-1} == what the reference's own inverse tools produced (digests in the goldens)."""
    from helpers import sha
    from qsvc_amd import hip
    z, m = load_golden(name)
    clip = golden_input(z, m)
    W, H, bs, TRLs = m["W"], m["H"], m["bs"], m["TRLs"]
    run_levels(eng, clip, W, H, TRLs, bs, m["S"])
    P_t, S_t, P, S = [], [], clip.shape[0], m["S"]
    for t in range(1, TRLs):
        P_t.append(P); S_t.append(S)
        P, S = (P + 1) // 2, min(2 * S, 128)
    for t in range(TRLs - 1, 0, -1):
        eng.synthesize_step(t, P_t[t - 1], bs, S_t[t - 1])
        low = eng.get_frames(hip.LOW, t - 1, 0, P_t[t - 1])
        assert sha(low) == str(z[f"synth_low_{t - 1}_sha"]), f"synth_low_{t - 1}"
        if f"synth_low_{t - 1}" in z.files:
            np.testing.assert_array_equal(low, z[f"synth_low_{t - 1}"])


SYNTH = [("pan_noise", 31, 9, 256, 160, 3, 16, 4), ("binary", 32, 5, 128, 96, 2, 16, 4), ("saturating", 33, 9, 192, 128, 3, 8, 4),
         ("edge", 34, 9, 120, 68, 2, 32, 4), ("pan_noise", 35, 9, 200, 120, 3, 64, 4)]


@pytest.mark.parametrize("kind,seed,frames,W,H,TRLs,bs,S", SYNTH)
def test_hip_inverse_level_matches_oracle_from_host_inputs(eng, oracle, kind, seed, frames, W, H, TRLs, bs, S):
    """The decoder's situation: subband pictures, motion and frame types arrive from the host (put_*), one level back."""
    from qsvc_amd import hip
    clip = make_clip(kind, frames, W, H, seed)
    a = oracle.analyze_step(clip, W, H, bs, S)
    want = oracle.synthesize_step(a["low"], a["high"], a["motion_filtered"], a["frame_types"], W, H, bs, S)
    nB = frames // 2
    eng.seq_config(W, H, True)
    eng.put_frames(hip.LOW, 1, a["low"])
    eng.put_frames(hip.HIGH, 1, a["high"])
    eng.put_motion(hip.MOTION_FILTERED, 1, a["motion_filtered"], bs)
    eng.put_frame_types(1, a["frame_types"])
    eng.un_update(1, frames, bs)
    np.testing.assert_array_equal(eng.get_frames(hip.LOW, 0, 0, frames)[0::2], want["even"])
    eng.correlate(1, frames, bs, S)
    got = eng.get_frames(hip.LOW, 0, 0, frames)
    np.testing.assert_array_equal(got[0::2], want["even"])
    np.testing.assert_array_equal(got[1::2], want["odd"])


@pytest.mark.parametrize("W,H,bs,R,seed", [(256, 160, 8, 4, 1), (256, 160, 8, 16, 2), (192, 128, 16, 12, 3), (128, 96, 8, 40, 4), (200, 120, 8, 7, 5)])
def test_update_with_random_fields_both_forms(eng, oracle, W, H, bs, R, seed):
    """The update step alone on ARBITRARY motion (every block its own random vector up to R: footprints overlap everywhere, the
    ordered clamp-per-addition of update.cpp:85-146 matters at most targets) and random residues, from host inputs: the scatter
    form (LDS images + ordered replay of the clashes), the gather form (QSVC_UPD_SCATTER=0) and -- reach 40: more than 255 blocks
    within reach -- the scatter form's own fall-back all equal the restatement."""
    from qsvc_amd import hip
    rng = np.random.default_rng(seed)
    P, nB = 5, 2
    F = W * H * 3 // 2
    low = rng.integers(0, 256, size=(P, F), dtype=np.uint8)
    high = rng.integers(96, 160, size=(nB, F), dtype=np.uint8)
    high[:, ::7] = rng.integers(0, 256, size=high[:, ::7].shape, dtype=np.uint8)   # some residues that saturate
    mv = rng.integers(-R, R + 1, size=(nB, 4, H // bs, W // bs), dtype=np.int16)
    mv[:, :, ::3, ::2] = 0   # ... and stretches of equal vectors between the discontinuities
    types = b"BB"
    want = oracle.update(low, W, H, bs, high, mv, types)
    got = {}
    for form in (1, 0):
        eng.set_tuning("upd_scatter", form)
        try:
            eng.seq_config(W, H, True)
            eng.put_frames(hip.LOW, 0, low)
            eng.put_frames(hip.HIGH, 1, high)
            eng.put_motion(hip.MOTION_FILTERED, 1, mv, bs)
            eng.put_frame_types(1, types)
            eng.update(1, P, bs)
            got[form] = eng.get_frames(hip.LOW, 1, 0, nB + 1)
        finally:
            eng.set_tuning("upd_scatter", 1)
        np.testing.assert_array_equal(got[form], want, err_msg=f"upd_scatter={form}")


def test_inverse_level_needs_its_inputs(eng):
    from qsvc_amd import hip
    eng.seq_config(64, 64)
    eng.put_frames(hip.LOW, 2, np.zeros((2, 64 * 64 * 3 // 2), dtype=np.uint8))
    with pytest.raises(hip.QsvcError):
        eng.un_update(2, 5, 16)  # needs 3 low pictures, high_2, motion_2, frame_types_2


def test_unsupported_geometries_fail_loudly(eng):
    """What the GPU path still refuses: a search range that is not a power of two (the reference's rint(log2 S) pyramid depth with a
    clamp that can fire), odd picture sizes (4:2:0), and a pyramid deeper than the picture allows."""
    from qsvc_amd import hip
    clip = make_clip("pan", 3, 64, 48, 1)
    eng.seq_config(64, 48)
    eng.put_frames(hip.LOW, 0, clip)
    with pytest.raises(hip.QsvcError):
        eng.motion_estimate(1, 3, 16, 6)
    with pytest.raises(hip.QsvcError):
        eng.seq_config(63, 48)
    eng.seq_config(72, 18)          # 18 >> 5 = 0: no room for the six levels of search range 128
    eng.put_frames(hip.LOW, 0, make_clip("pan", 3, 72, 18, 1))
    with pytest.raises(hip.QsvcError):
        eng.motion_estimate(1, 3, 8, 128)


CASES = [
    ("pan_noise", 21, 9, 256, 160, 3, 16, 4),
    ("edge", 22, 9, 320, 192, 4, 8, 4),
    ("saturating", 23, 5, 192, 128, 3, 8, 8),
    ("noise", 24, 5, 128, 128, 2, 32, 16),
    ("scene_cut", 25, 9, 352, 288, 4, 32, 4),
    ("binary", 26, 5, 128, 96, 2, 16, 4),
    # Q7: ragged pictures -- stale prediction margins, several pictures deep, over more than one level
    ("pan_noise", 18, 9, 72, 56, 3, 16, 4),
    ("edge", 19, 9, 120, 68, 2, 32, 4),
    ("scene_cut", 20, 9, 240, 136, 3, 16, 4),
    ("saturating", 21, 5, 80, 48, 2, 32, 8),
    ("pan_noise", 27, 9, 200, 120, 3, 64, 4),  # one block row: margin taller than the predicted region's half
]


@pytest.mark.parametrize("kind,seed,frames,W,H,TRLs,bs,S", CASES)
def test_hip_matches_oracle_fresh_clips(eng, oracle, kind, seed, frames, W, H, TRLs, bs, S):
    clip = make_clip(kind, frames, W, H, seed)
    got = run_levels(eng, clip, W, H, TRLs, bs, S)
    want = oracle.analyze(clip, W, H, TRLs, bs, S)
    for t, (g, w) in enumerate(zip(got, want), 1):
        np.testing.assert_array_equal(g["motion"], w["motion"], err_msg=f"motion_{t}")
        assert g["frame_types"] == w["frame_types"], f"frame_types_{t}"
        np.testing.assert_array_equal(g["motion_filtered"], w["motion_filtered"], err_msg=f"motion_filtered_{t}")
        np.testing.assert_array_equal(g["hists"].astype(np.int32), w["hists"], err_msg=f"hist_{t}")
        np.testing.assert_array_equal(g["high"], w["high"], err_msg=f"high_{t}")
        np.testing.assert_array_equal(g["low"], w["low"], err_msg=f"low_{t}")


def test_always_B_and_update_factor(eng, oracle):
    W, H, bs, S = 128, 96, 16, 4
    clip = make_clip("binary", 5, W, H, 31)
    got = run_levels(eng, clip, W, H, 2, bs, S, always_B=1, uf=0.5)[0]
    want = oracle.analyze_step(clip, W, H, bs, S, always_B=1, update_factor=0.5)
    assert got["frame_types"] == b"BB" == want["frame_types"]
    np.testing.assert_array_equal(got["high"], want["high"])
    np.testing.assert_array_equal(got["low"], want["low"])
    # a non-dyadic factor exercises the separately-rounded float multiply/add of update.cpp:99-106
    got = run_levels(eng, clip, W, H, 2, bs, S, always_B=1, uf=0.3)[0]
    want = oracle.analyze_step(clip, W, H, bs, S, always_B=1, update_factor=0.3)
    np.testing.assert_array_equal(got["low"], want["low"])


def test_dwt53_trunc_kernel(eng, oracle):
    rng = np.random.default_rng(9)
    for (h, w, lv) in [(16, 16, 2), (24, 40, 3), (9, 8, 1), (17, 33, 1), (270, 480, 2)]:
        x = rng.integers(-300, 300, size=(h, w)).astype(np.int16)
        if (h >> (lv - 1)) % 2 or (w >> (lv - 1)) % 2:
            lv = 1
        a = eng.dwt53_trunc(x, lv)
        np.testing.assert_array_equal(a, oracle.dwt53t(x, lv))
        np.testing.assert_array_equal(eng.dwt53_trunc(a, lv, inverse=True), oracle.dwt53t(a, lv, inverse=True))


def test_1080p_level_against_oracle(eng, oracle):
    """BASELINE config 3 geometry (1920x1080, bs 8), one level of 5 pictures: full-size bit-exactness."""
    W, H, bs, S = 1920, 1080, 8, 4
    clip = make_clip("pan_noise", 5, W, H, 43)
    got = run_levels(eng, clip, W, H, 2, bs, S, keep_prediction=False)[0]
    want = oracle.analyze_step(clip, W, H, bs, S)
    np.testing.assert_array_equal(got["motion"], want["motion"])
    assert got["frame_types"] == want["frame_types"]
    np.testing.assert_array_equal(got["high"], want["high"])
    np.testing.assert_array_equal(got["low"], want["low"])


def test_halo_split_equals_whole_sequence(eng):
    """Open-GOP sharding: two halves with exchanged halos == the whole sequence in one piece."""
    import ctypes as C
    from qsvc_amd import hip
    W, H, bs, S = 192, 128, 16, 4
    clip = make_clip("pan_noise", 9, W, H, 77)
    whole = run_levels(eng, clip, W, H, 2, bs, S, keep_prediction=False)[0]
    L = hip.load_library()
    # left half: pictures 0..4, right half: 4..8, on two contexts of the same GPU
    e2 = hip.Engine(0)
    try:
        halves = []
        for e, part in ((eng, clip[0:5]), (e2, clip[4:9])):
            e.seq_config(W, H)
            e.put_frames(hip.LOW, 0, part)
            e.motion_estimate(1, 5, bs, S)
            e.decorrelate(1, 5, bs, S)
        to_right = eng.halo_export(1, 1, bs)   # left shard's last B picture
        to_left = e2.halo_export(1, 0, bs)     # right shard's first B picture
        eng.halo_import(1, 1, to_left)
        e2.halo_import(1, 0, to_right)
        eng.update(1, 5, bs, 0.25)
        e2.update(1, 5, bs, 0.25)
        left = eng.get_frames(hip.LOW, 1, 0, 3)
        right = e2.get_frames(hip.LOW, 1, 0, 3)
        eng.halo_import(1, 1, None)
        np.testing.assert_array_equal(left[2], right[0], err_msg="boundary picture differs between the two shards")
        np.testing.assert_array_equal(np.concatenate([left, right[1:]]), whole["low"])
    finally:
        e2.close()


def test_halo_three_shards_the_middle_one_imports_on_both_sides(eng):
    """Six of the eight ranks of a node have a neighbour on either side: a middle shard exports its first and its last B picture's
    record and takes in one from each side before its update.  Three shards of one GPU, two temporal levels == the whole sequence."""
    from qsvc_amd import hip
    W, H, bs, S = 192, 128, 16, 4
    clip = make_clip("pan_noise", 25, W, H, 78)
    whole = run_levels(eng, clip, W, H, 3, bs, S, keep_prediction=False)
    es = [eng, hip.Engine(0), hip.Engine(0)]
    try:
        for k, e in enumerate(es):
            e.seq_config(W, H)
            e.put_frames(hip.LOW, 0, clip[8 * k:8 * k + 9])
        for t, n in ((1, 9), (2, 5)):                # level t analyses n pictures of every shard
            for e in es:
                e.motion_estimate(t, n, bs, S << (t - 1))
                e.decorrelate(t, n, bs, S << (t - 1))
            rec = [(e.halo_export(t, 0, bs), e.halo_export(t, 1, bs)) for e in es]   # (to the left, to the right)
            for k, e in enumerate(es):
                e.halo_import(t, 0, rec[k - 1][1] if k > 0 else None)
                e.halo_import(t, 1, rec[k + 1][0] if k < 2 else None)
                e.update(t, n, bs, 0.25)
            lows = [e.get_frames(hip.LOW, t, 0, (n + 1) // 2) for e in es]
            highs = [e.get_frames(hip.HIGH, t, 0, n // 2) for e in es]
            np.testing.assert_array_equal(lows[0][-1], lows[1][0], err_msg=f"level {t}: boundary picture 0 | 1")
            np.testing.assert_array_equal(lows[1][-1], lows[2][0], err_msg=f"level {t}: boundary picture 1 | 2")
            np.testing.assert_array_equal(np.concatenate([lows[0], lows[1][1:], lows[2][1:]]), whole[t - 1]["low"])
            np.testing.assert_array_equal(np.concatenate(highs), whole[t - 1]["high"])
        for e in es:
            e.halo_import(1, 0, None); e.halo_import(1, 1, None); e.halo_import(2, 0, None); e.halo_import(2, 1, None)
    finally:
        for e in es[1:]:
            e.close()


def test_degenerate_picture_counts(eng, oracle):
    """pictures = 1 (no B picture: low_t is a copy), an even count (the last even picture is missing: refused, never
    read out of bounds), and the minimum 3."""
    from qsvc_amd import hip
    W, H, bs, S = 64, 64, 16, 4
    clip = make_clip("pan", 4, W, H, 2)
    eng.seq_config(W, H)
    eng.put_frames(hip.LOW, 0, clip[:1])
    eng.analyze_step(1, 1, bs, S)
    np.testing.assert_array_equal(eng.get_frames(hip.LOW, 1, 0, 1), clip[:1])
    eng.put_frames(hip.LOW, 0, clip)            # 4 pictures: B picture 3 would need even picture 4
    with pytest.raises(hip.QsvcError):
        eng.motion_estimate(1, 4, bs, S)
    eng.put_frames(hip.LOW, 0, clip[:3])
    eng.analyze_step(1, 3, bs, S)
    want = oracle.analyze_step(clip[:3], W, H, bs, S)
    np.testing.assert_array_equal(eng.get_frames(hip.LOW, 1, 0, 2), want["low"])
    np.testing.assert_array_equal(eng.get_frames(hip.HIGH, 1, 0, 1), want["high"])


def test_getters_check_the_callers_capacity():
    """qsvc_get_motion / _frame_types / _histograms copy what the CALLER sized its buffer for, and refuse (QSVC_ERR_STATE) when
    that is more than -- or laid out differently from -- what is resident."""
    from qsvc_amd import hip
    from qsvc_amd.synth import make_clip
    W, H, bs = 128, 96, 16
    e = hip.Engine(0)
    try:
        e.seq_config(W, H)
        e.put_frames(hip.LOW, 0, make_clip("pan", 5, W, H, 1))
        e.analyze_step(1, 5, bs, 4)
        assert e.get_motion(hip.MOTION_RAW, 1, 1, bs).shape == (1, 4, H // bs, W // bs)   # fewer than resident: fine
        assert len(e.get_frame_types(1, 1)) == 1 and e.get_histograms(1, 2).shape == (2, 3, 256)
        for call in (lambda: e.get_motion(hip.MOTION_RAW, 1, 3, bs), lambda: e.get_motion(hip.MOTION_FILTERED, 1, 2, bs // 2),
                     lambda: e.get_frame_types(1, 3), lambda: e.get_histograms(1, 3)):
            with pytest.raises(hip.QsvcError) as ei:
                call()
            assert ei.value.code == -4
        with pytest.raises(hip.QsvcError):
            e._chk(e.L.qsvc_put_frame_types(e.h, 1, -1, b"B"))
    finally:
        e.close()
