"""GPU: the predict and update steps on RANDOM geometries and ARBITRARY motion fields against the restatement (the committed goldens
and the fresh clips use estimated motion, which is smooth; here every block has its own random vectors: predictions leave the
picture on every side, update footprints overlap, sources fold onto the border).  tools/stress_predict.py / stress_update.py are the
long-running forms of the same checks."""
import numpy as np
import pytest

from qsvc_amd.synth import make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from qsvc_amd import hip
    e = hip.Engine(0)
    yield e
    e.close()


def _geometry(rng):
    while True:
        bs = int(rng.choice([8, 16, 32]))
        W, H = bs * int(rng.integers(2, 21)), bs * int(rng.integers(1, 13))
        if W % 8 == 0 and W >= 16 and H >= 16:
            return bs, W, H


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_predict_on_random_fields(eng, oracle, seed):
    from qsvc_amd import hip
    rng = np.random.default_rng(seed)
    for case in range(12):
        bs, W, H = _geometry(rng)
        S = int(rng.choice([4, 8, 16]))
        nB = int(rng.integers(1, 4))
        P = 2 * nB + 1
        kind = str(rng.choice(["pan_noise", "noise", "edge", "binary"]))
        low = make_clip(kind, P, W, H, int(rng.integers(1, 1000)))
        mv = rng.integers(-S, S + 1, size=(nB, 4, H // bs, W // bs), dtype=np.int16)
        always_B = int(rng.random() < 0.5)
        want = oracle.decorrelate(low, W, H, bs, S, mv, always_B, want_hists=True)
        eng.seq_config(W, H, True)
        eng.put_frames(hip.LOW, 0, low)
        eng.put_motion(hip.MOTION_RAW, 1, mv, bs)
        eng.decorrelate(1, P, bs, S, always_B)
        tag = f"case {case}: {kind} {W}x{H} bs {bs} S {S} nB {nB} always_B {always_B}"
        assert bytes(eng.get_frame_types(1, nB)) == want["types"], tag
        np.testing.assert_array_equal(eng.get_frames(hip.HIGH, 1, 0, nB), want["high"], err_msg=tag)
        np.testing.assert_array_equal(eng.get_motion(hip.MOTION_FILTERED, 1, nB, bs), want["mv"], err_msg=tag)
        if not always_B:  # (with always_B the reference computes no entropies: decorrelate.cpp:945-979)
            np.testing.assert_array_equal(eng.get_histograms(1, nB).astype(np.int64), np.asarray(want["hists"]).astype(np.int64), err_msg=tag)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_update_on_random_fields_both_forms(eng, oracle, seed):
    from qsvc_amd import hip
    rng = np.random.default_rng(seed)
    try:
        for case in range(12):
            bs, W, H = _geometry(rng)
            R = int(rng.choice([1, 2, 4, 7, 16, 33, 60, 100]))
            nB = int(rng.integers(1, 4))
            P = 2 * nB + 1
            F = W * H * 3 // 2
            low = rng.integers(0, 256, size=(P, F), dtype=np.uint8)
            high = rng.integers(100, 156, size=(nB, F), dtype=np.uint8)
            high[:, ::5] = rng.integers(0, 256, size=high[:, ::5].shape, dtype=np.uint8)
            mv = rng.integers(-R, R + 1, size=(nB, 4, H // bs, W // bs), dtype=np.int16)
            types = bytes(rng.choice([ord("B"), ord("B"), ord("I")], size=nB).astype(np.uint8))
            uf = float(rng.choice([0.25, 0.25, 0.5, 0.3]))
            want = oracle.update(low, W, H, bs, high, mv, types, uf)
            for form in (1, 0):
                eng.set_tuning("upd_scatter", form)
                eng.seq_config(W, H, True)
                eng.put_frames(hip.LOW, 0, low)
                eng.put_frames(hip.HIGH, 1, high)
                eng.put_motion(hip.MOTION_FILTERED, 1, mv, bs)
                eng.put_frame_types(1, types)
                eng.update(1, P, bs, uf)
                np.testing.assert_array_equal(eng.get_frames(hip.LOW, 1, 0, nB + 1), want,
                                              err_msg=f"case {case}: {W}x{H} bs {bs} R {R} nB {nB} types {types} uf {uf} upd_scatter={form}")
    finally:
        eng.set_tuning("upd_scatter", 1)


@pytest.mark.parametrize("seed", [31, 32])
def test_motion_estimate_on_random_geometries(eng, oracle, seed):
    """The hierarchical search (pyramid, piece / ring kernels, literal replay at odd pyramid sizes, stand-alone kernels where the block
    grid does not tile the picture) on random sizes, block sizes and search ranges."""
    from qsvc_amd import hip
    rng = np.random.default_rng(seed)
    for case in range(10):
        bs = int(rng.choice([8, 16, 32]))
        W, H = 8 * int(rng.integers(4, 40)), 8 * int(rng.integers(4, 24))
        if W < bs or H < bs:
            continue
        S = int(rng.choice([1, 2, 4, 8, 16]))
        nB = int(rng.integers(1, 3))
        P = 2 * nB + 1
        kind = str(rng.choice(["pan_noise", "edge", "scene_cut", "binary"]))
        low = make_clip(kind, P, W, H, int(rng.integers(1, 1000)))
        want = oracle.motion_estimate(low, W, H, bs, S)
        eng.seq_config(W, H, True)
        eng.put_frames(hip.LOW, 0, low)
        try:
            eng.motion_estimate(1, P, bs, S)
        except hip.QsvcError as ex:   # a geometry the path refuses loudly is not a parity failure
            assert "refus" in str(ex).lower() or "unsupported" in str(ex).lower() or "not " in str(ex).lower(), str(ex)
            continue
        np.testing.assert_array_equal(eng.get_motion(hip.MOTION_RAW, 1, nB, bs), want, err_msg=f"case {case}: {kind} {W}x{H} bs {bs} S {S} nB {nB}")
