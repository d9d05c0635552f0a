"""GPU: the JPEG 2000 decoder (qsvc_j2k_decode_planes) through the C ABI against the restatement's decoder
(same arithmetic: identical samples), the original pictures (5/3: lossless) and an independent decoder (OpenJPEG: 9/7 within
1 LSB); code-streams of this path, of the restatement and of another encoder."""
import io
import os
import subprocess

import numpy as np
import pytest

from qsvc_amd.synth import make_clip

pytestmark = pytest.mark.gpu
J = pytest.importorskip("oracle.j2k")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from qsvc_amd import hip
    e = hip.Engine(0)
    yield e
    e.close()


def _luma(kind, W, H, seed):
    return make_clip(kind, 1, W, H, seed)[0][:W * H].reshape(H, W)


CASES = [("pan", 64, 64, 0), ("pan", 64, 64, 3), ("noise", 100, 75, 3), ("binary", 128, 96, 5), ("pan_noise", 352, 288, 4),
         ("edge", 37, 130, 2), ("saturating", 200, 120, 5), ("noise", 65, 129, 1), ("noise", 256, 256, 5)]


def test_reversible_streams_decode_losslessly_in_one_batch(eng):
    from qsvc_amd import hip
    ys = [_luma(k, W, H, 7) for (k, W, H, L) in CASES]
    for levels in (2,):
        streams = [eng.j2k_encode_plane(y, hip.j2k_params(reversible=True, levels=levels)) for y in ys]
        for y, d in zip(ys, eng.j2k_decode_planes(streams)):
            np.testing.assert_array_equal(d, y)
    # each with its own level count (separate calls), streams by the restatement
    for y, (k, W, H, L) in zip(ys, CASES):
        cs = J.encode_plane(y, reversible=True, levels=L)
        np.testing.assert_array_equal(eng.j2k_decode_planes([cs])[0], y)


def test_blocks_per_wave_of_the_block_decoder_do_not_change_the_planes(eng):
    """k_t1_decode_lds<NB>: 1 (the default), 2, 4, 6 or 8 code-blocks per wave, one decoding lane each (a per-context switch)."""
    from qsvc_amd import hip
    ys = [_luma(k, W, H, 11) for (k, W, H, L) in CASES]
    streams = [eng.j2k_encode_plane(y, hip.j2k_params(reversible=True, levels=3)) for y in ys]
    lossy = [eng.j2k_encode_plane(ys[4], hip.j2k_params(reversible=False, levels=4, layers=2, slopes=[42000, 0]))]   # (one call, one transform)
    ref = None
    try:
        for nb in (1, 2, 4, 6, 8):
            eng.set_tuning("dec_lanes", nb)
            got = eng.j2k_decode_planes(streams) + eng.j2k_decode_planes(lossy)
            if ref is None:
                ref = got
                for y, d in zip(ys, got):
                    np.testing.assert_array_equal(d, y)
            else:
                for a, b in zip(ref, got):
                    np.testing.assert_array_equal(a, b, err_msg=f"dec_lanes={nb}")
        with pytest.raises(hip.QsvcError):
            eng.set_tuning("dec_lanes", 3)
    finally:
        eng.set_tuning("dec_lanes", 1)


@pytest.mark.parametrize("kind,W,H,levels", CASES)
def test_irreversible_layers_equal_restatement_and_track_openjpeg(eng, kind, W, H, levels):
    from qsvc_amd import hip
    y = _luma(kind, W, H, 8)
    cs = eng.j2k_encode_plane(y, hip.j2k_params(reversible=False, levels=levels, layers=5, slopes=(45000, 46000, 47000, 48000, 49000)))
    for layers in (1, 3, 0):
        d = eng.j2k_decode_planes([cs], max_layers=layers)[0]
        np.testing.assert_array_equal(d, J.own_decode(cs, layers))           # same arithmetic as the restatement: identical
        o = J.decode(cs, layers=layers if layers else None)
        assert np.abs(d.astype(int) - o.astype(int)).max() <= 1               # independent decoder: within 1 LSB


def test_foreign_streams_and_motion_planes(eng):
    from PIL import Image
    from qsvc_amd import hip
    y = _luma("pan_noise", 352, 288, 9)
    buf = io.BytesIO()
    Image.fromarray(y).save(buf, format="JPEG2000", no_jp2=True, irreversible=False, num_resolutions=5, codeblock_size=(64, 64), mct=0)
    np.testing.assert_array_equal(eng.j2k_decode_planes([buf.getvalue()])[0], y)
    buf = io.BytesIO()
    Image.fromarray(y).save(buf, format="JPEG2000", no_jp2=True, irreversible=True, num_resolutions=4, codeblock_size=(32, 32), mct=0,
                            quality_mode="rates", quality_layers=[20, 8])
    d = eng.j2k_decode_planes([buf.getvalue()])[0]
    np.testing.assert_array_equal(d, J.own_decode(buf.getvalue()))
    assert np.abs(d.astype(int) - J.decode(buf.getvalue()).astype(int)).max() <= 1
    r = np.random.default_rng(5).integers(-3000, 3000, size=(6, 135, 240)).astype(np.int16)
    r[2] = 0
    streams = eng.j2k_encode_planes_s16(r, hip.j2k_params(reversible=True, levels=0, layers=1))
    for pl, d in zip(r, eng.j2k_decode_planes(streams)):
        assert d.dtype == np.int16
        np.testing.assert_array_equal(d, pl)
    with pytest.raises(hip.QsvcError):
        eng.j2k_decode_planes([b"\xff\x4f\xff\x51" + bytes(60)])


def test_1080p_sequence_round_trip_and_expand_tool(eng, tmp_path):
    """compress -> texture_expand_fb_j2k: every temporal subband file comes back byte for byte (5/3); a missing component file
    becomes the reference's luma-sized mid-grey block; output is appended."""
    W, H, TRLs, GOPs, bs = 352, 288, 3, 1, 16
    P = GOPs * 2 ** (TRLs - 1) + 1
    clip = make_clip("pan_noise", P, W, H, 4)
    clip.tofile(tmp_path / "low_0")
    mctf = os.path.join(ROOT, "bin", "mctf")

    def run(*args):
        r = subprocess.run([mctf] + [str(a) for a in args], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    run("compress", f"--pixels_in_x={W}", f"--pixels_in_y={H}", f"--TRLs={TRLs}", f"--GOPs={GOPs}", f"--block_size={bs}",
        f"--block_size_min={bs}", "--SRLs=4", "--lossless=1")
    for name, n in (("high_1", 2), ("high_2", 1), ("low_2", 2)):
        want = (tmp_path / name).read_bytes()
        os.rename(tmp_path / name, tmp_path / (name + ".orig"))
        run("texture_expand_fb_j2k", f"--file={name}", f"--pictures={n}", f"--pixels_in_x={W}", f"--pixels_in_y={H}")
        assert (tmp_path / name).read_bytes() == want
    os.remove(tmp_path / "high_1_U_0001.j2c")
    run("texture_expand_fb_j2k", "--file=high_1", "--pictures=2", f"--pixels_in_x={W}", f"--pixels_in_y={H}")
    got = (tmp_path / "high_1").read_bytes()
    F = W * H * 3 // 2
    want = (tmp_path / "high_1.orig").read_bytes()
    assert got[:2 * F] == want                                             # appended after the first expansion
    second = got[2 * F:]
    assert second[:F + W * H] == want[:F + W * H]                          # picture 0 and the luma of picture 1
    assert second[F + W * H:F + 2 * W * H] == bytes([128]) * (W * H)       # missing U: luma-sized neutral block (:170-176)
    assert second[F + 2 * W * H:] == want[F + W * H + W * H // 4:]


def test_corrupt_main_headers_are_refused(eng):
    """The same malformed headers as the CPU test, through the C ABI: QsvcError, no fault, and the context stays usable."""
    from qsvc_amd import hip
    y = make_clip("pan", 1, 96, 64, 5)[0][:96 * 64].reshape(64, 96)
    cs = J.encode_plane(y, reversible=True, levels=3)
    for name, bad in J.corrupt_headers(cs):
        with pytest.raises(hip.QsvcError):
            eng.j2k_decode_planes([bad])
    np.testing.assert_array_equal(eng.j2k_decode_planes([cs])[0], y)


def test_decode_into_resident_sequences_equals_host_planes():
    """qsvc_j2k_decode_multi: the subbands' pictures straight from their code-streams into device-resident sequences, one batch for
    several sequences -- the same pictures qsvc_j2k_decode_planes returns to the host."""
    from qsvc_amd import hip
    from qsvc_amd.pipeline import CompressParams, Compressor
    from qsvc_amd.synth import make_clip
    W, H = 192, 128
    e = hip.Engine(0)
    d = hip.Engine(0)
    try:
        comp = Compressor(e, CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=3, GOPs=1, block_size=16, block_size_min=16, SRLs=4, lossless=True))
        comp.configure(); comp.load(make_clip("edge", comp.pictures, W, H, 5))
        out = comp.compress()
        d.seq_config(W, H)
        d.j2k_decode_multi([(kind, t, ss) for (kind, t), ss in out.items()])
        for (kind, t), ss in out.items():
            n = len(ss) // 3
            got = d.get_frames(kind, t, 0, n)
            np.testing.assert_array_equal(got, e.get_frames(kind, t, 0, n), err_msg=f"{kind} {t}")     # lossless: the encoder's pictures
            planes = d.j2k_decode_planes(ss)
            for i in range(n):
                np.testing.assert_array_equal(got[i], np.concatenate([p.reshape(-1) for p in planes[3 * i:3 * i + 3]]))
        with pytest.raises(hip.QsvcError):   # a stream of another geometry is refused, nothing is written out of place
            d.seq_config(96, 64)
            d.j2k_decode_frames(hip.LOW, 2, out[(hip.LOW, 2)][:3])
    finally:
        e.close(); d.close()
