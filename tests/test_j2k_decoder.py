"""CPU: the restatement's JPEG 2000 DECODER (oracle/j2k_oracle.c, qo_j2k_decode_plane: T.800 Annex B parsing, C.3 MQ decoder,
Annex D in the decoding direction, E.1.1 midpoint reconstruction, F.3 inverse DWT) -- SURVEY.md 8f N3, the stage the reference
runs as the external `kdu_expand` (trunk/src/texture_expand_fb_j2k.py:153-162).  Pinned by an independent decoder: OpenJPEG
through Pillow decodes the same streams to the same samples -- bit-exact on the reversible path, within 1 LSB on the 9/7 path
(float rounding of two different implementations), for every layer count.  Parity with Kakadu's kdu_expand: unpinned (absent)."""
import io

import numpy as np
import pytest

from qsvc_amd.synth import make_clip

J = pytest.importorskip("oracle.j2k")
CASES = [("pan", 64, 64, 0), ("pan", 64, 64, 3), ("noise", 100, 75, 3), ("binary", 128, 96, 5), ("pan_noise", 352, 288, 4),
         ("edge", 37, 130, 2), ("saturating", 200, 120, 5), ("noise", 65, 129, 1), ("noise", 256, 256, 5)]


def _luma(kind, W, H, seed):
    return make_clip(kind, 1, W, H, seed)[0][:W * H].reshape(H, W)


@pytest.mark.parametrize("kind,W,H,levels", CASES)
def test_reversible_decodes_bit_exactly(kind, W, H, levels):
    y = _luma(kind, W, H, 7)
    cs = J.encode_plane(y, reversible=True, levels=levels)
    np.testing.assert_array_equal(J.own_decode(cs), y)
    np.testing.assert_array_equal(J.own_decode(cs), J.decode(cs))


@pytest.mark.parametrize("kind,W,H,levels", CASES)
def test_irreversible_within_one_lsb_of_openjpeg_for_every_layer_count(kind, W, H, levels):
    y = _luma(kind, W, H, 8)
    cs = J.encode_plane(y, reversible=False, levels=levels, layers=5, slopes=(45000, 46000, 47000, 48000, 49000))
    for layers in (1, 2, 5):
        d, o = J.own_decode(cs, layers).astype(int), J.decode(cs, layers=layers).astype(int)
        assert np.abs(d - o).max() <= 1, (layers, np.abs(d - o).max())
        assert (d != o).mean() < 0.01
    assert J.psnr(J.own_decode(cs), y) > 45


def test_truncated_reversible_streams_and_foreign_streams():
    """Layers of a 5/3 stream (midpoint for the undecoded bit-planes) and a code-stream written by ANOTHER encoder (OpenJPEG, no SOP)."""
    from PIL import Image
    y = _luma("pan_noise", 352, 288, 9)
    cs = J.encode_plane(y, reversible=True, levels=4, layers=4, slopes=(44000, 46000, 48000, 0))
    for layers in (1, 2, 3):
        d, o = J.own_decode(cs, layers).astype(int), J.decode(cs, layers=layers).astype(int)
        assert np.abs(d - o).max() <= 1
    np.testing.assert_array_equal(J.own_decode(cs), y)
    buf = io.BytesIO()
    Image.fromarray(y).save(buf, format="JPEG2000", no_jp2=True, irreversible=False, num_resolutions=5, codeblock_size=(64, 64), mct=0)
    np.testing.assert_array_equal(J.own_decode(buf.getvalue()), y)
    buf = io.BytesIO()
    Image.fromarray(y).save(buf, format="JPEG2000", no_jp2=True, irreversible=True, num_resolutions=4, codeblock_size=(32, 32), mct=0,
                            quality_mode="rates", quality_layers=[20, 8])
    d, o = J.own_decode(buf.getvalue()).astype(int), J.decode(buf.getvalue()).astype(int)
    assert np.abs(d - o).max() <= 1


def test_motion_planes_decode():
    r = np.random.default_rng(3).integers(-3000, 3000, size=(135, 240)).astype(np.int16)
    np.testing.assert_array_equal(J.own_decode(J.encode_plane_s16(r)), r)
    z = np.zeros((17, 30), dtype=np.int16)
    np.testing.assert_array_equal(J.own_decode(J.encode_plane_s16(z)), z)


def test_corrupt_main_headers_are_refused():
    """Short, truncated or inconsistent SIZ / COD / QCD segments (the decoder reads code-streams from files): an error, not a read
    past the buffer; a later duplicate SIZ is refused, not allowed to replace the checked one."""
    y = _luma("pan", 96, 64, 5)
    cs = J.encode_plane(y, reversible=True, levels=3)
    np.testing.assert_array_equal(J.own_decode(cs), y)
    for name, bad in J.corrupt_headers(cs):
        with pytest.raises(RuntimeError):
            J.own_decode(bad)
