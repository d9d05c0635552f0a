"""CPU: the T.800 restatement (oracle/j2k_oracle.c).  PARITY UNPINNED versus the reference's Kakadu
(no artefact of it exists); what pins the restatement here:
  * the MQ coder reproduces the published arithmetic-coder test sequence of ITU-T T.88 Annex H.2
    (same coder as T.800 Annex C);
  * every code-stream is decoded by an independent decoder (OpenJPEG via Pillow): bit-exact for the
    reversible 5/3 path, PSNR-bounded for 9/7, layer-progressive for multi-layer streams;
  * structural checks the reference's own consumer makes (header_size.c:46-58 scans for FF 91 00 04).
"""
import ctypes as C

import numpy as np
import pytest

from qsvc_amd.synth import make_clip

J = pytest.importorskip("oracle.j2k")


def _luma(kind, W, H, seed):
    return make_clip(kind, 1, W, H, seed)[0][:W * H].reshape(H, W)


def test_mq_coder_published_test_sequence(oracle):
    inp = bytes.fromhex("00020051000000C00352872AAAAAAAAA82C02000FCD79EF6BF7FED904F46A3BF")
    want = bytes.fromhex("84C73BFCE1A1430402200000410DBB86F4317FFF88FF37471ADB6ADF")
    bits = np.unpackbits(np.frombuffer(inp, dtype=np.uint8)).astype(np.uint8)
    out = np.zeros(64, dtype=np.uint8)
    n = oracle.lib().qo_mq_encode_bits(bits.ctypes.data_as(C.c_void_p), len(bits), out.ctypes.data_as(C.c_void_p))
    assert out[:n].tobytes() == want


CASES = [("pan", 64, 64, 0), ("pan", 64, 64, 3), ("noise", 100, 75, 3), ("binary", 128, 96, 5), ("pan_noise", 352, 288, 4),
         ("edge", 37, 130, 2), ("saturating", 200, 120, 5), ("noise", 65, 129, 1)]


@pytest.mark.parametrize("kind,W,H,levels", CASES)
def test_reversible_roundtrip_through_openjpeg(kind, W, H, levels):
    y = _luma(kind, W, H, 5)
    cs = J.encode_plane(y, reversible=True, levels=levels)
    assert cs[:2] == b"\xff\x4f" and cs[-2:] == b"\xff\xd9"
    np.testing.assert_array_equal(J.decode(cs), y)


@pytest.mark.parametrize("kind,W,H,levels", CASES[:6])
def test_irreversible_psnr_through_openjpeg(kind, W, H, levels):
    y = _luma(kind, W, H, 6)
    cs = J.encode_plane(y, reversible=False, levels=levels)
    assert J.psnr(J.decode(cs), y) > 50.0
    cs2 = J.encode_plane(y, reversible=False, levels=levels, base_step=1.0 / 16)
    assert len(cs2) < len(cs) and J.psnr(J.decode(cs2), y) > 30.0


def test_constant_and_extreme_planes():
    for v in (0, 128, 255):
        y = np.full((48, 80), v, np.uint8)
        np.testing.assert_array_equal(J.decode(J.encode_plane(y, reversible=True, levels=3)), y)
    y = np.zeros((64, 64), np.uint8); y[::2, 1::2] = 255  # worst-case high-pass energy
    np.testing.assert_array_equal(J.decode(J.encode_plane(y, reversible=True, levels=2)), y)


def test_quality_layers_are_progressive():
    y = _luma("pan", 352, 288, 42)
    cs = J.encode_plane(y, reversible=False, levels=4, layers=5, slopes=(49000, 48000, 47000, 46000, 45000))
    ps = [J.psnr(J.decode(cs, layers=l), y) for l in range(1, 6)]
    assert all(b >= a for a, b in zip(ps, ps[1:])) and ps[-1] - ps[0] > 20
    # the reference's ascending schedule (texture_compress.py:156-165): everything lands in layer 1
    cs = J.encode_plane(y, reversible=False, levels=4, layers=5, slopes=(45000, 46000, 47000, 48000, 49000))
    assert abs(J.psnr(J.decode(cs, layers=1), y) - J.psnr(J.decode(cs), y)) < 1e-9
    # reversible with a final "everything" layer stays lossless
    cs = J.encode_plane(y, reversible=True, levels=4, layers=3, slopes=(47000, 45000, 0))
    np.testing.assert_array_equal(J.decode(cs), y)


def test_sop_markers_as_header_size_counts_them():
    y = _luma("pan", 128, 96, 1)
    for layers, levels in ((1, 3), (4, 2)):
        cs = J.encode_plane(y, reversible=True, levels=levels, layers=layers, slopes=(48000, 47000, 46000, 0)[:layers] if layers > 1 else ())
        assert cs.count(b"\xff\x91\x00\x04") >= layers * (levels + 1)
        cs2 = J.encode_plane(y, reversible=True, levels=levels, layers=layers, use_sop=False,
                             slopes=(48000, 47000, 46000, 0)[:layers] if layers > 1 else ())
        assert len(cs) - len(cs2) == 6 * layers * (levels + 1)


def test_resolution_scalability():
    y = _luma("pan", 256, 192, 3)
    cs = J.encode_plane(y, reversible=True, levels=4)
    half = J.decode(cs, reduce=1)
    assert half.shape == (96, 128)
    co = J.coefficients(y, True, 1)[:96, :128] + 128   # LL of a 1-level 5/3 = what a reduce=1 decode shows
    np.testing.assert_array_equal(np.clip(co, 0, 255).astype(np.uint8), half)


def test_t1_single_block_basics():
    z = np.zeros((64, 64), np.int32)
    data, npasses, nbps, _ = J.t1_block(z, 0)
    assert (len(data), npasses, nbps) == (0, 0, 0)
    z[10, 10] = -5
    data, npasses, nbps, pl = J.t1_block(z, 3)
    assert nbps == 3 and npasses == 7 and len(data) > 0 and pl[-1] == len(data) and all(np.diff(pl) >= 0)
