"""CPU: the T.800 restatement (oracle/j2k_oracle.c) against a SECOND, independent ENCODER -- OpenJPEG 2.5 through Pillow --
at matching parameters (reversible 5/3, one lossless layer, LRCP, 64x64 code-blocks, 2 guard bits, no precincts).

Both code-streams are taken apart by the independent Tier-2 parser tests/j2k_t2parse.py and compared per code-block:
zero bit-planes, number of coding passes, and the MQ code-word segment byte for byte (which pins context formation, pass
membership, the MQ coder and its termination: one wrong decision changes every later byte).  In every case tried the segments
are IDENTICAL including the terminating bytes, and so is the whole packet sequence (packet headers, tag trees, Lblock
signalling) once the restatement's SOP markers -- which the reference asks for with Cuse_sop=yes and Pillow cannot emit -- are
removed.  What this does NOT pin: the reference's Kakadu (absent, version unpinned: parity with it stays "unpinned"), the 9/7
path (each encoder chooses its own step sizes and float lifting order) and multi-layer rate allocation (encoder-private).
"""
import io
import re

import numpy as np
import pytest

from qsvc_amd.synth import make_clip

J = pytest.importorskip("oracle.j2k")
PIL_Image = pytest.importorskip("PIL.Image")
import j2k_t2parse as T  # noqa: E402  (tests/ is on sys.path under pytest's rootdir conftest)


def _luma(kind, W, H, seed):
    return make_clip(kind, 1, W, H, seed)[0][:W * H].reshape(H, W)


def openjpeg_encode(arr, levels, signed=False):
    buf = io.BytesIO()
    PIL_Image.fromarray(arr).save(buf, format="JPEG2000", no_jp2=True, irreversible=False, num_resolutions=levels + 1,
                                  codeblock_size=(64, 64), mct=0, signed=signed)
    return buf.getvalue()


def packets(cs, strip_sop):
    body = cs[cs.index(b"\xff\x93") + 2:-2]
    return re.sub(b"\xff\x91\x00\x04..", b"", body, flags=re.S) if strip_sop else body


def assert_same_blocks(ours: bytes, theirs: bytes):
    a, b = T.parse(ours), T.parse(theirs)
    assert (a.W, a.H, a.prec, a.signed, a.levels, a.layers, a.reversible, a.guard, a.cbw, a.cbh) == \
           (b.W, b.H, b.prec, b.signed, b.levels, b.layers, b.reversible, b.guard, b.cbw, b.cbh)
    assert a.sop and not b.sop
    n = 0
    for (sa, ba), (sb, bb) in zip(a.blocks(), b.blocks()):
        where = f"res {sa.res} orient {sa.orient} block {n}"
        assert (sa.eps, sa.mant, sa.w, sa.h, ba.w, ba.h) == (sb.eps, sb.mant, sb.w, sb.h, bb.w, bb.h), where
        assert ba.zbp == bb.zbp, f"zero bit-planes, {where}"
        assert ba.npasses == bb.npasses, f"coding passes, {where}"
        assert bytes(ba.data) == bytes(bb.data), f"MQ segment, {where}"
        n += 1
    assert n == sum(len(s.blocks) for s in b.subbands)
    assert packets(ours, True) == packets(theirs, False), "packet headers"
    return n


CASES = [("pan", 64, 64, 0), ("pan", 64, 64, 3), ("noise", 100, 75, 3), ("binary", 128, 96, 5), ("pan_noise", 352, 288, 4),
         ("edge", 37, 130, 2), ("saturating", 200, 120, 5), ("noise", 65, 129, 1), ("noise", 256, 256, 5)]  # = test_gpu_j2k.CS_CASES


@pytest.mark.parametrize("kind,W,H,levels", CASES)
def test_restatement_equals_openjpeg_encoder_per_codeblock(kind, W, H, levels):
    y = _luma(kind, W, H, 7)
    assert_same_blocks(J.encode_plane(y, reversible=True, levels=levels), openjpeg_encode(y, levels))


def test_restatement_equals_openjpeg_encoder_1080p_plane():
    """One luma plane of BASELINE config 2/3: 577 code-blocks, Clevels=4."""
    y = _luma("pan_noise", 1920, 1080, 43)
    assert assert_same_blocks(J.encode_plane(y, reversible=True, levels=4), openjpeg_encode(y, 4)) == 577


def test_motion_plane_coding_equals_openjpeg_encoder():
    """N1: 16-bit signed, reversible, Clevels=0 planes (motion_compress_j2k.py:131-142), 1080p block-8 field size and an odd one."""
    rng = np.random.default_rng(3)
    for shape, amp in (((135, 240), 300), ((9, 11), 20), ((70, 65), 5000)):
        r = rng.integers(-amp, amp, size=shape).astype(np.int16)
        assert_same_blocks(J.encode_plane_s16(r), openjpeg_encode(r.view(np.uint16), 0, signed=True))
    # The ONE divergence found: a packet to which no code-block contributes (an all-zero field).  The restatement writes the
    # zero-length packet of B.10.3 (first header bit 0 -> byte 00); OpenJPEG writes a non-empty packet whose inclusion tag
    # tree says "nothing yet" (byte 80).  Both are legal T.800, both are one byte (so the SOP size accounting of
    # header_size.c:62-66, "7 = empty layer header", sees the same size), and both decode to zeros.
    z = np.zeros((17, 30), dtype=np.int16)
    ours, theirs = J.encode_plane_s16(z), openjpeg_encode(z.view(np.uint16), 0, signed=True)
    assert packets(ours, True) == b"\x00" and packets(theirs, False) == b"\x80"
    np.testing.assert_array_equal(J.decode_s16(ours), z)
    np.testing.assert_array_equal(J.decode_s16(theirs), z)


def test_parser_reads_multi_layer_streams():
    """The restatement's own 5-layer 9/7 streams parse, and the passes / bytes per layer add up to the block's total."""
    y = _luma("pan_noise", 352, 288, 4)
    cs = J.encode_plane(y, reversible=False, levels=4, layers=5, slopes=(45000, 46000, 47000, 48000, 49000))
    s = T.parse(cs)
    assert s.layers == 5 and not s.reversible
    tot = 0
    for sb, b in s.blocks():
        assert len(b.passes) == 5 and sum(b.lengths) == len(b.data)
        tot += b.npasses
    assert tot > 0
