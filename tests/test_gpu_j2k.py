"""GPU parity of the JPEG 2000 stage: HIP path (through the C ABI) == the T.800 restatement byte for byte,
and every code-stream decodes in an independent decoder (OpenJPEG via Pillow).  Parity with the
reference's Kakadu is UNPINNED (see oracle/j2k_oracle.c)."""
import numpy as np
import pytest

from qsvc_amd.synth import make_clip

pytestmark = pytest.mark.gpu
J = pytest.importorskip("oracle.j2k")


@pytest.fixture(scope="module")
def eng():
    from qsvc_amd import hip
    e = hip.Engine(0)
    yield e
    e.close()


def _luma(kind, W, H, seed):
    return make_clip(kind, 1, W, H, seed)[0][:W * H].reshape(H, W)


DWT_CASES = [("pan", 64, 64, 3), ("noise", 100, 75, 3), ("pan_noise", 352, 288, 5), ("edge", 37, 130, 2), ("noise", 65, 129, 1),
             ("pan", 1920, 1080, 5)]


@pytest.mark.parametrize("kind,W,H,levels", DWT_CASES)
def test_dwt53_bit_exact(eng, kind, W, H, levels):
    y = _luma(kind, W, H, 5)
    np.testing.assert_array_equal(eng.j2k_dwt(y, True, levels), J.coefficients(y, True, levels))


@pytest.mark.parametrize("kind,W,H,levels", DWT_CASES)
def test_dwt97_within_one_ulp(eng, kind, W, H, levels):
    """north_star tolerance: within 1 ULP per coefficient (float32); the kernel keeps the restatement's
    operation order without FMA contraction, so in practice it is bit-identical."""
    y = _luma(kind, W, H, 6)
    got, want = eng.j2k_dwt(y, False, levels), J.dwt97(y, levels)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    np.testing.assert_array_equal(got, want)


CS_CASES = [("pan", 64, 64, 0), ("pan", 64, 64, 3), ("noise", 100, 75, 3), ("binary", 128, 96, 5), ("pan_noise", 352, 288, 4),
            ("edge", 37, 130, 2), ("saturating", 200, 120, 5), ("noise", 65, 129, 1), ("noise", 256, 256, 5)]


@pytest.mark.parametrize("kind,W,H,levels", CS_CASES)
@pytest.mark.parametrize("reversible", [True, False])
def test_codestream_equals_restatement_and_decodes(eng, kind, W, H, levels, reversible):
    from qsvc_amd import hip
    y = _luma(kind, W, H, 7)
    cs = eng.j2k_encode_plane(y, hip.j2k_params(reversible=reversible, levels=levels))
    want = J.encode_plane(y, reversible=reversible, levels=levels)
    assert cs == want
    d = J.decode(cs)
    if reversible:
        np.testing.assert_array_equal(d, y)
    else:
        assert J.psnr(d, y) > 50


RB97_CASES = [("noise", 250, 190, 3, 64), ("pan_noise", 641, 363, 4, 64), ("edge", 133, 517, 3, 32), ("noise", 16, 16, 1, 16),
              ("saturating", 1000, 72, 5, 64), ("pan_noise", 1920, 1080, 5, 64), ("pan_noise", 3840, 2160, 5, 64)]   # the last: a luma plane of BASELINE config 4


@pytest.mark.parametrize("kind,W,H,levels,cblk", RB97_CASES)
def test_irreversible_streams_register_blocked_dwt(eng, kind, W, H, levels, cblk):
    """The 9/7 path to Tier-1 runs k_dwt97_rb (one pass per direction, lifting in registers, the signal's ends mirrored) on every
    level of at least 8 x 8 samples and the general tile kernel below that: odd widths and heights, picture ends inside a tile in
    both directions, levels that switch kernels, narrow code-blocks, a 1080p plane.  Byte-identical to the restatement, and to the
    general kernel alone (dwt_pk = 0)."""
    from qsvc_amd import hip
    y = _luma(kind, W, H, 11)
    prm = hip.j2k_params(reversible=False, levels=levels, cblk=cblk)
    cs = eng.j2k_encode_plane(y, prm)
    assert cs == J.encode_plane(y, reversible=False, levels=levels, cblk=cblk)
    try:
        eng.set_tuning("dwt_pk", 0)
        assert eng.j2k_encode_plane(y, prm) == cs
    finally:
        eng.set_tuning("dwt_pk", 1)


def test_reversible_dwt_two_tiles_per_workgroup(eng):
    """k_dwt53_rb<.., 2> (two tiles of a level's list per workgroup, what levels of at least 2 048 tiles run): forced on every level
    (dwt_pk = 4) for planes whose levels have odd and even tile counts, one tile, picture ends inside tiles; then a batch large enough
    to take it by itself (8 pictures of 1080p: 6 256 tiles at level 1, 1 664 at level 2) against the one-tile form and the restatement."""
    from qsvc_amd import hip
    try:
        eng.set_tuning("dwt_pk", 4)
        for kind, W, H, levels, cblk in (("noise", 250, 190, 3, 64), ("pan_noise", 641, 363, 4, 64), ("edge", 133, 517, 3, 32), ("noise", 16, 16, 1, 16),
                                         ("pan_noise", 352, 288, 5, 64), ("saturating", 1000, 72, 5, 64), ("pan_noise", 1920, 1080, 4, 64)):
            y = _luma(kind, W, H, 13)
            assert eng.j2k_encode_plane(y, hip.j2k_params(reversible=True, levels=levels, cblk=cblk)) == J.encode_plane(y, reversible=True, levels=levels, cblk=cblk), (W, H)
    finally:
        eng.set_tuning("dwt_pk", 1)
    W, H = 1920, 1080
    clip = make_clip("pan_noise", 8, W, H, 44)
    eng.seq_config(W, H)
    eng.put_frames(hip.LOW, 0, clip)
    prm = hip.j2k_params(reversible=True, levels=4)
    streams = eng.j2k_encode_frames(hip.LOW, 0, 0, 8, prm)
    try:
        eng.set_tuning("dwt_pk", 3)
        assert eng.j2k_encode_frames(hip.LOW, 0, 0, 8, prm) == streams
    finally:
        eng.set_tuning("dwt_pk", 1)
    assert streams[21] == J.encode_plane(clip[7][:W * H].reshape(H, W), reversible=True, levels=4)
    assert streams[5] == J.encode_plane(clip[1][W * H + (W // 2) * (H // 2):].reshape(H // 2, W // 2), reversible=True, levels=4)


def test_layers_slopes_and_options(eng):
    from qsvc_amd import hip
    y = _luma("pan", 352, 288, 42)
    for kw in (dict(reversible=False, levels=4, layers=5, slopes=(49000, 48000, 47000, 46000, 45000)),
               dict(reversible=False, levels=4, layers=5, slopes=(45000, 46000, 47000, 48000, 49000)),
               dict(reversible=True, levels=4, layers=3, slopes=(47000, 45000, 0)),
               dict(reversible=True, levels=3, use_sop=False),
               dict(reversible=False, levels=5, base_step=1.0 / 16),
               dict(reversible=True, levels=2, cblk=32)):
        cs = eng.j2k_encode_plane(y, hip.j2k_params(**kw))
        assert cs == J.encode_plane(y, **kw), kw
    st = eng.j2k_stats()
    assert st["codeblocks"] > 0 and st["passes"] > 0 and st["symbols"] > 0


def test_frames_1080p_lossless(eng):
    """BASELINE config 2: 1080p intra frame, 5/3 + EBCOT: per-component streams == restatement, lossless."""
    from qsvc_amd import hip
    W, H = 1920, 1080
    clip = make_clip("pan_noise", 2, W, H, 43)
    eng.seq_config(W, H)
    eng.put_frames(hip.LOW, 0, clip)
    streams = eng.j2k_encode_frames(hip.LOW, 0, 0, 2, hip.j2k_params(reversible=True, levels=4))
    assert len(streams) == 6
    assert eng.j2k_stats()["codeblocks"] == 2 * 921
    for i in range(2):
        Y = clip[i][:W * H].reshape(H, W)
        U = clip[i][W * H:W * H + (W // 2) * (H // 2)].reshape(H // 2, W // 2)
        V = clip[i][W * H + (W // 2) * (H // 2):].reshape(H // 2, W // 2)
        for k, pl in enumerate((Y, U, V)):
            np.testing.assert_array_equal(J.decode(streams[3 * i + k]), pl)
    assert streams[0] == J.encode_plane(clip[0][:W * H].reshape(H, W), reversible=True, levels=4)
    assert streams[4] == J.encode_plane(clip[1][W * H:W * H + (W // 2) * (H // 2)].reshape(H // 2, W // 2), reversible=True, levels=4)


def test_multi_sequence_batch_equals_single_calls(eng):
    """One batched launch set for all temporal subbands == the per-sequence calls (texture_compress.py:183-217)."""
    from qsvc_amd import hip
    W, H, bs = 192, 128, 16
    clip = make_clip("pan_noise", 9, W, H, 5)
    eng.seq_config(W, H)
    eng.put_frames(hip.LOW, 0, clip)
    eng.analyze_step(1, 9, bs, 4)
    eng.analyze_step(2, 5, bs, 8)
    items = [(hip.HIGH, 1, 0, 4, hip.j2k_params(reversible=False, levels=3, layers=3, slopes=(47500, 46500, 45500))),
             (hip.HIGH, 2, 0, 2, hip.j2k_params(reversible=False, levels=3, layers=3, slopes=(47300, 46300, 45300))),
             (hip.LOW, 2, 1, 2, hip.j2k_params(reversible=False, levels=3, layers=2, slopes=(46000, 45000)))]
    multi = eng.j2k_encode_multi(items)
    for it, got in zip(items, multi):
        assert got == eng.j2k_encode_frames(*it)
    hi = eng.get_frames(hip.HIGH, 2, 1, 1)[0]
    assert multi[1][3] == J.encode_plane(hi[:W * H].reshape(H, W), reversible=False, levels=3, layers=3, slopes=(47300, 46300, 45300))


@pytest.mark.parametrize("n,H,W,amp", [(8, 9, 11, 40), (4, 135, 240, 300), (3, 17, 30, 16000), (5, 1, 1, 7), (2, 70, 66, 2000)])
def test_motion_planes_s16_equal_restatement_and_decode(eng, n, H, W, amp):
    """16-bit signed lossless planes (motion residues, motion_compress_j2k.py:131-142): HIP == restatement, decodes exactly."""
    from qsvc_amd import hip
    rng = np.random.default_rng(H * 1000 + W)
    planes = rng.integers(-amp, amp + 1, size=(n, H, W)).astype(np.int16)
    planes[0] = 0  # an all-zero residue plane (static scene): empty packets
    got = eng.j2k_encode_planes_s16(planes, hip.j2k_params(reversible=True, levels=0, layers=1))
    assert len(got) == n
    for k in range(n):
        assert got[k] == J.encode_plane_s16(planes[k]), k
        np.testing.assert_array_equal(J.decode_s16(got[k]), planes[k])


def test_motion_planes_s16_too_many_bitplanes_is_an_error(eng):
    from qsvc_amd import hip
    planes = np.full((1, 8, 8), -32768, dtype=np.int16)
    with pytest.raises(hip.QsvcError):
        eng.j2k_encode_planes_s16(planes, hip.j2k_params(reversible=True, levels=0, layers=1))


def test_all_zero_planes_have_empty_packets(eng):
    """No block of the batch has a bit-plane (a flat mid-grey plane): the Tier-1 job list is empty, the stream still decodes."""
    from qsvc_amd import hip
    pl = np.full((48, 80), 128, dtype=np.uint8)
    cs = eng.j2k_encode_plane(pl, hip.j2k_params(reversible=True, levels=2))
    assert cs == J.encode_plane(pl, reversible=True, levels=2)
    np.testing.assert_array_equal(J.decode(cs), pl)


def test_trailing_blocks_without_bit_planes(eng):
    """The last code-blocks of the batch are all zero while earlier ones are not: such a block owns no Tier-1 job and its first-job
    index equals the job count (one past the tables) -- the MQ kernel's pass-table prefetch must not form an address from it."""
    from qsvc_amd import hip
    for (H, W, levels) in ((128, 128, 1), (192, 256, 2), (64, 320, 0)):
        y = np.full((H, W), 128, dtype=np.uint8)
        y[:40, :40] = _luma("noise", 40, 40, 9)  # texture in the top-left corner only: the high bands' last blocks stay empty
        for pipe in (1, 0):
            eng.set_tuning("mq_pipe", pipe)
            try:
                cs = eng.j2k_encode_plane(y, hip.j2k_params(reversible=True, levels=levels))
            finally:
                eng.set_tuning("mq_pipe", 1)
            assert cs == J.encode_plane(y, reversible=True, levels=levels)
            np.testing.assert_array_equal(J.decode(cs), y)


def test_plane_groups_do_not_change_the_streams(eng):
    """QSVC_T1_BUDGET_GB bounds the Tier-1 working set by coding the planes in groups; the streams are the same bytes."""
    import os
    import subprocess
    import sys
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from qsvc_amd import hip\nfrom qsvc_amd.synth import make_clip\n"
            "e = hip.Engine(0); W, H = 256, 192\n"
            "clip = make_clip('pan_noise', 5, W, H, 9); e.seq_config(W, H); e.put_frames(hip.LOW, 0, clip)\n"
            "out = e.j2k_encode_frames(hip.LOW, 0, 0, 5, hip.j2k_params(reversible=True, levels=3))\n"
            "print(len(out), hashlib.sha256(b''.join(out)).hexdigest())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = []
    for gb in ("32", "0.01"):  # 0.01 GB -> 29 blocks per group: every plane of this size becomes its own group
        env = dict(os.environ, QSVC_T1_BUDGET_GB=gb)
        runs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True))
        assert runs[-1].returncode == 0, runs[-1].stderr[-800:]
    assert runs[0].stdout == runs[1].stdout and runs[0].stdout.startswith("15 ")


def test_random_planes_equal_restatement(eng):
    """Random sizes (1..200), contents (noise, a few isolated samples on a flat plane, smooth, sparse), decomposition levels 0..5,
    code-block sizes, both filters: byte-identical to the restatement.  Sparse and partial blocks exercise the column-list form
    of the context modelling kernel and blocks of any height / width; tools/stress_j2k.py is the long version (700 cases)."""
    from qsvc_amd import hip
    rng = np.random.default_rng(20)
    for k in range(70):
        W, H = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        kind = k % 4
        if kind == 0:
            p = rng.integers(0, 256, size=(H, W))
        elif kind == 1:
            p = np.full((H, W), int(rng.integers(0, 256)))
            for _ in range(int(rng.integers(0, 6))):
                p[rng.integers(0, H), rng.integers(0, W)] = int(rng.integers(0, 256))
        elif kind == 2:
            yy, xx = np.mgrid[0:H, 0:W]
            p = (128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0)).astype(np.int64)
        else:
            p = np.where(rng.random((H, W)) < 0.03, 255, 0)
        p = np.ascontiguousarray(p, dtype=np.uint8)
        rev, levels, cblk = bool(rng.integers(0, 2)), int(rng.integers(0, 6)), int(rng.choice([64, 64, 32, 16]))
        got = eng.j2k_encode_plane(p, hip.j2k_params(reversible=rev, levels=levels, cblk=cblk))
        assert got == J.encode_plane(p, reversible=rev, levels=levels, cblk=cblk), (k, W, H, kind, rev, levels, cblk)


def test_both_mq_kernel_forms_give_the_same_streams(eng):
    """The MQ coder runs as a two-wave pipeline (k_t1_mq_pipe, the default, in two LDS sizes) or as one wave per 4 .. 64 blocks, with the
    blocks sorted by expected length or not: per-context switches (qsvc_set_tuning; a context starts from the QSVC_* environment as it
    was when it was created).  Same bytes, and equal to the restatement's.  So with the other kernel forms kept for comparison."""
    from qsvc_amd import hip
    y = _luma("pan_noise", 352, 288, 12)
    prm = hip.j2k_params(reversible=True, levels=4)
    want = J.encode_plane(y, reversible=True, levels=4)
    defaults = {"mq_pipe": 1, "mq_lds_small": 0, "mq_lanes": 64, "mq_sort": 1, "dwt_pk": 1, "t1_devlist": 1}
    try:
        for sw in ({}, {"mq_lds_small": 1}, {"mq_pipe": 0}, {"mq_pipe": 0, "mq_lanes": 16}, {"mq_sort": 0}, {"dwt_pk": 0}, {"dwt_pk": 2}, {"dwt_pk": 3}, {"dwt_pk": 4}, {"t1_devlist": 0}):
            for k, v in dict(defaults, **sw).items():
                eng.set_tuning(k, v)
            assert eng.j2k_encode_plane(y, prm) == want, sw
        with pytest.raises(hip.QsvcError):
            eng.set_tuning("no_such_switch", 1)
        with pytest.raises(hip.QsvcError):
            eng.set_tuning("mq_lanes", 5)
    finally:
        for k, v in defaults.items():
            eng.set_tuning(k, v)


def test_too_many_bitplanes_is_an_error_not_a_fault(eng):
    """A quantiser step so small that a subband needs more than its M_b (or more than 15) magnitude bit-planes must come back
    as QSVC_ERR_UNSUPPORTED before any Tier-1 kernel runs on the unclamped count (j2k_api.hip: err_d is read in the same sync
    as the bit-plane counts).  Reachable from the public --base_step extension and from int16 planes holding -32768."""
    from qsvc_amd import hip
    y = _luma("noise", 256, 192, 9)
    with pytest.raises(hip.QsvcError) as ei:
        eng.j2k_encode_plane(y, hip.j2k_params(reversible=False, levels=3, base_step=1.0 / 65536))
    assert ei.value.code == -2
    planes = np.zeros((2, 40, 48), dtype=np.int16)
    planes[1, 7, 9] = -32768
    with pytest.raises(hip.QsvcError) as ei:
        eng.j2k_encode_planes_s16(planes, hip.j2k_params(reversible=True, levels=0, layers=1))
    assert ei.value.code == -2
    # the context is still usable afterwards
    cs = eng.j2k_encode_plane(y, hip.j2k_params(reversible=True, levels=3))
    np.testing.assert_array_equal(J.decode(cs), y)


@pytest.mark.parametrize("kind,W,H,levels", [("pan_noise", 352, 288, 4), ("noise", 65, 129, 1), ("pan_noise", 1920, 1080, 4)])
def test_hip_codestream_equals_openjpeg_encoder(eng, kind, W, H, levels):
    """The HIP path against the SECOND encoder directly (not through the restatement): per code-block zero bit-planes, passes
    and MQ segments, and the whole packet sequence once our SOP markers are removed (tests/test_j2k_vs_openjpeg.py)."""
    from qsvc_amd import hip
    from test_j2k_vs_openjpeg import assert_same_blocks, openjpeg_encode
    y = _luma(kind, W, H, 43)
    assert_same_blocks(eng.j2k_encode_plane(y, hip.j2k_params(reversible=True, levels=levels)), openjpeg_encode(y, levels))


def test_hip_motion_planes_equal_openjpeg_encoder(eng):
    from qsvc_amd import hip
    from test_j2k_vs_openjpeg import assert_same_blocks, openjpeg_encode
    r = np.random.default_rng(5).integers(-40, 40, size=(8, 135, 240)).astype(np.int16)
    for cs, pl in zip(eng.j2k_encode_planes_s16(r, hip.j2k_params(reversible=True, levels=0, layers=1)), r):
        assert_same_blocks(cs, openjpeg_encode(pl.view(np.uint16), 0, signed=True))


def test_dwt_entry_point_between_two_encodes_of_one_geometry(eng):
    """qsvc_j2k_dwt writes its own one-plane descriptor into the table the encoder caches by content: an encode, a transform of
    another geometry and the same encode again must give the same stream (the table is uploaded again, not assumed resident)."""
    from qsvc_amd import hip
    pl = _luma("pan_noise", 176, 144, 5)
    p = hip.j2k_params(reversible=True, levels=3)
    a = eng.j2k_encode_plane(pl, p)
    other = _luma("noise", 100, 76, 6)
    np.testing.assert_array_equal(eng.j2k_dwt(other, True, 2), J.coefficients(other, True, 2))
    b = eng.j2k_encode_plane(pl, p)
    assert a == b == J.encode_plane(pl, reversible=True, levels=3)


def test_native_motion_coding_chain_equals_the_host_modules(eng):
    """qsvc_motion_encode (the whole motion-field coding chain of `mctf compress` in one call) == qsvc_amd/motion.py, which is pinned by
    the reference's own decorrelators (tests/test_motion_host.py), + the restatement's lossless coder on every component plane."""
    from qsvc_amd import hip, motion
    from qsvc_amd.pipeline import CompressParams, Compressor
    W, H, bs, TRLs = 128, 96, 16, 4
    comp = Compressor(eng, CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=TRLs, GOPs=1, block_size=bs, block_size_min=bs, SRLs=3, lossless=True))
    comp.configure(); comp.load(make_clip("edge", comp.pictures, W, H, 12))
    comp.analyze()
    res, streams = comp.motion(want_residues=True)
    filt = {t: eng.get_motion(hip.MOTION_FILTERED, t, P // 2, bs) for (t, P, S, b) in comp.levels}
    assert any(np.abs(f).max() > 0 for f in filt.values())
    want = motion.residue_chain(filt, TRLs, bs, bs)
    for t in want:
        np.testing.assert_array_equal(res[t], want[t], err_msg=f"motion_residue_{t}")
        comps = motion.demux_components(want[t])
        for n in range(want[t].shape[0]):
            for c in range(4):
                assert streams[t][4 * n + c] == J.encode_plane_s16(comps[c, n]), (t, n, c)


@pytest.mark.parametrize("lossless", [True, False])
def test_textures_and_motion_in_one_batch_equal_the_separate_calls(eng, lossless):
    """qsvc_compress_code: the motion fields' 16-bit component planes ride in the textures' batch (level shift + pack beside the DWT
    kernels, then the same Tier-1 launches); every stream equals what qsvc_j2k_encode_multi and qsvc_motion_encode give on their own."""
    from qsvc_amd.pipeline import CompressParams, Compressor
    W, H, bs, TRLs = 192, 128, 16, 4
    comp = Compressor(eng, CompressParams(pixels_in_x=W, pixels_in_y=H, TRLs=TRLs, GOPs=1, block_size=bs, block_size_min=bs, SRLs=4, lossless=lossless,
                                          nLayers=3))
    comp.configure(); comp.load(make_clip("edge", comp.pictures, W, H, 13))
    comp.analyze()
    _, mot = comp.motion()
    tex = comp.texture()
    tex2, mot2 = comp.code()
    assert sorted(tex) == sorted(tex2) and sorted(mot) == sorted(mot2)
    for k in tex:
        assert tex[k] == tex2[k], k
    for t in mot:
        assert mot[t] == mot2[t], t
    assert comp.texture() == tex   # and the texture batch alone again afterwards (the encoder's resident tables follow the batch's shape)
