"""Motion-field coding chain, host part: qsvc_amd/motion.py and the two decorrelator CLIs against the
reference's own binaries (goldens from oracle/_ref, tests/golden/make_goldens.py) -- no GPU involved."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from qsvc_amd import motion

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KAT = np.load(os.path.join(HERE, "golden", "motion_kat.npz"))


def test_tdiv2_is_c_division():
    a = np.array([-32768, -5, -3, -2, -1, 0, 1, 2, 3, 5, 32767], dtype=np.int16)
    assert motion.tdiv2(a).tolist() == [-16384, -2, -1, -1, 0, 0, 0, 1, 1, 2, 16383]


@pytest.mark.parametrize("i", [0, 1, 2])
def test_interlevel_vs_reference_binary(i):
    got = motion.interlevel_decorrelate(KAT[f"pred_{i}"], KAT[f"ref_{i}"])
    assert got.dtype == np.int16 and np.array_equal(got, KAT[f"inter_{i}"])


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_bidirectional_vs_reference_binary(i):
    src = KAT[f"ref_{i}"] if f"inter_{i}" in KAT else KAT[f"pred_{i}"]
    assert np.array_equal(motion.bidirectional_decorrelate(src), KAT[f"bidir_{i}"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "mctf_*.npz"))))
def test_residue_chain_on_golden_fields(path):
    g = np.load(path)
    TRLs, bs = int(g["meta"][3]), int(g["meta"][4])
    filt = {t: g[f"motion_filtered_{t}"] for t in range(1, TRLs)}
    res = motion.residue_chain(filt, TRLs, bs, bs)
    for t in range(1, TRLs):
        assert np.array_equal(res[t], g[f"motion_residue_{t}"]), (path, t)


def test_demux_is_the_reference_byte_cut():
    """motion_compress_j2k.py:103-107: `demux 8 2c 2 < file | split --bytes=BY*BX*2`, done here on the bytes."""
    rng = np.random.default_rng(3)
    F, BY, BX = 3, 4, 5
    r = rng.integers(-500, 500, size=(F, 4, BY, BX)).astype("<i2")
    raw = r.tobytes()
    comps = motion.demux_components(r)
    for c in range(4):
        stream = b"".join(raw[o + 2 * c:o + 2 * c + 2] for o in range(0, len(raw), 8))
        for n in range(F):
            assert comps[c, n].astype("<i2").tobytes() == stream[n * BY * BX * 2:(n + 1) * BY * BX * 2]


def test_shrinking_block_size_is_refused():
    with pytest.raises(NotImplementedError):
        motion.residue_chain({}, 3, 64, 32)


def _mctf(tool, args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, "-m", f"qsvc_amd.cli.{tool}"] + args, cwd=cwd, env=env, capture_output=True)


def test_decorrelator_clis_write_the_reference_files(tmp_path):
    i = 1  # 5 predicted fields over 2 reference fields: the last one reuses the stale reference, as the C++ loop does
    a, b = KAT[f"pred_{i}"], KAT[f"ref_{i}"]
    BY, BX = a.shape[2:]
    a.astype("<i2").tofile(tmp_path / "motion_filtered_1")
    b.astype("<i2").tofile(tmp_path / "motion_filtered_2")
    r = _mctf("interlevel_motion_decorrelate", [f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", "--fields_in_predicted=5",
              "--predicted=motion_filtered_1", "--reference=motion_filtered_2", "--residue=motion_residue_1"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "motion_residue_1").read_bytes() == KAT[f"inter_{i}"].astype("<i2").tobytes()
    r = _mctf("bidirectional_motion_decorrelate", [f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", "--fields=2",
              "--input=motion_filtered_2", "--output=motion_residue_2"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "motion_residue_2").read_bytes() == KAT[f"bidir_{i}"].astype("<i2").tobytes()


def test_interlevel_cli_missing_reference_reads_zeros(tmp_path):
    a = KAT["pred_0"]
    BY, BX = a.shape[2:]
    a.astype("<i2").tofile(tmp_path / "p")
    r = _mctf("interlevel_motion_decorrelate", [f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", "--fields_in_predicted=4",
              "--predicted=p", "--reference=absent", "--residue=r"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "r").read_bytes() == a.astype("<i2").tobytes()  # /dev/zero fallback: residue == predicted


# ---- the decoder's side (motion_expand.py:131-170): the same sources built without -D ANALYZE -----------------------------
@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_correlators_vs_reference_binaries(i):
    if f"corr_inter_{i}" in KAT:
        top = motion.bidirectional_correlate(KAT[f"ref_{i}"])
        assert top.dtype == np.int16 and np.array_equal(top, KAT[f"corr_bidir_{i}"])
        assert np.array_equal(motion.interlevel_correlate(KAT[f"pred_{i}"], top), KAT[f"corr_inter_{i}"])
    else:
        assert np.array_equal(motion.bidirectional_correlate(KAT[f"pred_{i}"]), KAT[f"corr_bidir_{i}"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "mctf_*.npz"))))
def test_field_chain_undoes_the_residue_chain_on_golden_fields(path):
    g = np.load(path)
    TRLs = int(g["meta"][3])
    res = {t: g[f"motion_residue_{t}"] for t in range(1, TRLs)}
    fields = motion.field_chain(res, TRLs)
    for t in range(1, TRLs):
        assert np.array_equal(fields[t], g[f"motion_filtered_{t}"]), (path, t)


def test_mux_undoes_demux():
    rng = np.random.default_rng(4)
    r = rng.integers(-32768, 32768, size=(3, 4, 5, 7)).astype(np.int16)
    assert np.array_equal(motion.mux_components(motion.demux_components(r)), r)


def test_correlator_clis_write_the_reference_files(tmp_path):
    i = 1  # 5 residue fields over 2 reference fields: the fifth reuses the last reference read
    a, b = KAT[f"pred_{i}"], KAT[f"ref_{i}"]
    BY, BX = a.shape[2:]
    a.astype("<i2").tofile(tmp_path / "motion_residue_1")
    b.astype("<i2").tofile(tmp_path / "motion_residue_2")
    r = _mctf("bidirectional_motion_correlate", [f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", "--fields=2",
              "--input=motion_residue_2", "--output=motion_2"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "motion_2").read_bytes() == KAT[f"corr_bidir_{i}"].astype("<i2").tobytes()
    r = _mctf("interlevel_motion_correlate", [f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", "--fields_in_predicted=5",
              "--reference=motion_2", "--predicted=motion_1", "--residue=motion_residue_1"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "motion_1").read_bytes() == KAT[f"corr_inter_{i}"].astype("<i2").tobytes()
