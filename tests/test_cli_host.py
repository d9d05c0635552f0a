"""CPU: host-side logic of the drop-in entry points -- schedules, flag semantics, size index."""
import json
import os

import pytest

from helpers import GOLDEN
from qsvc_amd import schedule
from qsvc_amd.cli import common as C
from qsvc_amd.cli.texture_compress_fb_j2k import header_size


def test_slope_table_known_answers():
    # TRLs=4, q=45000, nLayers=5 (texture_compress.py:114-165)
    s = schedule.slope_table(4, 45000, 5)
    assert s[0] == [45000, 46000, 47000, 48000, 49000]
    assert s[1][0] == 45000 + round(181.019336 * 1.1598810146) == 45210
    assert [r[0] for r in s] == [45000, 45210, 45384, 45573]
    assert schedule.slope_table(2, 44000, 1) == [[44000], [44000 + 226]]
    assert schedule.slope_table(1, 45000, 3, 500) == [[45000, 45500, 46000]]
    with pytest.raises(ValueError):
        schedule.slope_table(9, 45000, 5)


def test_level_schedule_and_pictures():
    assert schedule.pictures(2, 4) == 17 and schedule.gop_size(4) == 8
    assert schedule.level_schedule(2, 4, 8, 8, 4) == [(1, 17, 4, 8), (2, 9, 8, 8), (3, 5, 16, 8)]
    # block size halves per level down to block_size_min; search range doubles up to 128 (analyze.py:142-151)
    assert schedule.level_schedule(1, 5, 64, 16, 32) == [(1, 17, 32, 64), (2, 9, 64, 32), (3, 5, 128, 16), (4, 3, 128, 16)]
    assert schedule.default_block_size(352, 288) == 32 and schedule.default_block_size(1920, 1080) == 64


def test_flag_semantics_like_parse_known_args():
    spec = {"block_size": (32, int), "always_B": (0, int), "update_factor": (0.25, float), "even_fn": ("even", str)}
    a = C.parse("x", spec, ["--block_size=16", "--always_B=0", "--unknown_flag=7", "--update_factor", "0"])
    assert a == {"block_size": 16, "always_B": 0, "update_factor": 0.0, "even_fn": "even"}


def test_header_size_matches_reference_tool_goldens():
    for g in json.load(open(os.path.join(GOLDEN, "header_size.json"))):
        assert header_size(bytes.fromhex(g["hex"])) == g["out"]


def test_interleave_is_split_inverse():
    import numpy as np
    low = np.arange(5 * 6, dtype=np.uint8).reshape(5, 6)
    np.testing.assert_array_equal(C.interleave(low[0::2], low[1::2]), low)


def test_compress_refuses_the_shrinking_block_motion_chain(tmp_path):
    """compress.py:197-212 always runs motion_compress; with --block_size > --block_size_min and more than one interlevel
    iteration the reference's loop (motion_compress.py:159-166) garbles the fields.  This path refuses instead of leaving
    a code-stream without motion_residue_* behind -- before any GPU work, with the drivers' exit status 255."""
    import subprocess, sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    np.zeros(9 * 64 * 64 * 3 // 2, dtype=np.uint8).tofile(tmp_path / "low_0")
    r = subprocess.run([os.path.join(root, "bin", "mctf"), "compress", "--pixels_in_x=64", "--pixels_in_y=64", "--TRLs=4", "--GOPs=1",
                        "--block_size=32", "--block_size_min=16"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 255, (r.returncode, r.stderr[-500:])
    assert "block_size_min" in r.stderr
    assert not list(tmp_path.glob("*.j2c")) and not list(tmp_path.glob("frame_types_*"))


def test_residue_chain_single_iteration_keeps_the_level1_grid():
    import numpy as np
    from qsvc_amd import motion
    f = np.arange(2 * 4 * 3 * 5, dtype=np.int16).reshape(2, 4, 3, 5)
    res = motion.residue_chain({1: f}, 2, 32, 16)  # TRLs=2: only the bidirectional step (motion_compress.py:176-186)
    np.testing.assert_array_equal(res[1], motion.bidirectional_decorrelate(f))
    with pytest.raises(NotImplementedError):
        motion.residue_chain({1: f, 2: f[:1]}, 3, 32, 16)
