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
