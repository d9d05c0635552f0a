"""Host-side schedules of the reference's Python drivers, restated for Python 3.

  GOP size / picture count ...... GOP.py:23-24, analyze.py:110-112
  per-level (pictures, search range, block size) ... analyze.py:121-153
  default block size ............ compress.py:139-147
  slope table ................... texture_compress.py:114-165
"""
from __future__ import annotations

import math

SEARCH_RANGE_MAX = 128  # analyze.py:144-147
RESOLUTION_FHD = 1920 * 1080

# GAINS[TRLs] = temporal-subband energy gains [L/H_{T-1}, ..., L/H_1] (texture_compress.py:114-130)
GAINS = {
    2: [1.2460784922],
    3: [1.8652117304, 1.2500103877],
    4: [1.1598810146, 2.1224082769, 3.1669663339],
    5: [1.0877939347, 2.1250255455, 3.8884779989, 5.8022196044],
    6: [1.0456562538, 2.0788785438, 4.0611276369, 7.4312544148, 11.0885981772],
    7: [1.0232370223, 2.0434169985, 4.0625355976, 7.9362383342, 14.5221257323, 21.6692913386],
    8: [1.0117165706, 2.0226778348, 4.0393126714, 8.0305936232, 15.6879129862, 28.7065276104, 42.8346456693],
}


def gop_size(TRLs: int) -> int:
    return 2 ** (TRLs - 1)


def pictures(GOPs: int, TRLs: int) -> int:
    return GOPs * gop_size(TRLs) + 1


def default_block_size(pixels_in_x: int, pixels_in_y: int) -> int:
    return 32 if pixels_in_x * pixels_in_y < RESOLUTION_FHD else 64


def level_schedule(GOPs, TRLs, block_size, block_size_min, search_range):
    """[(t, pictures, search_range, block_size)] for t = 1..TRLs-1 (analyze.py:110-153)."""
    P = pictures(GOPs, TRLs)
    if block_size < block_size_min:
        block_size_min = block_size
    out = []
    for t in range(1, TRLs):
        out.append((t, P, search_range, block_size))
        P = (P + 1) // 2
        search_range = min(search_range * 2, SEARCH_RANGE_MAX)
        block_size = max(block_size // 2, block_size_min)
    return out


def _py2_round(x: float) -> int:
    """Python 2's round(): half away from zero (the reference ran under Python 2)."""
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def slope_table(TRLs, quantization, nLayers, quantization_step=0):
    """SLOPES[subband][layer], subbands ordered [L, H_{T-1}, ..., H_1] (texture_compress.py:140-165)."""
    range_quantization = 46000.0 - 42000.0
    if TRLs > 1 and TRLs not in GAINS:
        raise ValueError(f"Gains are not available for {TRLs} TRLs")
    if quantization_step == 0 and nLayers > 1:
        quantization_step = _py2_round(range_quantization / (nLayers - 1))
    step_subband = 256 / math.sqrt(2)
    slopes = [[int(quantization)]]
    for sub in range(TRLs - 1):
        slopes.append([_py2_round(slopes[0][0] + step_subband * GAINS[TRLs][sub])])
    for sub in range(TRLs):
        for layer in range(nLayers - 1):
            slopes[sub].append(_py2_round(slopes[sub][layer] + quantization_step))
    return slopes
