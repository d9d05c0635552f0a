"""Motion-field coding chain of `mctf compress` (compress.py:197-212 -> motion_compress.py ->
motion_compress_j2k.py), SURVEY.md section 8(f) N1.  The fields are kilobytes: everything here is host NumPy except the
lossless JPEG 2000 coding of the residue planes, which goes through the C ABI (qsvc_j2k_encode_planes_s16).

  interlevel .... residue_i = field_i - tdiv(field_{i+1}, 2), two consecutive fields share one reference field
                  (interlevel_motion_decorrelate.cpp:32-70, main :249-283)
  bidirectional . NEXT -= PREV on the coarsest level only (bidirectional_motion_decorrelate.cpp:25-52,
                  motion_compress.py:176-186)
  demux ......... the reference cuts the planar file as if it were 8-byte interleaved records
                  (motion_compress_j2k.py:103-107): "component" c of chunk n = every 4th int16, phase c, of field n
"""
from __future__ import annotations

import numpy as np


def tdiv2(a: np.ndarray) -> np.ndarray:
    """C integer division by 2 (truncation toward zero) on int arrays."""
    a = a.astype(np.int32)
    return np.where(a >= 0, a // 2, -((-a) // 2))


def interlevel_decorrelate(predicted: np.ndarray, reference: np.ndarray) -> np.ndarray:
    """predicted [F][4][BY][BX], reference [F/2 (or fewer)][4][BY][BX] -> residue like predicted.  A missing reference file
    reads as zeros in the reference (/dev/zero fallback, interlevel_motion_decorrelate.cpp:229-236)."""
    F = predicted.shape[0]
    out = np.empty_like(predicted, dtype=np.int16)
    for k in range(F):
        r = k // 2
        ref = reference[r] if r < reference.shape[0] else (reference[-1] if reference.shape[0] else np.zeros_like(predicted[0]))
        out[k] = (predicted[k].astype(np.int32) - tdiv2(ref)).astype(np.int16)
    return out


def bidirectional_decorrelate(fields: np.ndarray) -> np.ndarray:
    out = fields.astype(np.int16).copy()
    out[:, 2] = (out[:, 2].astype(np.int32) - out[:, 0]).astype(np.int16)  # NEXT-X -= PREV-X
    out[:, 3] = (out[:, 3].astype(np.int32) - out[:, 1]).astype(np.int16)  # NEXT-Y -= PREV-Y
    return out


def interlevel_correlate(residue: np.ndarray, reference: np.ndarray) -> np.ndarray:
    """The decoder's counterpart (interlevel_motion_decorrelate.cpp:55-59 without ANALYZE, main :249-283): field_k = residue_k +
    tdiv(reference_{k/2}, 2), int16 wrap; reference fields past the end of the file keep the last one read, a missing file is zeros."""
    F = residue.shape[0]
    out = np.empty_like(residue, dtype=np.int16)
    for k in range(F):
        r = k // 2
        ref = reference[r] if r < reference.shape[0] else (reference[-1] if reference.shape[0] else np.zeros_like(residue[0]))
        out[k] = (residue[k].astype(np.int32) + tdiv2(ref)).astype(np.int16)
    return out


def bidirectional_correlate(fields: np.ndarray) -> np.ndarray:
    """NEXT += PREV (bidirectional_motion_decorrelate.cpp:42-44 without ANALYZE)."""
    out = fields.astype(np.int16).copy()
    out[:, 2] = (out[:, 2].astype(np.int32) + out[:, 0]).astype(np.int16)
    out[:, 3] = (out[:, 3].astype(np.int32) + out[:, 1]).astype(np.int16)
    return out


def mux_components(comps: np.ndarray) -> np.ndarray:
    """[4 comps][F][BY][BX] -> [F][4][BY][BX] planar fields: the 8-byte records motion_expand_j2k.py:115-141 writes
    (2 bytes of each component in turn), i.e. demux_components undone."""
    _, F, BY, BX = comps.shape
    flat = np.empty((F, 4 * BY * BX), dtype=np.int16)
    for c in range(4):
        flat[:, c::4] = comps[c].reshape(F, BY * BX)
    return flat.reshape(F, 4, BY, BX)


def field_chain(residues: dict, TRLs: int) -> dict:
    """{t: motion_residue_t} -> {t: motion_t} as motion_expand.py:131-170 chains the correlators: bidirectional on the coarsest
    level, then interlevel from the level above down to level 1."""
    top = TRLs - 1
    fields = {top: bidirectional_correlate(residues[top])}
    for it in range(top - 1, 0, -1):
        fields[it] = interlevel_correlate(residues[it], fields[it + 1])
    return fields


def demux_components(residue: np.ndarray) -> np.ndarray:
    """[F][4][BY][BX] planar int16 -> [4 comps][F][BY][BX] as `demux 8 2c 2 | split --bytes=BY*BX*2` cuts it."""
    F, _, BY, BX = residue.shape
    flat = residue.reshape(F, 4 * BY * BX)
    return np.stack([flat[:, c::4].reshape(F, BY, BX) for c in range(4)])


def residue_chain(filtered: dict, TRLs: int, block_size: int, block_size_min: int) -> dict:
    """filtered: {t: motion_filtered_t}.  -> {t: motion_residue_t} following the loop of motion_compress.py:146-186 for
    the case block_size == block_size_min (the only one the reference's schedule produces through `compress`)."""
    iterations = TRLs - 1
    if block_size > block_size_min and iterations > 1:  # one iteration: the loop of :140-166 never runs, the grid is level 1's
        raise NotImplementedError("motion_compress with a shrinking block size (motion_compress.py:159-166 quirk) is not reproduced")
    res = {}
    for it in range(1, iterations):
        res[it] = interlevel_decorrelate(filtered[it], filtered[it + 1])
    res[iterations] = bidirectional_decorrelate(filtered[iterations])
    return res
