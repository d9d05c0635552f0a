"""`mctf interlevel_motion_decorrelate`, `bidirectional_motion_decorrelate`, `motion_compress`, `motion_compress_j2k`
(SURVEY.md 8f N1).  Flags and files as the reference tools (interlevel_motion_decorrelate.cpp main,
bidirectional_motion_decorrelate.cpp main, motion_compress.py:60-212, motion_compress_j2k.py:44-178)."""
import os

import numpy as np

from .. import motion, schedule
from . import common as C
from .texture_compress_fb_j2k import header_size


def _fields(fn, n, BY, BX, missing_zero=False, whole_only=False):
    """Up to n fields of fn.  whole_only: stop at the last complete field (the C++ tools stop at feof)."""
    if missing_zero and not os.path.exists(fn):
        return np.zeros((0, 4, BY, BX), dtype=np.int16)
    if whole_only and os.path.exists(fn):
        n = min(n, os.path.getsize(fn) // (4 * BY * BX * 2))
    return C.read_motion(fn, n, BY, BX)


def interlevel_motion_decorrelate(argv=None):
    a = C.parse("Interlevel decorrelation of the motion information.", {
        "blocks_in_x": (11, int), "blocks_in_y": (9, int), "fields_in_predicted": (2, int), "predicted": ("", str),
        "reference": ("", str), "residue": ("", str)}, argv)
    BY, BX = a["blocks_in_y"], a["blocks_in_x"]
    # the tool pairs two predicted fields with each reference field and stops at the end of the predicted file
    # (interlevel_motion_decorrelate.cpp:249-283), so --fields_in_predicted only bounds the work from above
    pred = _fields(a["predicted"], 2 * a["fields_in_predicted"], BY, BX, whole_only=True)
    ref = _fields(a["reference"], a["fields_in_predicted"], BY, BX, missing_zero=True, whole_only=True)
    C.write_motion(a["residue"], motion.interlevel_decorrelate(pred, ref))


def bidirectional_motion_decorrelate(argv=None):
    a = C.parse("Bidirectional decorrelation of the motion information.", {
        "blocks_in_x": (11, int), "blocks_in_y": (9, int), "fields": (1, int), "input": ("", str), "output": ("", str)}, argv)
    f = _fields(a["input"], a["fields"], a["blocks_in_y"], a["blocks_in_x"])
    C.write_motion(a["output"], motion.bidirectional_decorrelate(f))


def encode_residue(e, file, residue):
    """motion_compress_j2k.py:98-178: <file>_comp{0..3}_NNNN.j2c + cumulative size index <file>.mjc."""
    from .. import hip
    F, _, BY, BX = residue.shape
    comps = motion.demux_components(residue)                      # [4][F][BY][BX]
    planes = comps.transpose(1, 0, 2, 3).reshape(F * 4, BY, BX)   # (field, component) order for one batch
    streams = e.j2k_encode_planes_s16(planes, hip.j2k_params(reversible=True, levels=0, layers=1)) if F else []
    total = 0
    with open(file + ".mjc", "w") as idx:
        for n in range(F):
            for c in range(4):
                cs = streams[n * 4 + c]
                with open(f"{file}_comp{c}_{n:04d}.j2c", "wb") as f:
                    f.write(cs)
                total += len(cs) - header_size(cs)
            idx.write(str(total) + "\n")
    return streams


def motion_compress_j2k(argv=None):
    a = C.parse("Compress the motion data using JPEG 2000.", {
        "blocks_in_x": (0, int), "blocks_in_y": (0, int), "fields": (0, int), "quantization": ("45000", str), "clayers": ("1", str),
        "file": ("", str), "iteration": (1, int)}, argv)
    from .. import hip
    res = _fields(a["file"], a["fields"], a["blocks_in_y"], a["blocks_in_x"])
    e = C.engine()
    try:
        encode_residue(e, a["file"], res)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def run_motion_compress(e, W, H, TRLs, GOPs, block_size, block_size_min, filtered=None):
    """motion_compress.py:146-212 with the fields given (device results) or read from motion_filtered_t."""
    if os.environ.get("MCTF_MOTION_CODEC", "j2k") != "j2k":
        C.die("only the j2k motion codec is implemented on this path")
    if block_size < block_size_min:
        block_size_min = block_size
    BY, BX = H // block_size, W // block_size
    fields = schedule.pictures(GOPs, TRLs) // 2
    if filtered is None:
        filtered, f = {}, fields
        for it in range(1, TRLs):
            filtered[it] = _fields(f"motion_filtered_{it}", f, BY, BX)
            f //= 2
    try:
        res = motion.residue_chain(filtered, TRLs, block_size, block_size_min)
    except NotImplementedError as ex:
        C.die(str(ex))
    for it in range(1, TRLs):
        C.write_motion(f"motion_residue_{it}", res[it])
        encode_residue(e, f"motion_residue_{it}", res[it])
    return res


def motion_compress(argv=None):
    a = C.parse("Compress the motion data.", {
        "pixels_in_x": (352, int), "pixels_in_y": (288, int), "block_size": (None, int), "block_size_min": (None, int),
        "GOPs": (1, int), "TRLs": (4, int), "quantization": ("45000", str), "clayers": ("1", str), "SRLs": (5, int)}, argv)
    d = schedule.default_block_size(a["pixels_in_x"], a["pixels_in_y"])  # motion_compress.py:99-102 (after reading the size)
    bs = a["block_size"] or d
    bsm = a["block_size_min"] or d
    from .. import hip
    e = C.engine()
    try:
        run_motion_compress(e, a["pixels_in_x"], a["pixels_in_y"], a["TRLs"], a["GOPs"], bs, bsm)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()
