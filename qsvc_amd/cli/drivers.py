"""In-process, device-resident versions of the reference's Python drivers.  Each keeps the flags, defaults and
cwd files of the script it replaces, but the stages it would spawn as processes run back to back on the GPU.

  compress ............ trunk/src/compress.py:57-228
  analyze ............. trunk/src/analyze.py:30-153
  analyze_step ........ trunk/src/analyze_step.py:30-232
  texture_compress .... trunk/src/texture_compress.py:30-217
  synthesize_step ..... trunk/src/synthesize_step.py:30-151
  synthesize .......... trunk/src/synthesize.py:30-170
"""
from __future__ import annotations

import os

import numpy as np

from .. import schedule
from . import common as C
from .texture_compress_fb_j2k import write_streams

COMMON = {"pixels_in_x": (352, int), "pixels_in_y": (288, int), "always_B": (0, int), "block_overlaping": (0, int),
          "block_size": (None, int), "block_size_min": (None, int), "border_size": (0, int), "GOPs": (1, int),
          "search_range": (4, int), "subpixel_accuracy": (0, int), "TRLs": (4, int), "update_factor": (0.25, float)}
EXT = {"reversible": (0, int), "lossless": (0, int), "base_step": (1.0 / 256, float), "keep_intermediates": (1, int)}


def _write_level(e, t, P, bs, W, H, low_prev, want_pred=True):
    """The cwd files one analyze_step leaves behind (analyze_step.py:117-232)."""
    from .. import hip
    nB = P // 2
    C.write_frames(f"even_{t}", low_prev[0:2 * nB + 1:2])
    C.write_frames(f"odd_{t}", low_prev[1:2 * nB + 1:2])
    C.write_motion(f"motion_{t}", e.get_motion(hip.MOTION_RAW, t, nB, bs))
    C.write_motion(f"motion_filtered_{t}", e.get_motion(hip.MOTION_FILTERED, t, nB, bs))
    C.write_frames(f"high_{t}", e.get_frames(hip.HIGH, t, 0, nB))
    with open(f"frame_types_{t}", "wb") as f:
        f.write(e.get_frame_types(t, nB))
    low = e.get_frames(hip.LOW, t, 0, nB + 1)
    C.write_frames(f"low_{t}", low)
    if want_pred:
        C.write_frames(f"prediction_even_{t}", e.get_frames(hip.PRED, t, 0, nB))
    return low


def analyze_step(argv=None):
    a = C.parse("Performs a temporal analysis step.", dict(COMMON, pictures=(9, int), temporal_subband=(1, int)), argv)
    from .. import hip
    W, H, t, P = a["pixels_in_x"], a["pixels_in_y"], a["temporal_subband"], a["pictures"]
    bs = a["block_size"] or schedule.default_block_size(W, H)  # analyze_step.py:92-96
    low_prev = C.read_frames(f"low_{t - 1}", W, H, 2 * (P // 2) + 1)
    e = C.engine()
    try:
        e.seq_config(W, H, keep_prediction=True)
        C.apply_mctf_options(e, a)
        e.put_frames(hip.LOW, t - 1, low_prev)
        e.analyze_step(t, low_prev.shape[0], bs, a["search_range"], a["always_B"], a["update_factor"])
        _write_level(e, t, low_prev.shape[0], bs, W, H, low_prev)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def _load_subband(e, t, nB, bs, W, H, with_low):
    """The cwd files one inverse level reads (synthesize_step.py:96-136): low_t, high_t, motion_t, frame_types_t."""
    from .. import hip
    if with_low:
        e.put_frames(hip.LOW, t, C.read_frames(f"low_{t}", W, H, nB + 1))
    e.put_frames(hip.HIGH, t, C.read_frames(f"high_{t}", W, H, nB))
    e.put_motion(hip.MOTION_FILTERED, t, C.read_motion(f"motion_{t}", nB, H // bs, W // bs), bs)
    try:
        types = open(f"frame_types_{t}", "rb").read(nB)
    except OSError:
        C.die(f"unable to read \"frame_types_{t}\" ... aborting!")
    e.put_frame_types(t, types.ljust(nB, b"I"))


def _write_synth_level(e, t, P):
    from .. import hip
    low = e.get_frames(hip.LOW, t - 1, 0, P)
    C.write_frames(f"even_{t}", low[0::2])
    C.write_frames(f"odd_{t}", low[1::2])
    C.write_frames(f"low_{t - 1}", low)


def synthesize_step(argv=None):
    a = C.parse("Performs a step of the temporal synthesis.", {
        "block_overlaping": (0, int), "block_size": (16, int), "pictures": (33, int), "pixels_in_x": (352, int),
        "pixels_in_y": (288, int), "search_range": (4, int), "subpixel_accuracy": (0, int), "temporal_subband": (0, int),
        "update_factor": (1.0 / 4, float)}, argv)
    from .. import hip
    W, H, t, bs = a["pixels_in_x"], a["pixels_in_y"], a["temporal_subband"], a["block_size"]
    nB = a["pictures"] // 2
    e = C.engine()
    try:
        e.seq_config(W, H)
        C.apply_mctf_options(e, a)
        _load_subband(e, t, nB, bs, W, H, with_low=True)
        e.synthesize_step(t, 2 * nB + 1, bs, a["search_range"], a["update_factor"])
        _write_synth_level(e, t, 2 * nB + 1)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def synthesize(argv=None):
    """synthesize.py:103-170: from the coarsest subband down; the per-level lists (block size, picture size) of the
    reference's spatially scalable decoder are accepted, a change of picture size between levels is not."""
    a = C.parse("Performs the temporal synthesis of a picture sequence.", {
        "GOPs": (1, int), "TRLs": (4, int), "block_size": ("16,16,16,16", str), "pixels_in_x": ("352,352,352,352,352", str),
        "pixels_in_y": ("288,288,288,288,288", str), "subpixel_accuracy": ("0,0,0,0", str), "border_size": (0, int),
        "block_overlaping": (0, int), "search_range": (4, int), "update_factor": (1.0 / 4, float)}, argv)
    if a["TRLs"] < 2:
        return
    from .. import hip
    e = C.engine()
    try:
        run_synthesize(e, a)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def run_synthesize(e, a):
    """The level loop of synthesize.py:103-170 on an open engine (also the last stage of `expand`, cli/expand_tools.py)."""
    TRLs = a["TRLs"]

    def per_level(text, idx):  # the reference indexes its comma lists; a single value stands for every level
        parts = str(text).split(",")
        return int(parts[idx] if len(parts) > 1 else parts[0])

    sizes = {(per_level(a["pixels_in_x"], TRLs - t), per_level(a["pixels_in_y"], TRLs - t)) for t in range(1, TRLs)}
    if len(sizes) != 1:
        C.die("a different picture size per temporal subband (spatial scalability) is not implemented on this path")
    W, H = sizes.pop()
    P0 = schedule.pictures(a["GOPs"], TRLs)
    e.seq_config(W, H)
    for t in range(TRLs - 1, 0, -1):
        # synthesize.py:133: --subpixel_accuracy is a list indexed like the picture sizes, one entry per temporal subband
        C.apply_mctf_options(e, dict(a, subpixel_accuracy=per_level(a.get("subpixel_accuracy", 0) or 0, TRLs - t)))
        P, S = P0, a["search_range"]
        for _ in range(1, t):  # synthesize.py:118-124,160-168
            S = min(S * 2, 128)
            P = (P + 1) // 2
        bs = per_level(a["block_size"], (TRLs - 1) - t)
        _load_subband(e, t, P // 2, bs, W, H, with_low=(t == TRLs - 1))
        e.synthesize_step(t, P, bs, S, a["update_factor"])
        _write_synth_level(e, t, 2 * (P // 2) + 1)


def _run_analyze(e, a, low0, keep=True):
    from .. import hip
    W, H = a["pixels_in_x"], a["pixels_in_y"]
    e.put_frames(hip.LOW, 0, low0)
    low_prev = low0
    levels = schedule.level_schedule(a["GOPs"], a["TRLs"], a["block_size"], a["block_size_min"], a["search_range"])
    for (t, P, S, bs) in levels:
        e.analyze_step(t, P, bs, S, a["always_B"], a["update_factor"])
        if keep:
            low_prev = _write_level(e, t, P, bs, W, H, low_prev)
        else:
            with open(f"frame_types_{t}", "wb") as f:
                f.write(e.get_frame_types(t, P // 2))
            C.write_motion(f"motion_filtered_{t}", e.get_motion(hip.MOTION_FILTERED, t, P // 2, bs))
    return levels


def _resolve_blocks(a):
    # analyze.py:76-83 decides the default from ITS OWN default picture size (352x288 -> 32) because the test runs
    # before --pixels_in_* are read; compress.py:139-147 decides from the real size and always passes --block_size.
    d = a.pop("_default_bs")
    if a["block_size"] is None:
        a["block_size"] = d
    if a["block_size_min"] is None:
        a["block_size_min"] = d


def analyze(argv=None):
    a = C.parse("Performs the temporal analysis of a picture sequence.", dict(COMMON, **{"update_factor": (0.0, float)}), argv)
    a["_default_bs"] = 32
    _resolve_blocks(a)
    from .. import hip
    W, H = a["pixels_in_x"], a["pixels_in_y"]
    low0 = C.read_frames("low_0", W, H, schedule.pictures(a["GOPs"], a["TRLs"]))
    e = C.engine()
    try:
        e.seq_config(W, H, keep_prediction=True)
        C.apply_mctf_options(e, a)
        _run_analyze(e, a, low0)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def _texture(e, a, device=True):
    """texture_compress.py:114-217 on device-resident (or freshly uploaded) subbands."""
    from .. import hip
    TRLs, GOPs, W, H = a["TRLs"], a["GOPs"], a["pixels_in_x"], a["pixels_in_y"]
    codec = os.environ.get("MCTF_TEXTURE_CODEC", "j2k")
    if codec != "j2k":
        C.die(f"texture codec '{codec}' is not implemented on this path (only j2k)")
    slopes = schedule.slope_table(TRLs, a["quantization_texture"], a["nLayers"], a["quantization_step"])
    with open("slopes", "w") as f:  # texture_compress.py:174-176
        f.write("\n".join(map(str, slopes)))
    lossless = bool(a.get("lossless"))
    def prm(sub):
        if lossless:
            return hip.j2k_params(reversible=True, levels=max(a["SRLs"] - 1, 0), layers=1)
        return hip.j2k_params(reversible=bool(a.get("reversible")), levels=max(a["SRLs"] - 1, 0), layers=a["nLayers"],
                              slopes=slopes[sub], base_step=a.get("base_step", 1.0 / 256))
    P = schedule.pictures(GOPs, TRLs)
    items, names = [], []
    for t in range(1, TRLs):
        P = (P + 1) // 2
        if not device:
            e.put_frames(hip.HIGH, t, C.read_frames(f"high_{t}", W, H, P - 1))
        items.append((hip.HIGH, t, 0, P - 1, prm(TRLs - t)))
        names.append(f"high_{t}")
    if not device:
        e.put_frames(hip.LOW, TRLs - 1, C.read_frames(f"low_{TRLs - 1}", W, H, P))
    items.append((hip.LOW, TRLs - 1, 0, P, prm(0)))
    names.append(f"low_{TRLs - 1}")
    keep = [(n, it) for n, it in zip(names, items) if it[3] > 0]
    streams = e.j2k_encode_multi([it for _, it in keep])
    for (n, _), s in zip(keep, streams):
        write_streams(n, s)


TEXTURE = {"GOPs": (1, int), "nLayers": (5, int), "pixels_in_x": (352, int), "pixels_in_y": (288, int),
           "quantization": ("45000", str), "quantization_step": (0, int), "TRLs": (4, int), "SRLs": (5, int)}


def texture_compress(argv=None):
    a = C.parse("Compress the texture.", dict(TEXTURE, **EXT), argv)
    a["quantization_texture"] = int(str(a["quantization"]).replace('"', ""))
    from .. import hip
    e = C.engine()
    try:
        e.seq_config(a["pixels_in_x"], a["pixels_in_y"])
        _texture(e, a, device=False)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def compress(argv=None):
    spec = dict(COMMON, clayers_motion=(0, str), quantization_step=(0, int), quantization_motion=("45000", str),
                quantization_texture=("45000", str), SRLs=(5, int), nLayers=(5, int), **EXT)
    a = C.parse("Encodes a sequence of pictures.", spec, argv)
    a["_default_bs"] = schedule.default_block_size(a["pixels_in_x"], a["pixels_in_y"])  # compress.py:139-142
    _resolve_blocks(a)
    a["quantization_texture"] = int(str(a["quantization_texture"]).replace('"', ""))
    if a["block_size"] > a["block_size_min"] and a["TRLs"] > 2:
        # motion_compress.py:159-166 re-reads the levels with the wrong block grid when the block size shrinks (its loop
        # control sits inside the `if`): the reference then emits garbled motion streams.  Not reproduced; refuse loudly
        # rather than leave a code-stream without motion_residue_* / *.mjc behind.
        C.die("--block_size > --block_size_min with TRLs > 2: the motion-field coding of a shrinking block size "
              "(motion_compress.py:159-166) is not implemented on this path")
    from .. import hip
    W, H = a["pixels_in_x"], a["pixels_in_y"]
    low0 = C.read_frames("low_0", W, H, schedule.pictures(a["GOPs"], a["TRLs"]))
    e = C.engine()
    try:
        e.seq_config(W, H, keep_prediction=bool(a["keep_intermediates"]))
        C.apply_mctf_options(e, a)
        e.put_frames(hip.LOW, 0, low0)
        if a["TRLs"] > 1:  # compress.py:176-212
            _run_analyze(e, a, low0, keep=bool(a["keep_intermediates"]))
            from .motion_tools import run_motion_compress  # compress.py:197-212
            filt = {t: e.get_motion(hip.MOTION_FILTERED, t, P // 2, bs)
                    for (t, P, S, bs) in schedule.level_schedule(a["GOPs"], a["TRLs"], a["block_size"], a["block_size_min"], a["search_range"])}
            run_motion_compress(e, W, H, a["TRLs"], a["GOPs"], a["block_size"], a["block_size_min"], filtered=filt)
        _texture(e, a, device=True)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()
