"""The decoder chain of `mctf expand`, in process and on the GPU where the reference spawns one process per
stage and one `kdu_expand` per picture and component:

  expand ............................ trunk/src/expand.py:40-256 (texture_expand, motion_expand, synthesize)
  texture_expand .................... trunk/src/texture_expand.py:30-131 (-> texture_expand_fb_j2k per temporal subband)
  motion_expand ..................... trunk/src/motion_expand.py:56-179 (-> motion_expand_j2k per level, then the correlators)
  motion_expand_j2k ................. trunk/src/motion_expand_j2k.py:36-149 (kdu_expand per component plane, 8-byte record mux)
  bidirectional_motion_correlate .... trunk/src/bidirectional_motion_decorrelate.cpp without -D ANALYZE (:25-52, main)
  interlevel_motion_correlate ....... trunk/src/interlevel_motion_decorrelate.cpp without -D ANALYZE (:32-70, main :249-283)

Flags, defaults, list arguments and cwd files as the reference tools; the motion fields are kilobytes and stay host NumPy
(qsvc_amd/motion.py), the JPEG 2000 decoding of textures and motion residues is one batched GPU call per tool (or one for the
whole of `expand`)."""
from __future__ import annotations

import os

import numpy as np

from .. import motion, schedule
from . import common as C
from .motion_tools import _fields
from .texture_expand_fb_j2k import expand_file


def bidirectional_motion_correlate(argv=None):
    a = C.parse("Bidirectional correlation of the motion information.", {
        "blocks_in_x": (11, int), "blocks_in_y": (9, int), "fields": (1, int), "input": ("", str), "output": ("", str)}, argv)
    f = _fields(a["input"], a["fields"], a["blocks_in_y"], a["blocks_in_x"])
    C.write_motion(a["output"], motion.bidirectional_correlate(f))


def interlevel_motion_correlate(argv=None):
    a = C.parse("Interlevel correlation of the motion information.", {
        "blocks_in_x": (11, int), "blocks_in_y": (9, int), "fields_in_predicted": (1, int), "predicted": ("", str),
        "reference": ("", str), "residue": ("", str)}, argv)
    BY, BX = a["blocks_in_y"], a["blocks_in_x"]
    # one reference field per two residue fields, until the residue file ends (interlevel_motion_decorrelate.cpp:249-283)
    res = _fields(a["residue"], 2 * a["fields_in_predicted"], BY, BX, whole_only=True)
    ref = _fields(a["reference"], a["fields_in_predicted"], BY, BX, missing_zero=True, whole_only=True)
    C.write_motion(a["predicted"], motion.interlevel_correlate(res, ref))


def _residue_names(file, fields):
    return [f"{file}_comp{c}_{n:04d}.j2c" for c in range(4) for n in range(fields)]


def _residue_from_planes(planes_by_name, file, fields, BY, BX):
    """motion_expand_j2k.py:84-141: a missing component plane is a plane of zeros ("linear motion"); the four components of a field are
    multiplexed back into 8-byte records."""
    comps = np.zeros((4, fields, BY, BX), dtype=np.int16)
    for c in range(4):
        for n in range(fields):
            p = planes_by_name.get(f"{file}_comp{c}_{n:04d}.j2c")
            if p is None:
                continue
            if p.shape != (BY, BX) or p.dtype != np.int16:
                C.die(f"{file}_comp{c}_{n:04d}.j2c holds a {p.shape[1]}x{p.shape[0]} {p.dtype} plane, expected {BX}x{BY} int16")
            comps[c, n] = p
    return motion.mux_components(comps)


def _decode_named(e, names):
    present = [n for n in names if os.path.exists(n)]
    if not present:
        return {}
    return dict(zip(present, e.j2k_decode_planes([open(n, "rb").read() for n in present])))


def motion_expand_j2k(argv=None):
    a = C.parse("Expands the motion data using JPEG 2000.", {
        "blocks_in_x": (0, int), "blocks_in_y": (0, int), "fields": (0, int), "file": ("", str)}, argv)
    from .. import hip
    e = C.engine()
    try:
        planes = _decode_named(e, _residue_names(a["file"], a["fields"]))
        C.write_motion(a["file"], _residue_from_planes(planes, a["file"], a["fields"], a["blocks_in_y"], a["blocks_in_x"]))
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


# ---- list arguments (expand.py:143-207) ---------------------------------------------------------------------------------
def _lists(a, TRLs):
    """block_size per level (TRLs-1 entries, halving towards block_size_min), pixels_in_x/y and subpixel_accuracy (TRLs entries):
    a shorter list is extended with its last entry as set_parameters() does."""
    bs = [int(x) for x in str(a["block_size"]).split(",") if x != ""]
    for _ in range(len(bs), TRLs - 1):
        bs.append(bs[-1] // 2 if bs[-1] > a["block_size_min"] else bs[-1])
    out = {"block_size": bs}
    for k in ("pixels_in_x", "pixels_in_y", "subpixel_accuracy"):
        v = [x for x in str(a[k]).split(",") if x != ""]
        for _ in range(len(v), TRLs):
            v.append(v[-1])
        out[k] = [int(x) for x in v]
    return out


def _one_size(xs, ys, what):
    sizes = set(zip(xs, ys))
    if len(sizes) != 1:
        C.die(f"{what}: a different picture size per temporal subband (spatial scalability) is not implemented on this path")
    return sizes.pop()


def _texture_plan(TRLs, GOPs):
    """(file, pictures) per call of texture_expand.py:85-131: high_{TRLs-1} .. high_1, then low_{TRLs-1}."""
    P0 = schedule.pictures(GOPs, TRLs)
    plan = []
    for sub in range(TRLs - 1, 0, -1):
        P = P0
        for _ in range(sub):
            P = (P + 1) // 2
        plan.append((f"high_{sub}", P - 1))
    plan.append((f"low_{TRLs - 1}", GOPs + 1))
    return plan


def _check_rates(rates):
    if any(float(x) > 0 for x in str(rates).split(",") if x != ""):
        C.die("--rates (kdu_expand -rate, \"optional and developing\" in the reference) is not implemented on this path")


def texture_expand(argv=None):
    a = C.parse("Expands the texture.", {"GOPs": (1, int), "TRLs": (4, int), "SRLs": (5, int), "rates": ("0.0,0.0,0.0,0.0,0.0", str),
                                         "pixels_in_x": ("352,352,352,352,352", str), "pixels_in_y": ("288,288,288,288,288", str)}, argv)
    _check_rates(a["rates"])
    TRLs = a["TRLs"]
    xs = [int(x) for x in a["pixels_in_x"].split(",")]
    ys = [int(x) for x in a["pixels_in_y"].split(",")]
    from .. import hip
    e = C.engine()
    try:
        for file, pictures in _texture_plan(TRLs, a["GOPs"]):
            k = 0 if file.startswith("low") else TRLs - int(file.split("_")[1])  # texture_expand.py:104-105,124-125
            expand_file(e, file, pictures, xs[min(k, len(xs) - 1)], ys[min(k, len(ys) - 1)])
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def _motion_plan(TRLs, GOPs):
    """(iteration, fields) of motion_expand.py:113-131."""
    fields = schedule.pictures(GOPs, TRLs) // 2
    plan = []
    for it in range(1, TRLs):
        plan.append((it, fields))
        fields //= 2
    return plan


def _run_motion_expand(e, TRLs, GOPs, BY, BX):
    """motion_expand.py:113-170 for one block grid: residues of every level from their code-streams (one GPU batch), the coarsest level
    through the bidirectional correlator, the others through the interlevel correlator from the top down.  Writes motion_residue_t
    and motion_t like the tools it replaces; returns {t: motion_t}."""
    plan = _motion_plan(TRLs, GOPs)
    planes = _decode_named(e, [n for it, f in plan for n in _residue_names(f"motion_residue_{it}", f)])
    res = {}
    for it, f in plan:
        res[it] = _residue_from_planes(planes, f"motion_residue_{it}", f, BY, BX)
        C.write_motion(f"motion_residue_{it}", res[it])
    res[TRLs - 1] = res[TRLs - 1][:GOPs]          # motion_expand.py:139: --fields=GOPs
    fields = motion.field_chain(res, TRLs)
    for it in fields:
        C.write_motion(f"motion_{it}", fields[it])
    return fields


def _grid(a, TRLs):
    L = _lists(a, TRLs)
    W, H = _one_size(L["pixels_in_x"], L["pixels_in_y"], "motion_expand")
    bs = set(L["block_size"])
    if len(bs) != 1 and TRLs > 2:
        # motion_expand.py:103-105,121-122,152-153 index the block grids of the levels inconsistently (expand by (TRLs-1)-it,
        # correlate by it-1): with a block size that changes from level to level the reference correlates fields of different grids
        C.die("a block size that changes between temporal levels: the reference's motion_expand mixes the levels' block grids "
              "(motion_expand.py:121-153); not implemented on this path")
    b = L["block_size"][0]
    return W, H, b, L


def motion_expand(argv=None):
    a = C.parse("Expands the motion data.", {"GOPs": (1, int), "TRLs": (4, int), "block_size": ("16,16,16,16", str), "block_size_min": (0, int),
                                             "pixels_in_x": ("352,352,352,352,352", str), "pixels_in_y": ("288,288,288,288,288", str),
                                             "subpixel_accuracy": ("0", str)}, argv)
    if os.environ.get("MCTF_MOTION_CODEC", "j2k") != "j2k":
        C.die("only the j2k motion codec is implemented on this path")
    TRLs = a["TRLs"]
    if TRLs < 2:
        return
    if not a["block_size_min"]:      # motion_expand.py:87 takes the list as given (expand.py has extended it): a short list repeats its last entry
        a["block_size_min"] = 1 << 30
    W, H, bs, _ = _grid(a, TRLs)
    from .. import hip
    e = C.engine()
    try:
        _run_motion_expand(e, TRLs, a["GOPs"], H // bs, W // bs)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()


def expand(argv=None):
    """expand.py:209-256: textures, motion, synthesis -- here with one engine, the decoded subbands written to the files the reference's
    stages exchange (high_t, low_T, motion_residue_t, motion_t) and handed to the inverse temporal levels."""
    spec = {"TRLs": (4, int), "pixels_in_x": ("352,352,352,352", str), "pixels_in_y": ("288,288,288,288", str), "block_size": (None, str),
            "block_size_min": (None, int), "GOPs": (1, int), "SRLs": (5, int), "rates": ("0.0,0.0,0.0,0.0,0.0", str),
            "subpixel_accuracy": ("0,0,0,0,0", str), "search_range": (4, int), "border_size": (0, int), "block_overlaping": (0, int),
            "update_factor": (1.0 / 4, float)}
    a = C.parse("Decodes a sequence of pictures.", spec, argv)
    _check_rates(a["rates"])
    if os.environ.get("MCTF_TEXTURE_CODEC", "j2k") != "j2k" or os.environ.get("MCTF_MOTION_CODEC", "j2k") != "j2k":
        C.die("only the j2k texture and motion codecs are implemented on this path")
    TRLs, GOPs = a["TRLs"], a["GOPs"]
    W0, H0 = int(a["pixels_in_x"].split(",")[0]), int(a["pixels_in_y"].split(",")[0])
    d = schedule.default_block_size(W0, H0)                    # expand.py:110-116
    if a["block_size"] is None:
        a["block_size"] = ",".join([str(d)] * max(TRLs - 1, 1))
    if a["block_size_min"] is None:
        a["block_size_min"] = d
    from . import drivers
    from .. import hip
    e = C.engine()
    try:
        if TRLs < 2:
            W, H = W0, H0
            expand_file(e, f"low_{TRLs - 1}", GOPs + 1, W, H)
            return
        W, H, bs, L = _grid(a, TRLs)
        # The motion streams on a context (and stream) of their own, beside the textures: their component planes are the reference's raw
        # 16-bit words, a negative residue makes a 16-bit-plane code-block, and one such block is ten milliseconds of one decoding lane
        # whatever else the GPU does.  (Same files, same order of writes within each side.)
        from concurrent.futures import ThreadPoolExecutor
        em = C.engine()
        try:
            with ThreadPoolExecutor(max_workers=1) as pool:
                fut = pool.submit(_run_motion_expand, em, TRLs, GOPs, H // bs, W // bs)
                for file, pictures in _texture_plan(TRLs, GOPs):
                    expand_file(e, file, pictures, W, H)
                fut.result()
        finally:
            em.close()
        sa = dict(a, block_size=",".join(map(str, L["block_size"])), pixels_in_x=",".join(map(str, L["pixels_in_x"])),
                  pixels_in_y=",".join(map(str, L["pixels_in_y"])), subpixel_accuracy=",".join(map(str, L["subpixel_accuracy"])))
        drivers.run_synthesize(e, sa)
    except hip.QsvcError as ex:
        C.die(str(ex))
    finally:
        e.close()
