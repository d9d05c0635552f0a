"""Shared pieces of the Python-3 entry points that mirror the reference's executables.

Flag names, defaults and cwd file names follow the reference tool each entry point replaces (cited in each
module).  Unknown flags are ignored like the reference's `parse_known_args()[0]` (compress.py:128), and a flag
given as `--x=0` IS honoured because the test is on the non-empty string ("Config").
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np


def parse(description: str, spec: dict, argv=None) -> dict:
    """spec: name -> (default, caster).  Returns the resolved values."""
    p = argparse.ArgumentParser(description=description)
    for name, (default, _) in spec.items():
        p.add_argument("--" + name, help=f"(Default = {default})")
    args = p.parse_known_args(argv)[0]
    out = {}
    for name, (default, cast) in spec.items():
        v = getattr(args, name)
        out[name] = cast(v) if v else default  # `if args.x:` of the reference drivers
    return out


def frame_bytes(W, H):
    return W * H + 2 * (W // 2) * (H // 2)


def read_frames(fn, W, H, count=None, first=0) -> np.ndarray:
    F = frame_bytes(W, H)
    if not os.path.exists(fn):
        die(f"\"{fn}\" does not exist ... aborting!")
    n = os.path.getsize(fn) // F if count is None else count
    a = np.fromfile(fn, dtype=np.uint8, count=n * F, offset=first * F)
    if a.size < n * F:  # the reference reads short files as zeros past EOF (fread result unchecked, split.cpp:265-341)
        a = np.concatenate([a, np.zeros(n * F - a.size, dtype=np.uint8)])
    return a.reshape(n, F)


def write_frames(fn, frames: np.ndarray):
    np.ascontiguousarray(frames, dtype=np.uint8).tofile(fn)


def read_motion(fn, nB, BY, BX) -> np.ndarray:
    if not os.path.exists(fn):
        die(f"unable to read \"{fn}\" ... aborting!")
    a = np.fromfile(fn, dtype="<i2", count=nB * 4 * BY * BX)
    if a.size < nB * 4 * BY * BX:
        a = np.concatenate([a, np.zeros(nB * 4 * BY * BX - a.size, dtype=np.int16)])
    return a.reshape(nB, 4, BY, BX).astype(np.int16)


def write_motion(fn, mv: np.ndarray):
    np.ascontiguousarray(mv, dtype="<i2").tofile(fn)


def die(msg, code=255):
    sys.stderr.write(f"{os.path.basename(sys.argv[0])}: {msg}\n")
    sys.exit(code)  # the reference drivers exit(-1) == 255, the C++ tools abort()


def engine():
    """The HIP backend.  No CPU fallback: a missing library or GPU is a hard error (non-zero exit)."""
    from .. import hip
    try:
        return hip.Engine(int(os.environ.get("QSVC_DEVICE", "0")))
    except Exception as ex:  # noqa: BLE001
        die(f"cannot initialise the MI355X backend: {ex}")


def trace(argv):
    """mctf.sh:35-36 appends every command to ./trace."""
    try:
        with open("trace", "a") as f:
            f.write(" ".join(argv) + "\n")
    except OSError:
        pass


def interleave(even: np.ndarray, odd: np.ndarray) -> np.ndarray:
    """even_t (nB+1 pictures) + odd_t (nB pictures) -> low_{t-1} order (split.cpp:226-341 undone)."""
    nB = odd.shape[0]
    low = np.empty((2 * nB + 1, even.shape[1]), dtype=np.uint8)
    low[0::2] = even[:nB + 1]
    low[1::2] = odd
    return low


def apply_mctf_options(e, a):
    """--block_overlaping / --border_size / --subpixel_accuracy of the reference's tools -> the engine.
    What the library refuses (QSVC_ERR_UNSUPPORTED) ends the tool with the drivers' exit status."""
    from .. import hip
    sp = a.get("subpixel_accuracy", 0)
    if isinstance(sp, str):  # synthesize takes a comma list, one entry per level
        vals = {int(x) for x in sp.split(",") if x != ""}
        sp = max(vals) if vals else 0
    try:
        e.set_mctf_options(int(a.get("block_overlaping", 0) or 0), int(a.get("border_size", 0) or 0), int(sp or 0))
    except hip.QsvcError as ex:
        die(str(ex))
