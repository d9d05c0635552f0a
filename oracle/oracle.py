"""TEST INFRASTRUCTURE ONLY -- ctypes bindings of oracle/libqsvc_oracle.so (the CPU restatement)
and a runner for the reference's own tools compiled into oracle/_ref (when present).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (qsvc_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")
_LIB = None

_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the C restatement (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "libqsvc_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mctf_oracle.c", "mctf_subpel_oracle.c", "j2k_oracle.c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libqsvc_oracle.so"])
    if os.path.exists("/root/reference/trunk/src/split.cpp"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        L.qo_dwt53t_analyze.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.qo_dwt53t_synthesize.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.qo_haar_analyze.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.qo_haar_synthesize.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.qo_bordered_from_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _i16p]
        L.qo_q1_safe.argtypes = [C.c_int, C.c_int]
        L.qo_motion_estimate.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p]
        L.qo_motion_estimate_b.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p]
        L.qo_decorrelate.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p, _u8p,
                                     C.c_char_p, _i16p, C.c_void_p, C.c_void_p]
        L.qo_decorrelate_ov.argtypes = L.qo_decorrelate.argtypes + [C.c_int]
        L.qo_update.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _i16p, C.c_char_p, C.c_float, _u8p]
        L.qo_un_update.argtypes = L.qo_update.argtypes
        L.qo_correlate.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p, _u8p, C.c_char_p, _u8p, C.c_void_p]
        L.qo_correlate_ov.argtypes = L.qo_correlate.argtypes + [C.c_int]
        L.qo_motion_estimate_sub.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p]
        L.qo_decorrelate_sub.argtypes = L.qo_decorrelate.argtypes + [C.c_int, C.c_int]
        L.qo_correlate_sub.argtypes = L.qo_correlate.argtypes + [C.c_int, C.c_int]
        L.qo_entropy.argtypes = [_i32p, C.c_int]
        L.qo_entropy.restype = C.c_float
        L.qo_frame_type.argtypes = [_i32p, _i32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def frame_bytes(W, H):
    return W * H + 2 * (W // 2) * (H // 2)


# ------------------------------------------------------------------ restatement wrappers
def dwt53t(plane: np.ndarray, levels: int, inverse: bool = False, H=None, W=None) -> np.ndarray:
    p = np.ascontiguousarray(plane, dtype=np.int16).copy()
    h, w = p.shape
    fn = lib().qo_dwt53t_synthesize if inverse else lib().qo_dwt53t_analyze
    fn(p, w, H or h, W or w, levels)
    return p


def haar(plane: np.ndarray, levels: int, inverse: bool = False) -> np.ndarray:
    p = np.ascontiguousarray(plane, dtype=np.int16).copy()
    h, w = p.shape
    (lib().qo_haar_synthesize if inverse else lib().qo_haar_analyze)(p, w, h, w, levels)
    return p


def bordered(luma: np.ndarray, b: int) -> np.ndarray:
    H, W = luma.shape
    out = np.empty((H + 2 * b, W + 2 * b), dtype=np.int16)
    assert lib().qo_bordered_from_u8(np.ascontiguousarray(luma, dtype=np.uint8), H, W, b, out) == 0
    return out


def motion_estimate(low: np.ndarray, W, H, bs, S, border_size=0, subpixel_accuracy=0) -> np.ndarray:
    low = np.ascontiguousarray(low, dtype=np.uint8)
    P = low.shape[0]
    mv = np.zeros((P // 2, 4, H // bs, W // bs), dtype=np.int16)
    if subpixel_accuracy:  # the heap-image restatement (oracle/mctf_subpel_oracle.c)
        assert lib().qo_motion_estimate_sub(low, P, W, H, bs, S, border_size, subpixel_accuracy, mv) == 0
    else:
        assert lib().qo_motion_estimate_b(low, P, W, H, bs, S, border_size, mv) == 0
    return mv


def decorrelate(low, W, H, bs, S, mv, always_B=0, want_prediction=False, want_hists=False, block_overlaping=0, subpixel_accuracy=0):
    low = np.ascontiguousarray(low, dtype=np.uint8)
    P = low.shape[0]
    nB = P // 2
    F = frame_bytes(W, H)
    high = np.zeros((nB, F), dtype=np.uint8)
    types = C.create_string_buffer(nB + 1)
    mvo = np.zeros_like(mv)
    pred = np.zeros((nB, F), dtype=np.uint8) if want_prediction else None
    hists = np.zeros((nB, 3, 256), dtype=np.int32) if want_hists else None
    if subpixel_accuracy:
        r = lib().qo_decorrelate_sub(low, P, W, H, bs, S, always_B, np.ascontiguousarray(mv), high, types, mvo,
                                     pred.ctypes.data if pred is not None else None,
                                     hists.ctypes.data if hists is not None else None, block_overlaping, subpixel_accuracy)
    else:
        r = lib().qo_decorrelate_ov(low, P, W, H, bs, S, always_B, np.ascontiguousarray(mv), high, types, mvo,
                                    pred.ctypes.data if pred is not None else None,
                                    hists.ctypes.data if hists is not None else None, block_overlaping)
    assert r == 0
    return dict(high=high, types=types.raw[:nB], mv=mvo, prediction=pred, hists=hists)


def update(low, W, H, bs, high, mv, types: bytes, update_factor=0.25) -> np.ndarray:
    low = np.ascontiguousarray(low, dtype=np.uint8)
    P = low.shape[0]
    out = np.zeros((P // 2 + 1, frame_bytes(W, H)), dtype=np.uint8)
    r = lib().qo_update(low, P, W, H, bs, np.ascontiguousarray(high), np.ascontiguousarray(mv), types,
                        update_factor, out)
    assert r == 0
    return out


def un_update(low_t, P, W, H, bs, high, mv, types: bytes, update_factor=0.25) -> np.ndarray:
    """low_t [P/2+1] -> even_t [P/2+1] (update.cpp without ANALYZE)."""
    low_t = np.ascontiguousarray(low_t, dtype=np.uint8)
    out = np.zeros((P // 2 + 1, frame_bytes(W, H)), dtype=np.uint8)
    assert lib().qo_un_update(low_t, P, W, H, bs, np.ascontiguousarray(high), np.ascontiguousarray(mv), types, update_factor, out) == 0
    return out


def correlate(even, P, W, H, bs, S, high, mv, types: bytes, block_overlaping=0, subpixel_accuracy=0) -> np.ndarray:
    """even_t [P/2+1], high_t [P/2] -> odd_t [P/2] (decorrelate.cpp without ANALYZE)."""
    even = np.ascontiguousarray(even, dtype=np.uint8)
    out = np.zeros((P // 2, frame_bytes(W, H)), dtype=np.uint8)
    if subpixel_accuracy:
        assert lib().qo_correlate_sub(even, P, W, H, bs, S, np.ascontiguousarray(mv), np.ascontiguousarray(high), types, out, None,
                                      block_overlaping, subpixel_accuracy) == 0
    else:
        assert lib().qo_correlate_ov(even, P, W, H, bs, S, np.ascontiguousarray(mv), np.ascontiguousarray(high), types, out, None, block_overlaping) == 0
    return out


def synthesize_step(low_t, high, mv, types: bytes, W, H, bs, S, update_factor=0.25, block_overlaping=0, subpixel_accuracy=0):
    """One inverse temporal level (synthesize_step.py:96-151): un_update, correlate, merge -> low_{t-1}."""
    nB = high.shape[0]
    P = 2 * nB + 1
    even = un_update(low_t, P, W, H, bs, high, mv, types, update_factor)
    odd = correlate(even, P, W, H, bs, S, high, mv, types, block_overlaping, subpixel_accuracy)
    out = np.empty((P, frame_bytes(W, H)), dtype=np.uint8)
    out[0::2] = even
    out[1::2] = odd
    return dict(even=even, odd=odd, low=out)


def analyze_step(low, W, H, bs, S, always_B=0, update_factor=0.25, want_prediction=False, block_overlaping=0, border_size=0,
                 subpixel_accuracy=0):
    """One temporal level (analyze_step.py:117-232) with the restatement.  update.cpp ignores --subpixel_accuracy: it applies the
    (sub-pixel) vectors as whole pixels."""
    mv = motion_estimate(low, W, H, bs, S, border_size, subpixel_accuracy)
    d = decorrelate(low, W, H, bs, S, mv, always_B, want_prediction=want_prediction, want_hists=True, block_overlaping=block_overlaping,
                    subpixel_accuracy=subpixel_accuracy)
    lo = update(low, W, H, bs, d["high"], d["mv"], d["types"], update_factor)
    return dict(motion=mv, motion_filtered=d["mv"], high=d["high"], frame_types=d["types"], low=lo,
                prediction=d["prediction"], hists=d["hists"])


def analyze(low0, W, H, TRLs, bs, S=4, bs_min=None, always_B=0, update_factor=0.25, block_overlaping=0, border_size=0, subpixel_accuracy=0):
    """The level loop of analyze.py:110-153."""
    bs_min = bs if bs_min is None else min(bs_min, bs)
    levels = []
    low = low0
    for t in range(1, TRLs):
        r = analyze_step(low, W, H, bs, S, always_B, update_factor, block_overlaping=block_overlaping, border_size=border_size,
                         subpixel_accuracy=subpixel_accuracy)
        r.update(block_size=bs, search_range=S, pictures=low.shape[0])
        levels.append(r)
        low = r["low"]
        S = min(2 * S, 128)
        bs = max(bs // 2, bs_min)
    return levels


# ------------------------------------------------------------------ compiled reference tools
def have_ref() -> bool:
    return all(os.path.exists(os.path.join(REF_DIR, t)) for t in ("split", "motion_estimate", "decorrelate", "update"))


def ref_analyze_step(low: np.ndarray, W, H, bs, S, always_B=0, update_factor=0.25, t=1, workdir=None, timing=None,
                     block_overlaping=0, border_size=0, subpixel_accuracy=0):
    """Run the reference's split/motion_estimate/decorrelate/update binaries (oracle/_ref) on one level,
    exactly as analyze_step.py:117-232 invokes them.  Default glibc malloc (no MALLOC_PERTURB_)."""
    import time
    low = np.ascontiguousarray(low, dtype=np.uint8)
    P = low.shape[0]
    F = frame_bytes(W, H)
    own = workdir is None
    d = workdir or tempfile.mkdtemp(prefix="qsvc_ref_")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MALLOC_")}
    try:
        low.tofile(os.path.join(d, f"low_{t-1}"))
        for f in (f"motion_{t}",):
            if os.path.exists(os.path.join(d, f)):
                os.remove(os.path.join(d, f))  # Q8: motion_estimate exits 1 if it exists

        def run(tool, *args):
            t0 = time.perf_counter()
            subprocess.check_call([os.path.join(REF_DIR, tool)] + [str(a) for a in args], cwd=d, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if timing is not None:
                timing[tool] = timing.get(tool, 0.0) + time.perf_counter() - t0

        run("split", f"--even_fn=even_{t}", f"--low_fn=low_{t-1}", f"--odd_fn=odd_{t}", f"--pictures={P}",
            f"--pixels_in_x={W}", f"--pixels_in_y={H}")
        run("motion_estimate", f"--block_size={bs}", f"--border_size={border_size}", f"--even_fn=even_{t}",
            f"--imotion_fn=imotion_{t}", f"--motion_fn=motion_{t}", f"--odd_fn=odd_{t}", f"--pictures={P}",
            f"--pixels_in_x={W}", f"--pixels_in_y={H}", f"--search_range={S}", f"--subpixel_accuracy={subpixel_accuracy}")
        run("decorrelate", f"--block_overlaping={block_overlaping}", f"--block_size={bs}", f"--even_fn=even_{t}",
            f"--frame_types_fn=frame_types_{t}", f"--high_fn=high_{t}", f"--motion_in_fn=motion_{t}",
            f"--motion_out_fn=motion_filtered_{t}", f"--odd_fn=odd_{t}", f"--pictures={P}", f"--pixels_in_x={W}",
            f"--pixels_in_y={H}", f"--search_range={S}", f"--subpixel_accuracy={subpixel_accuracy}", f"--always_B={always_B}")
        run("update", f"--block_size={bs}", f"--even_fn=even_{t}", f"--frame_types_fn=frame_types_{t}",
            f"--high_fn=high_{t}", f"--low_fn=low_{t}", f"--motion_fn=motion_filtered_{t}", f"--pictures={P}",
            f"--pixels_in_x={W}", f"--pixels_in_y={H}", f"--subpixel_accuracy={subpixel_accuracy}", f"--update_factor={update_factor}")
        nB = P // 2
        BY, BX = H // bs, W // bs
        rd = lambda n, dt=np.uint8: np.fromfile(os.path.join(d, n), dtype=dt)
        return dict(
            motion=rd(f"motion_{t}", np.int16).reshape(nB, 4, BY, BX),
            motion_filtered=rd(f"motion_filtered_{t}", np.int16).reshape(nB, 4, BY, BX),
            high=rd(f"high_{t}").reshape(nB, F),
            frame_types=rd(f"frame_types_{t}").tobytes(),
            low=rd(f"low_{t}").reshape(nB + 1, F),
            prediction=rd(f"prediction_even_{t}").reshape(nB, F),
        )
    finally:
        if own:
            shutil.rmtree(d, ignore_errors=True)


def ref_analyze(low0, W, H, TRLs, bs, S=4, bs_min=None, always_B=0, update_factor=0.25, timing=None, block_overlaping=0, border_size=0,
                subpixel_accuracy=0):
    bs_min = bs if bs_min is None else min(bs_min, bs)
    levels, low = [], low0
    for t in range(1, TRLs):
        r = ref_analyze_step(low, W, H, bs, S, always_B, update_factor, t=t, timing=timing, block_overlaping=block_overlaping,
                             border_size=border_size, subpixel_accuracy=subpixel_accuracy)
        r.update(block_size=bs, search_range=S, pictures=low.shape[0])
        levels.append(r)
        low = r["low"]
        S = min(2 * S, 128)
        bs = max(bs // 2, bs_min)
    return levels


def ref_synthesize_step(low_t, high, mv, types: bytes, W, H, bs, S, update_factor=0.25, t=1, block_overlaping=0, subpixel_accuracy=0):
    """The reference's un_update / correlate / merge binaries (oracle/_ref) as synthesize_step.py:96-151 chains them."""
    nB = high.shape[0]
    P = 2 * nB + 1
    d = tempfile.mkdtemp(prefix="qsvc_refs_")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MALLOC_")}
    try:
        np.ascontiguousarray(low_t, dtype=np.uint8).tofile(os.path.join(d, f"low_{t}"))
        np.ascontiguousarray(high, dtype=np.uint8).tofile(os.path.join(d, f"high_{t}"))
        np.ascontiguousarray(mv, dtype="<i2").tofile(os.path.join(d, f"motion_{t}"))
        open(os.path.join(d, f"frame_types_{t}"), "wb").write(types)

        def run(tool, *args):
            subprocess.check_call([os.path.join(REF_DIR, tool)] + [str(a) for a in args], cwd=d, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

        run("un_update", f"--block_size={bs}", f"--even_fn=even_{t}", f"--frame_types_fn=frame_types_{t}", f"--high_fn=high_{t}",
            f"--low_fn=low_{t}", f"--motion_fn=motion_{t}", f"--pictures={P}", f"--pixels_in_x={W}", f"--pixels_in_y={H}",
            f"--subpixel_accuracy={subpixel_accuracy}", f"--update_factor={update_factor}")
        run("correlate", f"--block_overlaping={block_overlaping}", f"--block_size={bs}", f"--even_fn=even_{t}", f"--frame_types_fn=frame_types_{t}",
            f"--high_fn=high_{t}", f"--motion_in_fn=motion_{t}", f"--odd_fn=odd_{t}", f"--pictures={P}", f"--pixels_in_x={W}",
            f"--pixels_in_y={H}", f"--search_range={S}", f"--subpixel_accuracy={subpixel_accuracy}")
        run("merge", f"--even=even_{t}", f"--low=low_{t - 1}", f"--odd=odd_{t}", f"--pictures={P}", f"--pixels_in_x={W}",
            f"--pixels_in_y={H}")
        F = frame_bytes(W, H)
        rd = lambda n: np.fromfile(os.path.join(d, n), dtype=np.uint8)
        return dict(even=rd(f"even_{t}").reshape(nB + 1, F), odd=rd(f"odd_{t}").reshape(nB, F), low=rd(f"low_{t - 1}").reshape(P, F))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ref_motion_residues(filtered: dict, BY: int, BX: int) -> dict:
    """motion_filtered_t -> motion_residue_t by the reference's interlevel_/bidirectional_motion_decorrelate binaries
    (oracle/_ref), called as motion_compress.py:146-186 does for block_size == block_size_min."""
    d = tempfile.mkdtemp(prefix="qsvc_refm_")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MALLOC_")}
    try:
        its = sorted(filtered)
        for t in its:
            np.ascontiguousarray(filtered[t], dtype="<i2").tofile(os.path.join(d, f"motion_filtered_{t}"))
        fields = filtered[its[0]].shape[0]
        out = {}

        def run(tool, *args):
            subprocess.check_call([os.path.join(REF_DIR, tool)] + list(args), cwd=d, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

        for t in its[:-1]:
            run("interlevel_motion_decorrelate", f"--blocks_in_x={BX}", f"--blocks_in_y={BY}",
                f"--fields_in_predicted={fields}", f"--predicted=motion_filtered_{t}",
                f"--reference=motion_filtered_{t + 1}", f"--residue=motion_residue_{t}")
            fields //= 2
        t = its[-1]
        run("bidirectional_motion_decorrelate", f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", f"--fields={fields}",
            f"--input=motion_filtered_{t}", f"--output=motion_residue_{t}")
        for t in its:
            out[t] = np.fromfile(os.path.join(d, f"motion_residue_{t}"), dtype="<i2").reshape(-1, 4, BY, BX)
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ref_motion_fields(residues: dict, BY: int, BX: int) -> dict:
    """motion_residue_t -> motion_t by the reference's bidirectional_/interlevel_motion_correlate binaries (oracle/_ref: the
    decorrelators' sources without -D ANALYZE), called as motion_expand.py:131-170 does."""
    d = tempfile.mkdtemp(prefix="qsvc_refmc_")
    env = {k: v for k, v in os.environ.items() if not k.startswith("MALLOC_")}
    try:
        its = sorted(residues)
        for t in its:
            np.ascontiguousarray(residues[t], dtype="<i2").tofile(os.path.join(d, f"motion_residue_{t}"))

        def run(tool, *args):
            subprocess.check_call([os.path.join(REF_DIR, tool)] + list(args), cwd=d, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

        top = its[-1]
        run("bidirectional_motion_correlate", f"--blocks_in_x={BX}", f"--blocks_in_y={BY}", f"--fields={residues[top].shape[0]}",
            f"--input=motion_residue_{top}", f"--output=motion_{top}")
        for t in reversed(its[:-1]):
            run("interlevel_motion_correlate", f"--blocks_in_x={BX}", f"--blocks_in_y={BY}",
                f"--fields_in_predicted={residues[t].shape[0]}", f"--reference=motion_{t + 1}", f"--predicted=motion_{t}",
                f"--residue=motion_residue_{t}")
        return {t: np.fromfile(os.path.join(d, f"motion_{t}"), dtype="<i2").reshape(-1, 4, BY, BX) for t in its}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def ref_dwt(plane: np.ndarray, levels: int, inverse=False, filt="53") -> np.ndarray:
    p = np.ascontiguousarray(plane, dtype=np.int16)
    H, W = p.shape
    out = subprocess.run([os.path.join(REF_DIR, "dwt_probe"), filt, "s" if inverse else "a", str(H), str(W), str(levels)],
                         input=p.tobytes(), stdout=subprocess.PIPE, check=True).stdout
    return np.frombuffer(out, dtype=np.int16).reshape(H, W).copy()
