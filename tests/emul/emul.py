"""TEST INFRASTRUCTURE: ctypes driver for the host loop harness (tests/emul/emul.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libqsvc_emul.so")
        srcs = [os.path.join(_HERE, "emul.cpp")] + [os.path.join(_HERE, "..", "..", "qsvc_amd", "csrc", f)
                                                    for f in ("mctf_core.h", "mctf_threads.h", "t1_core.h", "t1_cm2.h")] + [os.path.join(_HERE, "t1_legacy.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so,
                                   os.path.join(_HERE, "emul.cpp")])
        L = C.CDLL(so)
        L.emul_dwt53t.argtypes = [_i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.emul_motion_estimate.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p]
        L.emul_predict.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i16p, _u8p, _u8p, _u32p]
        L.emul_update.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _i16p, C.c_char_p, C.c_float, C.c_int,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_int, _u8p]
        L.emul_t1_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.emul_t1_block_split.argtypes = L.emul_t1_block.argtypes
        L.emul_t1_block_cm2.argtypes = L.emul_t1_block.argtypes
        L.emul_border_luma.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _i16p]
        L.emul_synthesize_step.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _i16p, C.c_char_p, C.c_float,
                                           C.c_int, _u8p]
        _LIB = L
    return _LIB


def fb(W, H):
    return W * H + 2 * (W // 2) * (H // 2)


def motion_estimate(low, W, H, bs, S):
    low = np.ascontiguousarray(low)
    mv = np.zeros((low.shape[0] // 2, 4, H // bs, W // bs), dtype=np.int16)
    r = lib().emul_motion_estimate(low, low.shape[0], W, H, bs, S, mv)
    return r, mv


def predict(low, W, H, bs, S, mv):
    low = np.ascontiguousarray(low)
    nB = low.shape[0] // 2
    high = np.zeros((nB, fb(W, H)), dtype=np.uint8)
    pred = np.zeros_like(high)
    hist = np.zeros((nB, 3, 256), dtype=np.uint32)
    r = lib().emul_predict(low, low.shape[0], W, H, bs, S, np.ascontiguousarray(mv), high, pred, hist)
    return r, high, pred, hist


def update(low, W, H, bs, high, mv, types, uf=0.25, halo_l=None, halo_r=None):
    low = np.ascontiguousarray(low)
    out = np.zeros((low.shape[0] // 2 + 1, fb(W, H)), dtype=np.uint8)
    reach = int(np.abs(mv).max()) if mv.size else 0
    hl = hr = None
    tl = tr = 0
    if halo_l is not None:
        hl = np.ascontiguousarray(halo_l[0]); tl = halo_l[1]; reach = max(reach, halo_l[2])
    if halo_r is not None:
        hr = np.ascontiguousarray(halo_r[0]); tr = halo_r[1]; reach = max(reach, halo_r[2])
    r = lib().emul_update(low, low.shape[0], W, H, bs, np.ascontiguousarray(high), np.ascontiguousarray(mv), types, uf, reach,
                          hl.ctypes.data if hl is not None else None, tl, hr.ctypes.data if hr is not None else None, tr, out)
    return r, out


def synthesize_step(low_t, W, H, bs, S, high, mv, types, uf=0.25):
    low_t = np.ascontiguousarray(low_t)
    nB = high.shape[0]
    out = np.zeros((2 * nB + 1, fb(W, H)), dtype=np.uint8)
    reach = int(np.abs(mv).max()) if mv.size else 0
    r = lib().emul_synthesize_step(low_t, 2 * nB + 1, W, H, bs, S, np.ascontiguousarray(high), np.ascontiguousarray(mv), types, uf,
                                   reach, out)
    return r, out


def t1_block(vals, orient, split=False, cm2=False):
    v = np.ascontiguousarray(vals, dtype=np.int32)
    h, w = v.shape
    out = np.empty(w * h * 4 + 64, dtype=np.uint8)
    npasses, nbps = C.c_int(), C.c_int()
    pl = np.zeros(64, dtype=np.int32)
    pd = np.zeros(64, dtype=np.int64)
    n = (lib().emul_t1_block_cm2 if cm2 else (lib().emul_t1_block_split if split else lib().emul_t1_block))(v.ctypes.data, w, h, orient, out.ctypes.data, C.byref(npasses), C.byref(nbps), pl.ctypes.data, pd.ctypes.data)
    return out[:max(n, 0)].tobytes(), npasses.value, nbps.value, pl[:npasses.value].copy(), pd[:npasses.value].copy()
