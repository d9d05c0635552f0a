"""TEST INFRASTRUCTURE ONLY -- ctypes bindings of the T.800 restatement (oracle/j2k_oracle.c) and an
independent decoder (OpenJPEG through Pillow) used to pin it.  Parity with the reference's Kakadu is
UNPINNED (no Kakadu artefact exists under /root/reference); see the header of j2k_oracle.c."""
from __future__ import annotations

import ctypes as C
import io

import numpy as np

from . import oracle as _O


class Params(C.Structure):
    _fields_ = [("reversible", C.c_int), ("levels", C.c_int), ("layers", C.c_int), ("cblk_w", C.c_int),
                ("cblk_h", C.c_int), ("use_sop", C.c_int), ("base_step", C.c_float), ("slopes", C.c_int * 16)]


_ready = False


def _lib():
    global _ready
    L = _O.lib()
    if not _ready:
        L.qo_j2k_encode_plane.restype = C.c_long
        L.qo_j2k_encode_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p]
        L.qo_j2k_encode_plane_s16.restype = C.c_long
        L.qo_j2k_encode_plane_s16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.c_void_p, C.c_size_t, C.c_void_p]
        L.qo_j2k_t1_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.qo_j2k_coefficients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.qo_j2k_dwt97_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.qo_j2k_norm.restype = C.c_double
        L.qo_j2k_norm.argtypes = [C.c_int, C.c_int, C.c_int]
        L.qo_j2k_slope_to_lambda.restype = C.c_double
        L.qo_j2k_slope_to_lambda.argtypes = [C.c_int]
        L.qo_j2k_decode_plane.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _ready = True
    return L


def params(reversible=False, levels=4, layers=1, cblk=64, use_sop=True, base_step=1.0 / 256, slopes=()):
    p = Params()
    p.reversible, p.levels, p.layers, p.cblk_w, p.cblk_h, p.use_sop = int(reversible), levels, layers, cblk, cblk, int(use_sop)
    p.base_step = base_step
    for k in range(16):
        p.slopes[k] = int(slopes[k]) if k < len(slopes) else 0
    return p


def encode_plane(plane: np.ndarray, reversible=False, levels=4, layers=1, cblk=64, use_sop=True, base_step=1.0 / 256,
                 slopes=(), want_stats=False):
    p = np.ascontiguousarray(plane, dtype=np.uint8)
    H, W = p.shape
    cap = H * W * 3 + (1 << 16)
    out = np.empty(cap, dtype=np.uint8)
    st = np.zeros(3, dtype=np.uint64)
    prm = params(reversible, levels, layers, cblk, use_sop, base_step, slopes)
    n = _lib().qo_j2k_encode_plane(p.ctypes.data, H, W, C.byref(prm), out.ctypes.data, cap, st.ctypes.data)
    if n < 0:
        raise RuntimeError(f"qo_j2k_encode_plane failed: {n}")
    cs = out[:n].tobytes()
    return (cs, dict(codeblocks=int(st[0]), passes=int(st[1]), symbols=int(st[2]))) if want_stats else cs


def encode_plane_s16(plane: np.ndarray, cblk=64, use_sop=True) -> bytes:
    """16-bit signed plane, reversible, Clevels=0 (motion fields, motion_compress_j2k.py:131-142)."""
    p = np.ascontiguousarray(plane, dtype=np.int16)
    H, W = p.shape
    out = np.empty(H * W * 4 + 4096, dtype=np.uint8)
    prm = params(True, 0, 1, cblk, use_sop)
    n = _lib().qo_j2k_encode_plane_s16(p.ctypes.data, H, W, C.byref(prm), out.ctypes.data, out.size, None)
    if n < 0:
        raise RuntimeError(f"qo_j2k_encode_plane_s16 failed: {n}")
    return out[:n].tobytes()


def decode_s16(codestream: bytes) -> np.ndarray:
    """Pillow hands 16-bit signed single-component streams back as uint16 with a 32768 offset."""
    return (decode(codestream).astype(np.int32) - 32768).astype(np.int16)


def coefficients(plane, reversible, levels, base_step=1.0 / 256) -> np.ndarray:
    p = np.ascontiguousarray(plane, dtype=np.uint8)
    H, W = p.shape
    out = np.empty((H, W), dtype=np.int32)
    _lib().qo_j2k_coefficients(p.ctypes.data, H, W, int(reversible), levels, base_step, out.ctypes.data)
    return out


def dwt97(plane, levels) -> np.ndarray:
    p = np.ascontiguousarray(plane, dtype=np.uint8)
    H, W = p.shape
    out = np.empty((H, W), dtype=np.float32)
    _lib().qo_j2k_dwt97_plane(p.ctypes.data, H, W, levels, out.ctypes.data)
    return out


def t1_block(vals: np.ndarray, orient: int):
    v = np.ascontiguousarray(vals, dtype=np.int32)
    h, w = v.shape
    out = np.empty(w * h * 4 + 64, dtype=np.uint8)
    npasses, nbps = C.c_int(), C.c_int()
    pl = np.zeros(128, dtype=np.int32)
    n = _lib().qo_j2k_t1_block(v.ctypes.data, w, h, orient, out.ctypes.data, C.byref(npasses), C.byref(nbps), pl.ctypes.data)
    return out[:n].tobytes(), npasses.value, nbps.value, pl[:npasses.value].copy()


def own_decode(codestream: bytes, layers=0) -> np.ndarray:
    """The restatement's own decoder (T.800 Annex B parse, C.3 MQ decoder, D, E.1.1, F.3): uint8, or int16 for 16-bit signed."""
    cs = bytes(codestream)
    W, H, prec, sg = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    # geometry first (SIZ is at a fixed place: SOC, SIZ marker, Lsiz, Rsiz, Xsiz, Ysiz)
    if len(cs) < 44 or cs[:4] != b"\xff\x4f\xff\x51":
        raise RuntimeError("not a JPEG 2000 code-stream (SOC/SIZ missing)")
    w, h = int.from_bytes(cs[8:12], "big"), int.from_bytes(cs[12:16], "big")
    if not (1 <= w <= 16384 and 1 <= h <= 16384):
        raise RuntimeError(f"code-stream announces a {w}x{h} picture")
    p16 = (cs[42] & 0x7F) + 1 > 8
    out = np.empty((h, w), dtype=np.int16 if p16 else np.uint8)
    r = _lib().qo_j2k_decode_plane(cs, len(cs), layers, out.ctypes.data, C.byref(W), C.byref(H), C.byref(prec), C.byref(sg))
    if r != 0:
        raise RuntimeError(f"qo_j2k_decode_plane failed: {r}")
    assert (W.value, H.value) == (w, h)
    return out


def decode(codestream: bytes, layers=None, reduce=0) -> np.ndarray:
    """Independent decoder: OpenJPEG via Pillow."""
    from PIL import Image
    im = Image.open(io.BytesIO(codestream))
    if reduce:
        im.reduce = reduce
    if layers is not None:
        # Pillow only forwards .layers to the decoder when .reduce is set; patch the tile descriptor directly.
        im.layers = layers
        t = im.tile[0]
        args = (t[3][0], t[3][1], layers) + tuple(t[3][3:])
        im.tile = [type(t)(t[0], t[1], t[2], args)] if hasattr(t, "_fields") else [(t[0], t[1], t[2], args)]
    im.load()
    return np.asarray(im)


def psnr(a, b) -> float:
    d = a.astype(np.float64) - b.astype(np.float64)
    m = float((d * d).mean())
    return 99.0 if m == 0 else 10 * np.log10(255.0 ** 2 / m)


def corrupt_headers(cs: bytes):
    """Malformed variants of a valid code-stream's main header (name, bytes): every one must be refused, none may crash."""
    cs = bytes(cs)
    cod = cs.index(b"\xff\x52")
    qcd = cs.index(b"\xff\x5c")
    sot = cs.index(b"\xff\x90")
    out = []
    out.append(("cut inside SIZ", cs[:30]))
    out.append(("cut before SOT", cs[:sot]))
    out.append(("Lsiz too short", cs[:4] + b"\x00\x0a" + cs[6:]))
    out.append(("Lcod runs past the end", cs[:cod + 2] + b"\xff\xf0" + cs[cod + 4:]))
    out.append(("code-block exponent 30", cs[:cod + 4 + 6] + b"\x1e" + cs[cod + 4 + 7:]))
    out.append(("Lqcd too short for the subbands", cs[:qcd + 2] + b"\x00\x03" + cs[qcd + 4:qcd + 5] + cs[sot:]))
    out.append(("zero width", cs[:8] + b"\x00\x00\x00\x00" + cs[12:]))
    out.append(("width 1 << 20", cs[:8] + b"\x00\x10\x00\x00" + cs[12:]))
    out.append(("zero layers", cs[:cod + 4 + 2] + b"\x00\x00" + cs[cod + 4 + 4:]))
    out.append(("garbage where a marker should be", cs[:cod] + b"\x12\x34" + cs[cod + 2:]))
    siz2 = b"\xff\x51" + cs[4:8] + (4096).to_bytes(4, "big") + (4096).to_bytes(4, "big") + cs[16:cod]
    out.append(("a second SIZ announcing another size", cs[:sot] + siz2 + cs[sot:]))
    return out
