"""ctypes binding of libqsvc_hip.so (include/qsvc_hip.h) -- the only compute backend of this package.

There is no CPU fallback: if the shared library is missing or no GPU is usable, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QSVC_HIP_LIB") or os.path.join(_HERE, "libqsvc_hip.so")  # the override is a kernel-tuning hook (A/B builds)

LOW, HIGH, PRED = 0, 1, 2
MOTION_RAW, MOTION_FILTERED = 0, 1

# every symbol include/qsvc_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "qsvc_init", "qsvc_destroy", "qsvc_last_error", "qsvc_stream", "qsvc_sync", "qsvc_seq_config",
    "qsvc_set_mctf_options", "qsvc_put_frames", "qsvc_get_frames", "qsvc_put_motion", "qsvc_get_motion", "qsvc_put_frame_types",
    "qsvc_get_frame_types", "qsvc_get_histograms", "qsvc_motion_estimate", "qsvc_decorrelate", "qsvc_update",
    "qsvc_analyze_step", "qsvc_un_update", "qsvc_correlate", "qsvc_synthesize_step", "qsvc_halo_bytes", "qsvc_halo_export_host", "qsvc_halo_import_host", "qsvc_comm_unique_id",
    "qsvc_comm_init", "qsvc_comm_destroy", "qsvc_halo_exchange", "qsvc_halo_batch_bytes", "qsvc_halo_batch_export_host", "qsvc_halo_batch_import_host",
    "qsvc_halo_exchange_batch", "qsvc_dwt53_trunc",
    "qsvc_j2k_encode_frames", "qsvc_j2k_encode_multi", "qsvc_j2k_encode_plane", "qsvc_j2k_encode_planes_s16", "qsvc_motion_encode", "qsvc_compress_code", "qsvc_j2k_dwt", "qsvc_j2k_decode_planes", "qsvc_j2k_decode_frames", "qsvc_j2k_decode_multi", "qsvc_j2k_stats", "qsvc_set_tier2_threads", "qsvc_set_tuning", "qsvc_set_profiling", "qsvc_get_timings",
]


class J2kParams(C.Structure):
    _fields_ = [("reversible", C.c_int), ("levels", C.c_int), ("layers", C.c_int), ("cblk_w", C.c_int),
                ("cblk_h", C.c_int), ("use_sop", C.c_int), ("base_step", C.c_float), ("slopes", C.c_int * 16)]


class QsvcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"qsvc error {code}: {msg}")
        self.code = code


_lib = None


def load_library() -> C.CDLL:
    """Load libqsvc_hip.so; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built: run `make -C qsvc_amd/csrc` (or __graft_entry__.build())")
    L = C.CDLL(LIB_PATH)
    vp, i, u8p, i16p = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p
    L.qsvc_init.argtypes = [i, C.POINTER(vp)]
    L.qsvc_destroy.argtypes = [vp]; L.qsvc_destroy.restype = None
    L.qsvc_last_error.argtypes = [vp]; L.qsvc_last_error.restype = C.c_char_p
    L.qsvc_stream.argtypes = [vp]; L.qsvc_stream.restype = vp
    L.qsvc_sync.argtypes = [vp]
    L.qsvc_seq_config.argtypes = [vp, i, i, i]
    L.qsvc_set_mctf_options.argtypes = [vp, i, i, i]
    L.qsvc_put_frames.argtypes = [vp, i, i, i, i, i, u8p]
    L.qsvc_get_frames.argtypes = [vp, i, i, i, i, u8p]
    L.qsvc_put_motion.argtypes = [vp, i, i, i, i, i16p]
    L.qsvc_get_motion.argtypes = [vp, i, i, i, i, i16p]
    L.qsvc_put_frame_types.argtypes = [vp, i, i, C.c_char_p]
    L.qsvc_get_frame_types.argtypes = [vp, i, i, C.c_char_p]
    L.qsvc_get_histograms.argtypes = [vp, i, i, vp]
    L.qsvc_motion_estimate.argtypes = [vp, i, i, i, i]
    L.qsvc_decorrelate.argtypes = [vp, i, i, i, i, i]
    L.qsvc_update.argtypes = [vp, i, i, i, C.c_float]
    L.qsvc_analyze_step.argtypes = [vp, i, i, i, i, i, C.c_float]
    L.qsvc_un_update.argtypes = [vp, i, i, i, C.c_float]
    L.qsvc_correlate.argtypes = [vp, i, i, i, i]
    L.qsvc_synthesize_step.argtypes = [vp, i, i, i, i, C.c_float]
    L.qsvc_halo_bytes.argtypes = [vp, i]; L.qsvc_halo_bytes.restype = C.c_size_t
    L.qsvc_halo_export_host.argtypes = [vp, i, i, vp]
    L.qsvc_halo_import_host.argtypes = [vp, i, i, vp]
    L.qsvc_comm_unique_id.argtypes = [vp, C.c_size_t]
    L.qsvc_comm_init.argtypes = [vp, i, i, vp, C.c_size_t]
    L.qsvc_comm_destroy.argtypes = [vp]
    L.qsvc_halo_exchange.argtypes = [vp, i]
    L.qsvc_halo_batch_bytes.argtypes = [C.POINTER(vp), i, i]; L.qsvc_halo_batch_bytes.restype = C.c_size_t
    L.qsvc_halo_batch_export_host.argtypes = [C.POINTER(vp), i, i, i, vp]
    L.qsvc_halo_batch_import_host.argtypes = [C.POINTER(vp), i, i, i, vp]
    L.qsvc_halo_exchange_batch.argtypes = [C.POINTER(vp), i, i]
    L.qsvc_dwt53_trunc.argtypes = [vp, i16p, i, i, i, i, i]
    L.qsvc_j2k_encode_frames.argtypes = [vp, i, i, i, i, C.POINTER(J2kParams), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qsvc_j2k_encode_multi.argtypes = [vp, i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(J2kParams), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qsvc_j2k_encode_plane.argtypes = [vp, u8p, i, i, C.POINTER(J2kParams), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qsvc_j2k_encode_planes_s16.argtypes = [vp, i16p, i, i, i, C.POINTER(J2kParams), u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qsvc_motion_encode.argtypes = [vp, i, C.POINTER(C.c_int), C.POINTER(C.c_int), i, vp, vp, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    IP = C.POINTER(C.c_int)
    L.qsvc_compress_code.argtypes = [vp, i, IP, IP, IP, IP, C.POINTER(J2kParams), i, IP, IP, i, vp, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qsvc_j2k_dwt.argtypes = [vp, u8p, i, i, i, i, vp]
    L.qsvc_j2k_decode_planes.argtypes = [vp, i, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), i, vp, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.qsvc_j2k_decode_frames.argtypes = [vp, i, i, i, i, i, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), i]
    L.qsvc_j2k_decode_multi.argtypes = [vp, i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), i]
    L.qsvc_j2k_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.qsvc_set_tier2_threads.argtypes = [vp, i]
    L.qsvc_set_tuning.argtypes = [vp, C.c_char_p, i]
    L.qsvc_set_profiling.argtypes = [vp, i]
    L.qsvc_get_timings.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_int), i]
    _lib = L
    return L


def frame_bytes(W: int, H: int) -> int:
    return W * H + 2 * (W // 2) * (H // 2)


def j2k_params(reversible=False, levels=4, layers=1, cblk=64, use_sop=True, base_step=1.0 / 256, slopes=()):
    p = J2kParams()
    p.reversible, p.levels, p.layers, p.cblk_w, p.cblk_h, p.use_sop = int(reversible), levels, layers, cblk, cblk, int(use_sop)
    p.base_step = base_step
    for k in range(16):
        p.slopes[k] = int(slopes[k]) if k < len(slopes) else 0
    return p


class Engine:
    """One GPU context (one per process / per GPU)."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        r = self.L.qsvc_init(device, C.byref(h))
        if r != 0 or not h:
            raise QsvcError(r, "qsvc_init failed: no usable HIP device (this package has no CPU fallback)")
        self.h = h
        self.W = self.H = self.F = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.qsvc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, r):
        if r != 0:
            raise QsvcError(r, self.L.qsvc_last_error(self.h).decode())

    # ---- sequence / files
    def seq_config(self, W, H, keep_prediction=False):
        self._chk(self.L.qsvc_seq_config(self.h, W, H, int(keep_prediction)))
        self.W, self.H, self.F = W, H, frame_bytes(W, H)

    def set_mctf_options(self, block_overlaping=0, border_size=0, subpixel_accuracy=0):
        self._chk(self.L.qsvc_set_mctf_options(self.h, block_overlaping, border_size, subpixel_accuracy))

    def put_frames(self, kind, t, frames: np.ndarray, first=0, total=None):
        a = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, self.F)
        total = a.shape[0] if total is None else total
        self._chk(self.L.qsvc_put_frames(self.h, kind, t, first, a.shape[0], total, a.ctypes.data))

    def get_frames(self, kind, t, first, count) -> np.ndarray:
        out = np.empty((count, self.F), dtype=np.uint8)
        self._chk(self.L.qsvc_get_frames(self.h, kind, t, first, count, out.ctypes.data))
        return out

    def put_motion(self, kind, t, mv: np.ndarray, block_size):
        a = np.ascontiguousarray(mv, dtype=np.int16)
        self._chk(self.L.qsvc_put_motion(self.h, kind, t, a.shape[0], block_size, a.ctypes.data))

    def get_motion(self, kind, t, nB, block_size) -> np.ndarray:
        out = np.empty((nB, 4, self.H // block_size, self.W // block_size), dtype=np.int16)
        self._chk(self.L.qsvc_get_motion(self.h, kind, t, nB, block_size, out.ctypes.data))
        return out

    def put_frame_types(self, t, types: bytes):
        self._chk(self.L.qsvc_put_frame_types(self.h, t, len(types), types))

    def get_frame_types(self, t, nB) -> bytes:
        buf = C.create_string_buffer(nB + 1)
        self._chk(self.L.qsvc_get_frame_types(self.h, t, nB, buf))
        return buf.raw[:nB]

    def get_histograms(self, t, nB) -> np.ndarray:
        out = np.empty((nB, 3, 256), dtype=np.uint32)
        self._chk(self.L.qsvc_get_histograms(self.h, t, nB, out.ctypes.data))
        return out

    # ---- stages
    def motion_estimate(self, t, pictures, block_size, search_range):
        self._chk(self.L.qsvc_motion_estimate(self.h, t, pictures, block_size, search_range))

    def decorrelate(self, t, pictures, block_size, search_range, always_B=0):
        self._chk(self.L.qsvc_decorrelate(self.h, t, pictures, block_size, search_range, always_B))

    def update(self, t, pictures, block_size, update_factor=0.25):
        self._chk(self.L.qsvc_update(self.h, t, pictures, block_size, update_factor))

    def analyze_step(self, t, pictures, block_size, search_range, always_B=0, update_factor=0.25):
        self._chk(self.L.qsvc_analyze_step(self.h, t, pictures, block_size, search_range, always_B, update_factor))

    # ---- inverse temporal level (synthesize_step.py): low_t, high_t, motion (FILTERED), frame types -> low_{t-1}
    def un_update(self, t, pictures, block_size, update_factor=0.25):
        self._chk(self.L.qsvc_un_update(self.h, t, pictures, block_size, update_factor))

    def correlate(self, t, pictures, block_size, search_range):
        self._chk(self.L.qsvc_correlate(self.h, t, pictures, block_size, search_range))

    def synthesize_step(self, t, pictures, block_size, search_range, update_factor=0.25):
        self._chk(self.L.qsvc_synthesize_step(self.h, t, pictures, block_size, search_range, update_factor))

    def sync(self):
        self._chk(self.L.qsvc_sync(self.h))

    # ---- open-GOP boundary records
    def halo_bytes(self, block_size) -> int:
        return int(self.L.qsvc_halo_bytes(self.h, block_size))

    def halo_export(self, t, side, block_size) -> np.ndarray:
        out = np.empty(self.halo_bytes(block_size), dtype=np.uint8)
        self._chk(self.L.qsvc_halo_export_host(self.h, t, side, out.ctypes.data))
        return out

    def halo_import(self, t, side, record):
        if record is None:
            self._chk(self.L.qsvc_halo_import_host(self.h, t, side, None))
        else:
            a = np.ascontiguousarray(record, dtype=np.uint8)
            self._chk(self.L.qsvc_halo_import_host(self.h, t, side, a.ctypes.data))

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        r = load_library().qsvc_comm_unique_id(buf, 128)
        if r != 0:
            raise QsvcError(r, "cannot create an RCCL unique id")
        return buf.raw

    def comm_init(self, rank, nranks, uid: bytes):
        self._chk(self.L.qsvc_comm_init(self.h, rank, nranks, uid, len(uid)))

    def halo_exchange(self, t):
        self._chk(self.L.qsvc_halo_exchange(self.h, t))

    # ---- the records of several contexts of one rank as one message per neighbour (engines[0] owns buffers and communicator)
    @staticmethod
    def _handles(engines):
        return (C.c_void_p * len(engines))(*[e.h for e in engines])

    @staticmethod
    def halo_batch_export(engines, t, side) -> np.ndarray:
        e0, hs = engines[0], Engine._handles(engines)
        n = e0.L.qsvc_halo_batch_bytes(hs, len(engines), t)
        buf = np.empty(n, dtype=np.uint8)
        e0._chk(e0.L.qsvc_halo_batch_export_host(hs, len(engines), t, side, buf.ctypes.data))
        return buf

    @staticmethod
    def halo_batch_import(engines, t, side, data):
        e0, hs = engines[0], Engine._handles(engines)
        if data is None:
            e0._chk(e0.L.qsvc_halo_batch_import_host(hs, len(engines), t, side, None))
            return
        a = np.ascontiguousarray(data, dtype=np.uint8)
        assert a.size == e0.L.qsvc_halo_batch_bytes(hs, len(engines), t)
        e0._chk(e0.L.qsvc_halo_batch_import_host(hs, len(engines), t, side, a.ctypes.data))

    @staticmethod
    def halo_exchange_batch(engines, t):
        e0 = engines[0]
        e0._chk(e0.L.qsvc_halo_exchange_batch(Engine._handles(engines), len(engines), t))

    # ---- transforms
    def dwt53_trunc(self, plane: np.ndarray, levels: int, inverse=False) -> np.ndarray:
        p = np.ascontiguousarray(plane, dtype=np.int16).copy()
        H, W = p.shape
        self._chk(self.L.qsvc_dwt53_trunc(self.h, p.ctypes.data, H, W, W, levels, int(inverse)))
        return p

    # ---- JPEG 2000
    def j2k_encode_frames(self, kind, t, first, count, params: J2kParams, cap=None):
        cap = cap or (count * self.F * 2 + (1 << 20))
        out = np.empty(cap, dtype=np.uint8)
        offs = (C.c_size_t * (3 * count + 1))()
        self._chk(self.L.qsvc_j2k_encode_frames(self.h, kind, t, first, count, C.byref(params), out.ctypes.data, cap, offs))
        o = list(offs)
        return [out[o[k]:o[k + 1]].tobytes() for k in range(3 * count)]

    def j2k_encode_multi(self, items, copy=True):
        """items: [(kind, t, first, count, J2kParams)] -> list (per item) of lists of code-streams, one batch on the GPU.
        copy=False returns memoryviews into a buffer owned by the engine, valid until its next j2k call."""
        items = [it for it in items if it[3] > 0]
        n = len(items)
        if n == 0:
            return []
        IA = C.c_int * n
        kinds, ts = IA(*[it[0] for it in items]), IA(*[it[1] for it in items])
        firsts, counts = IA(*[it[2] for it in items]), IA(*[it[3] for it in items])
        prms = (J2kParams * n)(*[it[4] for it in items])
        total = sum(it[3] for it in items)
        cap = total * self.F * 2 + (1 << 20)
        if getattr(self, "_j2k_out", None) is None or self._j2k_out.size < cap:
            self._j2k_out = np.empty(cap, dtype=np.uint8)
        out = self._j2k_out
        offs = (C.c_size_t * (3 * total + 1))()
        self._chk(self.L.qsvc_j2k_encode_multi(self.h, n, kinds, ts, firsts, counts, prms, out.ctypes.data, out.size, offs))
        o = list(offs)
        res, k = [], 0
        for it in items:
            if copy:
                res.append([out[o[k + j]:o[k + j + 1]].tobytes() for j in range(3 * it[3])])
            else:
                mv = memoryview(out)
                res.append([mv[o[k + j]:o[k + j + 1]] for j in range(3 * it[3])])
            k += 3 * it[3]
        return res

    def j2k_encode_plane(self, plane: np.ndarray, params: J2kParams) -> bytes:
        p = np.ascontiguousarray(plane, dtype=np.uint8)
        H, W = p.shape
        cap = H * W * 2 + (1 << 16)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t()
        self._chk(self.L.qsvc_j2k_encode_plane(self.h, p.ctypes.data, H, W, C.byref(params), out.ctypes.data, cap, C.byref(n)))
        return out[:n.value].tobytes()

    def j2k_encode_planes_s16(self, planes: np.ndarray, params: J2kParams):
        """planes: (n, H, W) int16 signed -> n lossless code-streams (Sprecision=16, Ssigned=yes, Clevels=0)."""
        p = np.ascontiguousarray(planes, dtype=np.int16)
        n, H, W = p.shape
        cap = n * (H * W * 4 + 4096)
        out = np.empty(cap, dtype=np.uint8)
        offs = (C.c_size_t * (n + 1))()
        self._chk(self.L.qsvc_j2k_encode_planes_s16(self.h, p.ctypes.data, n, H, W, C.byref(params), out.ctypes.data, cap, offs))
        o = list(offs)
        return [out[o[k]:o[k + 1]].tobytes() for k in range(n)]

    def compress_code(self, items, levels, block_size, copy=True):
        """items: [(kind, t, first, count, J2kParams)] as j2k_encode_multi; levels: [(t, fields)] as motion_encode ->
        (texture streams per item, {t: motion streams}): textures and motion fields of a sequence in one batch (qsvc_compress_code)."""
        items = [it for it in items if it[3] > 0]
        n, nl = len(items), len(levels)
        IA = C.c_int * max(n, 1)
        kinds, ts = IA(*[it[0] for it in items]), IA(*[it[1] for it in items])
        firsts, counts = IA(*[it[2] for it in items]), IA(*[it[3] for it in items])
        prms = (J2kParams * max(n, 1))(*[it[4] for it in items])
        MA = C.c_int * max(nl, 1)
        mts, mfs = MA(*[t for t, _ in levels]), MA(*[f for _, f in levels])
        total, mtotal = sum(it[3] for it in items), sum(f for _, f in levels)
        BY, BX = self.H // block_size, self.W // block_size
        cap = total * self.F * 2 + mtotal * 4 * (BY * BX * 4 + 4096) + (1 << 20)
        if getattr(self, "_j2k_out", None) is None or self._j2k_out.size < cap:
            self._j2k_out = np.empty(cap, dtype=np.uint8)
        out = self._j2k_out
        offs = (C.c_size_t * (3 * total + 4 * mtotal + 1))()
        self._chk(self.L.qsvc_compress_code(self.h, n, kinds, ts, firsts, counts, prms, nl, mts, mfs, block_size, None, out.ctypes.data, out.size, offs))
        o, mv = list(offs), memoryview(out)
        cut = (lambda a, b: out[a:b].tobytes()) if copy else (lambda a, b: mv[a:b])
        tex, k = [], 0
        for it in items:
            tex.append([cut(o[k + j], o[k + j + 1]) for j in range(3 * it[3])])
            k += 3 * it[3]
        mot = {}
        for t, f in levels:
            mot[t] = [cut(o[k + j], o[k + j + 1]) for j in range(4 * f)]
            k += 4 * f
        return tex, mot

    def motion_encode(self, levels, block_size, want_residues=False, copy=True, filtered=None):
        """levels: [(t, fields)] in ascending t -> ({t: [code-stream per (field, component)]}, {t: motion_residue_t} or None): the whole
        motion-field coding chain of `mctf compress` in one call (qsvc_motion_encode).  filtered: {t: motion_filtered_t} from the host
        instead of this context's resident fields."""
        levels = [(t, f) for (t, f) in levels]
        n = len(levels)
        BY, BX = self.H // block_size, self.W // block_size
        total = sum(f for _, f in levels)
        ts, fs = (C.c_int * n)(*[t for t, _ in levels]), (C.c_int * n)(*[f for _, f in levels])
        cap = total * 4 * (BY * BX * 4 + 4096) + 4096
        if getattr(self, "_mv_out", None) is None or self._mv_out.size < cap:
            self._mv_out = np.empty(cap, dtype=np.uint8)
        out = self._mv_out
        offs = (C.c_size_t * (4 * total + 1))()
        res = np.empty((total, 4, BY, BX), dtype=np.int16) if want_residues else None
        fh = None
        if filtered is not None:
            fh = np.ascontiguousarray(np.concatenate([np.asarray(filtered[t], dtype=np.int16).reshape(f, 4, BY, BX) for t, f in levels]))
        self._chk(self.L.qsvc_motion_encode(self.h, n, ts, fs, block_size, fh.ctypes.data if fh is not None else None,
                                            res.ctypes.data if res is not None else None, out.ctypes.data, out.size, offs))
        o = list(offs)
        mv = memoryview(out)
        streams, residues, k, r = {}, ({} if want_residues else None), 0, 0
        for t, f in levels:
            streams[t] = [out[o[k + j]:o[k + j + 1]].tobytes() if copy else mv[o[k + j]:o[k + j + 1]] for j in range(4 * f)]
            k += 4 * f
            if want_residues:
                residues[t] = res[r:r + f]
                r += f
        return streams, residues

    def j2k_dwt(self, plane: np.ndarray, reversible: bool, levels: int) -> np.ndarray:
        p = np.ascontiguousarray(plane, dtype=np.uint8)
        H, W = p.shape
        out = np.empty((H, W), dtype=np.int32 if reversible else np.float32)
        self._chk(self.L.qsvc_j2k_dwt(self.h, p.ctypes.data, H, W, int(reversible), levels, out.ctypes.data))
        return out

    def j2k_decode_planes(self, streams, max_layers=0):
        """Code-streams (bytes) -> list of sample planes: uint8 (8-bit unsigned) or int16 (16-bit signed)."""
        streams = [bytes(s) for s in streams]
        n = len(streams)
        if n == 0:
            return []
        geo = []
        for cs in streams:  # SOC, SIZ marker, Lsiz, Rsiz, Xsiz, Ysiz ... Ssiz at byte 42
            if len(cs) < 44 or cs[:4] != b"\xff\x4f\xff\x51":
                raise QsvcError(-1, "not a JPEG 2000 code-stream (SOC/SIZ missing)")
            w, h = int.from_bytes(cs[8:12], "big"), int.from_bytes(cs[12:16], "big")
            if not (1 <= w <= 16384 and 1 <= h <= 16384):  # the output buffer below is sized from these header fields
                raise QsvcError(-1, f"code-stream announces a {w}x{h} picture: outside this path")
            geo.append((w, h, (cs[42] & 0x7F) + 1 > 8))
        cap = sum(w * h * (2 if p16 else 1) + 2 for (w, h, p16) in geo) + 64
        out = np.empty(cap, dtype=np.uint8)
        ptrs = (C.c_char_p * n)(*streams)
        lens = (C.c_size_t * n)(*[len(s) for s in streams])
        offs = (C.c_size_t * (n + 1))()
        Ws, Hs = (C.c_int * n)(), (C.c_int * n)()
        self._chk(self.L.qsvc_j2k_decode_planes(self.h, n, ptrs, lens, int(max_layers), out.ctypes.data, cap, offs, Ws, Hs))
        res = []
        for k, (w, h, p16) in enumerate(geo):
            a = out[offs[k]:offs[k] + w * h * (2 if p16 else 1)]
            res.append(a.view(np.int16).reshape(h, w).copy() if p16 else a.reshape(h, w).copy())
        return res

    def j2k_decode_frames(self, kind, t, streams, first=0, total=None, max_layers=0):
        """3 code-streams per picture (Y, U, V) -> pictures [first, first + n) of the device-resident sequence (kind, t) of `total`."""
        streams = [bytes(s) for s in streams]
        n = len(streams) // 3
        assert len(streams) == 3 * n and n >= 1
        ptrs = (C.c_char_p * len(streams))(*streams)
        lens = (C.c_size_t * len(streams))(*[len(s) for s in streams])
        self._chk(self.L.qsvc_j2k_decode_frames(self.h, kind, t, first, n, n + first if total is None else total, ptrs, lens, int(max_layers)))

    def j2k_decode_multi(self, items, max_layers=0):
        """items: [(kind, t, [streams: 3 per picture])] -> every sequence resident (its picture count = len(streams) // 3), one batch."""
        items = [(k, t, [bytes(s) for s in ss]) for (k, t, ss) in items if len(ss)]
        n = len(items)
        if not n:
            return
        IA = C.c_int * n
        cnt = [len(ss) // 3 for (_, _, ss) in items]
        allcs = [s for (_, _, ss) in items for s in ss]
        ptrs = (C.c_char_p * len(allcs))(*allcs)
        lens = (C.c_size_t * len(allcs))(*[len(s) for s in allcs])
        self._chk(self.L.qsvc_j2k_decode_multi(self.h, n, IA(*[k for k, _, _ in items]), IA(*[t for _, t, _ in items]), IA(*([0] * n)), IA(*cnt), IA(*cnt),
                                               ptrs, lens, int(max_layers)))

    def j2k_stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.qsvc_j2k_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(codeblocks=a.value, passes=b.value, symbols=c.value)

    def set_tier2_threads(self, n: int):
        """Host threads writing the code-streams of one encode call (0 = default: QSVC_T2_THREADS at the first encode, else 8)."""
        self._chk(self.L.qsvc_set_tier2_threads(self.h, int(n)))

    def set_tuning(self, name: str, value: int):
        """An A/B switch of the kernel forms for this context (include/qsvc_hip.h: qsvc_set_tuning); results never depend on it."""
        self._chk(self.L.qsvc_set_tuning(self.h, name.encode(), int(value)))

    # ---- profiling
    def set_profiling(self, on=True):
        self._chk(self.L.qsvc_set_profiling(self.h, int(on)))

    def timings(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n_l = (C.c_int * cap)()
        n = self.L.qsvc_get_timings(self.h, names, ms, n_l, cap)
        return {names[k].decode(): (ms[k], n_l[k]) for k in range(max(n, 0))}
