"""Seeded synthetic I420 clips (no real video is available offline).

The clip families follow SURVEY.md section 8(d): they are inputs, not part of the codec.
All generators use numpy's PCG64 (`default_rng(seed)`), which is platform-stable, so the same
(kind, seed, geometry) gives the same bytes here, on the GPU box and in the committed goldens.

A clip is a C-contiguous uint8 array of shape (frames, W*H*3//2): planar I420, the layout of
the reference's `low_0` (split.cpp:55-56).
"""
from __future__ import annotations

import numpy as np

KINDS = ("pan", "pan_noise", "scene_cut", "saturating", "noise", "binary", "edge")


def frame_bytes(W: int, H: int) -> int:
    return W * H + 2 * (W // 2) * (H // 2)


def _texture(rng, h, w):
    t = rng.integers(0, 256, size=(h, w)).astype(np.float32)
    for _ in range(4):  # 5-point box blur
        p = np.pad(t, 1, mode="edge")
        t = (p[1:-1, 1:-1] + p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:]) / 5.0
    t = (t - 128.0) * 4.0 + 128.0
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def _pack(y, u, v):
    return np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]).astype(np.uint8)


def make_clip(kind: str, frames: int, W: int, H: int, seed: int = 42) -> np.ndarray:
    """Return (frames, frame_bytes) uint8."""
    if kind not in KINDS:
        raise ValueError(f"unknown clip kind {kind!r}")
    rng = np.random.default_rng(seed)
    cw, ch = W // 2, H // 2
    out = np.empty((frames, frame_bytes(W, H)), dtype=np.uint8)
    if kind in ("pan", "pan_noise", "scene_cut", "edge"):
        m = 4 * frames + 16
        big = [_texture(rng, H + 2 * m, W + 2 * m) for _ in range(2 if kind == "scene_cut" else 1)]
        bigc = [(_texture(rng, ch + 2 * m, cw + 2 * m), _texture(rng, ch + 2 * m, cw + 2 * m)) for _ in big]
        for t in range(frames):
            k = 0
            if kind == "scene_cut" and t >= frames // 2:
                k = 1
            if kind == "edge":
                dy = (3 * t) * (1 if (t // 2) % 2 == 0 else -1)
                dx = (2 * t) * (1 if (t // 3) % 2 == 0 else -1)
            else:
                dy, dx = t, 2 * t
            oy, ox = m + dy, m + dx
            y = big[k][oy:oy + H, ox:ox + W].astype(np.int16)
            u = bigc[k][0][m + dy // 2:m + dy // 2 + ch, m + dx // 2:m + dx // 2 + cw].astype(np.int16)
            v = bigc[k][1][m + dy // 2:m + dy // 2 + ch, m + dx // 2:m + dx // 2 + cw].astype(np.int16)
            if kind == "pan_noise":
                y = y + np.rint(rng.normal(0, 2, size=y.shape)).astype(np.int16)
                u = u + np.rint(rng.normal(0, 2, size=u.shape)).astype(np.int16)
                v = v + np.rint(rng.normal(0, 2, size=v.shape)).astype(np.int16)
            if kind == "edge":
                y = y + rng.integers(-3, 4, size=y.shape).astype(np.int16)
            out[t] = _pack(np.clip(y, 0, 255), np.clip(u, 0, 255), np.clip(v, 0, 255))
    elif kind == "saturating":
        bh, bw = (H + 15) // 16 + 8, (W + 15) // 16 + 8
        mosaic = (rng.integers(0, 2, size=(bh, bw)) * 255).astype(np.uint8)
        full = np.kron(mosaic, np.ones((16, 16), dtype=np.uint8))
        mos_c = (rng.integers(0, 2, size=(bh, bw)) * 255).astype(np.uint8)
        fullc = np.kron(mos_c, np.ones((8, 8), dtype=np.uint8))
        for t in range(frames):
            oy, ox = 32 + (t % 5), 32 + 2 * (t % 7)
            y = full[oy:oy + H, ox:ox + W].copy()
            # a sub-rectangle moving differently so displaced blocks overlap
            ry, rx = H // 4, W // 4
            sy, sx = 32 + 3 * (t % 4), 32 - 2 * (t % 5) + 16
            y[ry:ry + H // 3, rx:rx + W // 3] = full[sy:sy + H // 3, sx:sx + W // 3]
            u = fullc[16 + t % 3:16 + t % 3 + ch, 16 + t % 4:16 + t % 4 + cw]
            v = fullc[20 + t % 2:20 + t % 2 + ch, 24 + t % 3:24 + t % 3 + cw]
            out[t] = _pack(y, u, v)
    elif kind == "noise":
        out[:] = rng.integers(0, 256, size=out.shape, dtype=np.uint8)
    elif kind == "binary":
        for t in range(frames):
            my = (rng.integers(0, 2, size=((H + 15) // 16, (W + 15) // 16)) * 255).astype(np.uint8)
            y = np.kron(my, np.ones((16, 16), dtype=np.uint8))[:H, :W]
            mu = (rng.integers(0, 2, size=((ch + 7) // 8, (cw + 7) // 8)) * 255).astype(np.uint8)
            u = np.kron(mu, np.ones((8, 8), dtype=np.uint8))[:ch, :cw]
            mv = (rng.integers(0, 2, size=((ch + 7) // 8, (cw + 7) // 8)) * 255).astype(np.uint8)
            v = np.kron(mv, np.ones((8, 8), dtype=np.uint8))[:ch, :cw]
            out[t] = _pack(y, u, v)
    return out
