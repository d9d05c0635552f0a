"""TEST INFRASTRUCTURE: an independent JPEG 2000 Tier-2 parser (ITU-T T.800 Annex A markers, Annex B packets).

Written from the standard alone -- it shares no code with oracle/j2k_oracle.c or with the product's Tier-2 writer
(qsvc_amd/csrc/j2k_api.hip) -- so that code-streams of DIFFERENT encoders (the restatement, the HIP path, OpenJPEG through
Pillow) can be taken apart into per-code-block facts and compared: zero bit-planes, coding passes per layer, and the MQ
code-word segment bytes.  Scope: what the hot path emits and what OpenJPEG emits at matching parameters -- one tile, one
tile-part, one component, LRCP, no precinct partition (PPx = PPy = 15), code-block style 0 (one code-word segment per
block), SOP / EPH markers optional.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field


@dataclass
class Subband:
    res: int          # resolution level r (0 = LL_L)
    orient: int       # 0 LL, 1 HL, 2 LH, 3 HH
    w: int
    h: int
    eps: int          # exponent of the QCD entry
    mant: int
    gx: int = 0
    gy: int = 0
    blocks: list = field(default_factory=list)  # row-major code-blocks


@dataclass
class Block:
    w: int
    h: int
    zbp: int = -1                 # zero bit-planes (tag tree value), -1 while never included
    passes: list = field(default_factory=list)   # coding passes contributed per layer
    lengths: list = field(default_factory=list)  # bytes contributed per layer
    data: bytearray = field(default_factory=bytearray)
    lblock: int = 3

    @property
    def npasses(self):
        return sum(self.passes)


@dataclass
class Stream:
    W: int
    H: int
    prec: int
    signed: bool
    levels: int
    layers: int
    reversible: bool
    guard: int
    sop: bool
    eph: bool
    cbw: int
    cbh: int
    subbands: list

    def blocks(self):
        for sb in self.subbands:
            for b in sb.blocks:
                yield sb, b


class _Bits:
    """Packet-header bit reader with the stuffing rule of B.10.1 (after 0xFF the next byte carries 7 bits)."""

    def __init__(self, data, pos):
        self.d, self.p, self.acc, self.n, self.last = data, pos, 0, 0, 0

    def bit(self):
        if self.n == 0:
            b = self.d[self.p]
            self.p += 1
            self.n = 7 if self.last == 0xFF else 8
            self.acc, self.last = b, b
        self.n -= 1
        return (self.acc >> self.n) & 1

    def bits(self, k):
        v = 0
        for _ in range(k):
            v = (v << 1) | self.bit()
        return v

    def align(self):
        """End of the header: drop the rest of the byte; a final 0xFF is followed by a stuffed byte."""
        self.n = 0
        if self.last == 0xFF:
            self.p += 1
            self.last = 0
        return self.p


class _TagTree:
    """B.10.2 decoder."""

    def __init__(self, w, h):
        self.lv = []
        while True:
            self.lv.append((w, h))
            if w == 1 and h == 1:
                break
            w, h = (w + 1) // 2, (h + 1) // 2
        self.val = [[0] * (a * b) for a, b in self.lv]     # current lower bound
        self.done = [[False] * (a * b) for a, b in self.lv]

    def decode(self, br, x, y, threshold):
        """Returns True when value(x, y) < threshold is established (value then in self.value(x, y))."""
        lo = 0
        for k in range(len(self.lv) - 1, -1, -1):
            w, _ = self.lv[k]
            i = (y >> k) * w + (x >> k)
            if self.val[k][i] < lo:
                self.val[k][i] = lo
            while not self.done[k][i] and self.val[k][i] < threshold:
                if br.bit():
                    self.done[k][i] = True
                else:
                    self.val[k][i] += 1
            lo = self.val[k][i]
            if not self.done[k][i]:
                return False
        return self.val[0][y * self.lv[0][0] + x] < threshold

    def value(self, x, y):
        return self.val[0][y * self.lv[0][0] + x]


def _npasses(br):
    """Table B.4."""
    if not br.bit():
        return 1
    if not br.bit():
        return 2
    v = br.bits(2)
    if v != 3:
        return 3 + v
    v = br.bits(5)
    if v != 31:
        return 6 + v
    return 37 + br.bits(7)


def parse(cs: bytes) -> Stream:
    cs = bytes(cs)
    assert cs[:2] == b"\xff\x4f", "no SOC"
    p = 2
    siz = cod = qcd = None
    while True:
        m = struct.unpack(">H", cs[p:p + 2])[0]
        if m == 0xFF90:  # SOT
            break
        ln = struct.unpack(">H", cs[p + 2:p + 4])[0]
        body = cs[p + 4:p + 2 + ln]
        if m == 0xFF51:
            siz = body
        elif m == 0xFF52:
            cod = body
        elif m == 0xFF5C:
            qcd = body
        p += 2 + ln
    _, X, Y, XO, YO, XT, YT, XTO, YTO, C = struct.unpack(">HIIIIIIIIH", siz[:36])
    assert C == 1 and XO == 0 and YO == 0 and XTO == 0 and YTO == 0 and XT >= X and YT >= Y, "one tile, one component at the origin"
    ssiz, xr, yr = siz[36], siz[37], siz[38]
    assert xr == 1 and yr == 1
    prec, signed = (ssiz & 0x7F) + 1, bool(ssiz & 0x80)
    scod = cod[0]
    prog, layers, mct = cod[1], struct.unpack(">H", cod[2:4])[0], cod[4]
    L, xcb, ycb, style, xform = cod[5], cod[6] + 2, cod[7] + 2, cod[8], cod[9]
    assert prog == 0, "LRCP only"
    assert style == 0, "code-block style 0 only (one code-word segment per block)"
    if scod & 1:
        assert all(b == 0xFF for b in cod[10:10 + L + 1]), "precinct partition not supported"
    cbw, cbh = 1 << xcb, 1 << ycb
    sq = qcd[0]
    guard, qstyle = sq >> 5, sq & 31
    nsb = 1 + 3 * L
    if qstyle == 0:
        qs = [(qcd[1 + i] >> 3, 0) for i in range(nsb)]
    elif qstyle == 2:
        qs = [((v := struct.unpack(">H", qcd[1 + 2 * i:3 + 2 * i])[0]) >> 11, v & 0x7FF) for i in range(nsb)]
    else:
        raise AssertionError("derived quantisation not supported")
    ws, hs = [X], [Y]
    for _ in range(L):
        ws.append((ws[-1] + 1) // 2)
        hs.append((hs[-1] + 1) // 2)
    sbs = [Subband(0, 0, ws[L], hs[L], *qs[0])]
    k = 1
    for r in range(1, L + 1):
        d = L - r + 1  # decomposition level of this resolution's subbands
        lw, lh = ws[d], hs[d]
        hw, hh = ws[d - 1] - lw, hs[d - 1] - lh
        for orient, (w, h) in ((1, (hw, lh)), (2, (lw, hh)), (3, (hw, hh))):
            sbs.append(Subband(r, orient, w, h, *qs[k]))
            k += 1
    for sb in sbs:
        sb.gx, sb.gy = (sb.w + cbw - 1) // cbw, (sb.h + cbh - 1) // cbh
        if sb.w == 0 or sb.h == 0:
            sb.gx = sb.gy = 0
        for by in range(sb.gy):
            for bx in range(sb.gx):
                sb.blocks.append(Block(min(cbw, sb.w - bx * cbw), min(cbh, sb.h - by * cbh)))
    st = Stream(X, Y, prec, signed, L, layers, xform == 1, guard, bool(scod & 2), bool(scod & 4), cbw, cbh, sbs)
    # tile-part
    lsot, isot, psot, tp, ntp = struct.unpack(">HHIBB", cs[p + 2:p + 12])
    assert isot == 0 and tp == 0
    end = p + psot if psot else len(cs) - 2
    p += 12
    while cs[p:p + 2] != b"\xff\x93":  # tile-part header markers up to SOD
        ln = struct.unpack(">H", cs[p + 2:p + 4])[0]
        p += 2 + ln
    p += 2
    incl = [_TagTree(sb.gx, sb.gy) if sb.blocks else None for sb in sbs]
    zbpt = [_TagTree(sb.gx, sb.gy) if sb.blocks else None for sb in sbs]
    for layer in range(layers):
        for r in range(L + 1):
            if cs[p:p + 2] == b"\xff\x91":  # SOP
                assert cs[p + 2:p + 4] == b"\x00\x04"
                p += 6
            band_ix = [i for i, sb in enumerate(sbs) if sb.res == r]
            br = _Bits(cs, p)
            contrib = []
            if br.bit():
                for i in band_ix:
                    sb = sbs[i]
                    for n, b in enumerate(sb.blocks):
                        bx, by = n % sb.gx, n // sb.gx
                        if b.zbp < 0:
                            if not incl[i].decode(br, bx, by, layer + 1):
                                b.passes.append(0); b.lengths.append(0)
                                continue
                            t = 1
                            while not zbpt[i].decode(br, bx, by, t):
                                t += 1
                            b.zbp = zbpt[i].value(bx, by)
                        elif not br.bit():
                            b.passes.append(0); b.lengths.append(0)
                            continue
                        npass = _npasses(br)
                        while br.bit():
                            b.lblock += 1
                        nbits = b.lblock + (npass.bit_length() - 1)
                        ln = br.bits(nbits)
                        b.passes.append(npass); b.lengths.append(ln)
                        contrib.append((b, ln))
            else:
                for i in band_ix:
                    for b in sbs[i].blocks:
                        b.passes.append(0); b.lengths.append(0)
            p = br.align()
            if cs[p:p + 2] == b"\xff\x92" and st.eph:
                p += 2
            for b, ln in contrib:
                b.data += cs[p:p + ln]
                p += ln
    assert p == end, f"packets end at {p}, tile-part at {end}"
    assert cs[p:p + 2] == b"\xff\xd9", "no EOC after the tile-part"
    return st
