// qsvc_amd/csrc/j2k_decode.hip -- JPEG 2000 DECODER of the texture / motion code-streams this path writes:
// replaces the reference's external `kdu_expand` call (trunk/src/texture_expand_fb_j2k.py:153-162, motion_expand_j2k.py).
//   host  : Tier-2 -- markers, packet headers, tag trees (T.800 Annex A / B) -> per code-block zero bit-planes, passes, segment;
//   device: EBCOT Tier-1 decoding (Annex C.3 MQ decoder + Annex D), one code-block per lane; midpoint dequantisation (E.1.1);
//           inverse DWT 5/3 / 9/7 (F.3), level shift and clamp.
// Scope: single tile, single component, LRCP, no precinct partition, code-block style 0 -- what the encoder here and OpenJPEG at
// matching parameters emit.  The arithmetic follows oracle/j2k_oracle.c (qo_j2k_decode_plane) operation for operation; that
// restatement is pinned by OpenJPEG's decoder.  First correct form: the block decoder is one LANE per code-block with its state
// in global memory (the decode direction of EBCOT has no sample-parallel shortcut: contexts depend on decoded data).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <exception>

#include <algorithm>
#include <vector>

#include "j2k_kernels.h"
#include "qsvc_ctx.h"

using namespace qsvc;

#define HIPCHK(ctx, call)                                    \
  do {                                                       \
    hipError_t e_ = (call);                                  \
    if (e_ != hipSuccess) return (ctx)->hipfail(e_, #call);  \
  } while (0)

namespace {

#define D_GUARD_MAXLEV 10
#define F97_ALPHA (-1.586134342059924f)
#define F97_BETA (-0.052980118572961f)
#define F97_GAMMA (0.882911075530934f)
#define F97_DELTA (0.443506852043971f)
#define F97_K (1.230174104914001f)

struct DecCb {          // one code-block to decode
  unsigned long long seg;  // offset of its code-word segment in the segment buffer
  size_t plane_off;     // element offset of its plane in the coefficient buffer
  int len, npasses, nbps;
  int x0, y0, w, h, W;  // position in the plane's Mallat layout, plane width
  int orient, rev;
  float delta;
};
struct DecPlane { size_t off; int W, H, L, rev, prec, sgnd; size_t out_off; };

// ---- host: bit reader, tag trees, packet headers -------------------------------------------------------------------------
struct BitR {
  const uint8_t* d; size_t n, p; int acc = 0, nb = 0, last = 0;
  int bit() {
    if (nb == 0) { int v = p < n ? d[p] : 0; p++; nb = last == 0xFF ? 7 : 8; acc = v; last = v; }
    nb--;
    return (acc >> nb) & 1;
  }
  unsigned bits(int k) { unsigned v = 0; while (k-- > 0) v = (v << 1) | (unsigned)bit(); return v; }
  size_t align() { nb = 0; if (last == 0xFF) { p++; last = 0; } return p; }
};
struct TagTreeD {
  int nlev = 0, lw[16], lh[16], off[16];
  std::vector<int> val; std::vector<uint8_t> known;
  void init(int w, int h) {
    nlev = 0; int n = 0;
    for (;;) { lw[nlev] = w; lh[nlev] = h; off[nlev] = n; n += w * h; nlev++; if (w == 1 && h == 1) break; w = (w + 1) >> 1; h = (h + 1) >> 1; }
    val.assign(n, 0); known.assign(n, 0);
  }
  bool decode(BitR& b, int x, int y, int threshold) {
    int lo = 0;
    for (int k = nlev - 1; k >= 0; k--) {
      int nd = off[k] + (y >> k) * lw[k] + (x >> k);
      if (val[nd] < lo) val[nd] = lo;
      while (!known[nd] && val[nd] < threshold) { if (b.bit()) known[nd] = 1; else val[nd]++; }
      lo = val[nd];
      if (!known[nd]) return false;
    }
    return val[off[0] + y * lw[0] + x] < threshold;
  }
  int value(int x, int y) const { return val[off[0] + y * lw[0] + x]; }
};
int npasses_of(BitR& b) {  // Table B.4
  if (!b.bit()) return 1;
  if (!b.bit()) return 2;
  unsigned v = b.bits(2);
  if (v != 3) return 3 + (int)v;
  v = b.bits(5);
  if (v != 31) return 6 + (int)v;
  return 37 + (int)b.bits(7);
}
int ilog2d(int v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }
uint32_t be32(const uint8_t* q) { return ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]; }

// One code-stream -> plane geometry + its code-blocks (appended to cbs, segments appended to segs).  Returns 0 or a message.
const char* parse_stream(const uint8_t* cs, size_t n, int max_layers, DecPlane& pl, std::vector<DecCb>& cbs, std::vector<uint8_t>& segs) {
  if (n < 4 || cs[0] != 0xFF || cs[1] != 0x4F) return "no SOC marker";
  size_t p = 2;
  // main header: every marker segment must lie inside the buffer and be long enough for the fields read below; SIZ comes first
  // (A.5.1) and only the first SIZ / COD / QCD count (a later duplicate must not replace a checked one)
  const uint8_t *siz = nullptr, *cod = nullptr, *qcd = nullptr;
  size_t lsiz = 0, lcod = 0, lqcd = 0;
  bool sot = false;
  while (p + 4 <= n) {
    if (cs[p] != 0xFF) return "corrupt main header (marker expected)";
    const int mk = (cs[p] << 8) | cs[p + 1];
    if (mk == 0xFF90) { sot = true; break; }
    const size_t ln = ((size_t)cs[p + 2] << 8) | cs[p + 3];
    if (ln < 2 || p + 2 + ln > n) return "marker segment runs past the end of the code-stream";
    if (mk == 0xFF51) { if (p != 2) return "SIZ must follow SOC"; if (!siz) { siz = cs + p + 4; lsiz = ln - 2; } }
    else if (mk == 0xFF52) { if (!cod) { cod = cs + p + 4; lcod = ln - 2; } }
    else if (mk == 0xFF5C) { if (!qcd) { qcd = cs + p + 4; lqcd = ln - 2; } }
    p += 2 + ln;
  }
  if (!siz || !cod || !qcd || !sot || p + 12 > n) return "missing SIZ / COD / QCD / SOT";
  if (lsiz < 39 || lcod < 10 || lqcd < 1) return "SIZ / COD / QCD segment too short";
  const int W = (int)be32(siz + 2), H = (int)be32(siz + 6);
  if (((siz[34] << 8) | siz[35]) != 1 || be32(siz + 10) || be32(siz + 14) || (int)be32(siz + 18) < W || (int)be32(siz + 22) < H)
    return "one component and one tile at the origin only";
  if (W < 1 || H < 1 || W > 16384 || H > 16384) return "geometry outside this path";
  const int prec = (siz[36] & 0x7F) + 1, sgnd = siz[36] >> 7;
  if (cod[6] > 4 || cod[7] > 4) return "code-blocks larger than 64x64 are not supported";  // also keeps the shifts below defined
  const int scod = cod[0], layers = (cod[2] << 8) | cod[3], L = cod[5], cbw = 1 << (cod[6] + 2), cbh = 1 << (cod[7] + 2), rev = cod[9] == 1;
  if (layers < 1) return "no quality layers";
  if (cod[1] != 0) return "LRCP progression only";
  if (cod[8] != 0) return "code-block style 0 only";
  if (scod & 1) return "precinct partitions are not supported";
  if (cbw > 64 || cbh > 64 || L > D_GUARD_MAXLEV || W < 1 || H < 1 || W > 16384 || H > 16384) return "geometry outside this path";
  if (!(prec == 8 && !sgnd) && !(prec == 16 && sgnd)) return "8-bit unsigned or 16-bit signed samples only";
  const int guard = qcd[0] >> 5, qstyle = qcd[0] & 31;
  if (qstyle != 0 && qstyle != 2) return "derived quantisation is not supported";
  if (lqcd < 1 + (size_t)(3 * L + 1) * (qstyle == 0 ? 1 : 2)) return "QCD shorter than the subband count";
  pl.W = W; pl.H = H; pl.L = L; pl.rev = rev; pl.prec = prec; pl.sgnd = sgnd;
  // subbands in code-stream order
  struct Sb { int x0, y0, w, h, orient, Mb; float delta; int gx, gy, first; };
  std::vector<Sb> sb;
  int ws[D_GUARD_MAXLEV + 1], hs[D_GUARD_MAXLEV + 1];
  ws[0] = W; hs[0] = H;
  for (int d = 1; d <= L; d++) { ws[d] = (ws[d - 1] + 1) >> 1; hs[d] = (hs[d - 1] + 1) >> 1; }
  sb.push_back({0, 0, ws[L], hs[L], 0, 0, 1.0f, 0, 0, 0});
  for (int d = L; d >= 1; d--) {
    int lw = ws[d], lh = hs[d], hw = ws[d - 1] - lw, hh = hs[d - 1] - lh;
    sb.push_back({lw, 0, hw, lh, 1, 0, 1.0f, 0, 0, 0});
    sb.push_back({0, lh, lw, hh, 2, 0, 1.0f, 0, 0, 0});
    sb.push_back({lw, lh, hw, hh, 3, 0, 1.0f, 0, 0, 0});
  }
  int ncb = 0;
  for (size_t s = 0; s < sb.size(); s++) {
    int gain = sb[s].orient == 0 ? 0 : (sb[s].orient == 3 ? 2 : 1), Rb = prec + gain, eps, mant = 0;
    if (qstyle == 0) eps = qcd[1 + s] >> 3;
    else { int v = (qcd[1 + 2 * s] << 8) | qcd[2 + 2 * s]; eps = v >> 11; mant = v & 0x7FF; sb[s].delta = (float)(ldexp(1.0, Rb - eps) * (1.0 + mant / 2048.0)); }
    sb[s].Mb = guard + eps - 1;
    sb[s].gx = sb[s].w > 0 && sb[s].h > 0 ? (sb[s].w + cbw - 1) / cbw : 0;
    sb[s].gy = sb[s].w > 0 && sb[s].h > 0 ? (sb[s].h + cbh - 1) / cbh : 0;
    sb[s].first = ncb; ncb += sb[s].gx * sb[s].gy;
  }
  // tile-part
  p += 12;
  while (p + 4 <= n && !(cs[p] == 0xFF && cs[p + 1] == 0x93)) p += 2 + (size_t)((cs[p + 2] << 8) | cs[p + 3]);
  p += 2;
  struct Blk { int zbp = 0, npasses = 0, lblock = 3, incl = 0; std::vector<uint8_t> data; };
  std::vector<Blk> blk(ncb);
  std::vector<TagTreeD> incl(sb.size()), zbpt(sb.size());
  for (size_t s = 0; s < sb.size(); s++) if (sb[s].gx * sb[s].gy > 0) { incl[s].init(sb[s].gx, sb[s].gy); zbpt[s].init(sb[s].gx, sb[s].gy); }
  const int use_layers = (max_layers > 0 && max_layers < layers) ? max_layers : layers;
  std::vector<int> plen(ncb);
  for (int l = 0; l < layers; l++)
    for (int r = 0; r <= L; r++) {
      if (p + 6 <= n && cs[p] == 0xFF && cs[p + 1] == 0x91) p += 6;  // SOP
      const size_t s0 = r == 0 ? 0 : 1 + 3 * (size_t)(r - 1), s1 = r == 0 ? 1 : s0 + 3;
      BitR br{cs, n, p};
      for (size_t s = s0; s < s1; s++) for (int i = 0; i < sb[s].gx * sb[s].gy; i++) plen[sb[s].first + i] = 0;
      if (br.bit()) {
        for (size_t s = s0; s < s1; s++)
          for (int i = 0; i < sb[s].gx * sb[s].gy; i++) {
            Blk& c = blk[sb[s].first + i];
            const int bx = i % sb[s].gx, by = i / sb[s].gx;
            if (!c.incl) {
              if (!incl[s].decode(br, bx, by, l + 1)) continue;
              int t = 1;
              while (!zbpt[s].decode(br, bx, by, t)) { if (++t > 64) return "corrupt zero-bit-plane tag tree"; }
              c.zbp = zbpt[s].value(bx, by); c.incl = 1;
            } else if (!br.bit()) continue;
            const int np = npasses_of(br);
            while (br.bit()) { if (++c.lblock > 32) return "corrupt Lblock"; }
            const int ln = (int)br.bits(c.lblock + ilog2d(np));
            if (l < use_layers) { c.npasses += np; plen[sb[s].first + i] = ln; }
            else plen[sb[s].first + i] = -ln;
          }
      }
      p = br.align();
      if ((scod & 4) && p + 2 <= n && cs[p] == 0xFF && cs[p + 1] == 0x92) p += 2;  // EPH
      for (size_t s = s0; s < s1; s++)
        for (int i = 0; i < sb[s].gx * sb[s].gy; i++) {
          const int ln = plen[sb[s].first + i];
          if (ln > 0) {
            if (p + (size_t)ln > n) return "packet body beyond the end of the code-stream";
            Blk& c = blk[sb[s].first + i];
            c.data.insert(c.data.end(), cs + p, cs + p + ln);
          }
          p += (size_t)(ln < 0 ? -ln : ln);
        }
    }
  for (size_t s = 0; s < sb.size(); s++)
    for (int i = 0; i < sb[s].gx * sb[s].gy; i++) {
      Blk& c = blk[sb[s].first + i];
      if (!c.incl || c.npasses == 0) continue;
      const int bx = i % sb[s].gx, by = i / sb[s].gx, nbps = sb[s].Mb - c.zbp;
      if (nbps < 1 || nbps > 30) return "code-block bit-plane count out of range";
      DecCb d;
      d.seg = segs.size(); d.plane_off = pl.off; d.len = (int)c.data.size(); d.npasses = c.npasses; d.nbps = nbps;
      d.x0 = sb[s].x0 + bx * cbw; d.y0 = sb[s].y0 + by * cbh; d.W = W;
      d.w = (bx + 1) * cbw <= sb[s].w ? cbw : sb[s].w - bx * cbw;
      d.h = (by + 1) * cbh <= sb[s].h ? cbh : sb[s].h - by * cbh;
      d.orient = sb[s].orient; d.rev = rev; d.delta = sb[s].delta;
      segs.insert(segs.end(), c.data.begin(), c.data.end());
      segs.push_back(0xFF); segs.push_back(0xFF);  // a decoder reads 0xFF beyond the end of a segment (C.3.4)
      cbs.push_back(d);
    }
  return nullptr;
}

// ---- device: MQ decoder + Tier-1 decoding, one code-block per lane ----------------------------------------------------------
__constant__ uint32_t c_dmq[47] = T1_MQ_TABLE_INIT;  // qe | nmps << 16 | nlps << 22 | switch << 28

struct MqDec {
  uint32_t A, C; int CT;
  const uint8_t *bp, *end;
  uint8_t* ctx;   // LDS: ctx[cx * 64] = state << 1 | mps
  __device__ void bytein() {
    if (bp < end) {
      if (*bp == 0xFF) {
        const int nx = bp + 1 < end ? bp[1] : 0xFF;
        if (nx > 0x8F) { C += 0xFF00; CT = 8; } else { bp++; C += (uint32_t)nx << 9; CT = 7; }
      } else { bp++; C += (uint32_t)(bp < end ? *bp : 0xFF) << 8; CT = 8; }
    } else { C += 0xFF00; CT = 8; }
  }
  __device__ void init(const uint8_t* data, int len, uint8_t* lds_ctx) {
    ctx = lds_ctx;
    for (int i = 0; i < 19; i++) ctx[i * 64] = 0;
    ctx[0] = 4 << 1; ctx[17 * 64] = 3 << 1; ctx[18 * 64] = 46 << 1;
    bp = data; end = data + len;
    C = (uint32_t)(len > 0 ? data[0] : 0xFF) << 16;
    bytein();
    C <<= 7; CT -= 7; A = 0x8000;
  }
  __device__ int decode(int cx) {
    const int sm = ctx[cx * 64], st = sm >> 1, mps = sm & 1;
    const uint32_t e = c_dmq[st], qe = e & 0xFFFF;
    int d;
    A -= qe;
    if ((C >> 16) < qe) {
      if (A < qe) { d = mps; ctx[cx * 64] = (uint8_t)((((e >> 16) & 63) << 1) | mps); }
      else { d = 1 - mps; ctx[cx * 64] = (uint8_t)((((e >> 22) & 63) << 1) | (mps ^ ((e >> 28) & 1))); }
      A = qe;
    } else {
      C -= qe << 16;
      if (A & 0x8000) return mps;
      if (A < qe) { d = 1 - mps; ctx[cx * 64] = (uint8_t)((((e >> 22) & 63) << 1) | (mps ^ ((e >> 28) & 1))); }
      else { d = mps; ctx[cx * 64] = (uint8_t)((((e >> 16) & 63) << 1) | mps); }
    }
    do { if (CT == 0) bytein(); A <<= 1; C <<= 1; CT--; } while ((A & 0x8000) == 0);
    return d;
  }
};

enum { DF_SIG = 1, DF_VISIT = 2, DF_REFINED = 4, DF_NEG = 8 };
#define DFL(y, x) fl[((y) + 1) * 66 + (x) + 1]

__device__ __forceinline__ int d_zc(const uint8_t* fl, int y, int x, int orient) {  // Table D.1
  int h = (DFL(y, x - 1) & DF_SIG) + (DFL(y, x + 1) & DF_SIG);
  int v = (DFL(y - 1, x) & DF_SIG) + (DFL(y + 1, x) & DF_SIG);
  int d = (DFL(y - 1, x - 1) & DF_SIG) + (DFL(y - 1, x + 1) & DF_SIG) + (DFL(y + 1, x - 1) & DF_SIG) + (DFL(y + 1, x + 1) & DF_SIG);
  if (orient == 1) { int t = h; h = v; v = t; }
  if (orient == 3) {
    int hv = h + v;
    if (d >= 3) return 8;
    if (d == 2) return hv >= 1 ? 7 : 6;
    if (d == 1) return hv >= 2 ? 5 : (hv == 1 ? 4 : 3);
    return hv >= 2 ? 2 : (hv == 1 ? 1 : 0);
  }
  if (h == 2) return 8;
  if (h == 1) return v >= 1 ? 7 : (d >= 1 ? 6 : 5);
  if (v == 2) return 4;
  if (v == 1) return 3;
  return d >= 2 ? 2 : (d == 1 ? 1 : 0);
}
__device__ __forceinline__ int d_sc(const uint8_t* fl, int y, int x, int* xorbit) {  // Tables D.2 / D.3
  int hc = 0, vc = 0, f;
  f = DFL(y, x - 1); if (f & DF_SIG) hc += (f & DF_NEG) ? -1 : 1;
  f = DFL(y, x + 1); if (f & DF_SIG) hc += (f & DF_NEG) ? -1 : 1;
  f = DFL(y - 1, x); if (f & DF_SIG) vc += (f & DF_NEG) ? -1 : 1;
  f = DFL(y + 1, x); if (f & DF_SIG) vc += (f & DF_NEG) ? -1 : 1;
  hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
  vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
  const int k = (hc + 1) * 3 + (vc + 1);  // ctx {13,12,11,10,9,10,11,12,13}, xor {1,1,1,1,0,0,0,0,0}
  *xorbit = k < 4;
  return k == 4 ? 9 : (k == 3 || k == 5 ? 10 : (k == 2 || k == 6 ? 11 : (k == 1 || k == 7 ? 12 : 13)));
}
__device__ __forceinline__ int d_anysig(const uint8_t* fl, int y, int x) {
  return (DFL(y, x - 1) | DFL(y, x + 1) | DFL(y - 1, x) | DFL(y + 1, x) | DFL(y - 1, x - 1) | DFL(y - 1, x + 1) | DFL(y + 1, x - 1) | DFL(y + 1, x + 1)) & DF_SIG;
}

// Round 3 form of the block decoder.  The decoding direction is serial per block and its control flow is data dependent at every
// sample (which pass a sample belongs to, run mode, renormalisation shifts), so lanes that decode different blocks in one wave
// each pay for every other lane's path: with 64 blocks per wave the wave executed the union of 64 control flows (measured:
// 159 ms for the 15 657 blocks of config 3, and no faster with the flags in LDS -- it was instruction stream, not memory).  A
// GPU has more wave slots than a sequence has code-blocks per round: so ONE CODE-BLOCK PER WAVE (a single active lane; the
// other 63 idle), a wave runs only its own block's path.  The per-sample flag plane lives in LDS (66 x 66 bytes per wave, up to
// 29 waves per CU), and the magnitude bits go to the coefficient plane as OR-atomics that return nothing (bit 31 = the sign):
// the kernel never waits for global memory except for its code bytes.  k_dec_dequant then turns the words into coefficients.
#define DEC_LANES 1
#define DEC_FL_BYTES (66 * 66)
__global__ void __launch_bounds__(64) k_t1_decode_lds(const DecCb* cbs, int ncb, const uint8_t* segs, uint8_t* lastp_all, void* coefv, int lane0, int nlanes) {
  __shared__ uint8_t s_ctx[19 * 64];
  __shared__ __attribute__((aligned(4))) uint8_t s_fl[DEC_LANES * DEC_FL_BYTES];
  // a whole wave is launched, lanes lane0 .. lane0 + nlanes - 1 decode (0 and DEC_LANES, passed as arguments: with the lane index a
  // compile-time constant the build of ROCm 7.2 produced a kernel that decoded garbage -- 16 VGPRs, everything scalarised)
  const int dl = (int)threadIdx.x - lane0;
  if (dl < 0 || dl >= nlanes) return;
  const int b = blockIdx.x * DEC_LANES + dl;
  if (b >= ncb) return;
  const DecCb cb = cbs[b];
  uint8_t* fl = s_fl + dl * DEC_FL_BYTES;   // this lane's flag plane (a word stride of 1089: lanes start one bank apart)
  for (int k = 0; k < DEC_FL_BYTES / 4; k++) ((uint32_t*)fl)[k] = 0;
  uint8_t* lastp = lastp_all + (size_t)b * 64 * 64;
  int32_t* mag = (int32_t*)coefv + cb.plane_off + (size_t)cb.y0 * cb.W + cb.x0;  // row stride cb.W
  const int w = cb.w, h = cb.h, orient = cb.orient, W = cb.W;
  MqDec m;
  m.init(segs + cb.seg, cb.len, s_ctx + dl);
  int np = 0;
#define MAG_OR(y, x, bits) atomicOr((unsigned*)&mag[(size_t)(y) * W + (x)], (unsigned)(bits))   /* no value comes back: never waited for */
#define DEC_SIGN(y, x) do { int xb_, cx_ = d_sc(fl, (y), (x), &xb_); int s_ = m.decode(cx_) ^ xb_; DFL((y), (x)) |= (uint8_t)(DF_SIG | (s_ ? DF_NEG : 0)); \
                            if (s_) MAG_OR((y), (x), 0x80000000u); } while (0)
  for (int p = cb.nbps - 1; p >= 0 && np < cb.npasses; p--) {
    if (p != cb.nbps - 1) {
      for (int y0 = 0; y0 < h; y0 += 4)
        for (int x = 0; x < w; x++)
          for (int y = y0; y < y0 + 4 && y < h; y++) {
            const uint8_t f = DFL(y, x);
            if ((f & DF_SIG) || !d_anysig(fl, y, x)) continue;
            if (m.decode(d_zc(fl, y, x, orient))) { MAG_OR(y, x, 1u << p); DEC_SIGN(y, x); }
            lastp[y * 64 + x] = (uint8_t)p;
            DFL(y, x) |= DF_VISIT;
          }
      if (++np >= cb.npasses) break;
      for (int y0 = 0; y0 < h; y0 += 4)
        for (int x = 0; x < w; x++)
          for (int y = y0; y < y0 + 4 && y < h; y++) {
            const uint8_t f = DFL(y, x);
            if (!(f & DF_SIG) || (f & DF_VISIT)) continue;
            const int cx = (f & DF_REFINED) ? 16 : (d_anysig(fl, y, x) ? 15 : 14);
            if (m.decode(cx)) MAG_OR(y, x, 1u << p);
            lastp[y * 64 + x] = (uint8_t)p;
            DFL(y, x) |= DF_REFINED;
          }
      if (++np >= cb.npasses) break;
    }
    for (int y0 = 0; y0 < h; y0 += 4)
      for (int x = 0; x < w; x++) {
        int y = y0;
        if (y0 + 3 < h) {
          bool run = true;
          for (int k = 0; k < 4; k++)
            if ((DFL(y0 + k, x) & (DF_SIG | DF_VISIT)) || d_anysig(fl, y0 + k, x)) { run = false; break; }
          if (run) {
            if (!m.decode(17)) { for (int k = 0; k < 4; k++) lastp[(y0 + k) * 64 + x] = (uint8_t)p; continue; }
            int first = m.decode(18) << 1;
            first |= m.decode(18);
            for (int k = 0; k < first; k++) lastp[(y0 + k) * 64 + x] = (uint8_t)p;
            y = y0 + first;
            MAG_OR(y, x, 1u << p); lastp[y * 64 + x] = (uint8_t)p;
            DEC_SIGN(y, x);
            y++;
          }
        }
        for (; y < y0 + 4 && y < h; y++) {
          const uint8_t f = DFL(y, x);
          if (f & (DF_SIG | DF_VISIT)) continue;
          if (m.decode(d_zc(fl, y, x, orient))) { MAG_OR(y, x, 1u << p); DEC_SIGN(y, x); }
          lastp[y * 64 + x] = (uint8_t)p;
        }
      }
    np++;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) DFL(y, x) &= (uint8_t)~DF_VISIT;
  }
#undef DEC_SIGN
}
#undef MAG_OR

// magnitude word (bit 31 sign, bits 30..0 magnitude bits decoded) -> coefficient, in place (E.1.1.2: r = 1/2 for the bit-planes
// that were not decoded; exact for a fully decoded reversible sample).  One workgroup per code-block.
__global__ void __launch_bounds__(256) k_dec_dequant(const DecCb* cbs, int ncb, const uint8_t* lastp_all, void* coefv) {
  const int b = blockIdx.x;
  if (b >= ncb) return;
  const DecCb cb = cbs[b];
  const uint8_t* lastp = lastp_all + (size_t)b * 64 * 64;
  int32_t* mag = (int32_t*)coefv + cb.plane_off + (size_t)cb.y0 * cb.W + cb.x0;
  for (int k = threadIdx.x; k < cb.w * cb.h; k += 256) {
    const int y = k / cb.w, x = k % cb.w;
    const uint32_t wd = (uint32_t)mag[(size_t)y * cb.W + x];
    const int mg = (int)(wd & 0x7FFFFFFFu), neg = (int)(wd >> 31);
    if (!mg) { mag[(size_t)y * cb.W + x] = 0; continue; }
    const int pl = lastp[y * 64 + x];
    if (cb.rev) {
      const int v = mg + (pl > 0 ? (1 << (pl - 1)) : 0);
      mag[(size_t)y * cb.W + x] = neg ? -v : v;
    } else {
      const float v = __fmul_rn(__fadd_rn((float)mg, pl > 0 ? (float)(1 << (pl - 1)) : 0.5f), cb.delta);
      ((float*)mag)[(size_t)y * cb.W + x] = neg ? -v : v;
    }
  }
}

// ---- device: inverse DWT, one 1-D pass per launch, thread per output sample --------------------------------------------------
template <class Ld> __device__ __forceinline__ int idwt53_sample(const Ld& X /* Mallat: low half first */, int k, int n) {
  const int nl = (n + 1) >> 1, nh = n >> 1;
  auto Hc = [&](int i) { i = i < 0 ? 0 : (i >= nh ? nh - 1 : i); return X(nl + i); };
  auto E = [&](int i) { return X(i) - ((Hc(i - 1) + Hc(i) + 2) >> 2); };
  if ((k & 1) == 0) return E(k >> 1);
  const int i = k >> 1, a = E(i), b = (2 * i + 2 < n) ? E(i + 1) : a;
  return X(nl + i) + ((a + b) >> 1);
}
// Idwt97<STEP>::val: value at interleaved position j after STEP inverse lifting steps (0: scaled input; 1: delta on even; 2: gamma on odd; 3: beta on
// even; 4: alpha on odd); a neighbour outside the signal is replaced by the one on the other side
template <int STEP, class Ld> struct Idwt97 {
  static __device__ __forceinline__ float val(const Ld& X, int j, int n) {
    const bool mine = (STEP & 1) ? !(j & 1) : (j & 1);  // steps 1, 3 update even positions; 2, 4 odd ones
    const float own = Idwt97<STEP - 1, Ld>::val(X, j, n);
    if (!mine) return own;
    const float c = STEP == 1 ? F97_DELTA : (STEP == 2 ? F97_GAMMA : (STEP == 3 ? F97_BETA : F97_ALPHA));
    const float lo = j - 1 >= 0 ? Idwt97<STEP - 1, Ld>::val(X, j - 1, n) : Idwt97<STEP - 1, Ld>::val(X, j + 1, n);
    const float hi = j + 1 < n ? Idwt97<STEP - 1, Ld>::val(X, j + 1, n) : Idwt97<STEP - 1, Ld>::val(X, j - 1, n);
    return __fsub_rn(own, __fmul_rn(c, __fadd_rn(lo, hi)));
  }
};
template <class Ld> struct Idwt97<0, Ld> {
  static __device__ __forceinline__ float val(const Ld& X, int j, int n) {
    const int nl = (n + 1) >> 1;
    return (j & 1) ? __fmul_rn(X(nl + (j >> 1)), 1.0f / F97_K) : __fmul_rn(X(j >> 1), F97_K);
  }
};
template <bool REV, bool VERT>
__global__ void __launch_bounds__(256) k_idwt(const DecPlane* planes, const void* srcv, void* dstv, int lev /* region = level lev-1 size */) {
  const DecPlane pd = planes[blockIdx.z];
  if (lev > pd.L) return;
  int wr = pd.W, hr = pd.H;
  for (int l = 0; l < lev - 1; l++) { wr = (wr + 1) >> 1; hr = (hr + 1) >> 1; }
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= wr || y >= hr) return;
  const size_t o = pd.off + (size_t)y * pd.W + x;
  const int n = VERT ? hr : wr, k = VERT ? y : x;
  if (REV) {
    const int32_t* s = (const int32_t*)srcv + pd.off;
    auto X = [&](int i) { return VERT ? s[(size_t)i * pd.W + x] : s[(size_t)y * pd.W + i]; };
    ((int32_t*)dstv)[o] = n >= 2 ? idwt53_sample(X, k, n) : X(k);
  } else {
    const float* s = (const float*)srcv + pd.off;
    auto X = [&](int i) { return VERT ? s[(size_t)i * pd.W + x] : s[(size_t)y * pd.W + i]; };
    ((float*)dstv)[o] = n >= 2 ? Idwt97<4, decltype(X)>::val(X, k, n) : X(k);
  }
}
__global__ void __launch_bounds__(256) k_dec_finish(const DecPlane* planes, const void* srcv, uint8_t* out) {
  const DecPlane pd = planes[blockIdx.z];
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= pd.W || y >= pd.H) return;
  const size_t i = pd.off + (size_t)y * pd.W + x;
  int v = pd.rev ? ((const int32_t*)srcv)[i] : __float2int_rn(((const float*)srcv)[i]);  // lrintf: round half to even
  if (pd.sgnd) {
    v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
    ((int16_t*)(out + pd.out_off))[(size_t)y * pd.W + x] = (int16_t)v;
  } else {
    v += 128;
    out[pd.out_off + (size_t)y * pd.W + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace

struct J2kDecState {
  DevBuf cbs_d, segs_d, flags_d, lastp_d, coefA, coefB, planes_d, out_d;
};

extern "C" int qsvc_j2k_decode_planes(qsvc_ctx* c, int nstreams, const uint8_t* const* streams, const size_t* lens, int max_layers,
                                      void* out, size_t cap, size_t* offs, int* widths, int* heights) {
  if (!c || nstreams < 1 || !streams || !lens || !out || !offs) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  if (!c->j2kdec) c->j2kdec = new J2kDecState();
  J2kDecState& S = *c->j2kdec;
  std::vector<DecPlane> planes(nstreams);
  std::vector<DecCb> cbs;
  std::vector<uint8_t> segs;
  size_t total = 0, out_total = 0;
  int maxW = 0, maxH = 0, maxL = 0;
  for (int i = 0; i < nstreams; i++) {
    planes[i].off = total; planes[i].out_off = out_total;
    if (!streams[i]) return c->fail(QSVC_ERR_INVALID, "code-stream %d is null", i);
    const char* e = nullptr;
    try {  // the parser sizes containers from header fields: an allocation failure is an error code, never an exception across the C ABI
      e = streams[i] ? parse_stream(streams[i], lens[i], max_layers, planes[i], cbs, segs) : "null stream";
    } catch (const std::exception&) {
      return c->fail(QSVC_ERR_NOMEM, "code-stream %d: out of host memory while parsing", i);
    }
    if (e) return c->fail(QSVC_ERR_UNSUPPORTED, "code-stream %d: %s", i, e);
    if (planes[i].rev != planes[0].rev) return c->fail(QSVC_ERR_UNSUPPORTED, "the streams of one call must share the transform");
    total += (size_t)planes[i].W * planes[i].H;
    offs[i] = out_total;
    out_total += (size_t)planes[i].W * planes[i].H * (planes[i].prec > 8 ? 2 : 1);
    out_total = (out_total + 1) & ~(size_t)1;
    if (widths) widths[i] = planes[i].W;
    if (heights) heights[i] = planes[i].H;
    maxW = std::max(maxW, planes[i].W); maxH = std::max(maxH, planes[i].H); maxL = std::max(maxL, planes[i].L);
  }
  offs[nstreams] = out_total;
  if (out_total > cap) return c->fail(QSVC_ERR_NOMEM, "output buffer too small (%zu bytes needed, %zu given)", out_total, cap);
  const int ncb = (int)cbs.size();
  const bool rev = planes[0].rev != 0;
  int r;
  if ((r = S.coefA.ensure(c, total * 4))) return r;
  if ((r = S.coefB.ensure(c, total * 4))) return r;
  if ((r = S.planes_d.ensure(c, planes.size() * sizeof(DecPlane)))) return r;
  if ((r = S.out_d.ensure(c, out_total + 16))) return r;
  HIPCHK(c, hipMemsetAsync(S.coefA.p, 0, total * 4, c->st));
  HIPCHK(c, hipMemcpyAsync(S.planes_d.p, planes.data(), planes.size() * sizeof(DecPlane), hipMemcpyHostToDevice, c->st));
  Prof pf;
  if (ncb) {
    if ((r = S.cbs_d.ensure(c, (size_t)ncb * sizeof(DecCb)))) return r;
    if ((r = S.segs_d.ensure(c, segs.size() + 16))) return r;
    if ((r = S.lastp_d.ensure(c, (size_t)ncb * 64 * 64))) return r;
    HIPCHK(c, hipMemcpyAsync(S.cbs_d.p, cbs.data(), (size_t)ncb * sizeof(DecCb), hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(S.segs_d.p, segs.data(), segs.size(), hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemsetAsync(S.lastp_d.p, 0, (size_t)ncb * 64 * 64, c->st));
    pf.begin(c, "j2k_t1_decode");
    hipLaunchKernelGGL(k_t1_decode_lds, dim3((ncb + DEC_LANES - 1) / DEC_LANES), dim3(64), 0, c->st, (const DecCb*)S.cbs_d.p, ncb,
                       (const uint8_t*)S.segs_d.p, (uint8_t*)S.lastp_d.p, S.coefA.p, 0, DEC_LANES);
    hipLaunchKernelGGL(k_dec_dequant, dim3(ncb), dim3(256), 0, c->st, (const DecCb*)S.cbs_d.p, ncb, (const uint8_t*)S.lastp_d.p, S.coefA.p);
    pf.end(c);
  }
  // inverse DWT, levels L .. 1: horizontal (A -> B) then vertical (B -> A); samples outside the level's region are not touched,
  // so B must hold the rest of the plane too: it starts as a copy of A
  pf.begin(c, rev ? "j2k_idwt53" : "j2k_idwt97");
  if (maxL > 0) HIPCHK(c, hipMemcpyAsync(S.coefB.p, S.coefA.p, total * 4, hipMemcpyDeviceToDevice, c->st));
  for (int lev = maxL; lev >= 1; lev--) {
    int w = maxW, h = maxH;
    for (int l = 0; l < lev - 1; l++) { w = (w + 1) >> 1; h = (h + 1) >> 1; }
    dim3 grid((w + 63) / 64, (h + 3) / 4, nstreams), block(64, 4);
    if (rev) {
      hipLaunchKernelGGL((k_idwt<true, false>), grid, block, 0, c->st, (const DecPlane*)S.planes_d.p, S.coefA.p, S.coefB.p, lev);
      hipLaunchKernelGGL((k_idwt<true, true>), grid, block, 0, c->st, (const DecPlane*)S.planes_d.p, S.coefB.p, S.coefA.p, lev);
    } else {
      hipLaunchKernelGGL((k_idwt<false, false>), grid, block, 0, c->st, (const DecPlane*)S.planes_d.p, S.coefA.p, S.coefB.p, lev);
      hipLaunchKernelGGL((k_idwt<false, true>), grid, block, 0, c->st, (const DecPlane*)S.planes_d.p, S.coefB.p, S.coefA.p, lev);
    }
  }
  hipLaunchKernelGGL(k_dec_finish, dim3((maxW + 63) / 64, (maxH + 3) / 4, nstreams), dim3(64, 4), 0, c->st, (const DecPlane*)S.planes_d.p,
                     S.coefA.p, (uint8_t*)S.out_d.p);
  pf.end(c);
  HIPCHK(c, hipMemcpyAsync(out, S.out_d.p, out_total, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, hipStreamSynchronize(c->st));
  HIPCHK(c, hipGetLastError());
  return QSVC_OK;
}

void j2k_decode_release(qsvc_ctx* c) {
  J2kDecState* s = c->j2kdec;
  if (!s) return;
  DevBuf* all[] = {&s->cbs_d, &s->segs_d, &s->flags_d, &s->lastp_d, &s->coefA, &s->coefB, &s->planes_d, &s->out_d};
  for (auto* b : all) b->release();
  delete s;
  c->j2kdec = nullptr;
}
