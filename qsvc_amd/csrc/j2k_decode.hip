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
#include <chrono>
#include <atomic>
#include <thread>
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

// Round 4 form of the block decoder.  One code-block per wave with ONE decoding lane, as in round 3 (the decoding direction is
// serial per block and its control flow is data dependent at every sample: lanes that decode different blocks in one wave each
// pay for every other lane's path), but the per-decision cost is cut from about 600 instructions-and-waits to about a hundred:
//   * state = one 32-bit FLAG WORD PER STRIPE COLUMN in LDS (significance of the column's 6 x 3 neighbourhood, signs of its six
//     rows, "visited" and "refined" of its four samples) instead of a byte per sample: a column nobody can code in this pass is
//     skipped with one test (no significant neighbour for the significance pass, no significant sample for the refinement, an
//     all-zero word = run mode for the clean-up), and a sample's context is a table look-up on 9 bits of the word;
//   * the MQ decoder's probability table lives in LDS and every context slot holds its table ENTRY (Qe and both successors), so a
//     decision costs one LDS read, and one more only when the state changes (round 3: a byte from LDS, then the entry through the
//     vector memory path of a __constant__ array, 300+ cycles, per decision);
//   * renormalisation shifts by the whole count at once; the code bytes come through a 16-byte register window refilled 8 bytes
//     ahead, not one global byte load per byte.
// The magnitude bits still go to the coefficient plane as OR-atomics that return nothing (bit 31 = the sign); k_dec_dequant turns
// the words into coefficients.  The per-sample "last coded plane" bytes are only written for blocks whose passes were truncated
// (quality layers left out): a fully decoded block has none.  The wave's other 63 lanes build the tables and clear the flag words.
#define DEC_FW_WORDS (18 * 66)     // flag words: stripes -1 .. 16, columns -1 .. 64
#define DFW(s, x) fw[((s) + 1) * 66 + (x) + 1]
#define DF_SIGALL 0x3FFFFu
#define DF_CHI 18
#define DF_PI 24
#define DF_MU 28
struct MqDec2 {
  uint32_t A, C; int CT;
  const uint8_t* base; int len, pos;   // the segment; bytes consumed
  uint64_t w0, w1; int wpos;            // bytes [wpos, wpos + 8) and [wpos + 8, wpos + 16) of the segment (0xFF past its end); pos >= wpos
  uint32_t* cw;                         // LDS: context words = table entry of the context's state | mps << 31
  const uint32_t* tbl;                  // LDS: the 47 entries
  __device__ __forceinline__ uint64_t fetch8(int at) const {
    uint64_t v = ~0ull;
    if (at + 8 <= len) __builtin_memcpy(&v, base + at, 8);
    else for (int k = 0; k < 8; k++) if (at + k < len) v = (v & ~(0xFFull << (8 * k))) | ((uint64_t)base[at + k] << (8 * k));
    return v;
  }
  __device__ __forceinline__ int byte_at(int i) {   // byte i of the segment, pos <= i <= pos + 1; 0xFF past the end
    if (i >= wpos + 16) { w0 = w1; wpos += 8; w1 = fetch8(wpos + 8); }   // (the next 8 bytes are requested 8 bytes before they are needed)
    const int o = i - wpos;
    return (int)(((o < 8 ? w0 : w1) >> (8 * (o & 7))) & 0xFF);
  }
  __device__ __forceinline__ void bytein() {   // T.800 C.3.4
    if (pos < len) {
      const int b = byte_at(pos);
      if (b == 0xFF) {
        const int nx = pos + 1 < len ? byte_at(pos + 1) : 0xFF;
        if (nx > 0x8F) { C += 0xFF00; CT = 8; } else { pos++; C += (uint32_t)nx << 9; CT = 7; }
      } else { pos++; C += (uint32_t)(pos < len ? byte_at(pos) : 0xFF) << 8; CT = 8; }
    } else { C += 0xFF00; CT = 8; }
  }
  __device__ void init(const uint8_t* data, int n, uint32_t* lds_cw, const uint32_t* lds_tbl) {
    cw = lds_cw; tbl = lds_tbl; base = data; len = n; pos = 0; wpos = 0;
    w0 = fetch8(0); w1 = fetch8(8);
    for (int i = 0; i < 19; i++) cw[i] = tbl[0];
    cw[0] = tbl[4]; cw[17] = tbl[3]; cw[18] = tbl[46];
    C = (uint32_t)(len > 0 ? byte_at(0) : 0xFF) << 16;
    bytein();
    C <<= 7; CT -= 7; A = 0x8000;
  }
  __device__ __forceinline__ int decode(int cx) {   // T.800 C.3.2
    const uint32_t e = cw[cx], qe = e & 0xFFFF;
    const int mps = (int)(e >> 31);
    int d, lps;   // lps: the state moves along the LPS branch
    A -= qe;
    if ((C >> 16) < qe) {
      lps = !(A < qe);
      A = qe;
    } else {
      C -= qe << 16;
      if (A & 0x8000) return mps;
      lps = A < qe;
    }
    d = lps ? 1 - mps : mps;
    const uint32_t ns = lps ? (e >> 22) & 63 : (e >> 16) & 63;
    const uint32_t nm = lps ? (uint32_t)mps ^ ((e >> 28) & 1) : (uint32_t)mps;
    cw[cx] = tbl[ns] | nm << 31;
    int sh = __builtin_clz(A) - 16;   // A < 0x8000 here: 1 .. 15 shifts
    while (sh > 0) {
      if (CT == 0) bytein();
      const int n = sh < CT ? sh : CT;
      A <<= n; C <<= n; CT -= n; sh -= n;
    }
    return d;
  }
};
// zero-coding context (Table D.1) of the 8 neighbours in a 9-bit window (bit 3 r + c, r = row above / own / below, c = left / own / right)
__device__ __forceinline__ int d_zc_of(int n9, int orient) {
  int h = ((n9 >> 3) & 1) + ((n9 >> 5) & 1), v = ((n9 >> 1) & 1) + ((n9 >> 7) & 1);
  const int d = (n9 & 1) + ((n9 >> 2) & 1) + ((n9 >> 6) & 1) + ((n9 >> 8) & 1);
  if (orient == 1) { const int t = h; h = v; v = t; }
  if (orient == 3) {
    const int hv = h + v;
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
// NB code-blocks per wave, one decoding lane each (lanes lane0 .. lane0 + NB - 1): the lanes run the same loops -- bit-plane, pass,
// stripe, column, row -- in step, every lane on its own block's flag words and MQ state, so a decision of one lane costs the wave
// the same instructions as a decision of all NB.  With one block per wave the kernel was bound by instruction issue (32 waves per
// CU, each issuing for one lane of 64).
template <int NB>
__global__ void __launch_bounds__(64) k_t1_decode_lds(const DecCb* cbs, int ncb, const uint8_t* segs, uint8_t* lastp_all, void* coefv, int lane0, int nlanes) {
  __shared__ uint32_t s_fw[NB * DEC_FW_WORDS];
  __shared__ uint32_t s_tbl[48];
  __shared__ uint32_t s_cw[NB * 20];
  // zero coding by neighbourhood: NB > 1: the three orientation classes LL / LH, HL, HH; NB == 1: the block's own class only (LDS sets
  // the number of one-lane waves a CU holds: 5.5 KB per wave = 29 of them, with three tables 24)
  constexpr int NZC = NB == 1 ? 1 : 3;
  __shared__ uint8_t s_zc3[NZC * 512];
  {  // the whole wave: tables and clean flag planes
    const int t = threadIdx.x;
    for (int k = t; k < NB * DEC_FW_WORDS; k += 64) s_fw[k] = 0;
    if (t < 47) s_tbl[t] = c_dmq[t];
    if (NB == 1) {
      const int o1 = blockIdx.x < (unsigned)ncb ? cbs[blockIdx.x].orient : 0;
      for (int k = t; k < 512; k += 64) s_zc3[k] = (uint8_t)d_zc_of(k, o1);
    } else {
      for (int k = t; k < 3 * 512; k += 64) s_zc3[k] = (uint8_t)d_zc_of(k & 511, k < 512 ? 0 : (k < 1024 ? 1 : 3));
    }
  }
  __syncthreads();
  // lanes lane0 .. lane0 + nlanes - 1 decode.  (History: round 3's kernel kept a flag BYTE per sample and cleared the plane with 32-bit
  // stores through the byte array; with the lane index a compile-time constant that build decoded garbage, and the index became an
  // argument.  The flag-word kernel has no such aliasing: a build with the constant index passes every decoder test (round 4); the
  // argument stays as the first lane of the NB > 1 forms.)
  const int li = (int)threadIdx.x - lane0;
  const int b = NB == 1 ? (int)blockIdx.x : (int)blockIdx.x * NB + li;   // (one block per wave: its record and pointers are wave-uniform -- scalar loads, SGPRs)
  if (li < 0 || li >= nlanes || b >= ncb) return;
  const DecCb cb = cbs[b];
  uint32_t* fw = s_fw + li * DEC_FW_WORDS;
  const uint8_t* s_zc = s_zc3 + (NB == 1 ? 0 : 512 * (cb.orient == 1 ? 1 : (cb.orient == 3 ? 2 : 0)));
  uint8_t* lastp = lastp_all + (size_t)b * 64 * 64;
  int32_t* mag = (int32_t*)coefv + cb.plane_off + (size_t)cb.y0 * cb.W + cb.x0;  // row stride cb.W
  const int w = cb.w, h = cb.h, W = cb.W, nstr = (h + 3) >> 2;
  const bool full = cb.npasses >= 3 * cb.nbps - 2;   // every pass present: every sample's last coded plane is 0
  MqDec2 m;
  m.init(segs + cb.seg, cb.len, s_cw + li * 20, s_tbl);
  int np = 0;
#define MAG_OR(y, x, bits) atomicOr((unsigned*)&mag[(size_t)(y) * W + (x)], (unsigned)(bits))   /* no value comes back: never waited for */
  // sample (stripe s, column x, row k) becomes significant with sign neg: every word whose neighbourhood holds it (t1_set_sig of
  // tests/emul/t1_legacy.h in the decoding direction; `f` is the column's own word, kept in a register by the caller)
  auto set_sig = [&](int s, int x, int k, int neg, uint32_t& f) {
    f |= (1u << (3 * (k + 1) + 1)) | ((uint32_t)neg << (DF_CHI + k + 1));
    atomicOr(&DFW(s, x - 1), 1u << (3 * (k + 1) + 2));
    atomicOr(&DFW(s, x + 1), 1u << (3 * (k + 1) + 0));
    if (k == 0) {
      atomicOr(&DFW(s - 1, x - 1), 1u << 17);
      atomicOr(&DFW(s - 1, x), (1u << 16) | ((uint32_t)neg << (DF_CHI + 5)));
      atomicOr(&DFW(s - 1, x + 1), 1u << 15);
    } else if (k == 3) {
      atomicOr(&DFW(s + 1, x - 1), 1u << 2);
      atomicOr(&DFW(s + 1, x), (1u << 1) | ((uint32_t)neg << DF_CHI));
      atomicOr(&DFW(s + 1, x + 1), 1u << 0);
    }
  };
  // sign decoding (Tables D.2 / D.3): horizontal neighbours from the words left and right, vertical ones from the own word
  auto dec_sign = [&](int s, int x, int k, uint32_t f) {
    const uint32_t l = DFW(s, x - 1), r = DFW(s, x + 1);
    const int rr = k + 1;
    int hc = 0, vc = 0;
    if (f & (1u << (3 * rr + 0))) hc += (l >> (DF_CHI + rr)) & 1 ? -1 : 1;
    if (f & (1u << (3 * rr + 2))) hc += (r >> (DF_CHI + rr)) & 1 ? -1 : 1;
    if (f & (1u << (3 * (rr - 1) + 1))) vc += (f >> (DF_CHI + rr - 1)) & 1 ? -1 : 1;
    if (f & (1u << (3 * (rr + 1) + 1))) vc += (f >> (DF_CHI + rr + 1)) & 1 ? -1 : 1;
    hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
    vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
    const int q = (hc + 1) * 3 + (vc + 1);  // ctx {13,12,11,10,9,10,11,12,13}, xor {1,1,1,1,0,0,0,0,0}
    const int cx = q == 4 ? 9 : (q == 3 || q == 5 ? 10 : (q == 2 || q == 6 ? 11 : (q == 1 || q == 7 ? 12 : 13)));
    return m.decode(cx) ^ (q < 4);
  };
  for (int p = cb.nbps - 1; p >= 0 && np < cb.npasses; p--) {
    if (p != cb.nbps - 1) {
      // significance propagation
      for (int s = 0; s < nstr; s++) {
        const int hs = h - 4 * s < 4 ? h - 4 * s : 4;
        for (int x = 0; x < w; x++) {
          uint32_t f = DFW(s, x);
          if ((f & DF_SIGALL) == 0) continue;   // nothing significant anywhere around this column
          const uint32_t f0 = f;
          for (int k = 0; k < hs; k++) {
            if (f & (1u << (3 * k + 4))) continue;                 // already significant
            const uint32_t n9 = (f >> (3 * k)) & 0x1EF;
            if (!n9) continue;
            if (m.decode(s_zc[n9])) {
              const int neg = dec_sign(s, x, k, f);
              set_sig(s, x, k, neg, f);
              MAG_OR(4 * s + k, x, (1u << p) | (neg ? 0x80000000u : 0u));
            }
            f |= 1u << (DF_PI + k);
            if (!full) lastp[(4 * s + k) * 64 + x] = (uint8_t)p;
          }
          if (f != f0) DFW(s, x) = f;
        }
      }
      if (++np >= cb.npasses) break;
      // magnitude refinement
      for (int s = 0; s < nstr; s++) {
        const int hs = h - 4 * s < 4 ? h - 4 * s : 4;
        for (int x = 0; x < w; x++) {
          uint32_t f = DFW(s, x);
          if ((f & 0x2490u) == 0) continue;     // no significant sample in this column (bits 4, 7, 10, 13)
          const uint32_t f0 = f;
          for (int k = 0; k < hs; k++) {
            if (!(f & (1u << (3 * k + 4))) || (f & (1u << (DF_PI + k)))) continue;
            const int cx = (f & (1u << (DF_MU + k))) ? 16 : (((f >> (3 * k)) & 0x1EF) ? 15 : 14);
            if (m.decode(cx)) MAG_OR(4 * s + k, x, 1u << p);
            f |= 1u << (DF_MU + k);
            if (!full) lastp[(4 * s + k) * 64 + x] = (uint8_t)p;
          }
          if (f != f0) DFW(s, x) = f;
        }
      }
      if (++np >= cb.npasses) break;
    }
    // clean-up
    for (int s = 0; s < nstr; s++) {
      const int hs = h - 4 * s < 4 ? h - 4 * s : 4;
      for (int x = 0; x < w; x++) {
        uint32_t f = DFW(s, x);
        const uint32_t f0 = f;
        int k = 0;
        if (hs == 4 && (f & (DF_SIGALL | (0xFu << DF_PI))) == 0) {   // run mode
          if (!m.decode(17)) {
            if (!full) for (int q = 0; q < 4; q++) lastp[(4 * s + q) * 64 + x] = (uint8_t)p;
            continue;
          }
          k = m.decode(18) << 1;
          k |= m.decode(18);
          if (!full) for (int q = 0; q <= k; q++) lastp[(4 * s + q) * 64 + x] = (uint8_t)p;
          const int neg = dec_sign(s, x, k, f);
          set_sig(s, x, k, neg, f);
          MAG_OR(4 * s + k, x, (1u << p) | (neg ? 0x80000000u : 0u));
          k++;
        }
        for (; k < hs; k++) {
          if (f & ((1u << (3 * k + 4)) | (1u << (DF_PI + k)))) continue;
          if (m.decode(s_zc[(f >> (3 * k)) & 0x1EF])) {
            const int neg = dec_sign(s, x, k, f);
            set_sig(s, x, k, neg, f);
            MAG_OR(4 * s + k, x, (1u << p) | (neg ? 0x80000000u : 0u));
          }
          if (!full) lastp[(4 * s + k) * 64 + x] = (uint8_t)p;
        }
        f &= ~(0xFu << DF_PI);
        if (f != f0) DFW(s, x) = f;
      }
    }
    np++;
  }
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
  // page-locked staging of the merged tables (grow-only): the streams' segments and block records are put together by the parser's
  // threads straight into these and go up by DMA (round 4: a serial merge into pageable vectors was 10 ms of a 55 ms decode)
  uint8_t* segs_h = nullptr; size_t segs_h_cap = 0;
  DecCb* cbs_h = nullptr; size_t cbs_h_cap = 0;
  bool pin(size_t seg_bytes, size_t nblocks) {
    if (seg_bytes > segs_h_cap) {
      if (segs_h) (void)hipHostFree(segs_h);
      segs_h = nullptr; segs_h_cap = 0;
      const size_t want = seg_bytes + seg_bytes / 4 + 4096;
      if (hipHostMalloc((void**)&segs_h, want, hipHostMallocDefault) != hipSuccess) return false;
      segs_h_cap = want;
    }
    if (nblocks > cbs_h_cap) {
      if (cbs_h) (void)hipHostFree(cbs_h);
      cbs_h = nullptr; cbs_h_cap = 0;
      const size_t want = nblocks + nblocks / 4 + 64;
      if (hipHostMalloc((void**)&cbs_h, want * sizeof(DecCb), hipHostMallocDefault) != hipSuccess) return false;
      cbs_h_cap = want;
    }
    return true;
  }
};

// dev_dst != nullptr: the planes go straight to device memory at dev_dst + dev_offs[i] (qsvc_j2k_decode_frames), geometry checked
// against (fw, fh); else to the host buffer `out`, back to back.
static int decode_impl(qsvc_ctx* c, int nstreams, const uint8_t* const* streams, const size_t* lens, int max_layers, void* out, size_t cap,
                       size_t* offs, int* widths, int* heights, uint8_t* dev_dst, const size_t* dev_offs, int fw, int fh) {
  (void)hipSetDevice(c->dev);
  TuningScope tscope(&c->tune);
  if (!c->j2kdec) c->j2kdec = new J2kDecState();
  J2kDecState& S = *c->j2kdec;
  std::vector<DecPlane> planes(nstreams);
  size_t total = 0, out_total = 0;
  int maxW = 0, maxH = 0, maxL = 0;
  // Tier-2 parsing: the streams are independent -- each into its own tables on a few host threads, then merged in stream order
  struct Parsed { std::vector<DecCb> cbs; std::vector<uint8_t> segs; const char* err = nullptr; bool nomem = false; };
  std::vector<Parsed> parsed(nstreams);
  for (int i = 0; i < nstreams; i++) if (!streams[i]) return c->fail(QSVC_ERR_INVALID, "code-stream %d is null", i);
  const auto tp0 = std::chrono::steady_clock::now();
  {
    const int nthr = std::max(1, std::min({nstreams, 16, c->t2_threads > 0 ? c->t2_threads : 16}));
    std::atomic<int> next{0};
    auto work = [&]() {
      for (int i; (i = next.fetch_add(1)) < nstreams;) {
        planes[i].off = 0;
        try {  // the parser sizes containers from header fields: an allocation failure is an error code, never an exception across the C ABI
          parsed[i].err = parse_stream(streams[i], lens[i], max_layers, planes[i], parsed[i].cbs, parsed[i].segs);
        } catch (const std::exception&) { parsed[i].nomem = true; }
      }
    };
    std::vector<std::thread> th;
    try {
      for (int k = 1; k < nthr; k++) th.emplace_back(work);
    } catch (const std::exception&) {}   // (fewer helpers than asked for: the caller's thread does the rest)
    work();
    for (auto& t : th) t.join();
  }
  const auto tp1 = std::chrono::steady_clock::now();
  c->host_time("host_dec_parse", std::chrono::duration<double, std::milli>(tp1 - tp0).count());
  std::vector<size_t> seg_base(nstreams), cb_base(nstreams);
  size_t seg_total = 0, cb_total = 0;
  for (int i = 0; i < nstreams; i++) {
    planes[i].off = total; planes[i].out_off = out_total;
    if (parsed[i].nomem) return c->fail(QSVC_ERR_NOMEM, "code-stream %d: out of host memory while parsing", i);
    if (parsed[i].err) return c->fail(QSVC_ERR_UNSUPPORTED, "code-stream %d: %s", i, parsed[i].err);
    seg_base[i] = seg_total; cb_base[i] = cb_total;
    seg_total += parsed[i].segs.size(); cb_total += parsed[i].cbs.size();
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
  if (dev_dst) {
    for (int i = 0; i < nstreams; i++) {
      const int ew = (i % 3) ? fw / 2 : fw, eh = (i % 3) ? fh / 2 : fh;
      if (planes[i].W != ew || planes[i].H != eh || planes[i].prec != 8)
        return c->fail(QSVC_ERR_INVALID, "code-stream %d holds a %dx%d plane of %d bits, the sequence expects %dx%d of 8", i, planes[i].W, planes[i].H, planes[i].prec, ew, eh);
      planes[i].out_off = dev_offs[i];
    }
  } else if (out_total > cap) return c->fail(QSVC_ERR_NOMEM, "output buffer too small (%zu bytes needed, %zu given)", out_total, cap);
  if (cb_total > 0x7fffffff) return c->fail(QSVC_ERR_UNSUPPORTED, "too many code-blocks in one call");
  const int ncb = (int)cb_total;
  if (!S.pin(seg_total + 16, cb_total + 1)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc (page-locked tables of the decoder)");
  {  // the streams' tables into the merged ones, by the same few threads; then the blocks in the order the kernel takes them
    std::vector<DecCb> tmp;
    try { tmp.resize(cb_total); } catch (const std::exception&) { return c->fail(QSVC_ERR_NOMEM, "out of host memory while parsing"); }
    const int nthr = std::max(1, std::min({nstreams, 16, c->t2_threads > 0 ? c->t2_threads : 16}));
    std::atomic<int> next{0};
    auto work = [&]() {
      for (int i; (i = next.fetch_add(1)) < nstreams;) {
        if (!parsed[i].segs.empty()) memcpy(S.segs_h + seg_base[i], parsed[i].segs.data(), parsed[i].segs.size());
        DecCb* o = tmp.data() + cb_base[i];
        for (const DecCb& d0 : parsed[i].cbs) { DecCb d = d0; d.seg += seg_base[i]; d.plane_off = planes[i].off; *o++ = d; }
      }
    };
    std::vector<std::thread> th;
    try {
      for (int k = 1; k < nthr; k++) th.emplace_back(work);
    } catch (const std::exception&) {}
    work();
    for (auto& t : th) t.join();
    // (the lanes of a wave decode neighbouring entries and the grid is taken in order: the longest blocks first)
    std::vector<uint32_t> ord(cb_total);
    for (size_t k = 0; k < cb_total; k++) ord[k] = (uint32_t)k;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return tmp[a].len > tmp[b].len; });
    for (size_t k = 0; k < cb_total; k++) S.cbs_h[k] = tmp[ord[k]];
  }
  const bool rev = planes[0].rev != 0;
  c->host_time("host_dec_merge", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp1).count());
  const auto tp2 = std::chrono::steady_clock::now();
  int r;
  if ((r = S.coefA.ensure(c, total * 4))) return r;
  if ((r = S.coefB.ensure(c, total * 4))) return r;
  if ((r = S.planes_d.ensure(c, planes.size() * sizeof(DecPlane)))) return r;
  if (!dev_dst && (r = S.out_d.ensure(c, out_total + 16))) return r;
  HIPCHK(c, hipMemsetAsync(S.coefA.p, 0, total * 4, c->st));
  HIPCHK(c, hipMemcpyAsync(S.planes_d.p, planes.data(), planes.size() * sizeof(DecPlane), hipMemcpyHostToDevice, c->st));
  Prof pf;
  if (ncb) {
    if ((r = S.cbs_d.ensure(c, (size_t)ncb * sizeof(DecCb)))) return r;
    if ((r = S.segs_d.ensure(c, seg_total + 16))) return r;
    if ((r = S.lastp_d.ensure(c, (size_t)ncb * 64 * 64))) return r;
    HIPCHK(c, hipMemcpyAsync(S.cbs_d.p, S.cbs_h, (size_t)ncb * sizeof(DecCb), hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(S.segs_d.p, S.segs_h, seg_total, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemsetAsync(S.lastp_d.p, 0, (size_t)ncb * 64 * 64, c->st));
    pf.begin(c, "j2k_t1_decode");
    {
      const int nb = tuning().dec_lanes;   // code-blocks per wave
      const dim3 grid((ncb + nb - 1) / nb), block(64);
#define DEC_LAUNCH(NB) hipLaunchKernelGGL(k_t1_decode_lds<NB>, grid, block, 0, c->st, (const DecCb*)S.cbs_d.p, ncb, (const uint8_t*)S.segs_d.p, (uint8_t*)S.lastp_d.p, S.coefA.p, 0, NB)
      if (nb == 1) DEC_LAUNCH(1); else if (nb == 2) DEC_LAUNCH(2); else if (nb == 4) DEC_LAUNCH(4); else if (nb == 6) DEC_LAUNCH(6); else DEC_LAUNCH(8);
#undef DEC_LAUNCH
    }
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
                     S.coefA.p, dev_dst ? dev_dst : (uint8_t*)S.out_d.p);
  pf.end(c);
  if (!dev_dst) HIPCHK(c, hipMemcpyAsync(out, S.out_d.p, out_total, hipMemcpyDeviceToHost, c->st));
  c->host_time("host_dec_queue", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp2).count());
  HIPCHK(c, ctx_sync(c));  // (the parsed tables are local: the kernels must be through with them)
  HIPCHK(c, hipGetLastError());
  c->host_time("host_dec_total", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
  return QSVC_OK;
}

// internal (qsvc_api.hip: qsvc_j2k_decode_frames): the 3 * count planes of `count` pictures into a device-resident I420 sequence
extern "C" int qsvc_j2k_decode_to_device(qsvc_ctx* c, int nstreams, const uint8_t* const* streams, const size_t* lens, int max_layers,
                                         uint8_t* dev_dst, const size_t* dev_offs, int fw, int fh) {
  std::vector<size_t> offs((size_t)nstreams + 1);
  return decode_impl(c, nstreams, streams, lens, max_layers, nullptr, 0, offs.data(), nullptr, nullptr, dev_dst, dev_offs, fw, fh);
}

extern "C" int qsvc_j2k_decode_planes(qsvc_ctx* c, int nstreams, const uint8_t* const* streams, const size_t* lens, int max_layers,
                                      void* out, size_t cap, size_t* offs, int* widths, int* heights) {
  if (!c || nstreams < 1 || !streams || !lens || !out || !offs) return QSVC_ERR_INVALID;
  return decode_impl(c, nstreams, streams, lens, max_layers, out, cap, offs, widths, heights, nullptr, nullptr, 0, 0);
}

void j2k_decode_release(qsvc_ctx* c) {
  J2kDecState* s = c->j2kdec;
  if (!s) return;
  DevBuf* all[] = {&s->cbs_d, &s->segs_d, &s->flags_d, &s->lastp_d, &s->coefA, &s->coefB, &s->planes_d, &s->out_d};
  for (auto* b : all) b->release();
  if (s->segs_h) (void)hipHostFree(s->segs_h);
  if (s->cbs_h) (void)hipHostFree(s->cbs_h);
  delete s;
  c->j2kdec = nullptr;
}
