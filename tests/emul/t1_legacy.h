// tests/emul/t1_legacy.h -- TEST INFRASTRUCTURE.  The superseded Tier-1 forms of rounds 1-2 (one code-block per LANE with the
// flag words of T.800 Annex D in memory: the fused lane coder t1_encode_lane, the bit-plane-parallel context modelling
// t1_cm_plane over t1_pass, the lane-walks-segments MQ driver t1_mq_block), kept as an independent second implementation for
// tests/emul/emul.cpp: the product's sample-parallel context modelling (qsvc_amd/csrc/t1_cm2.h) and its MQ kernels are checked
// against them on the host.  Nothing in qsvc_amd/ includes this file.
#pragma once
#include "../../qsvc_amd/csrc/t1_core.h"

namespace qsvc {

// sample (stripe-row r, column c, row-in-stripe k) has just become significant with sign `neg`:
// publish it to every flag word whose neighbourhood contains it.  The three words of the row that are in
// registers -- `prv` (r, c-1), `own` (r, c), `nxt` (r, c+1) -- are updated in place (they are stored once, when
// they leave the register window); only the words of the stripe rows above/below go through memory.
template <class Store>
T1HD void t1_set_sig(Store& S, int r, int c, uint32_t& prv, uint32_t& own, uint32_t& nxt, int k, int neg) {
  own |= (1u << (3 * (k + 1) + 1)) | ((uint32_t)neg << (T1_CHI_SHIFT + k + 1));
  prv |= 1u << (3 * (k + 1) + 2);
  nxt |= 1u << (3 * (k + 1) + 0);
  if (k == 0) {  // also row 4 (index 5) of the stripe above
    S.orr(r - 1, c, (1u << 16) | ((uint32_t)neg << (T1_CHI_SHIFT + 5)));
    S.orr(r - 1, c - 1, 1u << 17);
    S.orr(r - 1, c + 1, 1u << 15);
  } else if (k == 3) {  // also row -1 (index 0) of the stripe below
    S.orr(r + 1, c, (1u << 1) | ((uint32_t)neg << (T1_CHI_SHIFT + 0)));
    S.orr(r + 1, c - 1, 1u << 2);
    S.orr(r + 1, c + 1, 1u << 0);
  }
}
// One coding pass (0 significance propagation D.3.1, 1 magnitude refinement D.3.3, 2 clean-up D.3.4) of
// bit-plane p over the whole block.  Control flow over (stripe, column) is wave-uniform; `lane_pass`
// predicates the lane.  Decisions go to the Sink in scan order; returns the pass's distortion decrease.
template <class Store, class Sink>
T1HD long long t1_pass(Sink& K, Store& S, const uint8_t* zc, int pass, int p, int w, int h, int wmax, int nstripes,
                       bool lane_pass, bool want_dist) {
  long long dist = 0;
  S.pass_begin(nstripes);
  for (int s = 0; s < nstripes; s++) {
    S.stripe_begin(s, nstripes);
    const int r = s + 1;
    const bool row_on = lane_pass && 4 * s < h;
    // software pipeline: the flag word and magnitudes of column x+1 are fetched while column x is coded
    // register window over the row's flag words: fprv = column x-1 (apron word for x = 0), f = column x,
    // fnx = column x+1 (prefetched).  *_mem = the value memory holds, so each word is stored once, when it
    // leaves the window, and only if it changed.
    uint32_t fpre = 0, fprv = 0, fpre_mem = 0, fprv_mem = 0;
    uint64_t mpre = 0;
    if (row_on && 0 < w) { fpre = fpre_mem = S.ld(r, 1); mpre = S.mag(s, 0); }
    for (int x = 0; x < wmax; x++) {
      const bool on = row_on && x < w;
      const int c = x + 1;
      uint32_t f = fpre;
      uint32_t mlo = (uint32_t)mpre, mhi = (uint32_t)(mpre >> 32);
      T1_PIN32(f); T1_PIN32(mlo); T1_PIN32(mhi);
      const uint64_t m = (uint64_t)mlo | ((uint64_t)mhi << 32);
      const uint32_t f_mem = fpre_mem;
      uint32_t fnx = 0;
      uint64_t mnx = 0;
      if (row_on && x + 1 < w) { fnx = S.ld(r, c + 1); mnx = S.mag(s, x + 1); }
      const uint32_t fnx_mem = fnx;
      do {  // one stripe column (break = nothing (more) to code here)
      if (!on) break;
      if (pass == 0) {
        if ((f & T1_SIG_ALL) == 0) break;
        for (int k = 0; k < 4 && 4 * s + k < h; k++) {
          int nb = (f >> (3 * k)) & 0x1FF;
          if ((nb & 0x10) || (nb & 0x1EF) == 0) continue;
          int mag = t1_mag(m, k), bit = (mag >> p) & 1;
          K.sym(zc[nb & 0x1EF], bit);
          if (bit) {
            int neg = t1_neg(m, k);
            t1_code_sign(K, fprv, fnx, f, k, neg);
            t1_set_sig(S, r, c, fprv, f, fnx, k, neg);
            if (want_dist) dist += t1_dist_sig(mag, p);
          }
          f |= 1u << (T1_PI_SHIFT + k);
        }
      } else if (pass == 1) {
        uint32_t own = ((f >> 4) & 1) | ((f >> 6) & 2) | ((f >> 8) & 4) | ((f >> 10) & 8);  // sigma of rows 0..3
        uint32_t todo = own & ~(f >> T1_PI_SHIFT) & 0xF;
        if (!todo) break;
        for (int k = 0; k < 4; k++) {
          if (!((todo >> k) & 1)) continue;
          int nb = (f >> (3 * k)) & 0x1FF;
          int cx = ((f >> (T1_MU_SHIFT + k)) & 1) ? 16 : ((nb & 0x1EF) ? 15 : 14);
          int mag = t1_mag(m, k);
          K.sym(cx, (mag >> p) & 1);
          f |= 1u << (T1_MU_SHIFT + k);
          if (want_dist) dist += t1_dist_ref(mag, p);
        }
      } else {
        uint32_t own = ((f >> 4) & 1) | ((f >> 6) & 2) | ((f >> 8) & 4) | ((f >> 10) & 8);
        uint32_t done = (own | (f >> T1_PI_SHIFT)) & 0xF;
        int rows = h - 4 * s; if (rows > 4) rows = 4;
        if (done != 0xF || rows < 4) {
          int k = 0;
          if (rows == 4 && (f & (T1_SIG_ALL | (0xFu << T1_PI_SHIFT))) == 0) {  // run-length mode
            int b0 = (t1_mag(m, 0) >> p) & 1, b1 = (t1_mag(m, 1) >> p) & 1, b2 = (t1_mag(m, 2) >> p) & 1, b3 = (t1_mag(m, 3) >> p) & 1;
            if (!(b0 | b1 | b2 | b3)) { K.sym(T1_CTX_RUN, 0); k = 4; }
            else {
              int fk = b0 ? 0 : (b1 ? 1 : (b2 ? 2 : 3));
              K.sym(T1_CTX_RUN, 1);
              K.sym(T1_CTX_UNI, fk >> 1);
              K.sym(T1_CTX_UNI, fk & 1);
              int neg = t1_neg(m, fk);
              t1_code_sign(K, fprv, fnx, f, fk, neg);
              t1_set_sig(S, r, c, fprv, f, fnx, fk, neg);
              if (want_dist) dist += t1_dist_sig(t1_mag(m, fk), p);
              k = fk + 1;
            }
          }
          for (; k < rows; k++) {
            int nb = (f >> (3 * k)) & 0x1FF;
            if ((nb & 0x10) || ((f >> (T1_PI_SHIFT + k)) & 1)) continue;
            int mag = t1_mag(m, k), bit = (mag >> p) & 1;
            K.sym(zc[nb & 0x1EF], bit);
            if (bit) {
              int neg = t1_neg(m, k);
              t1_code_sign(K, fprv, fnx, f, k, neg);
              t1_set_sig(S, r, c, fprv, f, fnx, k, neg);
              if (want_dist) dist += t1_dist_sig(mag, p);
            }
          }
        }
        f &= ~(0xFu << T1_PI_SHIFT);  // the "visited" marks last one bit-plane
      }
      } while (0);
      if (row_on && x >= 1 && x - 1 < w && fprv != fprv_mem) S.st(r, c - 1, fprv);  // column x-1 is final for this pass
      if (on) { fprv = f; fprv_mem = f_mem; }
      fpre = fnx; fpre_mem = fnx_mem; mpre = mnx;  // includes whatever this column put into its right neighbour
    }
    if (row_on && w == wmax && w > 0 && fprv != fprv_mem) S.st(r, w, fprv);  // the row's last column
    S.stripe_end(s);
  }
  S.pass_end(nstripes);
  return dist;
}
struct T1MqSink {  // decisions go straight into the lane's MQ coder (fused kernel)
  T1Lane* t;
  T1HD_M void sym(int cx, int d) { mq_encode(*t, cx, d); }
};

T1HD void t1_lane_init(T1Lane& t) {  // context reset (T.800 Table D.7) + INITENC (C.2.8)
  for (int c = 0; c < 19; c++) t.cx[c * t.cs] = t.mqtab[0];
  t.cx[0 * t.cs] = t.mqtab[4]; t.cx[T1_CTX_RUN * t.cs] = t.mqtab[3]; t.cx[T1_CTX_UNI * t.cs] = t.mqtab[46];
  t.A = 0x8000; t.C = 0; t.CT = 12; t.B = 0; t.n = 0; t.ob = 0; t.nsym = 0;
}
// The whole block, context modelling and MQ coding fused.  `top` = highest bit-plane index any lane of the
// wave codes (uniform loop bound); wmax/hmax = uniform scan bounds.  A lane with nbps == 0 emits nothing.
template <class Store>
T1HD void t1_encode_lane(T1Lane& t, Store& S, int top, int wmax, int hmax, bool want_dist, T1Result& res) {
  const uint8_t* zc = t.zclut + 512 * (t.orient == 1 ? 1 : (t.orient == 3 ? 2 : 0));
  t1_lane_init(t);
  T1MqSink K{&t};
  int np = 0;
  const int nstripes = (hmax + 3) >> 2;
  for (int p = top; p >= 0; p--) {
    const bool active = p < t.nbps;
    const bool first = p == t.nbps - 1;
    for (int pass = 0; pass < 3; pass++) {
      if (pass < 2 && p == top) continue;  // every lane is at (or above) its first plane: clean-up only
      const bool lane_pass = active && (pass == 2 || !first);
      long long dist = t1_pass(K, S, zc, pass, p, t.w, t.h, wmax, nstripes, lane_pass, want_dist);
      if (lane_pass) { res.pass_len[np] = t.n + 4; if (res.pass_dist) res.pass_dist[np] = dist; np++; }
    }
  }
  t1_lane_finish(t, np, res);
}
// initial flag word (r = stripe+1, c = x+1) of bit-plane p from the magnitudes; M(s, x) returns the packed
// magnitudes (0 outside the block).
template <class MagF>
T1HD uint32_t t1_initial_flag(const MagF& M, int s, int x, int p, int nstripes_blk, int w) {
  uint32_t f = 0;
  for (int dc = 0; dc < 3; dc++) {
    int xx = x - 1 + dc;
    if (xx < 0 || xx >= w) continue;
    uint64_t mu = s > 0 ? M(s - 1, xx) : 0, mc = M(s, xx), md = s + 1 < nstripes_blk ? M(s + 1, xx) : 0;
    // rows -1..4 of the neighbourhood: last row of the stripe above, the 4 own rows, first row of the stripe below
    int mags[6] = {t1_mag(mu, 3), t1_mag(mc, 0), t1_mag(mc, 1), t1_mag(mc, 2), t1_mag(mc, 3), t1_mag(md, 0)};
    int negs[6] = {t1_neg(mu, 3), t1_neg(mc, 0), t1_neg(mc, 1), t1_neg(mc, 2), t1_neg(mc, 3), t1_neg(md, 0)};
    for (int rr = 0; rr < 6; rr++) {
      if ((mags[rr] >> (p + 1)) == 0) continue;
      f |= 1u << (3 * rr + dc);
      if (dc == 1) {
        f |= (uint32_t)negs[rr] << (T1_CHI_SHIFT + rr);
        if (rr >= 1 && rr <= 4 && (mags[rr] >> (p + 2))) f |= 1u << (T1_MU_SHIFT + rr - 1);
      }
    }
  }
  return f;
}
struct T1SymSink {  // decisions are appended to the (block, plane, pass) segment, 4 per u32 store
  uint8_t* seg;
  int n;
  uint32_t acc;
  T1HD_M void begin(uint8_t* s) { seg = s; n = 0; acc = 0; }
  T1HD_M void sym(int cx, int d) {
    acc |= (uint32_t)(cx | (d << 5)) << (8 * (n & 3));
    n++;
    if ((n & 3) == 0) { *(uint32_t*)(seg + n - 4) = acc; acc = 0; }
  }
  T1HD_M int end() {
    if (n & 3) *(uint32_t*)(seg + (n & ~3)) = acc;
    return n;
  }
};

// Context modelling of ONE bit-plane of one block (lane): the flag words must hold t1_initial_flag().
// cnt[3]/dist[3] receive the decisions and the distortion decrease of each pass.
template <class Store>
T1HD void t1_cm_plane(Store& S, const uint8_t* zclut, uint8_t* seg, int p, int w, int h, int orient, int nbps, int wmax,
                      int hmax, bool want_dist, int* cnt, long long* dist) {
  const uint8_t* zc = zclut + 512 * (orient == 1 ? 1 : (orient == 3 ? 2 : 0));
  const int nstripes = (hmax + 3) >> 2;
  const bool active = p < nbps, first = p == nbps - 1;
  T1SymSink K;
  for (int pass = 0; pass < 3; pass++) {
    const bool lane_pass = active && (pass == 2 || !first);
    K.begin(seg + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2)));
    long long d = t1_pass(K, S, zc, pass, p, w, h, wmax, nstripes, lane_pass, want_dist);
    int n = K.end();
    if (lane_pass) { cnt[pass] = n; dist[pass] = d; } else { cnt[pass] = 0; dist[pass] = 0; }
  }
}

// MQ coding of one block from its decision segments.  Seg(p) -> base of plane p's segments,
// Cnt(p, pass) -> decisions in that pass.  Sequential per lane; different lanes run different lengths.
template <class SegF, class CntF, class DistF>
T1HD void t1_mq_block(T1Lane& t, const SegF& Seg, const CntF& Cnt, const DistF& Dist, T1Result& res, bool front_back = false) {
  uint32_t stage[T1_STAGE / 4];
  t.stg = (uint8_t*)stage;
  mq2_lane_init(t);  // t.cx: 20 words
  int np = 0;
  for (int p = t.nbps - 1; p >= 0; p--)
    for (int pass = (p == t.nbps - 1 ? 2 : 0); pass < 3; pass++) {
      const uint8_t* sg = Seg(p) + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      int n = Cnt(p, pass);
      for (int i = 0; i < n; i += 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int k = 0; k < 16; k++) w[k >> 2] |= (i + k < n ? (uint32_t)sg[i + k] : T1_PAD) << (8 * (k & 3));
        if (front_back) { uint32_t rec[16]; mq2_front16(t, w[0], w[1], w[2], w[3], rec, 1); mq2_back_chunk(t, rec, 1); }
        else mq2_chunk16(t, w[0], w[1], w[2], w[3]);
        if (t.sp - t.stg >= T1_STAGE_FLUSH) mq2_stage_flush(t);
      }
      t.nsym += (uint32_t)n;
      res.pass_len[np] = mq2_count(t) + 4;
      if (res.pass_dist) res.pass_dist[np] = Dist(p, pass);
      np++;
    }
  t1_lane_finish<true>(t, np, res);
}

// Host/array store: lane stride ls (1 on the host harness).
struct T1ArrayStore {
  uint32_t* flags;       // word (r, c) at flags[(r*T1_FW + c)*ls]
  const uint64_t* mags;  // word (s, x) at mags[(s*64 + x)*ls]
  int ls;
  T1HD_M uint32_t ld(int r, int c) const { return flags[(r * T1_FW + c) * ls]; }
  T1HD_M void st(int r, int c, uint32_t v) { flags[(r * T1_FW + c) * ls] = v; }
  T1HD_M void orr(int r, int c, uint32_t b) { flags[(r * T1_FW + c) * ls] |= b; }
  T1HD_M uint64_t mag(int s, int x) const { return mags[(s * 64 + x) * ls]; }
  T1HD_M void pass_begin(int) {}
  T1HD_M void stripe_begin(int, int) {}
  T1HD_M void stripe_end(int) {}
  T1HD_M void pass_end(int) {}
};

}  // namespace qsvc
