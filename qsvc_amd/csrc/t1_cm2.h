// qsvc_amd/csrc/t1_cm2.h -- EBCOT context modelling (T.800 Annex D), SAMPLE-parallel form.
//
// Within one bit-plane p every sample's fate is a function of the magnitudes except one thing: WHICH insignificant samples
// the significance-propagation pass visits.  A sample that is coded while insignificant becomes significant iff its bit at
// p is 1 -- whatever pass codes it -- so
//     sig0 = mag >> (p+1) != 0                 significant before the plane (-> magnitude refinement pass)
//     M    = !sig0 & "has a significant neighbour when the significance pass reaches it"
//     S1   = M & bit(p)                         becomes significant in the significance pass
//     C    = !sig0 & !M                         left to the clean-up pass;   S2 = C & bit(p)
// and the significance of a neighbour n at the moment a sample is coded is
//     significance pass: sig0(n) | S1(n)[n earlier in the scan]     refinement: sig0(n) | S1(n)
//     clean-up:          sig0(n) | S1(n) | S2(n)[n earlier in the scan].
// Only M has a scan-order recurrence (through S1 of earlier samples): it runs first, one lane per stripe, each stripe two
// columns behind the one above (cm2_member).  After that every stripe column emits its decisions of every pass
// independently (cm2_column) and a prefix sum of the per-column counts puts them in scan order.  No flag words in memory,
// no neighbour updates, no atomics: the state is one 32-bit word per stripe column, built from the magnitudes.
//
// The per-column code is the column body of t1_pass (t1_core.h) on a locally assembled flag word, so both forms produce
// the same decisions (tests/emul checks all of them against the restatement's Tier-1).
#pragma once
#include "t1_core.h"

namespace qsvc {

// column word: 4-bit fields, bit k = row 4s+k
#define CW_SIG0 0
#define CW_BIT 4
#define CW_NEG 8
#define CW_MU 12
#define CW_VAL 16
#define CW_M 20
#define CW_S1 24
#define CW_S2 28

T1HD uint32_t cm2_colword(uint64_t m, int p, int rows, bool col_ok) {
  uint32_t w = 0;
  for (int k = 0; k < 4; k++) {
    const int mag = t1_mag(m, k);
    w |= (uint32_t)((mag >> (p + 1)) != 0) << (CW_SIG0 + k);
    w |= (uint32_t)((mag >> p) & 1) << (CW_BIT + k);
    w |= (uint32_t)t1_neg(m, k) << (CW_NEG + k);
    w |= (uint32_t)((mag >> (p + 2)) != 0) << (CW_MU + k);
    w |= (uint32_t)(col_ok && k < rows) << (CW_VAL + k);
  }
  return w;
}
// 6-row vector (bit r = row r-1 of the stripe: the row above, the four rows, the row below) of one 4-bit field
T1HD uint32_t cm2_six(uint32_t up, uint32_t mid, uint32_t dn, int f) {
  return ((up >> (f + 3)) & 1) | (((mid >> f) & 0xF) << 1) | (((dn >> f) & 1) << 5);
}
// 6-row vector -> every third bit (bit r -> 3r): one column of the 18-bit neighbourhood word of t1_core.h
#define CM2_SPREAD_INIT(v) (((v) & 1) | (((v) & 2) << 2) | (((v) & 4) << 4) | (((v) & 8) << 6) | (((v) & 16) << 8) | (((v) & 32) << 10))

// Significance-pass membership of stripe column (s, x).  u/m/d = the column words of stripes s-1, s, s+1 at x-1, x, x+1
// (zero outside the block); M/S1 of stripe s-1 (all columns) and of (s, x-1) must be final.  Returns the word's M, S1, S2.
T1HD uint32_t cm2_member(const uint32_t u[3], const uint32_t m[3], const uint32_t d[3]) {
  const uint32_t L6 = cm2_six(u[0], m[0], d[0], CW_SIG0) | cm2_six(u[0], m[0], 0, CW_S1);  // x-1: rows -1..3 are earlier
  uint32_t C6 = cm2_six(u[1], m[1], d[1], CW_SIG0) | ((u[1] >> (CW_S1 + 3)) & 1);          // x: the row above is earlier
  const uint32_t R6 = cm2_six(u[2], m[2], d[2], CW_SIG0) | ((u[2] >> (CW_S1 + 3)) & 1);    // x+1: only the row above
  const uint32_t w = m[1];
  uint32_t M = 0, S1 = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t nbh = ((L6 >> k) & 7) | ((R6 >> k) & 7) | ((C6 >> k) & 5);
    const uint32_t in = (w >> (CW_VAL + k)) & ~(w >> (CW_SIG0 + k)) & (uint32_t)(nbh != 0) & 1;
    const uint32_t s1 = in & (w >> (CW_BIT + k));
    M |= in << k; S1 |= s1 << k;
    C6 |= s1 << (k + 1);
  }
  const uint32_t C = (w >> CW_VAL) & ~(w >> CW_SIG0) & ~M & 0xF;
  const uint32_t S2 = C & (w >> CW_BIT) & 0xF;
  return (M << CW_M) | (S1 << CW_S1) | (S2 << CW_S2);
}

// The same membership as a fixpoint over ROW masks (bit x of row y; 64 rows = the 64 lanes of the wave).  With
// S1 = M & bit the recurrence of cm2_member reads, for row y = 4s + k,
//   M[y] = cand[y] & ( seedZ[y]                                       neighbours significant before the plane
//                    | S1[y] << 1                                      (y, x-1): a carry chain along the row
//                    | S1[y-1] << 1 | S1[y-1] | (k == 0 ? S1[y-1] >> 1 : 0)   row above: x-1, x always earlier in the scan,
//                                                                      x+1 only when it lies in the stripe above
//                    | (k < 3 ? S1[y+1] << 1 : 0) )                    (y+1, x-1) of the same stripe
// The chain along the row is an adder's carry chain (generate = g & bit, propagate = cand & bit), resolved by one 64-bit
// addition; the coupling between rows converges by iteration from M = 0 (monotone, least fixpoint = the scan-order result).
T1HD uint64_t cm2_row_update(uint64_t cand, uint64_t bt, uint64_t seedZ, uint64_t s1_up, uint64_t s1_dn, int k) {
  const uint64_t prop = (s1_up << 1) | s1_up | (k == 0 ? (s1_up >> 1) : 0) | (k < 3 ? (s1_dn << 1) : 0);
  const uint64_t g = cand & (seedZ | prop);
  const uint64_t gp = g & bt, pp = cand & bt;
  const uint64_t A = gp | pp, B = gp;
  const uint64_t cin = (A + B) ^ A ^ B;  // carry INTO each bit = S1 of the sample to the left
  return g | (cand & cin);
}
T1HD uint64_t cm2_row_seed(uint64_t z, uint64_t z_up, uint64_t z_dn) {
  return (z << 1) | (z >> 1) | (z_up << 1) | z_up | (z_up >> 1) | (z_dn << 1) | z_dn | (z_dn >> 1);
}

struct Cm2Ctx {
  const uint8_t* zc;     // [512] zero-coding LUT of the block's orientation class
  const uint32_t* sp;    // [64] CM2_SPREAD_INIT
  int p;
  bool want_dist;
};
struct Cm2CountSink {
  int n;
  T1HD_M void sym(int, int) { n++; }
};
struct Cm2ByteSink {  // decisions as bytes cx | d << 5, in order
  uint8_t* out;
  int n;
  T1HD_M void sym(int cx, int d) { out[n++] = (uint8_t)(cx | (d << 5)); }
};

// Everything the three passes of one stripe column need, assembled once from the 3 x 3 neighbourhood of column words.
struct Cm2Col {
  uint32_t w;                 // the column's own word
  uint32_t s0[3], s1[3], s2[3];  // 6-row vectors of sig0 / S1 / S2 for columns x-1, x, x+1
  uint32_t chiL, chiC, chiR;  // 6-row sign vectors at T1_CHI_SHIFT (the layout t1_code_sign reads)
};
T1HD void cm2_gather(Cm2Col& c, const uint32_t u[3], const uint32_t m[3], const uint32_t d[3]) {
  c.w = m[1];
  for (int k = 0; k < 3; k++) {
    c.s0[k] = cm2_six(u[k], m[k], d[k], CW_SIG0);
    c.s1[k] = cm2_six(u[k], m[k], d[k], CW_S1);
    c.s2[k] = cm2_six(u[k], m[k], d[k], CW_S2);
  }
  c.chiL = cm2_six(u[0], m[0], d[0], CW_NEG) << T1_CHI_SHIFT;
  c.chiC = cm2_six(u[1], m[1], d[1], CW_NEG) << T1_CHI_SHIFT;
  c.chiR = cm2_six(u[2], m[2], d[2], CW_NEG) << T1_CHI_SHIFT;
}

// Sliding form for a lane that walks consecutive columns of one stripe: the column that was x+1 becomes x, x becomes x-1, and
// only the new right-hand column (words u2, m2, d2 of stripes s-1, s, s+1 at x+2 of the old position) is taken apart.
T1HD void cm2_gather_first(Cm2Col& c, const uint32_t u[2], const uint32_t m[2], const uint32_t d[2]) {  // columns x-1, x into slots 1, 2
  for (int k = 0; k < 2; k++) {
    c.s0[k + 1] = cm2_six(u[k], m[k], d[k], CW_SIG0);
    c.s1[k + 1] = cm2_six(u[k], m[k], d[k], CW_S1);
    c.s2[k + 1] = cm2_six(u[k], m[k], d[k], CW_S2);
  }
  c.chiC = cm2_six(u[0], m[0], d[0], CW_NEG) << T1_CHI_SHIFT;
  c.chiR = cm2_six(u[1], m[1], d[1], CW_NEG) << T1_CHI_SHIFT;
  c.w = m[1];
}
T1HD void cm2_gather_slide(Cm2Col& c, uint32_t own, uint32_t u2, uint32_t m2, uint32_t d2) {  // own = the word of the new centre column
  c.s0[0] = c.s0[1]; c.s0[1] = c.s0[2]; c.s0[2] = cm2_six(u2, m2, d2, CW_SIG0);
  c.s1[0] = c.s1[1]; c.s1[1] = c.s1[2]; c.s1[2] = cm2_six(u2, m2, d2, CW_S1);
  c.s2[0] = c.s2[1]; c.s2[1] = c.s2[2]; c.s2[2] = cm2_six(u2, m2, d2, CW_S2);
  c.chiL = c.chiC; c.chiC = c.chiR; c.chiR = cm2_six(u2, m2, d2, CW_NEG) << T1_CHI_SHIFT;
  c.w = own;
}

// The decisions of one stripe column in one pass (0 significance propagation, 1 magnitude refinement, 2 clean-up), in
// coding order, with the column words final (M/S1/S2 set everywhere).  mag = the column's four packed magnitudes (only the
// distortion estimates read it).
template <class Sink>
T1HD void cm2_pass(Sink& K, const Cm2Ctx& cx, int pass, const Cm2Col& c, uint64_t mag, long long& dist) {
  const uint32_t w = c.w;
  const uint32_t val4 = (w >> CW_VAL) & 0xF;
  if (!val4) return;
  const int rows = val4 == 0xF ? 4 : (val4 == 7 ? 3 : (val4 == 3 ? 2 : 1));
  const uint32_t bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & 0xF, sig04 = (w >> CW_SIG0) & 0xF;
  const int p = cx.p;
  if (pass == 1) {
    const uint32_t todo = sig04 & val4;
    if (!todo) return;
    const uint32_t f = cx.sp[c.s0[0] | c.s1[0]] | (cx.sp[c.s0[1] | c.s1[1]] << 1) | (cx.sp[c.s0[2] | c.s1[2]] << 2);
    for (int k = 0; k < 4; k++) {
      if (!((todo >> k) & 1)) continue;
      const int nb = (f >> (3 * k)) & 0x1FF;
      const int ctx = ((w >> (CW_MU + k)) & 1) ? 16 : ((nb & 0x1EF) ? 15 : 14);
      K.sym(ctx, (bit4 >> k) & 1);
      if (cx.want_dist) dist += t1_dist_ref(t1_mag(mag, k), p);
    }
    return;
  }
  if (pass == 0) {
    const uint32_t L6 = c.s0[0] | (c.s1[0] & 0x1F), C6 = c.s0[1] | (c.s1[1] & 1), R6 = c.s0[2] | (c.s1[2] & 1);
    uint32_t f = cx.sp[L6] | (cx.sp[C6] << 1) | (cx.sp[R6] << 2);
    if (!f) return;
    f |= c.chiC;
    for (int k = 0; k < rows; k++) {
      const int nb = (f >> (3 * k)) & 0x1FF;
      if ((nb & 0x10) || (nb & 0x1EF) == 0) continue;
      const int bit = (bit4 >> k) & 1;
      K.sym(cx.zc[nb & 0x1EF], bit);
      if (bit) {
        t1_code_sign(K, c.chiL, c.chiR, f, k, (int)((w >> (CW_NEG + k)) & 1));
        f |= 1u << (3 * (k + 1) + 1);
        if (cx.want_dist) dist += t1_dist_sig(t1_mag(mag, k), p);
      }
    }
    return;
  }
  // clean-up
  if (((sig04 | M4) & val4) == val4) return;  // every row was coded by one of the other passes
  const uint32_t L6 = c.s0[0] | c.s1[0] | (c.s2[0] & 0x1F), C6 = c.s0[1] | c.s1[1] | (c.s2[1] & 1), R6 = c.s0[2] | c.s1[2] | (c.s2[2] & 1);
  uint32_t f = cx.sp[L6] | (cx.sp[C6] << 1) | (cx.sp[R6] << 2);
  int k = 0;
  if (rows == 4 && f == 0 && M4 == 0) {  // run-length mode (D.3.4)
    if (!bit4) { K.sym(T1_CTX_RUN, 0); return; }
    const int fk = (bit4 & 1) ? 0 : ((bit4 & 2) ? 1 : ((bit4 & 4) ? 2 : 3));
    K.sym(T1_CTX_RUN, 1);
    K.sym(T1_CTX_UNI, fk >> 1);
    K.sym(T1_CTX_UNI, fk & 1);
    f |= c.chiC;
    t1_code_sign(K, c.chiL, c.chiR, f, fk, (int)((w >> (CW_NEG + fk)) & 1));
    f |= 1u << (3 * (fk + 1) + 1);
    if (cx.want_dist) dist += t1_dist_sig(t1_mag(mag, fk), p);
    k = fk + 1;
  } else {
    f |= c.chiC;
  }
  for (; k < rows; k++) {
    const int nb = (f >> (3 * k)) & 0x1FF;
    if ((nb & 0x10) || ((M4 >> k) & 1)) continue;
    const int bit = (bit4 >> k) & 1;
    K.sym(cx.zc[nb & 0x1EF], bit);
    if (bit) {
      t1_code_sign(K, c.chiL, c.chiR, f, k, (int)((w >> (CW_NEG + k)) & 1));
      f |= 1u << (3 * (k + 1) + 1);
      if (cx.want_dist) dist += t1_dist_sig(t1_mag(mag, k), p);
    }
  }
}
// The number of decisions cm2_pass emits for the column in each pass, from the masks alone (every member of a pass costs one
// decision, every sample that becomes significant one more for its sign; a run-length column 1, or 3 + the sign).
T1HD int cm2_popc4(uint32_t v) { v &= 0xF; v = (v & 5) + ((v >> 1) & 5); return (int)((v & 3) + (v >> 2)); }
T1HD void cm2_counts(const Cm2Col& c, int n[3]) {
  const uint32_t w = c.w;
  const uint32_t val4 = (w >> CW_VAL) & 0xF;
  n[0] = n[1] = n[2] = 0;
  if (!val4) return;
  const uint32_t bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & 0xF, sig04 = (w >> CW_SIG0) & 0xF;
  const uint32_t S14 = (w >> CW_S1) & 0xF, S24 = (w >> CW_S2) & 0xF;
  n[0] = cm2_popc4(M4) + cm2_popc4(S14);
  n[1] = cm2_popc4(sig04 & val4);
  const uint32_t C4 = val4 & ~sig04 & ~M4;
  if (!C4) return;
  const uint32_t any = c.s0[0] | c.s1[0] | (c.s2[0] & 0x1F) | c.s0[1] | c.s1[1] | (c.s2[1] & 1) | c.s0[2] | c.s1[2] | (c.s2[2] & 1);
  if (val4 == 0xF && any == 0 && M4 == 0) {
    if (!bit4) { n[2] = 1; return; }
    const int fk = (bit4 & 1) ? 0 : ((bit4 & 2) ? 1 : ((bit4 & 4) ? 2 : 3));
    const uint32_t rest = 0xFu & ~((2u << fk) - 1);  // rows below fk
    n[2] = 4 + cm2_popc4(C4 & rest) + cm2_popc4(S24 & rest);
    return;
  }
  n[2] = cm2_popc4(C4) + cm2_popc4(S24);
}

template <class Sink>
T1HD void cm2_column(Sink& K, const Cm2Ctx& cx, int pass, const uint32_t u[3], const uint32_t m[3], const uint32_t d[3],
                     uint64_t mag, long long& dist) {
  if (!((m[1] >> CW_VAL) & 0xF)) return;
  Cm2Col c;
  cm2_gather(c, u, m, d);
  cm2_pass(K, cx, pass, c, mag, dist);
}

}  // namespace qsvc
