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
// The per-column code is the column body of t1_pass (tests/emul/t1_legacy.h) on a locally assembled flag word, so both forms produce
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
// v_perm_b32: byte k of the result = byte sel[k] of {b (0..3), a (4..7)}
T1HD uint32_t cm2_perm(uint32_t a, uint32_t b, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_perm(a, b, sel);
#else
  const uint64_t ab = (uint64_t)b | (uint64_t)a << 32;
  uint32_t r = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t q = (sel >> (8 * k)) & 0xFF;   // 0..7: that byte; 12: 0x00; 13 and above: 0xFF (8..11, the sign replications, are not used here)
    r |= (q >= 13 ? 0xFFu : (q == 12 ? 0u : (uint32_t)((ab >> (8 * (q & 7))) & 0xFF))) << (8 * k);
  }
  return r;
#endif
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

// ---------------------------------------------------------------------------------------------
// The three passes of one stripe column AT ONCE, without a branch per sample: every sample belongs to exactly one pass (sig0 ->
// refinement, M -> significance, else clean-up) and, the masks being final, its neighbourhood "at the moment it is coded" is a
// function of the masks alone -- the in-column update of cm2_pass ("f |= ..." when a sample turns significant) only ever reaches
// the next row, as its upper neighbour, and equals S1 (S2) of the row above.  A column's decisions leave as three short byte
// strings (at most 8 bytes of the significance pass, 4 of the refinement pass, a run-length header of up to 3 bytes and 8 more
// of the clean-up pass), which the caller appends to the lane's three streams with whole-word accesses.
struct Cm2Bytes { uint64_t a; int n; };             // n bytes, the first in the low byte
struct Cm2Out3 { Cm2Bytes p0, p1, hdr, p2; };
// sign coding (T.800 D.3.2) as a table: index = sigU | sgnU<<1 | sigL<<2 | sgnL<<3 | sigR<<4 | sgnR<<5 | sigD<<6 | sgnD<<7,
// value = context | xor-bit << 5 (the decision is value ^ (negative << 5))
T1HD uint8_t cm2_sign_entry(int idx) {
  auto contrib = [&](int sig, int sgn) { return sig ? (sgn ? -1 : 1) : 0; };
  int hc = contrib((idx >> 2) & 1, (idx >> 3) & 1) + contrib((idx >> 4) & 1, (idx >> 5) & 1);
  int vc = contrib(idx & 1, (idx >> 1) & 1) + contrib((idx >> 6) & 1, (idx >> 7) & 1);
  hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
  vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
  const int xr = hc < 0 || (hc == 0 && vc < 0);
  if (xr) { hc = -hc; vc = -vc; }
  const int cx = hc == 1 ? 12 + vc : (vc == 1 ? 10 : 9);
  return (uint8_t)(cx | (xr << 5));
}
// mag / dist[3]: the column's packed magnitudes and the per-pass distortion sums (touched only with cx.want_dist).
// Written with 0 / ~0 masks and bit-field inserts rather than conditions: a GPU lane pays for both sides of a branch anyway.
T1HD uint32_t cm2_m(uint32_t field4, int k) { return (uint32_t)((int32_t)(field4 << (31 - k)) >> 31); }  // bit k of a 4-bit field -> 0 / ~0
T1HD uint32_t cm2_pick(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }         // mask ? a : b
T1HD void cm2_emit3(Cm2Out3& O, const Cm2Ctx& cx, const uint8_t* signlut, const Cm2Col& c, uint64_t mag = 0, long long* dist = nullptr) {
  const uint32_t w = c.w;
  const uint32_t val4 = (w >> CW_VAL) & 0xF;
  const uint32_t bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & val4, sig04 = (w >> CW_SIG0) & val4;
  const uint32_t mu4 = (w >> CW_MU) & 0xF, neg4 = (w >> CW_NEG) & 0xF;
  const uint32_t C4 = val4 & ~sig04 & ~M4;
  const uint32_t S14 = M4 & bit4, S24 = C4 & bit4;
  // the 18-bit neighbourhood words (bit 3r + col, r = 0 the row above the stripe) as each pass sees them
  const uint32_t A0 = cx.sp[c.s0[0]], A1 = cx.sp[c.s0[1]], A2 = cx.sp[c.s0[2]];
  const uint32_t B0 = cx.sp[c.s1[0]], B1 = cx.sp[c.s1[1]], B2 = cx.sp[c.s1[2]];
  const uint32_t D0 = cx.sp[c.s2[0]], D1 = cx.sp[c.s2[1]], D2 = cx.sp[c.s2[2]];
  const uint32_t fref = (A0 | B0) | (A1 | B1) << 1 | (A2 | B2) << 2;
  const uint32_t f0 = (A0 | (B0 & 0x7FFF)) | (A1 | (B1 & 1)) << 1 | (A2 | (B2 & 1)) << 2;
  const uint32_t f2 = fref | (D0 & 0x7FFF) | (D1 & 1) << 1 | (D2 & 1) << 2;
  const uint32_t fchi = cx.sp[(c.chiL >> T1_CHI_SHIFT) & 63] | cx.sp[(c.chiC >> T1_CHI_SHIFT) & 63] << 1 | cx.sp[(c.chiR >> T1_CHI_SHIFT) & 63] << 2;
  // a sample of this column that turned significant earlier in the SAME pass is the upper neighbour of the next row only:
  // S1 / S2 of the column's own rows at the centre positions, cut off below the row in work
  const uint32_t G1 = (B1 & 0x1248) << 1, G2 = (D1 & 0x1248) << 1;
  // run-length mode (D.3.4): four rows, no significant neighbour (so none significant, none in the significance pass)
  const uint32_t rl = (val4 == 0xF && f2 == 0) ? ~0u : 0u;
  const uint32_t lowest = bit4 & (0u - bit4);                     // the first row with a 1 (0: none)
  const uint32_t skipA4 = rl & (2 * lowest - 1) & 0xF, skipB4 = rl & (lowest - 1) & 0xF;   // rows the run covers: no zero-coding decision / no sign yet
  const uint32_t fk = (lowest >> 1) - (lowest >> 3);              // 1, 2, 4, 8 -> 0, 1, 2, 3
  const uint32_t hdr3 = (uint32_t)(T1_CTX_RUN | 32) | (uint32_t)(T1_CTX_UNI | ((fk >> 1) << 5)) << 8 | (uint32_t)(T1_CTX_UNI | ((fk & 1) << 5)) << 16;
  O.hdr.a = rl & (bit4 ? hdr3 : (uint32_t)T1_CTX_RUN);   // bytes beyond .n must be zero: the streams OR whole words
  O.hdr.n = (int)(rl & (bit4 ? 3u : 1u));
  const uint32_t EA0 = M4, EB0 = S14, EA1 = sig04, EA2 = C4 & ~skipA4, EB2 = S24 & ~skipB4;
  uint32_t x0[4], x1[4], x2[4];   // the rows' bytes for each pass (0 where the row is not of that pass)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t mM = cm2_m(M4, k), m1 = cm2_m(sig04, k), mbit = cm2_m(bit4, k);
    const uint32_t cmask = (1u << (3 * k + 2)) - 1;
    const uint32_t nbr = (fref >> (3 * k)) & 0x1EF;
    const uint32_t rctx = cm2_pick(cm2_m(mu4, k), 16u, 14u + (nbr ? 1u : 0u));
    const uint32_t nb0 = ((f0 | (G1 & cmask)) >> (3 * k)) & 0x1FF;
    const uint32_t nb2 = ((f2 | (G2 & cmask)) >> (3 * k)) & 0x1FF;
    const uint32_t nbz = cm2_pick(mM, nb0, nb2);
    const uint32_t A = cm2_pick(m1, rctx, (uint32_t)cx.zc[nbz & 0x1EF]) | (mbit & 32);
    const uint32_t idx = ((nbz & 0xAA) >> 1) | ((fchi >> (3 * k)) & 0xAA);
    const uint32_t Bv = (uint32_t)signlut[idx] ^ (cm2_m(neg4, k) & 32);
    const uint32_t ab = A | Bv << 8;
    x1[k] = A & m1;
    x0[k] = ab & cm2_m(EA0, k) & (0xFFu | (cm2_m(EB0, k) & 0xFF00u));
    const uint32_t ea2 = cm2_m(EA2, k), eb2 = cm2_m(EB2, k);
    x2[k] = cm2_pick(ea2, ab & (0xFFu | (eb2 & 0xFF00u)), Bv & eb2);
  }
  auto cnt = [](uint32_t EA, uint32_t EB, int k) { return (int)(((EA >> k) & 1) + ((EB >> k) & 1)); };
  {  // refinement: one byte per row of sig0
    const int c0 = (int)(EA1 & 1), c1 = c0 + (int)((EA1 >> 1) & 1), c2 = c1 + (int)((EA1 >> 2) & 1);
    O.p1.a = x1[0] | x1[1] << (8 * c0) | x1[2] << (8 * c1) | x1[3] << (8 * c2);
    O.p1.n = c2 + (int)((EA1 >> 3) & 1);
  }
  {  // significance: rows 0, 1 in one word, rows 2, 3 in another (at most four bytes each)
    const int c0 = cnt(EA0, EB0, 0), c1 = cnt(EA0, EB0, 1), c2 = cnt(EA0, EB0, 2), c3 = cnt(EA0, EB0, 3);
    const uint32_t lo = x0[0] | x0[1] << (8 * c0), hi = x0[2] | x0[3] << (8 * c2);
    O.p0.a = (uint64_t)lo | (uint64_t)hi << (8 * (c0 + c1));
    O.p0.n = c0 + c1 + c2 + c3;
  }
  {  // clean-up
    const int c0 = cnt(EA2, EB2, 0), c1 = cnt(EA2, EB2, 1), c2 = cnt(EA2, EB2, 2), c3 = cnt(EA2, EB2, 3);
    const uint32_t lo = x2[0] | x2[1] << (8 * c0), hi = x2[2] | x2[3] << (8 * c2);
    O.p2.a = (uint64_t)lo | (uint64_t)hi << (8 * (c0 + c1));
    O.p2.n = c0 + c1 + c2 + c3;
  }
  if (cx.want_dist && dist) {
    long long d0 = 0, d1 = 0, d2 = 0;
    for (int k = 0; k < 4; k++) {
      const int m = t1_mag(mag, k);
      if ((sig04 >> k) & 1) d1 += t1_dist_ref(m, cx.p);
      else if ((S14 >> k) & 1) d0 += t1_dist_sig(m, cx.p);
      else if ((S24 >> k) & 1) d2 += t1_dist_sig(m, cx.p);
    }
    dist[0] += d0; dist[1] += d1; dist[2] += d2;
  }
}

// The same with a sliding window of SPREAD neighbourhood columns (a lane walks consecutive columns of one stripe): only the new
// right-hand column is taken apart and spread per step, and the merged table zcx ([0..511] zero coding by neighbourhood,
// [512..1023] refinement: bit 4 = "refined before", the other neighbourhood bits = significant neighbours) gives any row's
// context with one read.
struct Cm2Win {
  uint32_t w;               // the centre column's own word
  uint32_t AB[3], P[3], Q[3], DL[3], DB[3], X[3];  // per column x-1, x, x+1: sp(s0|s1), sp(s0)|sp(s1)&1, sp(s0)|sp(s1) rows -1..3,
                                                   // sp(s2) rows -1..3, sp(s2)&1, sp(chi)
  uint32_t B1, D1;          // sp(s1), sp(s2) of the centre column (its own rows' S1 / S2)
  uint32_t Bn, Dn;          // the same of column x+1 (becomes the centre at the next step)
};
T1HD void cm2_win_push(Cm2Win& c, const uint32_t* sp, uint32_t own, uint32_t u2, uint32_t m2, uint32_t d2) {  // own = the word of the new centre
  for (int k = 0; k < 2; k++) { c.AB[k] = c.AB[k + 1]; c.P[k] = c.P[k + 1]; c.Q[k] = c.Q[k + 1]; c.DL[k] = c.DL[k + 1]; c.DB[k] = c.DB[k + 1]; c.X[k] = c.X[k + 1]; }
  c.B1 = c.Bn; c.D1 = c.Dn;
  const uint32_t A = sp[cm2_six(u2, m2, d2, CW_SIG0)], B = sp[cm2_six(u2, m2, d2, CW_S1)], D = sp[cm2_six(u2, m2, d2, CW_S2)];
  c.AB[2] = A | B; c.P[2] = A | (B & 1); c.Q[2] = A | (B & 0x7FFF); c.DL[2] = D & 0x7FFF; c.DB[2] = D & 1;
  c.X[2] = sp[cm2_six(u2, m2, d2, CW_NEG)];
  c.Bn = B; c.Dn = D;
  c.w = own;
}
// the window around an arbitrary stripe column (cw: the column-word array with its one-cell apron, row stride cwstride words;
// column x of stripe s is cw[(s + 1) * cwstride + x + 1]) -- for the sparse form, where the columns in work are not consecutive
T1HD void cm2_win_at(Cm2Win& c, const uint32_t* sp, const uint32_t* cw, int cwstride, int s, int x) {
  c.w = 0; c.B1 = c.D1 = c.Bn = c.Dn = 0;
  for (int k = 0; k < 3; k++) c.AB[k] = c.P[k] = c.Q[k] = c.DL[k] = c.DB[k] = c.X[k] = 0;
  const uint32_t *u = cw + s * cwstride + x, *m = u + cwstride, *d = m + cwstride;
  cm2_win_push(c, sp, 0, u[0], m[0], d[0]);
  cm2_win_push(c, sp, 0, u[1], m[1], d[1]);
  cm2_win_push(c, sp, m[1], u[2], m[2], d[2]);
}
// Sparse bit-planes (most columns "plain"): the columns that are not plain are listed -- column, and where its decisions go in
// each of the three streams -- and worked off 64 at a time, instead of every lane walking all of its 16 columns.
struct Cm2Entry { uint16_t col, o0, o1, o2; };   // col = stripe * 64 + x; byte offsets into the three streams
#ifndef CM2_LIST_MAX
#define CM2_LIST_MAX 256
#define CM2_LIST_AT 3840            // the list lives in the upper part of the (then small) stream buffer
#endif
// the centre column is four insignificant zeros without a significant neighbour: its only decision is RUN = 0 (clean-up pass)
T1HD bool cm2_plain_run(const Cm2Win& c) {
  const uint32_t w = c.w;
  const uint32_t f2 = c.AB[0] | c.AB[1] << 1 | c.AB[2] << 2 | c.DL[0] | c.DB[1] << 1 | c.DB[2] << 2;
  return ((w >> CW_VAL) & 0xF) == 0xF && ((w >> CW_BIT) & 0xF) == 0 && f2 == 0;
}
T1HD void cm2_emit3w(Cm2Out3& O, const uint8_t* zcx, const uint8_t* signlut, const Cm2Win& c) {
  const uint32_t w = c.w;
  const uint32_t val4 = (w >> CW_VAL) & 0xF;
  const uint32_t bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & val4, sig04 = (w >> CW_SIG0) & val4;
  const uint32_t mu4 = (w >> CW_MU) & 0xF, neg4 = (w >> CW_NEG) & 0xF;
  const uint32_t C4 = val4 & ~sig04 & ~M4;
  const uint32_t S14 = M4 & bit4, S24 = C4 & bit4;
  const uint32_t fref = c.AB[0] | c.AB[1] << 1 | c.AB[2] << 2;
  const uint32_t f0 = c.Q[0] | c.P[1] << 1 | c.P[2] << 2;
  const uint32_t f2 = fref | c.DL[0] | c.DB[1] << 1 | c.DB[2] << 2;
  const uint32_t fchi = c.X[0] | c.X[1] << 1 | c.X[2] << 2;
  const uint32_t G1 = (c.B1 & 0x1248) << 1, G2 = (c.D1 & 0x1248) << 1;
  const uint32_t rl = (val4 == 0xF && f2 == 0 && M4 == 0) ? ~0u : 0u;
  const uint32_t lowest = bit4 & (0u - bit4);
  const uint32_t skipA4 = rl & (2 * lowest - 1) & 0xF, skipB4 = rl & (lowest - 1) & 0xF;
  const uint32_t fk = (lowest >> 1) - (lowest >> 3);
  const uint32_t hdr3 = (uint32_t)(T1_CTX_RUN | 32) | (uint32_t)(T1_CTX_UNI | ((fk >> 1) << 5)) << 8 | (uint32_t)(T1_CTX_UNI | ((fk & 1) << 5)) << 16;
  O.hdr.a = rl & (bit4 ? hdr3 : (uint32_t)T1_CTX_RUN);
  O.hdr.n = (int)(rl & (bit4 ? 3u : 1u));
  const uint32_t EA0 = M4, EB0 = S14, EA1 = sig04, EA2 = C4 & ~skipA4, EB2 = S24 & ~skipB4;
  uint32_t x0[4], x1[4], x2[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t mM = cm2_m(M4, k), m1 = cm2_m(sig04, k), mbit = cm2_m(bit4, k);
    const uint32_t cmask = (1u << (3 * k + 2)) - 1;
    const uint32_t fz = cm2_pick(m1, fref, cm2_pick(mM, f0 | (G1 & cmask), f2 | (G2 & cmask)));
    const uint32_t nb = (fz >> (3 * k)) & 0x1FF;
    const uint32_t li = (nb & 0x1EF) | (m1 & (0x200u | (cm2_m(mu4, k) & 0x10u)));
    const uint32_t A = (uint32_t)zcx[li] | (mbit & 32);
    const uint32_t idx = ((nb & 0xAA) >> 1) | ((fchi >> (3 * k)) & 0xAA);
    const uint32_t Bv = (uint32_t)signlut[idx] ^ (cm2_m(neg4, k) & 32);
    const uint32_t ab = A | Bv << 8;
    x1[k] = A & m1;
    x0[k] = ab & mM & (0xFFu | (mM & mbit & 0xFF00u));
    const uint32_t ea2 = cm2_m(EA2, k), eb2 = cm2_m(EB2, k);
    x2[k] = cm2_pick(ea2, ab & (0xFFu | (eb2 & 0xFF00u)), Bv & eb2);
  }
  auto cnt = [](uint32_t EA, uint32_t EB, int k) { return (int)(((EA >> k) & 1) + ((EB >> k) & 1)); };
  {
    const int c0 = (int)(EA1 & 1), c1 = c0 + (int)((EA1 >> 1) & 1), c2 = c1 + (int)((EA1 >> 2) & 1);
    O.p1.a = x1[0] | x1[1] << (8 * c0) | x1[2] << (8 * c1) | x1[3] << (8 * c2);
    O.p1.n = c2 + (int)((EA1 >> 3) & 1);
  }
  {
    const int c0 = cnt(EA0, EB0, 0), c1 = cnt(EA0, EB0, 1), c2 = cnt(EA0, EB0, 2), c3 = cnt(EA0, EB0, 3);
    const uint32_t lo = x0[0] | x0[1] << (8 * c0), hi = x0[2] | x0[3] << (8 * c2);
    O.p0.a = (uint64_t)lo | (uint64_t)hi << (8 * (c0 + c1));
    O.p0.n = c0 + c1 + c2 + c3;
  }
  {
    const int c0 = cnt(EA2, EB2, 0), c1 = cnt(EA2, EB2, 1), c2 = cnt(EA2, EB2, 2), c3 = cnt(EA2, EB2, 3);
    const uint32_t lo = x2[0] | x2[1] << (8 * c0), hi = x2[2] | x2[3] << (8 * c2);
    O.p2.a = (uint64_t)lo | (uint64_t)hi << (8 * (c0 + c1));
    O.p2.n = c0 + c1 + c2 + c3;
  }
}
// The same column with its four rows side by side in one register (a context byte and a sign byte per row): the rows' decision
// bytes are looked up row by row, everything after that -- the decision bits, which bytes belong to which pass and their order
// -- happens on all four rows at once: a pass's string is a byte compaction of [A0, B0, A1, B1, A2, B2, A3, B3] (A = zero coding /
// refinement decision, B = sign decision) by the pass's membership masks, done by v_perm with selectors from a table.
// cmp8[EA | EB << 4]: the two selector words that move the kept bytes (A_k where EA has bit k, B_k where EB has) to the front, zeros
// behind (the refinement pass, A bytes only, reads entry EA | 0 and keeps the first word).
T1HD uint64_t cm2_cmp8_entry(int m) {
  uint64_t sel = 0;
  int n = 0;
  for (int k = 0; k < 4; k++) {
    if ((m >> k) & 1) sel |= (uint64_t)(2 * k) << (8 * n++);
    if ((m >> (4 + k)) & 1) sel |= (uint64_t)(2 * k + 1) << (8 * n++);
  }
  for (; n < 8; n++) sel |= (uint64_t)0x0C << (8 * n);
  return sel;
}
T1HD int cm2_popc8(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)__popc(v);
#else
  return cm2_popc4(v) + cm2_popc4(v >> 4);
#endif
}
T1HD void cm2_emit3s(Cm2Out3& O, const uint8_t* zcx, const uint8_t* signlut, const uint64_t* cmp8, const Cm2Win& c) {
  const uint32_t w = c.w;
  const uint32_t val4 = (w >> CW_VAL) & 0xF;
  const uint32_t bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & val4, sig04 = (w >> CW_SIG0) & val4;
  const uint32_t mu4 = (w >> CW_MU) & 0xF, neg4 = (w >> CW_NEG) & 0xF;
  const uint32_t C4 = val4 & ~sig04 & ~M4;
  const uint32_t S14 = M4 & bit4, S24 = C4 & bit4;
  const uint32_t fref = c.AB[0] | c.AB[1] << 1 | c.AB[2] << 2;
  const uint32_t f0 = c.Q[0] | c.P[1] << 1 | c.P[2] << 2;
  const uint32_t f2 = fref | c.DL[0] | c.DB[1] << 1 | c.DB[2] << 2;
  const uint32_t fchi = c.X[0] | c.X[1] << 1 | c.X[2] << 2;
  const uint32_t G1 = (c.B1 & 0x1248) << 1, G2 = (c.D1 & 0x1248) << 1;
  // run-length mode (four insignificant rows without a significant neighbour): rare in the planes that cost the time, so its
  // header and row masks are only worked out when some lane of the wave has such a column
  const uint32_t rl = (val4 == 0xF && f2 == 0 && M4 == 0) ? ~0u : 0u;
  uint32_t EA2 = C4, EB2 = S24;
  O.hdr.a = 0; O.hdr.n = 0;
  if (t1_any_lane(rl != 0)) {
    const uint32_t lowest = bit4 & (0u - bit4);
    const uint32_t skipA4 = rl & (2 * lowest - 1) & 0xF, skipB4 = rl & (lowest - 1) & 0xF;
    const uint32_t fk = (lowest >> 1) - (lowest >> 3);
    const uint32_t hdr3 = (uint32_t)(T1_CTX_RUN | 32) | (uint32_t)(T1_CTX_UNI | ((fk >> 1) << 5)) << 8 | (uint32_t)(T1_CTX_UNI | ((fk & 1) << 5)) << 16;
    O.hdr.a = rl & (bit4 ? hdr3 : (uint32_t)T1_CTX_RUN);
    O.hdr.n = (int)(rl & (bit4 ? 3u : 1u));
    EA2 = C4 & ~skipA4; EB2 = S24 & ~skipB4;
  }
  uint32_t A4 = 0, B4 = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t mM = cm2_m(M4, k), m1 = cm2_m(sig04, k);
    const uint32_t cmask = (1u << (3 * k + 2)) - 1;
    const uint32_t fz = cm2_pick(m1, fref, cm2_pick(mM, f0 | (G1 & cmask), f2 | (G2 & cmask)));
    const uint32_t nb = (fz >> (3 * k)) & 0x1FF;
    const uint32_t li = (nb & 0x1EF) | (m1 & (0x200u | (cm2_m(mu4, k) & 0x10u)));
    const uint32_t idx = ((nb & 0xAA) >> 1) | ((fchi >> (3 * k)) & 0xAA);
    A4 |= (uint32_t)zcx[li] << (8 * k);
    B4 |= (uint32_t)signlut[idx] << (8 * k);
  }
  // the decision bits: bit 5 of every row's byte (0x00204081 puts bit k of a nibble at bit 8 k)
  A4 |= ((bit4 * 0x00204081u) & 0x01010101u) << 5;
  B4 ^= ((neg4 * 0x00204081u) & 0x01010101u) << 5;
  const uint32_t Ilo = cm2_perm(B4, A4, 0x05010400u), Ihi = cm2_perm(B4, A4, 0x07030602u);   // A0 B0 A1 B1 | A2 B2 A3 B3
  O.p1.a = cm2_perm(Ihi, Ilo, (uint32_t)cmp8[sig04]);
  O.p1.n = cm2_popc4(sig04);
  {
    const uint32_t m = M4 | S14 << 4;
    const uint64_t sel = cmp8[m];
    O.p0.a = (uint64_t)cm2_perm(Ihi, Ilo, (uint32_t)sel) | (uint64_t)cm2_perm(Ihi, Ilo, (uint32_t)(sel >> 32)) << 32;
    O.p0.n = cm2_popc8(m);
  }
  {
    const uint32_t m = EA2 | EB2 << 4;
    const uint64_t sel = cmp8[m];
    O.p2.a = (uint64_t)cm2_perm(Ihi, Ilo, (uint32_t)sel) | (uint64_t)cm2_perm(Ihi, Ilo, (uint32_t)(sel >> 32)) << 32;
    O.p2.n = cm2_popc8(m);
  }
}
// the distortion sums of one column (lossy layers only), by pass
T1HD void cm2_dist3(const Cm2Ctx& cx, uint32_t w, uint64_t mag, long long* dist) {
  const uint32_t val4 = (w >> CW_VAL) & 0xF, bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & val4, sig04 = (w >> CW_SIG0) & val4;
  const uint32_t S14 = M4 & bit4, S24 = val4 & ~sig04 & ~M4 & bit4;
  long long d0 = 0, d1 = 0, d2 = 0;
  for (int k = 0; k < 4; k++) {
    const int m = t1_mag(mag, k);
    if ((sig04 >> k) & 1) d1 += t1_dist_ref(m, cx.p);
    else if ((S14 >> k) & 1) d0 += t1_dist_sig(m, cx.p);
    else if ((S24 >> k) & 1) d2 += t1_dist_sig(m, cx.p);
  }
  dist[0] += d0; dist[1] += d1; dist[2] += d2;
}
// decisions per pass of the centre column from the raw words of its 3 x 3 neighbourhood (u/m/d = stripes s-1, s, s+1; index 0..2 =
// columns x-1, x, x+1): what cm2_counts derives from the gathered vectors
// *plain (optional): the column's only decision is RUN = 0 (four insignificant zeros, no significant neighbour)
T1HD void cm2_counts_w(const uint32_t u[3], const uint32_t m[3], const uint32_t d[3], int n[3], bool* plain = nullptr) {
  const uint32_t w = m[1];
  const uint32_t val4 = (w >> CW_VAL) & 0xF;
  const uint32_t bit4 = (w >> CW_BIT) & 0xF, M4 = (w >> CW_M) & val4, sig04 = (w >> CW_SIG0) & val4;
  const uint32_t C4 = val4 & ~sig04 & ~M4, S14 = M4 & bit4, S24 = C4 & bit4;
  auto pop = [](uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__popc(v);
#else
    return cm2_popc4(v);
#endif
  };
  n[0] = pop(M4) + pop(S14);
  n[1] = pop(sig04);
  // significant neighbours as the clean-up pass sees them at the column's first row (run-length test): sig0 | S1 everywhere,
  // S2 of the left column's rows -1..3 and of the row above in the centre and right columns
  const uint32_t ROW3 = 8u, ROW0 = 1u;  // within a 4-bit field
  const uint32_t SS = (0xFu << CW_SIG0) | (0xFu << CW_S1), S2F = 0xFu << CW_S2;
  const uint32_t up3 = (ROW3 << CW_SIG0) | (ROW3 << CW_S1) | (ROW3 << CW_S2), dn0 = (ROW0 << CW_SIG0) | (ROW0 << CW_S1);
  const uint32_t any = (u[0] & up3) | (m[0] & (SS | S2F)) | (d[0] & dn0) | (u[1] & up3) | (m[1] & SS) | (d[1] & dn0) | (u[2] & up3) | (m[2] & SS) | (d[2] & dn0);
  const uint32_t lowest = bit4 & (0u - bit4);
  const bool rl = val4 == 0xF && any == 0 && M4 == 0;
  const uint32_t skipA4 = rl ? ((2 * lowest - 1) & 0xF) : 0u, skipB4 = rl ? ((lowest - 1) & 0xF) : 0u;
  n[2] = pop(C4 & ~skipA4) + pop(S24 & ~skipB4) + (rl ? (bit4 ? 3 : 1) : 0);
  if (plain) *plain = rl && !bit4;
}

// ---------------------------------------------------------------------------------------------
// The same counts for ALL 16 stripe columns of one column x at once, on 64-bit masks over the rows (bit y; nibble s = the four
// rows of stripe s): V valid rows, Z significant before the plane, B the plane's bits, M significance-pass membership.  The
// neighbourhood test of the run-length mode needs T = Z | S1 and T2 = T | S2 of the columns x - 1 and x + 1 (zero outside the
// block), S1 = M & B, S2 = V & ~Z & ~M & B -- in the kernel a DPP shift of the wave away.  Results as nibble arrays: decisions
// of pass 0 (<= 8), 1 (<= 4), 2 (<= 11) and bit 0 of `act` = the column is listed by the sparse form (it has decisions and
// is not "plain").  Bit-parallel within the word (SWAR): about 2 operations per stripe column against cm2_counts_w's 80.
#define CM2_N1 0x1111111111111111ull
T1HD uint64_t cm2_nib_any(uint64_t v) { return (v | v >> 1 | v >> 2 | v >> 3) & CM2_N1; }
T1HD uint64_t cm2_nib_all(uint64_t v) { return (v & v >> 1 & v >> 2 & v >> 3) & CM2_N1; }
T1HD uint64_t cm2_nib_pop(uint64_t v) {
  v = (v & 0x5555555555555555ull) + ((v >> 1) & 0x5555555555555555ull);
  return (v & 0x3333333333333333ull) + ((v >> 2) & 0x3333333333333333ull);
}
struct Cm2Counts { uint64_t n0, n1, n2, act; };
T1HD void cm2_counts_swar(Cm2Counts& o, uint64_t V, uint64_t Z, uint64_t B, uint64_t M, uint64_t TL, uint64_t T2L, uint64_t TR, uint64_t T2R) {
  const uint64_t Zv = Z & V, Mv = M & V, C = V & ~Zv & ~Mv, S1 = Mv & B, S2 = C & B;
  const uint64_t T = Z | (M & B), T2 = T | (V & ~Z & ~M & B);   // the S1 / S2 fields as the column words hold them
  // a significant neighbour as the clean-up pass sees the column's first row: rows 0..3 of x - 1 (incl. S2), x, x + 1; the row above
  // in all three columns (incl. S2); the row below (before the clean-up pass)
  const uint64_t any = cm2_nib_any(T2L | T | TR) | (((T2L | T2 | T2R) >> 3) & CM2_N1) << 4 | ((TL | T | TR) & CM2_N1) >> 4;
  const uint64_t rl = cm2_nib_all(V) & ~any & ~cm2_nib_any(Mv);   // run-length mode (bit 0 of the nibble)
  const uint64_t RLm = (rl << 4) - rl;                               // ... as a full nibble
  const uint64_t LO = ((B << 1) & 0xEEEEEEEEEEEEEEEEull) | ((B << 2) & 0xCCCCCCCCCCCCCCCCull) | ((B << 3) & 0x8888888888888888ull);  // rows above the first 1
  const uint64_t skipA = ~LO & RLm, skipB = skipA & ~B;              // rows the run covers: no zero-coding decision / no sign
  const uint64_t hasbit = cm2_nib_any(B);
  o.n0 = cm2_nib_pop(Mv) + cm2_nib_pop(S1);
  o.n1 = cm2_nib_pop(Zv);
  o.n2 = cm2_nib_pop(C & ~skipA) + cm2_nib_pop(S2 & ~skipB) + rl + 2 * (rl & hasbit);
  o.act = cm2_nib_any(o.n0 | o.n1 | o.n2) & ~(rl & ~hasbit);
}
// the four nibble arrays as one 16-bit word per stripe -- n0 | n1 << 4 | n2 << 8 | act << 12 -- in the order CM2_PK_AT gives
#define CM2_PK_AT(s) (8 * ((s) >> 3) + 4 * ((s) & 1) + (((s) & 7) >> 1))   // halfword index of stripe s within the column's 32 bytes
T1HD void cm2_pack_pk(const Cm2Counts& c, uint32_t out[8]) {
  for (int h = 0; h < 2; h++) {
    const uint32_t A = (uint32_t)(c.n0 >> (32 * h)), B = (uint32_t)(c.n1 >> (32 * h)), C = (uint32_t)(c.n2 >> (32 * h)), D = (uint32_t)(c.act >> (32 * h));
    const uint32_t X = (A & 0x0F0F0F0Fu) | (B & 0x0F0F0F0Fu) << 4, Y = ((A >> 4) & 0x0F0F0F0Fu) | (B & 0xF0F0F0F0u);     // even / odd stripes: n0 | n1 << 4
    const uint32_t X2 = (C & 0x0F0F0F0Fu) | (D & 0x0F0F0F0Fu) << 4, Y2 = ((C >> 4) & 0x0F0F0F0Fu) | (D & 0xF0F0F0F0u);  // n2 | act << 4
    out[4 * h + 0] = cm2_perm(X2, X, 0x05010400u); out[4 * h + 1] = cm2_perm(X2, X, 0x07030602u);
    out[4 * h + 2] = cm2_perm(Y2, Y, 0x05010400u); out[4 * h + 3] = cm2_perm(Y2, Y, 0x07030602u);
  }
}

// The bit-planes of one column's 64 samples at once: m[2 s], m[2 s + 1] = the four 16-bit words (sign << 15 | magnitude) of stripe
// s's rows 0, 1 | 2, 3 as the Tier-1 layout packs them; P[p] = bit p of every sample (bit y = row y), so P[15] is the column's sign
// mask.  Four 16 x 16 bit-matrix transposes on halfword pairs (the classic masked block swaps, two rows per register) instead of
// extracting one plane's bits sample by sample for every plane: ~500 operations per block against ~190 per bit-plane.
T1HD void cm2_bitplanes(const uint32_t m[32], uint64_t P[16]) {
  uint32_t R[4][8];
  for (int g = 0; g < 4; g++) {
    uint32_t* r = R[g];
    for (int i = 0; i < 8; i++) {
      uint32_t x = m[8 * g + i];
      const uint32_t t = ((x >> 15) ^ x) & 0x0000AAAAu;   // rows 2 i / 2 i + 1 are the halves of one word: swap across the halves
      x ^= t; x ^= t << 15;
      r[i] = x;
    }
    for (int i = 0; i < 8; i += 2) { const uint32_t t = ((r[i] >> 2) ^ r[i + 1]) & 0x33333333u; r[i + 1] ^= t; r[i] ^= t << 2; }
    for (int i = 0; i < 8; i++) if (!(i & 2)) { const uint32_t t = ((r[i] >> 4) ^ r[i + 2]) & 0x0F0F0F0Fu; r[i + 2] ^= t; r[i] ^= t << 4; }
    for (int i = 0; i < 4; i++) { const uint32_t t = ((r[i] >> 8) ^ r[i + 4]) & 0x00FF00FFu; r[i + 4] ^= t; r[i] ^= t << 8; }
  }
  for (int p = 0; p < 16; p++) {   // plane p of group g: halfword p & 1 of R[g][p >> 1]
    const uint32_t sel = (p & 1) ? 0x07060302u : 0x05040100u;
    P[p] = (uint64_t)cm2_perm(R[1][p >> 1], R[0][p >> 1], sel) | (uint64_t)cm2_perm(R[3][p >> 1], R[2][p >> 1], sel) << 32;
  }
}

// The column words of eight stripes (8 h .. 8 h + 7) of one column from the 32-bit halves of its eight field masks, in the order of
// the fields in the word (SIG0, BIT, NEG, MU, VAL, M, S1, S2; nibble s of a half = stripe 8 h + s): an 8 x 8 nibble transpose --
// pairs of fields to bytes, then byte permutes -- instead of eight extractions and inserts per stripe.
T1HD void cm2_colwords8(const uint32_t f[8], uint32_t out[8]) {
  uint32_t X[4], Y[4];   // byte k of X[q]: fields 2 q, 2 q + 1 of stripe 2 k; of Y[q]: of stripe 2 k + 1
  for (int q = 0; q < 4; q++) {
    X[q] = (f[2 * q] & 0x0F0F0F0Fu) | (f[2 * q + 1] & 0x0F0F0F0Fu) << 4;
    Y[q] = ((f[2 * q] >> 4) & 0x0F0F0F0Fu) | (f[2 * q + 1] & 0xF0F0F0F0u);
  }
  for (int odd = 0; odd < 2; odd++) {
    const uint32_t* S = odd ? Y : X;
    const uint32_t lo01 = cm2_perm(S[1], S[0], 0x05010400u), hi01 = cm2_perm(S[3], S[2], 0x05010400u);   // low / high halfwords of stripes 0, 2 (+ odd)
    const uint32_t lo23 = cm2_perm(S[1], S[0], 0x07030602u), hi23 = cm2_perm(S[3], S[2], 0x07030602u);   // of stripes 4, 6 (+ odd)
    out[0 + odd] = cm2_perm(hi01, lo01, 0x05040100u); out[2 + odd] = cm2_perm(hi01, lo01, 0x07060302u);
    out[4 + odd] = cm2_perm(hi23, lo23, 0x05040100u); out[6 + odd] = cm2_perm(hi23, lo23, 0x07060302u);
  }
}

// The lookup tables of the context-modelling kernel, built once per context in device memory (k_t1_tables) and copied into LDS by
// every workgroup: byte offsets into the table block.
#define CM2_TAB_ZC 0        // [3][512] zero coding by neighbourhood, orientation classes LL/LH, HL, HH (t1_zc_from_nb)
#define CM2_TAB_REF 1536    // [512] refinement: bit 4 = "refined before", the other bits = significant neighbours
#define CM2_TAB_SIGN 2048   // [256] cm2_sign_entry
#define CM2_TAB_SP 2304     // [64] u32 CM2_SPREAD_INIT
#define CM2_TAB_CMP8 2560   // [256] u64 cm2_cmp8_entry
#define CM2_TAB_BYTES 4608

// A lane's position in one pass's byte stream: whole words are OR-ed into a zeroed buffer (the first and the last word of a
// lane's range are shared with its neighbours; a word may be written before it is complete and again when it is).
struct Cm2Stream {
  uint32_t* wp;    // the word the next byte falls into
  uint32_t pend;   // its bytes so far
  uint32_t fill;   // how many (0..3)
};
T1HD void cm2_or_word(uint32_t* p, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  atomicOr(p, v);
#else
  *p |= v;
#endif
}
T1HD void cm2_stream_begin(Cm2Stream& s, uint8_t* base /*4-byte aligned*/, int byte_off) {
  s.wp = (uint32_t*)base + (byte_off >> 2); s.pend = 0; s.fill = (uint32_t)(byte_off & 3);
}
T1HD void cm2_stream_put8(Cm2Stream& s, const Cm2Bytes& b) {  // up to 8 bytes
  const uint64_t sh = b.a << (8 * s.fill);
  const uint32_t lo = s.pend | (uint32_t)sh, mid = (uint32_t)(sh >> 32), hi = ((uint32_t)(b.a >> 32) >> 8) >> (24 - 8 * s.fill);
  cm2_or_word(s.wp, lo);
  cm2_or_word(s.wp + 1, mid);
  const uint32_t total = s.fill + (uint32_t)b.n, adv = total >> 2;
  s.pend = adv == 0 ? lo : (adv == 1 ? mid : hi);
  s.wp += adv; s.fill = total & 3;
}
T1HD void cm2_stream_put4(Cm2Stream& s, const Cm2Bytes& b) {  // up to 4 bytes
  const uint64_t sh = (uint64_t)(uint32_t)b.a << (8 * s.fill);
  const uint32_t lo = s.pend | (uint32_t)sh, mid = (uint32_t)(sh >> 32);
  cm2_or_word(s.wp, lo);
  const uint32_t total = s.fill + (uint32_t)b.n, adv = total >> 2;
  s.pend = adv ? mid : lo;
  s.wp += adv; s.fill = total & 3;
}
T1HD void cm2_stream_end(Cm2Stream& s) { if (s.fill) cm2_or_word(s.wp, s.pend); }

// The same for a lane that appends a whole contiguous range (its 16 columns' strings one after the other), without LDS atomics
// (measured: the kernel's largest single cost): a word leaves with a plain store once it is complete -- until then the store is
// pointed at a per-lane dummy word, which is cheaper than an exec-mask branch -- and only what is left at the end, the partial
// last word the next lane's range begins in, is OR-ed in.  A completed FIRST word may lack the bytes of the previous lane's tail:
// that lane ORs them in at its end, and since the wave runs its lanes in lockstep every plain store of the loop is older than
// every OR behind the loop (the LDS keeps a wave's accesses in order).
struct Cm2Seq {
  uint32_t* wp;     // the word the next byte falls into
  uint32_t pend;    // its bytes so far
  uint32_t fill;    // how many (0..3)
  uint32_t* trash;  // a word of this lane's own nobody reads
};
T1HD void cm2_seq_begin(Cm2Seq& s, uint8_t* base /*4-byte aligned*/, int byte_off, uint32_t* trash) {
  s.wp = (uint32_t*)base + (byte_off >> 2); s.pend = 0; s.fill = (uint32_t)(byte_off & 3); s.trash = trash;
}
T1HD void cm2_seq_put8(Cm2Seq& s, const Cm2Bytes& b) {  // up to 8 bytes (zeros behind them)
  const uint64_t sh = b.a << (8 * s.fill);
  const uint32_t lo = s.pend | (uint32_t)sh, mid = (uint32_t)(sh >> 32), hi = ((uint32_t)(b.a >> 32) >> 8) >> (24 - 8 * s.fill);
  const uint32_t total = s.fill + (uint32_t)b.n, adv = total >> 2;
  *(adv >= 1 ? s.wp : s.trash) = lo;
  *(adv >= 2 ? s.wp + 1 : s.trash) = mid;
  s.pend = adv == 0 ? lo : (adv == 1 ? mid : hi);
  s.wp += adv; s.fill = total & 3;
}
T1HD void cm2_seq_put4(Cm2Seq& s, const Cm2Bytes& b) {  // up to 4 bytes
  const uint64_t sh = (uint64_t)(uint32_t)b.a << (8 * s.fill);
  const uint32_t lo = s.pend | (uint32_t)sh, mid = (uint32_t)(sh >> 32);
  const uint32_t total = s.fill + (uint32_t)b.n, adv = total >> 2;
  *(adv ? s.wp : s.trash) = lo;
  s.pend = adv ? mid : lo;
  s.wp += adv; s.fill = total & 3;
}
T1HD void cm2_seq_skip(Cm2Seq& s, int n) {  // n bytes that others fill in later (they OR into the zeros left here)
  const uint32_t total = s.fill + (uint32_t)n, adv = total >> 2;
  *(adv ? s.wp : s.trash) = s.pend;
  s.pend = adv ? 0u : s.pend;
  s.wp += adv; s.fill = total & 3;
}
// (the forms with a branch around the store: where there is no dummy word to point it at)
T1HD void cm2_seq_put1_b(Cm2Seq& s, uint32_t byte) {
  const uint32_t lo = s.pend | byte << (8 * s.fill);
  const uint32_t total = s.fill + 1, adv = total >> 2;
  if (adv) *s.wp = lo;
  s.pend = adv ? 0u : lo;
  s.wp += adv; s.fill = total & 3;
}
T1HD void cm2_seq_skip_b(Cm2Seq& s, int n) {
  const uint32_t total = s.fill + (uint32_t)n, adv = total >> 2;
  if (adv && s.pend) *s.wp = s.pend;
  s.pend = adv ? 0u : s.pend;
  s.wp += adv; s.fill = total & 3;
}
T1HD void cm2_seq_end(Cm2Seq& s) { if (s.fill) cm2_or_word(s.wp, s.pend); }

template <class Sink>
T1HD void cm2_column(Sink& K, const Cm2Ctx& cx, int pass, const uint32_t u[3], const uint32_t m[3], const uint32_t d[3],
                     uint64_t mag, long long& dist) {
  if (!((m[1] >> CW_VAL) & 0xF)) return;
  Cm2Col c;
  cm2_gather(c, u, m, d);
  cm2_pass(K, cx, pass, c, mag, dist);
}

}  // namespace qsvc
