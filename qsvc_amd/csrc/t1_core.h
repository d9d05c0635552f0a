// qsvc_amd/csrc/t1_core.h -- EBCOT Tier-1 (ITU-T T.800 Annex C + D): the MQ coder and what the Tier-1 kernels share.
//
//   * the MQ coder (Annex C) of one code-block per LANE: T1Lane (registers A, C, CT, the byte staging pointer and the 19 context
//     states as packed words), the straight-line 16-decision step mq2_chunk16 and its two-wave split (front: contexts + interval,
//     back: code register + bytes; mq2_front16 / mq2_back16 with the exact replay mq2_back16_exact), termination (t1_lane_finish);
//   * the decision format between the context-modelling kernel (t1_cm2.h) and the MQ kernels: one byte per decision
//     (cx | d << 5, T1_PAD = the idle context), per (block, bit-plane) a segment of T1_SEG_BYTES with the three passes at +0,
//     +T1_SEG1, +T1_SEG2;
//   * the Tier-1 magnitude words the DWT kernel writes: per stripe column a u64 = 4 x (sign << 15 | 15-bit magnitude) (t1_mag,
//     t1_neg), the sign-coding rule (t1_code_sign) and the distortion estimates of a coding pass (t1_dist_sig / t1_dist_ref).
// The reference has no source for this stage -- it calls Kakadu (texture_compress_fb_j2k.py:183-195); the algorithm is the published
// one, restated independently in oracle/j2k_oracle.c.  The same functions compile for the host (tests/emul) with lane stride 1; the
// superseded lane-per-block forms with flag words in memory live in tests/emul/t1_legacy.h as a second implementation for the tests.
#pragma once
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
// "this value has landed": forces the wait for a pending load to sit HERE, before younger loads are issued.
// hipcc cannot count the divergent stores between a load and its use, so it falls back to vmcnt(0) at the
// first use -- which would also wait for every prefetch issued in between (cdna guide 5, "mixing load kinds").
#define T1_PIN32(x) asm volatile("" : "+v"(x)::"memory")
#else
#define T1_PIN32(x) ((void)0)
#endif
#if defined(__HIPCC__)
#define T1HD __host__ __device__ __forceinline__
#define T1HD_M __host__ __device__ __forceinline__
#else
#define T1HD static inline
#define T1HD_M inline
#endif

namespace qsvc {

#define T1_FW 66            // flag words per stripe row (64 + apron)
#define T1_MAXPASS 48       // 3*16-2 coding passes at most
#define T1_SIG_ALL 0x3FFFFu
#define T1_CHI_SHIFT 18
#define T1_PI_SHIFT 24
#define T1_MU_SHIFT 28
#define T1_CTX_RUN 17
#define T1_CTX_UNI 18

// MQ probability state table (T.800 Table C.2), packed: qe | nmps<<16 | nlps<<22 | switch<<28
#define T1_MQ_TABLE_INIT                                                                                                  \
  {0x5601 | 1 << 16 | 1 << 22 | 1 << 28,   0x3401 | 2 << 16 | 6 << 22,    0x1801 | 3 << 16 | 9 << 22,    0x0AC1 | 4 << 16 | 12 << 22,  \
   0x0521 | 5 << 16 | 29 << 22,            0x0221 | 38 << 16 | 33 << 22,  0x5601 | 7 << 16 | 6 << 22 | 1 << 28, 0x5401 | 8 << 16 | 14 << 22, \
   0x4801 | 9 << 16 | 14 << 22,            0x3801 | 10 << 16 | 14 << 22,  0x3001 | 11 << 16 | 17 << 22,  0x2401 | 12 << 16 | 18 << 22, \
   0x1C01 | 13 << 16 | 20 << 22,           0x1601 | 29 << 16 | 21 << 22,  0x5601 | 15 << 16 | 14 << 22 | 1 << 28, 0x5401 | 16 << 16 | 14 << 22, \
   0x5101 | 17 << 16 | 15 << 22,           0x4801 | 18 << 16 | 16 << 22,  0x3801 | 19 << 16 | 17 << 22,  0x3401 | 20 << 16 | 18 << 22, \
   0x3001 | 21 << 16 | 19 << 22,           0x2801 | 22 << 16 | 19 << 22,  0x2401 | 23 << 16 | 20 << 22,  0x2201 | 24 << 16 | 21 << 22, \
   0x1C01 | 25 << 16 | 22 << 22,           0x1801 | 26 << 16 | 23 << 22,  0x1601 | 27 << 16 | 24 << 22,  0x1401 | 28 << 16 | 25 << 22, \
   0x1201 | 29 << 16 | 26 << 22,           0x1101 | 30 << 16 | 27 << 22,  0x0AC1 | 31 << 16 | 28 << 22,  0x09C1 | 32 << 16 | 29 << 22, \
   0x08A1 | 33 << 16 | 30 << 22,           0x0521 | 34 << 16 | 31 << 22,  0x0441 | 35 << 16 | 32 << 22,  0x02A1 | 36 << 16 | 33 << 22, \
   0x0221 | 37 << 16 | 34 << 22,           0x0141 | 38 << 16 | 35 << 22,  0x0111 | 39 << 16 | 36 << 22,  0x0085 | 40 << 16 | 37 << 22, \
   0x0049 | 41 << 16 | 38 << 22,           0x0025 | 42 << 16 | 39 << 22,  0x0015 | 43 << 16 | 40 << 22,  0x0009 | 44 << 16 | 41 << 22, \
   0x0005 | 45 << 16 | 42 << 22,           0x0001 | 45 << 16 | 43 << 22,  0x5601 | 46 << 16 | 46 << 22}

// zero-coding context (T.800 Table D.1) from the 9-bit neighbourhood (bit 3*dr+dc, centre = bit 4)
T1HD int t1_zc_from_nb(int nb, int orient) {
  int h = ((nb >> 3) & 1) + ((nb >> 5) & 1);
  int v = ((nb >> 1) & 1) + ((nb >> 7) & 1);
  int d = (nb & 1) + ((nb >> 2) & 1) + ((nb >> 6) & 1) + ((nb >> 8) & 1);
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

struct T1Result {  // per code-block outputs besides the bytes
  int len;
  int npasses;
  int nbps;
  int overflow;         // the segment did not fit the lane's output capacity
  int* pass_len;        // [T1_MAXPASS] cumulative truncation length after each pass
  long long* pass_dist; // [T1_MAXPASS] integer distortion decrease per pass, quantiser-step^2 units (or nullptr)
  unsigned long long nsym;
};

// One lane's MQ coder + block geometry.  Block state (flag words, magnitudes) is reached through a
// `Store` policy so the same code runs over plain arrays on the host and over the LDS window on the GPU:
//   uint32_t ld(int r, int c)            flag word of stripe-row r (= stripe+1), column c (= x+1)
//   void     st(int r, int c, uint32_t)  store it
//   void     orr(int r, int c, uint32_t) OR bits into it
//   uint64_t mag(int s, int x)           4 x (sign<<15 | magnitude) of stripe s, column x
//   void pass_begin(int nstripes) / stripe_begin(int s, int nstripes) / stripe_end(int s) / pass_end(int nstripes)
//        wave-uniform window management (no-ops on the host)
struct T1Lane {
  uint32_t* cx;           // this lane's context states (19; 20 in the second coder form), one packed word each: the state's table entry
                          // (qe | nmps<<16 | nlps<<22 | switch<<28) | mps<<29; context c at cx[c*cs]
  const uint32_t* mqtab;  // [47]
  const uint64_t* mq2;    // [94] successor pairs of the second coder form (mq2_*), or nullptr
  const uint8_t* zclut;   // [3][512] zero-coding LUT (class 0: LL/LH, 1: HL, 2: HH)
  uint8_t* out;           // 4-byte aligned; out[0] is a scratch byte; the segment is out[1..]
  int cs;
  int cap;                // bytes available at out (incl. the scratch byte), multiple of 4
  int w, h, orient, nbps;
  // MQ registers
  uint32_t A, C;
  int CT, B, n;           // B: byte register (not yet in memory), n: bytes emitted incl. the scratch byte
  uint32_t ob;            // emitted bytes not yet stored (stores go out one aligned u32 at a time)
  uint8_t* stg;           // second coder form: this lane's staging area (T1_STAGE bytes, 4-byte aligned), cursor, bytes already in out
  uint8_t* sp;
  int nfl;
  uint32_t nsym;
};

T1HD void mq_emit(T1Lane& t) {
  t.ob |= (uint32_t)t.B << (8 * (t.n & 3));
  t.n++;
  if ((t.n & 3) == 0) {
    if (t.n <= t.cap) *(uint32_t*)(t.out + t.n - 4) = t.ob;
    t.ob = 0;
  }
}
// BYTEOUT (T.800 C.2.7) with the byte register held in t.B (always < 256), written with selects instead of the
// flow chart's nested branches: on a GPU every branch of a lane is paid by the whole wave.
T1HD void mq_byteout(T1Lane& t) {
  uint32_t carry = (t.C >> 27) & (uint32_t)(t.B != 0xFF);  // a carry never enters a byte that is 0xFF (bit stuffing)
  t.B += (int)carry;
  const bool ff = t.B == 0xFF;                               // was 0xFF, or became 0xFF by the carry
  if (carry && ff) t.C &= 0x7FFFFFF;
  mq_emit(t);
  t.B = ff ? (int)((t.C >> 20) & 0xFF) : (int)((t.C >> 19) & 0xFF);
  t.C &= ff ? 0xFFFFFu : 0x7FFFFu;
  t.CT = ff ? 7 : 8;
}
T1HD int t1_clz32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)v);
#else
  return __builtin_clz(v);
#endif
}
// ENCODE / CODEMPS / CODELPS / RENORME (T.800 C.2.2-C.2.6), select form:
//   MPS, A still >= 0x8000 ....... C += Qe, nothing else
//   MPS, A < 0x8000 .............. conditional exchange (A < Qe: A = Qe, else C += Qe), next = NMPS, renormalise
//   LPS .......................... conditional exchange (A < Qe: C += Qe, else A = Qe), MPS flips on SWITCH,
//                                  next = NLPS, renormalise
// `e` is the context's packed state, already fetched by the caller (so the fetch for the NEXT decision can be in
// flight while this one is coded); both successor entries are fetched up front, off the dependent chain.
// Returns the context's new packed state (also stored).
T1HD uint32_t mq_encode_e(T1Lane& t, int cx, int d, uint32_t e) {
  const uint32_t tm = t.mqtab[(e >> 16) & 63], tl = t.mqtab[(e >> 22) & 63];
  const uint32_t qe = e & 0xFFFF;
  uint32_t mps = (e >> 29) & 1;
  t.nsym++;
  uint32_t A = t.A - qe;
  const bool ismps = (uint32_t)d == mps;
  const bool small = (A & 0x8000) == 0;
  const bool lt = A < qe;
  const bool ren = !ismps || small;
  const bool addc = ismps ? !(small && lt) : lt;
  const bool seta = ismps ? (small && lt) : !lt;
  t.C += addc ? qe : 0;
  A = seta ? qe : A;
  if (ren) {
    mps ^= ismps ? 0u : (e >> 28) & 1;
    e = (ismps ? tm : tl) | (mps << 29);
    t.cx[cx * t.cs] = e;
    int sh = t1_clz32(A) - 16;  // A is in [1, 0x7FFF]: bring it back to bit 15 in one step
    A <<= sh;
    while (sh >= t.CT) { t.C <<= t.CT; sh -= t.CT; mq_byteout(t); }  // a byte leaves whenever CT bits have gone
    t.C <<= sh; t.CT -= sh;
  }
  t.A = A;
  return e;
}
T1HD void mq_encode(T1Lane& t, int cx, int d) { (void)mq_encode_e(t, cx, d, t.cx[cx * t.cs]); }
T1HD int mq_flush(T1Lane& t) {  // FLUSH, T.800 C.2.9; returns the segment length
  uint32_t tempc = t.C + t.A;
  t.C |= 0xFFFF;
  if (t.C >= tempc) t.C -= 0x8000;
  t.C <<= t.CT; mq_byteout(t);
  t.C <<= t.CT; mq_byteout(t);
  if (t.B != 0xFF) mq_emit(t);
  if ((t.n & 3) && (t.n & ~3) + 4 <= t.cap) *(uint32_t*)(t.out + (t.n & ~3)) = t.ob;  // the partial last word
  return t.n - 1;
}

// ---------------------------------------------------------------------------------------------
// Second coder form (the split MQ kernel): straight-line, 16 decisions per call, no branch on the common path.
// A context word describes itself: qe | 8*(2*state + mps) << 16 | mps << 31.  mq2[2*state + mps] holds the two possible successors
// (low half: after an MPS renormalisation; high half: after an LPS, MPS flipped where SWITCH is set), so a decision needs
// one context read and one table read, both issued ahead of their use: the context of decision j+1 is fetched while
// decision j is coded and replaced by j's result when both use the same context.
// With addc = "C += Qe", seta = "A = Qe" (T.800 C.2.4/C.2.5): seta == (ismps == (A - Qe < Qe)) and addc == !seta, because
// A - Qe < Qe implies A - Qe < 0x8000.
// Context T1_CTX_PAD (word 0: Qe = 0, MPS 0) codes nothing: decision byte T1_PAD fills the tails of the 16-byte chunks and
// the chunks of idle lanes, so no decision needs a validity test.
// Bytes leave through a per-lane staging area (t.stg, LDS on the GPU): BYTEOUT (C.2.7) is computed for every decision with
// selects, its byte is written to the staging slot unconditionally and the slot only advances when the byte was due (a slot
// is rewritten until then).  A lane's staged bytes go to its output buffer 16 at a time (mq2_stage_flush), checked once per
// chunk: the measured cost of the branchy byte-per-store form was 43 % of the kernel (in a 16-lane wave 85 % of the decisions
// see some lane emit a byte, and on gfx9 stores share vmcnt with the prefetched decision loads).
#define T1_CTX_PAD 19
#define T1_PAD 19u                   // cx = T1_CTX_PAD, d = 0
#define T1_PAD4 0x13131313u
#define T1_STAGE 196                 // bytes per lane (49 words: odd stride, conflict-free across lanes)
#define T1_STAGE_FLUSH 144           // flush once a lane holds this many: + 16 decisions x 3 bytes + FLUSH's 3 stays below T1_STAGE - 1
#define T1_STAGE_SMALL 116           // the same with a third of the LDS (29 words), for kernels that share a CU with others
#define T1_STAGE_SMALL_FLUSH 48
T1HD uint32_t mq2_word(const uint32_t* mqtab, int state, int mps) {
  return (mqtab[state] & 0xFFFF) | (uint32_t)(8 * (2 * state + mps)) << 16 | (uint32_t)mps << 31;
}
T1HD uint64_t mq2_pair(const uint32_t* mqtab, int idx) {  // idx = 2*state + mps
  const uint32_t e = mqtab[idx >> 1];
  const int mps = idx & 1, sw = (int)((e >> 28) & 1);
  return (uint64_t)mq2_word(mqtab, (int)((e >> 16) & 63), mps) | (uint64_t)mq2_word(mqtab, (int)((e >> 22) & 63), mps ^ sw) << 32;
}
T1HD void mq2_lane_init(T1Lane& t) {  // as t1_lane_init, context words in the second form; t.stg must be set
  for (int c = 0; c < 19; c++) t.cx[c * t.cs] = mq2_word(t.mqtab, 0, 0);
  t.cx[0 * t.cs] = mq2_word(t.mqtab, 4, 0); t.cx[T1_CTX_RUN * t.cs] = mq2_word(t.mqtab, 3, 0);
  t.cx[T1_CTX_UNI * t.cs] = mq2_word(t.mqtab, 46, 0);
  t.cx[T1_CTX_PAD * t.cs] = 0;
  t.A = 0x8000; t.C = 0; t.CT = 12; t.B = 0; t.n = 0; t.ob = 0; t.nsym = 0;
  t.sp = t.stg; t.nfl = 0;
}
T1HD int mq2_clz(uint32_t v) { return __builtin_clz(v); }  // v != 0
T1HD int mq2_count(const T1Lane& t) { return t.nfl + (int)(t.sp - t.stg); }  // bytes emitted so far, incl. the scratch byte
// whole 16-byte groups of the staged bytes -> out (skipped, but counted, past the capacity: -> overflow); the rest moves to the front
T1HD void mq2_stage_copy(T1Lane& t, int nbytes) {  // nbytes: multiple of 16, or the last bytes of the segment rounded up
  for (int i = 0; i < nbytes; i += 16) {
    const uint32_t* s = (const uint32_t*)(t.stg + i);
    const uint32_t a = s[0], b = s[1], c = s[2], d = s[3];
    if (t.nfl + i + 16 <= t.cap) {
#if defined(__HIP_DEVICE_COMPILE__)
      *(uint4*)(t.out + t.nfl + i) = make_uint4(a, b, c, d);
#else
      const uint32_t v[4] = {a, b, c, d};
      memcpy(t.out + t.nfl + i, v, 16);
#endif
    }
  }
}
T1HD void mq2_stage_flush(T1Lane& t) {
  const int sn = (int)(t.sp - t.stg), k = sn & ~15;
  if (k == 0) return;
  mq2_stage_copy(t, k);
  const uint32_t* s = (const uint32_t*)(t.stg + k);
  const uint32_t a = s[0], b = s[1], c = s[2], d = s[3];
  uint32_t* f = (uint32_t*)t.stg;
  f[0] = a; f[1] = b; f[2] = c; f[3] = d;
  t.nfl += k; t.sp = t.stg + (sn - k);
}
T1HD void mq2_emit(T1Lane& t) { *t.sp++ = (uint8_t)t.B; }
T1HD void mq2_byteout(T1Lane& t) {  // the branchy form: FLUSH and the rare second / third byte of one renormalisation
  uint32_t carry = (t.C >> 27) & (uint32_t)(t.B != 0xFF);
  t.B += (int)carry;
  const bool ff = t.B == 0xFF;
  if (carry && ff) t.C &= 0x7FFFFFF;
  mq2_emit(t);
  t.B = ff ? (int)((t.C >> 20) & 0xFF) : (int)((t.C >> 19) & 0xFF);
  t.C &= ff ? 0xFFFFFu : 0x7FFFFu;
  t.CT = ff ? 7 : 8;
}
T1HD int mq2_flush(T1Lane& t) {  // FLUSH, T.800 C.2.9, then the staged bytes; returns the segment length
  mq2_stage_flush(t);
  uint32_t tempc = t.C + t.A;
  t.C |= 0xFFFF;
  if (t.C >= tempc) t.C -= 0x8000;
  t.C <<= t.CT; mq2_byteout(t);
  t.C <<= t.CT; mq2_byteout(t);
  if (t.B != 0xFF) mq2_emit(t);
  mq2_stage_flush(t);
  const int sn = (int)(t.sp - t.stg);  // < 16
  mq2_stage_copy(t, (sn + 15) & ~15);
  t.n = t.nfl + sn;
  return t.n - 1;
}
// decisions: bytes cx | d << 5, little-endian in w0..w3, T1_PAD where there is none
// Selects are written as bit-field inserts under sign-extended masks (v_bfi_b32): a compare + v_cndmask pair pays wait states
// for the SGPR it goes through, and a single wave issues one instruction per ~4.3 cycles whatever its kind (tools/ubench.hip).
#if defined(__HIP_DEVICE_COMPILE__)
T1HD bool t1_any_lane(bool x) {  // wave-uniform; opaque, so that the test stays ONE scalar branch in front of the rare path
  unsigned long long m = __builtin_amdgcn_ballot_w64(x);
  asm volatile("" : "+s"(m));
  return m != 0;
}
#else
T1HD bool t1_any_lane(bool x) { return x; }
#endif
T1HD uint32_t mq2_sel(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }  // mask ? a : b
T1HD uint32_t mq2_neg_mask(int32_t v) {  // v < 0 ? ~0 : 0, opaque to the optimiser (which would turn the inserts back into compare + select)
  uint32_t m = (uint32_t)(v >> 31);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(m));
#endif
  return m;
}
T1HD void mq2_chunk16(T1Lane& t, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  const uint32_t w[4] = {w0, w1, w2, w3};
  uint32_t A = t.A;
  uint32_t e = t.cx[(int)(w0 & 31) * t.cs];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t wj = w[j >> 2];
    const int cx = (int)((wj >> (8 * (j & 3))) & 31);
    uint32_t epre = 0;
    int cx1 = -1;
    if (j < 15) { cx1 = (int)((w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 31); epre = t.cx[cx1 * t.cs]; }
    const uint64_t succ = *(const uint64_t*)((const char*)t.mq2 + ((e >> 16) & 0x7FFF));
    const uint32_t qe = e & 0xFFFF;
    // sign bit of x: the decision is the LPS (d at bit 31 against the MPS at bit 31 of the context word)
    const uint32_t x = (wj << (26 - 8 * (j & 3))) ^ e;
    A -= qe;
    // sign bit of (A - qe): A < qe (both below 2^17); seta = ismps == lt = lps != lt
    const uint32_t seta = mq2_neg_mask((int32_t)((A - qe) ^ x));
    const uint32_t Cin = t.C + (qe & ~seta);
    A = mq2_sel(seta, qe, A);
    const int sh = mq2_clz(A) - 16;  // 0 unless renormalising: A is in [0x8000, 0xFFFF] otherwise
    const uint32_t noren = mq2_neg_mask((int32_t)(A << 16));  // A >= 0x8000.  Every LPS renormalises (A = Qe, or A - Qe < Qe)
    A <<= sh;
    // RENORME's shifts and at most one BYTEOUT, with selects.  d = CT - sh <= 0: a byte is due after CT of the sh shifts.
    const int d = t.CT - sh;
    const uint32_t take = mq2_neg_mask(d - 1);
    const int s1 = sh < t.CT ? sh : t.CT;
    uint32_t C = Cin << s1;
    const uint32_t bn = (uint32_t)t.B + (C >> 27);   // B + carry; 0x100 only for B == 0xFF, whose carry stays in C (bit stuffing)
    const uint32_t em = bn < 255u ? bn : 255u;       // the byte that leaves
    *t.sp = (uint8_t)em;                             // rewritten until `take`
    C += ((uint32_t)t.B - em) << 27;                 // the carry that went into the byte leaves C
    const int bo = (int)((bn + 1 + 19 * 256) >> 8);  // 20 when the byte is 0xFF (the next one takes 7 bits and the carry position), else 19
    const uint32_t Bn = (C >> bo) & 0xFFu, Cm = C & ((1u << bo) - 1u);
    t.B = (int)mq2_sel(take, Bn, (uint32_t)t.B);
    C = mq2_sel(take, Cm, C);
    t.sp -= (int32_t)take;
    const int rest = sh - s1;                        // shifts still to do (0 unless a byte left)
    const int ctn = d + (int)((uint32_t)(27 - bo) & take);  // <= 0: another byte is due within `rest` (rare: sh up to 15)
    t.C = C << rest;
    t.CT = ctn;
    // the successor is only needed now: its table read had the whole decision to arrive
    const uint32_t lps = mq2_neg_mask((int32_t)x);
    const uint32_t en = mq2_sel(noren, e, mq2_sel(lps, (uint32_t)(succ >> 32), (uint32_t)succ));
    t.cx[cx * t.cs] = en;
    e = cx1 == cx ? en : epre;
    if (__builtin_expect(t1_any_lane(ctn <= 0), 0)) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("; rare: more than one byte in one renormalisation");  // keeps the uniform test a branch of its own
#endif
      if (ctn <= 0) {  // the second (third) byte of one renormalisation, with the flow chart's loop
        int ct = 27 - bo, r = rest;
        t.C = C;
        do { t.C <<= ct; r -= ct; mq2_byteout(t); ct = t.CT; } while (r >= ct);
        t.CT = ct - r;
        t.C <<= r;
      }
    }
  }
  t.A = A;
}

// The same decision cut in two for a pair of waves (k_t1_mq_pipe): the FRONT keeps the contexts and the interval register A and
// leaves one record per decision -- what is added to the code register (Qe or 0) | the renormalisation shift << 16 -- and the
// BACK keeps the code register, CT, the byte register and the output.  Nothing flows from the back to the front (T.800 C.2:
// A and the context states never depend on C), so the two halves run as a pipeline, each about half a decision long.
// rec: this lane's 16 record slots of the chunk, rs words apart.
T1HD void mq2_front16(T1Lane& t, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t* rec, int rs) {
  const uint32_t w[4] = {w0, w1, w2, w3};
  uint32_t A = t.A;
  uint32_t e = t.cx[(int)(w0 & 31) * t.cs];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t wj = w[j >> 2];
    const int cx = (int)((wj >> (8 * (j & 3))) & 31);
    uint32_t epre = 0;
    int cx1 = -1;
    if (j < 15) { cx1 = (int)((w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 31); epre = t.cx[cx1 * t.cs]; }
    const uint64_t succ = *(const uint64_t*)((const char*)t.mq2 + ((e >> 16) & 0x7FFF));
    const uint32_t qe = e & 0xFFFF;
    const uint32_t x = (wj << (26 - 8 * (j & 3))) ^ e;
    A -= qe;
    const uint32_t seta = mq2_neg_mask((int32_t)((A - qe) ^ x));
    const uint32_t add = qe & ~seta;
    A = mq2_sel(seta, qe, A);
    const int sh = mq2_clz(A) - 16;
    const uint32_t noren = mq2_neg_mask((int32_t)(A << 16));
    A <<= sh;
    rec[j * rs] = add | (uint32_t)sh << 16;
    const uint32_t lps = mq2_neg_mask((int32_t)x);
    const uint32_t en = mq2_sel(noren, e, mq2_sel(lps, (uint32_t)(succ >> 32), (uint32_t)succ));
    t.cx[cx * t.cs] = en;
    e = cx1 == cx ? en : epre;
  }
  t.A = A;
}
// The back half runs its 16 decisions WITHOUT looking for the rare second byte of one renormalisation (sh up to 15 against at most
// one BYTEOUT per decision here): it only remembers whether that happened (the smallest CT seen, <= 0), and the caller then
// restores the lane's four registers and codes the chunk again with the flow chart's loop (mq2_back16_exact) -- the back keeps
// no other state, and the records are still in the ring.  Measured on config 3: 4 decisions in a million.
T1HD int mq2_back16(T1Lane& t, const uint32_t* rec, int rs) {
  uint32_t r[16];
#pragma unroll
  for (int j = 0; j < 16; j++) r[j] = rec[j * rs];
  int lowest = 1;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t Cin = t.C + (r[j] & 0xFFFF);
    const int sh = (int)(r[j] >> 16);
    const int d = t.CT - sh;
    const uint32_t take = mq2_neg_mask(d - 1);
    const int s1 = sh < t.CT ? sh : t.CT;
    uint32_t C = Cin << s1;
    const uint32_t bn = (uint32_t)t.B + (C >> 27);
    const uint32_t em = bn < 255u ? bn : 255u;
    *t.sp = (uint8_t)em;
    C += ((uint32_t)t.B - em) << 27;
    const int bo = (int)((bn + 1 + 19 * 256) >> 8);
    const uint32_t Bn = (C >> bo) & 0xFFu, Cm = C & ((1u << bo) - 1u);
    t.B = (int)mq2_sel(take, Bn, (uint32_t)t.B);
    C = mq2_sel(take, Cm, C);
    t.sp -= (int32_t)take;
    const int rest = (sh - s1) & 15;   // (& 15: a lane that has already gone wrong must still shift by less than 32)
    const int ctn = d + (int)((uint32_t)(27 - bo) & take);
    t.C = C << rest;
    t.CT = ctn;
    lowest = ctn < lowest ? ctn : lowest;
  }
  return lowest;
}
T1HD void mq2_back16_exact(T1Lane& t, const uint32_t* rec, int rs) {  // RENORME (T.800 C.2.6) bit count by bit count
  for (int j = 0; j < 16; j++) {
    const uint32_t r = rec[j * rs];
    t.C += r & 0xFFFF;
    int sh = (int)(r >> 16);
    while (sh >= t.CT) { t.C <<= t.CT; sh -= t.CT; mq2_byteout(t); }
    t.C <<= sh; t.CT -= sh;
  }
}
// one chunk of the back half, exact in every case
T1HD void mq2_back_chunk(T1Lane& t, const uint32_t* rec, int rs) {
  const uint32_t C0 = t.C; const int CT0 = t.CT, B0 = t.B; uint8_t* const sp0 = t.sp;
  const int lowest = mq2_back16(t, rec, rs);
  if (__builtin_expect(t1_any_lane(lowest <= 0), 0)) {
    if (lowest <= 0) { t.C = C0; t.CT = CT0; t.B = B0; t.sp = sp0; mq2_back16_exact(t, rec, rs); }
  }
}

T1HD int t1_mag(uint64_t m, int k) { return (int)((m >> (16 * k)) & 0x7FFF); }
T1HD int t1_neg(uint64_t m, int k) { return (int)((m >> (16 * k + 15)) & 1); }

// sign coding (T.800 D.3.2, Tables D.2/D.3).  `K.sym(cx, d)` consumes one binary decision.
// l / rr: the flag words of the left / right neighbour columns (already in registers: the previous column's
// final word and the prefetched next one), own: this column's word.
template <class Sink>
T1HD void t1_code_sign(Sink& K, uint32_t l, uint32_t rr, uint32_t own, int k, int neg) {
  int nb = (own >> (3 * k)) & 0x1FF;
  int hc = 0, vc = 0;
  if (nb & (1 << 3)) hc += ((l >> (T1_CHI_SHIFT + k + 1)) & 1) ? -1 : 1;
  if (nb & (1 << 5)) hc += ((rr >> (T1_CHI_SHIFT + k + 1)) & 1) ? -1 : 1;
  if (nb & (1 << 1)) vc += ((own >> (T1_CHI_SHIFT + k)) & 1) ? -1 : 1;
  if (nb & (1 << 7)) vc += ((own >> (T1_CHI_SHIFT + k + 2)) & 1) ? -1 : 1;
  hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
  vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
  int xr = hc < 0 || (hc == 0 && vc < 0);
  if (xr) { hc = -hc; vc = -vc; }  // after the flip: hc in {0,1}; hc == 0 -> vc in {0,1}
  int cx = hc == 1 ? 12 + vc : (vc == 1 ? 10 : 9);
  K.sym(cx, neg ^ xr);
}
T1HD long long t1_dist_sig(int m, int p) {  // distortion decrease, midpoint reconstruction
  long long r = (long long)((m >> p) << p) + (p ? (1 << (p - 1)) : 0);
  return r * (2LL * m - r);
}
T1HD long long t1_dist_ref(int m, int p) {
  long long r0 = (long long)((m >> (p + 1)) << (p + 1)) + (1 << p);
  long long r1 = (long long)((m >> p) << p) + (p ? (1 << (p - 1)) : 0);
  long long d0 = m - r0, d1 = m - r1;
  return d0 * d0 - d1 * d1;
}


template <bool MQ2 = false>
T1HD void t1_lane_finish(T1Lane& t, int np, T1Result& res) {
  res.npasses = np;
  res.nbps = t.nbps;
  res.len = t.nbps ? (MQ2 ? mq2_flush(t) : mq_flush(t)) : 0;
  res.overflow = ((t.n + 3) & ~3) > t.cap;
  res.nsym = t.nsym;
  for (int i = 0; i < np; i++) if (res.pass_len[i] > res.len) res.pass_len[i] = res.len;
  if (np) res.pass_len[np - 1] = res.len;
}


// ---------------------------------------------------------------------------------------------
// Split form: bit-plane-parallel context modelling + a separate MQ kernel.
// The significance state at the START of bit-plane p is a pure function of the magnitudes
// (sigma = mag >> (p+1) != 0, "refined before" = mag >> (p+2) != 0), so every (block, bit-plane) pair
// can be modelled independently; only the MQ coder is serial over the block's whole decision stream.
// A decision is one byte: cx | d << 5.  Segments per (block, plane): pass 0 at +0, pass 1 at +T1_SEG1,
// pass 2 at +T1_SEG2; T1_SEG_BYTES in all (worst cases: 2*4096, 4096, 10*1024).
#define T1_SEG1 8192
#define T1_SEG2 12288
#define T1_SEG_BYTES 22528



}  // namespace qsvc
