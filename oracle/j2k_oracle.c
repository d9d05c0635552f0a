/*
 * oracle/j2k_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product path.
 *
 * PARITY UNPINNED versus the reference: the reference delegates this half of `mctf compress` to the
 * proprietary, un-vendored, version-unpinned Kakadu `kdu_compress` binary
 * (trunk/src/texture_compress_fb_j2k.py:183-195); no source, binary, code-stream or size of it exists
 * under /root/reference.  This file is therefore a plain-C restatement of the PUBLISHED algorithm
 * kdu_compress implements -- ITU-T T.800 (JPEG 2000 Part 1): Annex A/B (code-stream syntax, packets,
 * tag trees), C (MQ arithmetic coder), D (coefficient bit modelling / EBCOT Tier-1), E (quantisation),
 * F (5/3 reversible and 9/7 irreversible DWT) -- anchored on the reference's call site and the
 * parameters it pins there: single component, Sprecision=8 Ssigned=no, Clevels=SRLs-1, Clayers=nLayers,
 * Cuse_sop=yes, one tile, 64x64 code-blocks, LRCP, Creversible=no (9/7) with reversible 5/3 as the
 * build's extension.  What pins it instead of Kakadu: (1) every code-stream it writes is decoded by an
 * independent JPEG 2000 decoder (OpenJPEG 2.5 via Pillow) in tests/test_j2k_oracle.py -- bit-exact
 * reconstruction for 5/3, PSNR bound for 9/7; (2) a SECOND ENCODER agrees byte for byte: OpenJPEG's
 * code-streams of the same planes hold, per code-block, the same zero bit-planes, coding passes and MQ
 * segments, and the same packet sequence (tests/test_j2k_vs_openjpeg.py through the independent parser
 * tests/j2k_t2parse.py); (3) the decoder at the end of this file returns OpenJPEG's samples
 * (tests/test_j2k_decoder.py); (4) the published T.88 H.2 MQ test sequence; the MQ state table was
 * checked against libopenjp2's.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define J2K_GUARD 2   /* guard bits G: M_b = G + eps_b - 1 (T.800 E.1) */
#define MAX_LEVELS 10
#define MAX_PASSES 128

typedef struct {
  int reversible, levels, layers, cblk_w, cblk_h, use_sop;
  float base_step;
  int slopes[16];
} qo_j2k_params;

/* ------------------------------------------------------------------------------------------ */
/* T.800 Annex F: forward DWT, in place, Mallat layout (low half first).  Vertical pass first,
 * then horizontal (F.4.2), whole-sample symmetric extension.                                 */

/* 1-D reversible 5/3 on n samples with stride s (F.4.8.1): odd then even lifting, floor division. */
static void fdwt53_1d(int32_t *x, int n, ptrdiff_t s, int32_t *tmp) {
  if (n < 2) return;
  int nl = (n + 1) >> 1, nh = n >> 1;
  for (int i = 0; i < n; i++) tmp[i] = x[i * s];
  int32_t *lo = tmp + n, *hi = tmp + n + nl;
  for (int i = 0; i < nh; i++) {
    int32_t a = tmp[2 * i], b = (2 * i + 2 < n) ? tmp[2 * i + 2] : tmp[2 * i];
    hi[i] = tmp[2 * i + 1] - ((a + b) >> 1);
  }
  for (int i = 0; i < nl; i++) {
    int32_t a = (i > 0) ? hi[i - 1] : hi[0], b = (i < nh) ? hi[i] : hi[nh - 1];
    lo[i] = tmp[2 * i] + ((a + b + 2) >> 2);
  }
  for (int i = 0; i < nl; i++) x[i * s] = lo[i];
  for (int i = 0; i < nh; i++) x[(nl + i) * s] = hi[i];
}

/* 9/7 constants (T.800 Table F.4) as float32 */
#define F97_ALPHA (-1.586134342059924f)
#define F97_BETA (-0.052980118572961f)
#define F97_GAMMA (0.882911075530934f)
#define F97_DELTA (0.443506852043971f)
#define F97_K (1.230174104914001f)

/* 1-D irreversible 9/7 (F.4.8.2), float32, every product and sum rounded separately
 * (compile with -ffp-contract=off): the HIP kernel uses the same operation order. */
static void fdwt97_1d(float *x, int n, ptrdiff_t s, float *tmp) {
  if (n < 2) return;
  int nl = (n + 1) >> 1, nh = n >> 1;
  float *e = tmp, *o = tmp + nl; /* even (low) and odd (high) phases */
  for (int i = 0; i < nl; i++) e[i] = x[(2 * i) * s];
  for (int i = 0; i < nh; i++) o[i] = x[(2 * i + 1) * s];
  /* mirror rules: X[-1]=X[1], X[n]=X[n-2]:  odd index -1 -> o[0];  even index 2*nl (n even) -> e[nl-1];
   * odd index n (n odd) -> o[nh-1] */
  for (int i = 0; i < nh; i++) { float a = e[i], b = (i + 1 < nl) ? e[i + 1] : e[nl - 1]; o[i] = o[i] + F97_ALPHA * (a + b); }
  for (int i = 0; i < nl; i++) { float a = (i > 0) ? o[i - 1] : o[0], b = (i < nh) ? o[i] : o[nh - 1]; e[i] = e[i] + F97_BETA * (a + b); }
  for (int i = 0; i < nh; i++) { float a = e[i], b = (i + 1 < nl) ? e[i + 1] : e[nl - 1]; o[i] = o[i] + F97_GAMMA * (a + b); }
  for (int i = 0; i < nl; i++) { float a = (i > 0) ? o[i - 1] : o[0], b = (i < nh) ? o[i] : o[nh - 1]; e[i] = e[i] + F97_DELTA * (a + b); }
  const float invK = 1.0f / F97_K;
  for (int i = 0; i < nl; i++) x[i * s] = e[i] * invK;
  for (int i = 0; i < nh; i++) x[(nl + i) * s] = o[i] * F97_K;
}

void qo_j2k_fdwt53(int32_t *p, int stride, int H, int W, int levels) {
  int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 2 * ((W > H ? W : H) + 4));
  int w = W, h = H;
  for (int l = 0; l < levels; l++) {
    for (int x = 0; x < w; x++) fdwt53_1d(p + x, h, stride, tmp);       /* VER_SD */
    for (int y = 0; y < h; y++) fdwt53_1d(p + (size_t)y * stride, w, 1, tmp); /* HOR_SD */
    w = (w + 1) >> 1; h = (h + 1) >> 1;
  }
  free(tmp);
}
void qo_j2k_fdwt97(float *p, int stride, int H, int W, int levels) {
  float *tmp = (float *)malloc(sizeof(float) * 2 * ((W > H ? W : H) + 4));
  int w = W, h = H;
  for (int l = 0; l < levels; l++) {
    for (int x = 0; x < w; x++) fdwt97_1d(p + x, h, stride, tmp);
    for (int y = 0; y < h; y++) fdwt97_1d(p + (size_t)y * stride, w, 1, tmp);
    w = (w + 1) >> 1; h = (h + 1) >> 1;
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* Subband geometry for a tile anchored at (0,0) (T.800 B.5) in the Mallat layout.            */
typedef struct {
  int x0, y0, w, h;   /* position in the Mallat plane and size */
  int orient;         /* 0 LL, 1 HL, 2 LH, 3 HH */
  int dlev;           /* decomposition level 1..L (LL: L) */
  int eps, mant;      /* quantisation exponent / mantissa (QCD) */
  int Mb;             /* magnitude bit-planes */
  float delta;        /* step size (irreversible) */
} subband;

/* L2 norms of the 9/7 synthesis basis functions per (orient, decomposition level), computed
 * numerically (oracle/gen_norms.py, T.800 normalisation: low 1/K, high K).  Used only to shape
 * the step sizes; any positive step gives a valid stream. */
static const double norm97[4][MAX_LEVELS] = {
#include "j2k_norms97.inc"
};
static const double norm53[4][MAX_LEVELS] = {
#include "j2k_norms53.inc"
};
double qo_j2k_norm(int reversible, int orient, int dlev) {
  if (dlev < 1) dlev = 1;
  if (dlev > MAX_LEVELS) dlev = MAX_LEVELS;
  return (reversible ? norm53 : norm97)[orient][dlev - 1];
}

/* subbands in code-stream order: LL_L, then HL,LH,HH for d = L .. 1.  returns count = 3L+1 */
static int subbands_prec(int W, int H, int levels, int reversible, float base_step, int prec, subband *sb);
int qo_j2k_subbands(int W, int H, int levels, int reversible, float base_step, subband *sb) {
  return subbands_prec(W, H, levels, reversible, base_step, 8, sb);
}
static int subbands_prec(int W, int H, int levels, int reversible, float base_step, int prec, subband *sb) {
  int ws[MAX_LEVELS + 1], hs[MAX_LEVELS + 1];
  ws[0] = W; hs[0] = H;
  for (int d = 1; d <= levels; d++) { ws[d] = (ws[d - 1] + 1) >> 1; hs[d] = (hs[d - 1] + 1) >> 1; }
  int n = 0;
  sb[n].x0 = 0; sb[n].y0 = 0; sb[n].w = ws[levels]; sb[n].h = hs[levels]; sb[n].orient = 0; sb[n].dlev = levels; n++;
  for (int d = levels; d >= 1; d--) {
    int lw = ws[d], lh = hs[d], hw = ws[d - 1] - lw, hh = hs[d - 1] - lh;
    sb[n].x0 = lw; sb[n].y0 = 0; sb[n].w = hw; sb[n].h = lh; sb[n].orient = 1; sb[n].dlev = d; n++;
    sb[n].x0 = 0; sb[n].y0 = lh; sb[n].w = lw; sb[n].h = hh; sb[n].orient = 2; sb[n].dlev = d; n++;
    sb[n].x0 = lw; sb[n].y0 = lh; sb[n].w = hw; sb[n].h = hh; sb[n].orient = 3; sb[n].dlev = d; n++;
  }
  for (int i = 0; i < n; i++) {
    int gain = sb[i].orient == 0 ? 0 : (sb[i].orient == 3 ? 2 : 1); /* log2 nominal gain (E.1) */
    int Rb = prec + gain;
    if (reversible) {
      sb[i].eps = Rb; sb[i].mant = 0; sb[i].delta = 1.0f;
    } else {
      /* target step base_step*2^8/norm, then the nearest representable (eps, mant) below it (E.1.1) */
      double target = (double)base_step * 256.0 / qo_j2k_norm(0, sb[i].orient, sb[i].dlev ? sb[i].dlev : 1);
      int e = (int)floor(log2(target));
      double m = target / ldexp(1.0, e) - 1.0;
      int mant = (int)floor(m * 2048.0);
      if (mant < 0) mant = 0;
      if (mant > 2047) mant = 2047;
      int eps = Rb - e;
      if (eps < 0) { eps = 0; mant = 0; }
      if (eps > 31) { eps = 31; mant = 0; }
      sb[i].eps = eps; sb[i].mant = mant;
      sb[i].delta = (float)(ldexp(1.0, Rb - eps) * (1.0 + mant / 2048.0));
    }
    sb[i].Mb = J2K_GUARD + sb[i].eps - 1;
  }
  return n;
}

/* ------------------------------------------------------------------------------------------ */
/* T.800 Annex C: MQ encoder.  Table C.2 (Qe, NMPS, NLPS, SWITCH).                            */
static const uint16_t MQ_QE[47] = {0x5601, 0x3401, 0x1801, 0x0AC1, 0x0521, 0x0221, 0x5601, 0x5401, 0x4801, 0x3801,
                                   0x3001, 0x2401, 0x1C01, 0x1601, 0x5601, 0x5401, 0x5101, 0x4801, 0x3801, 0x3401,
                                   0x3001, 0x2801, 0x2401, 0x2201, 0x1C01, 0x1801, 0x1601, 0x1401, 0x1201, 0x1101,
                                   0x0AC1, 0x09C1, 0x08A1, 0x0521, 0x0441, 0x02A1, 0x0221, 0x0141, 0x0111, 0x0085,
                                   0x0049, 0x0025, 0x0015, 0x0009, 0x0005, 0x0001, 0x5601};
static const uint8_t MQ_NMPS[47] = {1, 2, 3, 4, 5, 38, 7, 8, 9, 10, 11, 12, 13, 29, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24,
                                    25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 45, 46};
static const uint8_t MQ_NLPS[47] = {1, 6, 9, 12, 29, 33, 6, 14, 14, 14, 17, 18, 20, 21, 14, 14, 15, 16, 17, 18, 19, 19, 20, 21,
                                    22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 46};
static const uint8_t MQ_SW[47] = {1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
const uint16_t *qo_mq_qe(void) { return MQ_QE; }
const uint8_t *qo_mq_nmps(void) { return MQ_NMPS; }
const uint8_t *qo_mq_nlps(void) { return MQ_NLPS; }
const uint8_t *qo_mq_sw(void) { return MQ_SW; }

#define CTX_RUN 17
#define CTX_UNI 18
#define NCTX 19

typedef struct {
  uint32_t A, C;
  int CT;
  uint8_t *start, *bp; /* bp points at the byte register B (start[-1] is a scratch byte) */
  uint8_t I[NCTX], MPS[NCTX];
  uint64_t nsym;
} mqenc;

static void mq_init(mqenc *m, uint8_t *buf) { /* INITENC (C.2.8) + context reset (Table D.7) */
  m->A = 0x8000; m->C = 0; m->CT = 12;
  m->start = buf + 1; m->bp = buf; buf[0] = 0;
  memset(m->I, 0, sizeof m->I); memset(m->MPS, 0, sizeof m->MPS);
  m->I[0] = 4; m->I[CTX_RUN] = 3; m->I[CTX_UNI] = 46;
  m->nsym = 0;
}
static void mq_byteout(mqenc *m) { /* BYTEOUT (C.2.7) */
  if (*m->bp == 0xFF) {
    m->bp++; *m->bp = (uint8_t)(m->C >> 20); m->C &= 0xFFFFF; m->CT = 7;
  } else if (m->C < 0x8000000) {
    m->bp++; *m->bp = (uint8_t)(m->C >> 19); m->C &= 0x7FFFF; m->CT = 8;
  } else {
    (*m->bp)++;
    if (*m->bp == 0xFF) {
      m->C &= 0x7FFFFFF;
      m->bp++; *m->bp = (uint8_t)(m->C >> 20); m->C &= 0xFFFFF; m->CT = 7;
    } else {
      m->bp++; *m->bp = (uint8_t)(m->C >> 19); m->C &= 0x7FFFF; m->CT = 8;
    }
  }
}
static void mq_renorm(mqenc *m) { /* RENORME (C.2.6) */
  do {
    m->A <<= 1; m->C <<= 1; m->CT--;
    if (m->CT == 0) mq_byteout(m);
  } while ((m->A & 0x8000) == 0);
}
static void mq_encode(mqenc *m, int cx, int d) { /* ENCODE / CODEMPS / CODELPS (C.2.2-C.2.5) */
  uint32_t qe = MQ_QE[m->I[cx]];
  m->nsym++;
  if (d == m->MPS[cx]) {
    m->A -= qe;
    if ((m->A & 0x8000) == 0) {
      if (m->A < qe) m->A = qe; else m->C += qe;
      m->I[cx] = MQ_NMPS[m->I[cx]];
      mq_renorm(m);
    } else {
      m->C += qe;
    }
  } else {
    m->A -= qe;
    if (m->A < qe) m->C += qe; else m->A = qe;
    if (MQ_SW[m->I[cx]]) m->MPS[cx] = 1 - m->MPS[cx];
    m->I[cx] = MQ_NLPS[m->I[cx]];
    mq_renorm(m);
  }
}
static int mq_flush(mqenc *m) { /* FLUSH (C.2.9); returns the code-word segment length */
  uint32_t tempc = m->C + m->A;
  m->C |= 0xFFFF;
  if (m->C >= tempc) m->C -= 0x8000;
  m->C <<= m->CT; mq_byteout(m);
  m->C <<= m->CT; mq_byteout(m);
  if (*m->bp != 0xFF) m->bp++;
  return (int)(m->bp - m->start);
}
/* bytes that certainly suffice to decode everything coded so far (truncation point estimate) */
static int mq_trunc_len(const mqenc *m) { return (int)(m->bp - m->start) + 1 + 4; }

/* ------------------------------------------------------------------------------------------ */
/* T.800 Annex D: EBCOT Tier-1 of one code-block (style 0: no bypass/reset/termall/causal/segsym).*/
enum { F_SIG = 1, F_VISIT = 2, F_REFINED = 4, F_NEG = 8 };

typedef struct {
  int w, h;
  uint8_t fl[66 * 66];   /* flags with a 1-sample apron */
  uint32_t mag[64 * 64];
} t1blk;
#define FL(b, y, x) ((b)->fl[((y) + 1) * 66 + (x) + 1])

/* Table D.1: zero-coding context from neighbour significance */
static int zc_ctx(const t1blk *b, int y, int x, int orient) {
  int h = (FL(b, y, x - 1) & F_SIG) + (FL(b, y, x + 1) & F_SIG);
  int v = (FL(b, y - 1, x) & F_SIG) + (FL(b, y + 1, x) & F_SIG);
  int d = (FL(b, y - 1, x - 1) & F_SIG) + (FL(b, y - 1, x + 1) & F_SIG) + (FL(b, y + 1, x - 1) & F_SIG) + (FL(b, y + 1, x + 1) & F_SIG);
  if (orient == 1) { int t = h; h = v; v = t; } /* HL: swap */
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
/* Tables D.2/D.3: sign context and XOR bit */
static int sc_ctx(const t1blk *b, int y, int x, int *xorbit) {
  int hc = 0, vc = 0, f;
  f = FL(b, y, x - 1); if (f & F_SIG) hc += (f & F_NEG) ? -1 : 1;
  f = FL(b, y, x + 1); if (f & F_SIG) hc += (f & F_NEG) ? -1 : 1;
  f = FL(b, y - 1, x); if (f & F_SIG) vc += (f & F_NEG) ? -1 : 1;
  f = FL(b, y + 1, x); if (f & F_SIG) vc += (f & F_NEG) ? -1 : 1;
  hc = hc > 1 ? 1 : (hc < -1 ? -1 : hc);
  vc = vc > 1 ? 1 : (vc < -1 ? -1 : vc);
  static const int ctx[3][3] = {{13, 12, 11}, {10, 9, 10}, {11, 12, 13}}; /* [hc+1][vc+1] */
  static const int xr[3][3] = {{1, 1, 1}, {1, 0, 0}, {0, 0, 0}};
  *xorbit = xr[hc + 1][vc + 1];
  return ctx[hc + 1][vc + 1];
}
static int any_sig_neighbour(const t1blk *b, int y, int x) {
  return (FL(b, y, x - 1) | FL(b, y, x + 1) | FL(b, y - 1, x) | FL(b, y + 1, x) | FL(b, y - 1, x - 1) | FL(b, y - 1, x + 1) |
          FL(b, y + 1, x - 1) | FL(b, y + 1, x + 1)) & F_SIG;
}
static void code_sign(mqenc *m, t1blk *b, int y, int x) {
  int xorbit, cx = sc_ctx(b, y, x, &xorbit);
  int neg = (FL(b, y, x) & F_NEG) ? 1 : 0;
  mq_encode(m, cx, neg ^ xorbit);
}

typedef struct {
  int npasses;                 /* coding passes produced */
  int nbps;                    /* magnitude bit-planes actually coded (Mb - zero bit-planes) */
  int len;                     /* code-word segment length */
  int pass_len[MAX_PASSES];    /* cumulative truncation length after each pass (<= len) */
  double pass_dist[MAX_PASSES];/* distortion reduction of each pass in quantiser-step^2 units */
  uint64_t nsym;
} t1result;

/* distortion decrease when a sample of magnitude m has bit-plane p revealed (midpoint reconstruction) */
static double dist_sig(uint32_t m, int p) {
  double r = (double)((m >> p) << p) + (p ? ldexp(1.0, p - 1) : 0.0);
  return (double)m * m - ((double)m - r) * ((double)m - r);
}
static double dist_ref(uint32_t m, int p) {
  double r0 = (double)((m >> (p + 1)) << (p + 1)) + ldexp(1.0, p);
  double r1 = (double)((m >> p) << p) + (p ? ldexp(1.0, p - 1) : 0.0);
  return ((double)m - r0) * ((double)m - r0) - ((double)m - r1) * ((double)m - r1);
}

/* coef: sign-magnitude samples (bit31 sign) of the code-block, row stride `stride`.
 * out must hold w*h*4 + 64 bytes.  Returns the segment length. */
static int t1_encode_block(const uint32_t *coef, int stride, int w, int h, int orient, uint8_t *out, t1result *res) {
  static t1blk blk; /* oracle is single-threaded */
  t1blk *b = &blk;
  b->w = w; b->h = h;
  memset(b->fl, 0, sizeof b->fl);
  uint32_t maxm = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      uint32_t c = coef[(size_t)y * stride + x];
      b->mag[y * 64 + x] = c & 0x7FFFFFFF;
      if (c & 0x80000000u) FL(b, y, x) |= F_NEG;
      if ((c & 0x7FFFFFFF) > maxm) maxm = c & 0x7FFFFFFF;
    }
  memset(res, 0, sizeof *res);
  int nbps = 0;
  while (nbps < 31 && (maxm >> nbps)) nbps++;
  res->nbps = nbps;
  if (nbps == 0) return 0;
  mqenc m;
  mq_init(&m, out);
  int np = 0;
  for (int p = nbps - 1; p >= 0; p--) {
    double dist;
    if (p != nbps - 1) {
      /* significance propagation pass (D.3.1) */
      dist = 0;
      for (int y0 = 0; y0 < h; y0 += 4)
        for (int x = 0; x < w; x++)
          for (int y = y0; y < y0 + 4 && y < h; y++) {
            uint8_t f = FL(b, y, x);
            if ((f & F_SIG) || !any_sig_neighbour(b, y, x)) continue;
            int bit = (b->mag[y * 64 + x] >> p) & 1;
            mq_encode(&m, zc_ctx(b, y, x, orient), bit);
            if (bit) { code_sign(&m, b, y, x); FL(b, y, x) |= F_SIG; dist += dist_sig(b->mag[y * 64 + x], p); }
            FL(b, y, x) |= F_VISIT;
          }
      res->pass_len[np] = mq_trunc_len(&m); res->pass_dist[np] = dist; np++;
      /* magnitude refinement pass (D.3.3) */
      dist = 0;
      for (int y0 = 0; y0 < h; y0 += 4)
        for (int x = 0; x < w; x++)
          for (int y = y0; y < y0 + 4 && y < h; y++) {
            uint8_t f = FL(b, y, x);
            if (!(f & F_SIG) || (f & F_VISIT)) continue;
            int cx = (f & F_REFINED) ? 16 : (any_sig_neighbour(b, y, x) ? 15 : 14);
            mq_encode(&m, cx, (b->mag[y * 64 + x] >> p) & 1);
            FL(b, y, x) |= F_REFINED;
            dist += dist_ref(b->mag[y * 64 + x], p);
          }
      res->pass_len[np] = mq_trunc_len(&m); res->pass_dist[np] = dist; np++;
    }
    /* clean-up pass (D.3.4) */
    dist = 0;
    for (int y0 = 0; y0 < h; y0 += 4)
      for (int x = 0; x < w; x++) {
        int y = y0;
        if (y0 + 3 < h) {
          int run = 1;
          for (int k = 0; k < 4; k++)
            if ((FL(b, y0 + k, x) & (F_SIG | F_VISIT)) || any_sig_neighbour(b, y0 + k, x)) { run = 0; break; }
          if (run) {
            int first = -1;
            for (int k = 0; k < 4; k++)
              if ((b->mag[(y0 + k) * 64 + x] >> p) & 1) { first = k; break; }
            if (first < 0) { mq_encode(&m, CTX_RUN, 0); continue; }
            mq_encode(&m, CTX_RUN, 1);
            mq_encode(&m, CTX_UNI, first >> 1);
            mq_encode(&m, CTX_UNI, first & 1);
            y = y0 + first;
            code_sign(&m, b, y, x);
            FL(b, y, x) |= F_SIG;
            dist += dist_sig(b->mag[y * 64 + x], p);
            y++;
          }
        }
        for (; y < y0 + 4 && y < h; y++) {
          uint8_t f = FL(b, y, x);
          if (f & (F_SIG | F_VISIT)) continue;
          int bit = (b->mag[y * 64 + x] >> p) & 1;
          mq_encode(&m, zc_ctx(b, y, x, orient), bit);
          if (bit) { code_sign(&m, b, y, x); FL(b, y, x) |= F_SIG; dist += dist_sig(b->mag[y * 64 + x], p); }
        }
      }
    res->pass_len[np] = mq_trunc_len(&m); res->pass_dist[np] = dist; np++;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) FL(b, y, x) &= ~F_VISIT;
  }
  res->npasses = np;
  res->len = mq_flush(&m);
  res->nsym = m.nsym;
  for (int i = 0; i < np; i++)
    if (res->pass_len[i] > res->len) res->pass_len[i] = res->len;
  res->pass_len[np - 1] = res->len;
  memmove(out, out + 1, (size_t)res->len); /* drop the scratch byte in front */
  return res->len;
}

/* exported single-block entry for tests: coef int32 two's complement values */
int qo_j2k_t1_block(const int32_t *vals, int w, int h, int orient, uint8_t *out, int *npasses, int *nbps, int *pass_len) {
  uint32_t *c = (uint32_t *)malloc(sizeof(uint32_t) * w * h);
  for (int i = 0; i < w * h; i++) c[i] = vals[i] < 0 ? (0x80000000u | (uint32_t)(-vals[i])) : (uint32_t)vals[i];
  uint8_t *tmp = (uint8_t *)malloc((size_t)w * h * 4 + 64);
  t1result r;
  int len = t1_encode_block(c, w, w, h, orient, tmp, &r);
  memcpy(out, tmp, len);
  *npasses = r.npasses; *nbps = r.nbps;
  if (pass_len) memcpy(pass_len, r.pass_len, sizeof(int) * r.npasses);
  free(c); free(tmp);
  return len;
}

/* ------------------------------------------------------------------------------------------ */
/* Tier-2: tag trees (B.10.2), packet headers (B.10), code-stream markers (Annex A).          */
typedef struct { uint8_t *p; size_t n, cap; int ovf; } obuf;
static void ob_put(obuf *o, int v) { if (o->n < o->cap) o->p[o->n] = (uint8_t)v; else o->ovf = 1; o->n++; }
static void ob_u16(obuf *o, int v) { ob_put(o, v >> 8); ob_put(o, v & 255); }
static void ob_u32(obuf *o, uint32_t v) { ob_u16(o, v >> 16); ob_u16(o, v & 0xFFFF); }

typedef struct { obuf *o; int acc, nbits, last_ff; } bitw; /* packet-header bit writer with 0xFF stuffing */
static void bw_init(bitw *b, obuf *o) { b->o = o; b->acc = 0; b->nbits = 0; b->last_ff = 0; }
static void bw_bit(bitw *b, int v) {
  int cap = b->last_ff ? 7 : 8;
  b->acc = (b->acc << 1) | (v & 1); b->nbits++;
  if (b->nbits == cap) { ob_put(b->o, b->acc); b->last_ff = (b->acc == 0xFF); b->acc = 0; b->nbits = 0; }
}
static void bw_bits(bitw *b, unsigned v, int n) { for (int i = n - 1; i >= 0; i--) bw_bit(b, (v >> i) & 1); }
static void bw_flush(bitw *b) {
  if (b->nbits) {
    int cap = b->last_ff ? 7 : 8;
    int v = b->acc << (cap - b->nbits);
    ob_put(b->o, v); b->last_ff = (v == 0xFF); b->acc = 0; b->nbits = 0;
  }
  if (b->last_ff) { ob_put(b->o, 0); b->last_ff = 0; }
}

typedef struct { int w, h; int nlev; int lw[12], lh[12], off[12]; int *val, *low; uint8_t *known; int nnodes; } tagtree;
static void tt_init(tagtree *t, int w, int h) {
  t->w = w; t->h = h; t->nlev = 0; t->nnodes = 0;
  int cw = w, ch = h;
  for (;;) {
    t->lw[t->nlev] = cw; t->lh[t->nlev] = ch; t->off[t->nlev] = t->nnodes; t->nnodes += cw * ch; t->nlev++;
    if (cw == 1 && ch == 1) break;
    cw = (cw + 1) >> 1; ch = (ch + 1) >> 1;
  }
  t->val = (int *)calloc(t->nnodes, sizeof(int)); t->low = (int *)calloc(t->nnodes, sizeof(int));
  t->known = (uint8_t *)calloc(t->nnodes, 1);
}
static void tt_free(tagtree *t) { free(t->val); free(t->low); free(t->known); }
static void tt_setvalues(tagtree *t, const int *leaf) { /* parent = min of children */
  for (int i = 0; i < t->w * t->h; i++) t->val[i] = leaf[i];
  for (int l = 1; l < t->nlev; l++)
    for (int y = 0; y < t->lh[l]; y++)
      for (int x = 0; x < t->lw[l]; x++) {
        int mn = 0x7FFFFFFF;
        for (int dy = 0; dy < 2; dy++)
          for (int dx = 0; dx < 2; dx++) {
            int cy = 2 * y + dy, cx = 2 * x + dx;
            if (cy < t->lh[l - 1] && cx < t->lw[l - 1]) { int v = t->val[t->off[l - 1] + cy * t->lw[l - 1] + cx]; if (v < mn) mn = v; }
          }
        t->val[t->off[l] + y * t->lw[l] + x] = mn;
      }
}
/* encode "is value(leaf) < threshold" information (B.10.2) */
static void tt_encode(tagtree *t, bitw *b, int lx, int ly, int threshold) {
  int path[12], n = 0;
  for (int l = 0; l < t->nlev; l++) { path[n++] = t->off[l] + (ly >> l) * t->lw[l] + (lx >> l); }
  int low = 0;
  for (int k = n - 1; k >= 0; k--) {
    int nd = path[k];
    if (low > t->low[nd]) t->low[nd] = low; else low = t->low[nd];
    while (low < threshold) {
      if (low >= t->val[nd]) {
        if (!t->known[nd]) { bw_bit(b, 1); t->known[nd] = 1; }
        break;
      }
      bw_bit(b, 0);
      low++;
    }
    t->low[nd] = low;
  }
}

typedef struct {
  int sb;                 /* subband index */
  int bx, by;             /* code-block position in its subband grid */
  int x0, y0, w, h;       /* in the Mallat plane */
  t1result r;
  uint8_t *data;
  int zbp;                /* missing MSBs: Mb - nbps */
  int layer_passes[16];   /* cumulative passes included up to each layer */
  int included;           /* already included in an earlier layer */
  int lblock;
  int first_layer;
} cblk;

static int ilog2f_(int v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }
static void put_npasses(bitw *b, int n) { /* Table B.4 */
  if (n == 1) bw_bit(b, 0);
  else if (n == 2) bw_bits(b, 2, 2);
  else if (n <= 5) bw_bits(b, 0xC | (n - 3), 4);
  else if (n <= 36) bw_bits(b, 0x1E0 | (n - 6), 9);
  else bw_bits(b, 0xFF80 | (n - 37), 16);
}

/* Kakadu-style logarithmic slope -> Lagrangian threshold (APPROXIMATE; the semantics of kdu_compress
 * `-slope` are Kakadu-private): slope s = 256*(192 + log2(dD/dL)) with D in units of
 * (full scale)^2, so in 8-bit sample^2 per byte  lambda = 2^(s/256 - 176). */
double qo_j2k_slope_to_lambda(int slope) { return pow(2.0, slope / 256.0 - 176.0); }

/* Choose, for one code-block, the number of passes to include for threshold lambda: the last pass on
 * the convex hull of (length, distortion) whose slope is >= lambda (PCRD, T.800 J.14 informative). */
static int passes_for_lambda(const t1result *r, double weight, double lambda) {
  if (lambda <= 0) return r->npasses;
  int best = 0;
  double best_d = 0; int best_l = 0;
  /* greedy hull walk */
  int cur = 0;
  while (cur < r->npasses) {
    double bs = -1; int bi = -1;
    double cd = 0;
    for (int i = cur; i < r->npasses; i++) {
      cd += r->pass_dist[i] * weight;
      int dl = r->pass_len[i] - best_l;
      if (dl <= 0) { if (cd > 0) { bs = 1e300; bi = i; } continue; }
      double s = cd / dl;
      if (s >= bs) { bs = s; bi = i; }
    }
    if (bi < 0 || bs < lambda) break;
    double dd = 0;
    for (int i = cur; i <= bi; i++) dd += r->pass_dist[i] * weight;
    best_d += dd; best_l = r->pass_len[bi]; best = bi + 1; cur = bi + 1;
  }
  (void)best_d;
  return best;
}

/* Tier-1 + Tier-2 of a plane whose sign-magnitude coefficients (Mallat layout) are given.  prec/sgnd describe the
 * samples for SIZ (Ssiz) and the dynamic range R_b. */
static long encode_coefs(uint32_t *coef, int H, int W, int L, const qo_j2k_params *prm, int prec, int sgnd, uint8_t *out,
                         size_t cap, uint64_t *stats);

/* Encode one 8-bit unsigned plane to a raw JPEG 2000 code-stream.  Returns the length or <0. */
long qo_j2k_encode_plane(const uint8_t *plane, int H, int W, const qo_j2k_params *prm, uint8_t *out, size_t cap,
                         uint64_t *stats /* nullable: [codeblocks, passes, symbols] */) {
  int L = prm->levels;
  if (L < 0 || L > MAX_LEVELS || W < 1 || H < 1) return -1;
  while (L > 0 && ((W >> (L - 1)) < 1 || (H >> (L - 1)) < 1)) L--;
  size_t N = (size_t)W * H;
  subband sb[3 * MAX_LEVELS + 1];
  int nsb = qo_j2k_subbands(W, H, L, prm->reversible, prm->base_step > 0 ? prm->base_step : 1.0f / 256, sb);
  /* level shift + DWT + quantisation -> sign-magnitude coefficients */
  uint32_t *coef = (uint32_t *)malloc(sizeof(uint32_t) * N);
  if (prm->reversible) {
    int32_t *p = (int32_t *)malloc(sizeof(int32_t) * N);
    for (size_t i = 0; i < N; i++) p[i] = (int32_t)plane[i] - 128;
    qo_j2k_fdwt53(p, W, H, W, L);
    for (size_t i = 0; i < N; i++) coef[i] = p[i] < 0 ? (0x80000000u | (uint32_t)(-p[i])) : (uint32_t)p[i];
    free(p);
  } else {
    float *p = (float *)malloc(sizeof(float) * N);
    for (size_t i = 0; i < N; i++) p[i] = (float)((int)plane[i] - 128);
    qo_j2k_fdwt97(p, W, H, W, L);
    for (int s = 0; s < nsb; s++)
      for (int y = 0; y < sb[s].h; y++)
        for (int x = 0; x < sb[s].w; x++) { /* dead-zone quantiser (E.1): q = sign(y) floor(|y| / delta) */
          size_t i = (size_t)(sb[s].y0 + y) * W + sb[s].x0 + x;
          float v = p[i];
          uint32_t q = (uint32_t)floorf(fabsf(v) / sb[s].delta);
          coef[i] = q | ((v < 0 && q) ? 0x80000000u : 0);
        }
    free(p);
  }
  long r = encode_coefs(coef, H, W, L, prm, 8, 0, out, cap, stats);
  free(coef);
  return r;
}

/* Signed 16-bit samples, reversible, no DC shift (Ssigned=yes): the motion-field planes of
 * motion_compress_j2k.py:131-142 (Sprecision=16, Clevels=0, Creversible=yes). */
long qo_j2k_encode_plane_s16(const int16_t *plane, int H, int W, const qo_j2k_params *prm, uint8_t *out, size_t cap,
                             uint64_t *stats) {
  int L = prm->levels;
  if (!prm->reversible || L < 0 || L > MAX_LEVELS || W < 1 || H < 1) return -1;
  while (L > 0 && ((W >> (L - 1)) < 1 || (H >> (L - 1)) < 1)) L--;
  size_t N = (size_t)W * H;
  int32_t *p = (int32_t *)malloc(sizeof(int32_t) * N);
  uint32_t *coef = (uint32_t *)malloc(sizeof(uint32_t) * N);
  for (size_t i = 0; i < N; i++) p[i] = plane[i];
  qo_j2k_fdwt53(p, W, H, W, L);
  for (size_t i = 0; i < N; i++) coef[i] = p[i] < 0 ? (0x80000000u | (uint32_t)(-p[i])) : (uint32_t)p[i];
  free(p);
  long r = encode_coefs(coef, H, W, L, prm, 16, 1, out, cap, stats);
  free(coef);
  return r;
}

static long encode_coefs(uint32_t *coef, int H, int W, int L, const qo_j2k_params *prm, int prec, int sgnd, uint8_t *out,
                         size_t cap, uint64_t *stats) {
  int layers = prm->layers < 1 ? 1 : prm->layers;
  if (layers > 16) return -1;
  int cbw = prm->cblk_w ? prm->cblk_w : 64, cbh = prm->cblk_h ? prm->cblk_h : 64;
  if (cbw > 64 || cbh > 64 || cbw < 4 || cbh < 4 || (cbw & (cbw - 1)) || (cbh & (cbh - 1))) return -1;
  subband sb[3 * MAX_LEVELS + 1];
  int nsb = subbands_prec(W, H, L, prm->reversible, prm->base_step > 0 ? prm->base_step : 1.0f / 256, prec, sb);

  /* code-blocks, in (subband, raster) order */
  int ncb = 0;
  for (int s = 0; s < nsb; s++) ncb += ((sb[s].w + cbw - 1) / cbw) * ((sb[s].h + cbh - 1) / cbh);
  cblk *cb = (cblk *)calloc(ncb ? ncb : 1, sizeof(cblk));
  int k = 0;
  uint64_t tot_passes = 0, tot_sym = 0;
  long ret = -2;
  for (int s = 0; s < nsb; s++) {
    int gx = (sb[s].w + cbw - 1) / cbw, gy = (sb[s].h + cbh - 1) / cbh;
    for (int by = 0; by < gy; by++)
      for (int bx = 0; bx < gx; bx++, k++) {
        cblk *c = &cb[k];
        c->sb = s; c->bx = bx; c->by = by;
        c->x0 = sb[s].x0 + bx * cbw; c->y0 = sb[s].y0 + by * cbh;
        c->w = (bx + 1) * cbw <= sb[s].w ? cbw : sb[s].w - bx * cbw;
        c->h = (by + 1) * cbh <= sb[s].h ? cbh : sb[s].h - by * cbh;
        c->data = (uint8_t *)malloc((size_t)c->w * c->h * 4 + 64);
        t1_encode_block(coef + (size_t)c->y0 * W + c->x0, W, c->w, c->h, sb[s].orient, c->data, &c->r);
        c->zbp = sb[s].Mb - c->r.nbps;
        if (c->zbp < 0) goto done; /* coefficient exceeds M_b: cannot be signalled */
        c->lblock = 3;
        tot_passes += c->r.npasses; tot_sym += c->r.nsym;
        /* layer allocation */
        double wgt = (double)sb[s].delta * qo_j2k_norm(prm->reversible, sb[s].orient, sb[s].dlev);
        wgt *= wgt;
        for (int l = 0; l < layers; l++) {
          int np;
          if (l == layers - 1 && prm->slopes[l] <= 0) np = c->r.npasses;
          else if (prm->slopes[l] <= 0) np = 0; /* only the last layer takes "everything" */
          else np = passes_for_lambda(&c->r, wgt, qo_j2k_slope_to_lambda(prm->slopes[l]));
          if (l > 0 && np < c->layer_passes[l - 1]) np = c->layer_passes[l - 1];
          c->layer_passes[l] = np;
        }
        c->first_layer = layers;
        for (int l = layers - 1; l >= 0; l--) if (c->layer_passes[l] > 0) c->first_layer = l;
      }
  }
  {
    obuf o = {out, 0, cap, 0};
    /* main header (A.4-A.6) */
    ob_u16(&o, 0xFF4F);                                    /* SOC */
    ob_u16(&o, 0xFF51); ob_u16(&o, 41); ob_u16(&o, 0);     /* SIZ, Rsiz=0 */
    ob_u32(&o, W); ob_u32(&o, H); ob_u32(&o, 0); ob_u32(&o, 0);
    ob_u32(&o, W); ob_u32(&o, H); ob_u32(&o, 0); ob_u32(&o, 0);
    ob_u16(&o, 1); ob_put(&o, (prec - 1) | (sgnd ? 0x80 : 0)); ob_put(&o, 1); ob_put(&o, 1);
    ob_u16(&o, 0xFF52); ob_u16(&o, 12);                    /* COD */
    ob_put(&o, prm->use_sop ? 0x02 : 0x00);
    ob_put(&o, 0); ob_u16(&o, layers); ob_put(&o, 0);      /* LRCP, layers, no MCT */
    ob_put(&o, L); ob_put(&o, ilog2f_(cbw) - 2); ob_put(&o, ilog2f_(cbh) - 2); ob_put(&o, 0);
    ob_put(&o, prm->reversible ? 1 : 0);
    ob_u16(&o, 0xFF5C);                                    /* QCD */
    if (prm->reversible) {
      ob_u16(&o, 3 + nsb); ob_put(&o, (J2K_GUARD << 5) | 0);
      for (int s = 0; s < nsb; s++) ob_put(&o, sb[s].eps << 3);
    } else {
      ob_u16(&o, 3 + 2 * nsb); ob_put(&o, (J2K_GUARD << 5) | 2);
      for (int s = 0; s < nsb; s++) ob_u16(&o, (sb[s].eps << 11) | sb[s].mant);
    }
    size_t sot = o.n;
    ob_u16(&o, 0xFF90); ob_u16(&o, 10); ob_u16(&o, 0); ob_u32(&o, 0); ob_put(&o, 0); ob_put(&o, 1); /* SOT, Psot patched */
    ob_u16(&o, 0xFF93);                                    /* SOD */
    /* packets: LRCP, one precinct per resolution */
    int nsop = 0;
    /* per-subband tag trees persist across layers */
    tagtree *incl = (tagtree *)calloc(nsb, sizeof(tagtree)), *zbpt = (tagtree *)calloc(nsb, sizeof(tagtree));
    int *first_cb = (int *)calloc(nsb + 1, sizeof(int));
    {
      int kk = 0;
      for (int s = 0; s < nsb; s++) {
        int gx = (sb[s].w + cbw - 1) / cbw, gy = (sb[s].h + cbh - 1) / cbh;
        first_cb[s] = kk;
        if (gx * gy > 0) {
          tt_init(&incl[s], gx, gy); tt_init(&zbpt[s], gx, gy);
          int *v1 = (int *)malloc(sizeof(int) * gx * gy), *v2 = (int *)malloc(sizeof(int) * gx * gy);
          for (int i = 0; i < gx * gy; i++) { v1[i] = cb[kk + i].first_layer; v2[i] = cb[kk + i].zbp; }
          tt_setvalues(&incl[s], v1); tt_setvalues(&zbpt[s], v2);
          free(v1); free(v2);
        }
        kk += gx * gy;
      }
      first_cb[nsb] = kk;
    }
    for (int l = 0; l < layers; l++)
      for (int r = 0; r <= L; r++) {
        int s0 = r == 0 ? 0 : 1 + 3 * (r - 1), s1 = r == 0 ? 1 : s0 + 3;
        if (prm->use_sop) { ob_u16(&o, 0xFF91); ob_u16(&o, 4); ob_u16(&o, nsop & 0xFFFF); }
        nsop++;
        bitw bw; bw_init(&bw, &o);
        int any = 0;
        for (int s = s0; s < s1; s++)
          for (int i = first_cb[s]; i < first_cb[s + 1]; i++) {
            int prev = l ? cb[i].layer_passes[l - 1] : 0;
            if (cb[i].layer_passes[l] > prev) any = 1;
          }
        bw_bit(&bw, any);
        if (any) {
          for (int s = s0; s < s1; s++)
            for (int i = first_cb[s]; i < first_cb[s + 1]; i++) {
              cblk *c = &cb[i];
              int prev = l ? c->layer_passes[l - 1] : 0, now = c->layer_passes[l], add = now - prev;
              if (!c->included) {
                tt_encode(&incl[s], &bw, c->bx, c->by, l + 1);   /* inclusion (B.10.4) */
                if (add <= 0) continue;
                tt_encode(&zbpt[s], &bw, c->bx, c->by, 0x7FFFFFF); /* zero bit-planes (B.10.5) */
                c->included = 1;
              } else {
                bw_bit(&bw, add > 0);
                if (add <= 0) continue;
              }
              put_npasses(&bw, add);
              int len = c->r.pass_len[now - 1] - (prev ? c->r.pass_len[prev - 1] : 0);
              int need = 1; while ((len >> need) && need < 32) need++;   /* bits to hold len */
              if (len == 0) need = 1;
              int have = c->lblock + ilog2f_(add);
              int inc = need > have ? need - have : 0;
              for (int q = 0; q < inc; q++) bw_bit(&bw, 1);
              bw_bit(&bw, 0);
              c->lblock += inc;
              bw_bits(&bw, (unsigned)len, c->lblock + ilog2f_(add));
            }
        }
        bw_flush(&bw);
        if (any)
          for (int s = s0; s < s1; s++)
            for (int i = first_cb[s]; i < first_cb[s + 1]; i++) {
              cblk *c = &cb[i];
              int prev = l ? c->layer_passes[l - 1] : 0, now = c->layer_passes[l];
              if (now <= prev) continue;
              int a = prev ? c->r.pass_len[prev - 1] : 0, e = c->r.pass_len[now - 1];
              for (int q = a; q < e; q++) ob_put(&o, c->data[q]);
            }
      }
    for (int s = 0; s < nsb; s++) if (incl[s].val) { tt_free(&incl[s]); tt_free(&zbpt[s]); }
    free(incl); free(zbpt); free(first_cb);
    /* Psot = bytes from SOT to the end of the tile-part */
    size_t psot = o.n - sot;
    if (sot + 10 <= cap) { out[sot + 6] = (uint8_t)(psot >> 24); out[sot + 7] = (uint8_t)(psot >> 16); out[sot + 8] = (uint8_t)(psot >> 8); out[sot + 9] = (uint8_t)psot; }
    ob_u16(&o, 0xFFD9); /* EOC */
    ret = o.ovf ? -3 : (long)o.n;
  }
  if (stats) { stats[0] = ncb; stats[1] = tot_passes; stats[2] = tot_sym; }
done:
  for (int i = 0; i < ncb; i++) free(cb[i].data);
  free(cb);
  return ret;
}

/* coefficients after level shift + DWT + quantisation (sign-magnitude -> two's complement), for tests */
int qo_j2k_coefficients(const uint8_t *plane, int H, int W, int reversible, int levels, float base_step, int32_t *out) {
  size_t N = (size_t)W * H;
  subband sb[3 * MAX_LEVELS + 1];
  int nsb = qo_j2k_subbands(W, H, levels, reversible, base_step > 0 ? base_step : 1.0f / 256, sb);
  if (reversible) {
    for (size_t i = 0; i < N; i++) out[i] = (int32_t)plane[i] - 128;
    qo_j2k_fdwt53(out, W, H, W, levels);
  } else {
    float *p = (float *)malloc(sizeof(float) * N);
    for (size_t i = 0; i < N; i++) p[i] = (float)((int)plane[i] - 128);
    qo_j2k_fdwt97(p, W, H, W, levels);
    for (int s = 0; s < nsb; s++)
      for (int y = 0; y < sb[s].h; y++)
        for (int x = 0; x < sb[s].w; x++) {
          size_t i = (size_t)(sb[s].y0 + y) * W + sb[s].x0 + x;
          float v = p[i];
          int32_t q = (int32_t)floorf(fabsf(v) / sb[s].delta);
          out[i] = v < 0 ? -q : q;
        }
    free(p);
  }
  return 0;
}
/* raw float 9/7 coefficients (before quantisation), for the 1-ULP check of the GPU transform */
int qo_j2k_dwt97_plane(const uint8_t *plane, int H, int W, int levels, float *out) {
  for (size_t i = 0; i < (size_t)W * H; i++) out[i] = (float)((int)plane[i] - 128);
  qo_j2k_fdwt97(out, W, H, W, levels);
  return 0;
}

/* MQ known-answer hook: code n decisions (one per byte of `bits`, values 0/1) in a single context that starts
 * in state 0 / MPS 0, then FLUSH.  Used with the arithmetic-coder test sequence of ITU-T T.88 Annex H.2
 * (the same MQ coder as T.800 Annex C). */
int qo_mq_encode_bits(const uint8_t *bits, int n, uint8_t *out) {
  uint8_t *buf = (uint8_t *)malloc((size_t)n + 64);
  mqenc m;
  mq_init(&m, buf);
  m.I[0] = 0; m.MPS[0] = 0;
  for (int i = 0; i < n; i++) mq_encode(&m, 0, bits[i]);
  int len = mq_flush(&m);
  memcpy(out, buf + 1, len);
  free(buf);
  return len;
}

/* ============================================================================================== */
/* DECODER restatement (the reference decodes with the external `kdu_expand`,     */
/* trunk/src/texture_expand_fb_j2k.py:153-162).  T.800 Annex B (packet parsing), C.3 (MQ decoder), */
/* D (bit modelling, decoding direction), E.1.1 (dequantisation, midpoint reconstruction) and F.3   */
/* (inverse DWT).  Pinned by an independent decoder: OpenJPEG through Pillow decodes the same       */
/* streams to the same samples (bit-exact for 5/3, within 1 LSB for 9/7: tests/test_j2k_decoder.py).*/

typedef struct { const uint8_t *d; size_t n, p; int acc, nb, last; } bitr;
static void br_init(bitr *b, const uint8_t *d, size_t n, size_t p) { b->d = d; b->n = n; b->p = p; b->acc = 0; b->nb = 0; b->last = 0; }
static int br_bit(bitr *b) {
  if (b->nb == 0) {
    int v = b->p < b->n ? b->d[b->p] : 0;
    b->p++;
    b->nb = b->last == 0xFF ? 7 : 8;
    b->acc = v; b->last = v;
  }
  b->nb--;
  return (b->acc >> b->nb) & 1;
}
static unsigned br_bits(bitr *b, int k) { unsigned v = 0; while (k-- > 0) v = (v << 1) | (unsigned)br_bit(b); return v; }
static size_t br_align(bitr *b) { b->nb = 0; if (b->last == 0xFF) { b->p++; b->last = 0; } return b->p; }

/* tag tree decoder (B.10.2): val = current lower bound, known = value established */
static int tt_decode(tagtree *t, bitr *b, int lx, int ly, int threshold) {
  int lo = 0;
  for (int k = t->nlev - 1; k >= 0; k--) {
    int nd = t->off[k] + (ly >> k) * t->lw[k] + (lx >> k);
    if (t->val[nd] < lo) t->val[nd] = lo;
    while (!t->known[nd] && t->val[nd] < threshold) {
      if (br_bit(b)) t->known[nd] = 1; else t->val[nd]++;
    }
    lo = t->val[nd];
    if (!t->known[nd]) return 0;
  }
  return t->val[t->off[0] + ly * t->lw[0] + lx] < threshold;
}
static int get_npasses(bitr *b) { /* Table B.4 */
  if (!br_bit(b)) return 1;
  if (!br_bit(b)) return 2;
  unsigned v = br_bits(b, 2);
  if (v != 3) return 3 + (int)v;
  v = br_bits(b, 5);
  if (v != 31) return 6 + (int)v;
  return 37 + (int)br_bits(b, 7);
}

/* MQ decoder (C.3) */
typedef struct {
  uint32_t A, C; int CT;
  const uint8_t *bp, *end;
  uint8_t I[NCTX], MPS[NCTX];
} mqdec;
static void mqd_bytein(mqdec *m) {
  if (m->bp < m->end) {
    if (*m->bp == 0xFF) {
      int nx = m->bp + 1 < m->end ? m->bp[1] : 0xFF;
      if (nx > 0x8F) { m->C += 0xFF00; m->CT = 8; }
      else { m->bp++; m->C += (uint32_t)nx << 9; m->CT = 7; }
    } else {
      m->bp++;
      m->C += (uint32_t)(m->bp < m->end ? *m->bp : 0xFF) << 8; m->CT = 8;
    }
  } else { m->C += 0xFF00; m->CT = 8; }
}
static void mqd_init(mqdec *m, const uint8_t *data, int len) {
  memset(m->I, 0, sizeof m->I); memset(m->MPS, 0, sizeof m->MPS);
  m->I[0] = 4; m->I[CTX_RUN] = 3; m->I[CTX_UNI] = 46;
  m->bp = data; m->end = data + len;
  m->C = (uint32_t)(len > 0 ? data[0] : 0xFF) << 16;
  mqd_bytein(m);
  m->C <<= 7; m->CT -= 7; m->A = 0x8000;
}
static int mqd_decode(mqdec *m, int cx) {
  uint32_t qe = MQ_QE[m->I[cx]];
  int d;
  m->A -= qe;
  if ((m->C >> 16) < qe) {           /* LPS exchange (C.3.2, Figure C.17) */
    if (m->A < qe) { d = m->MPS[cx]; m->I[cx] = MQ_NMPS[m->I[cx]]; }
    else { d = 1 - m->MPS[cx]; if (MQ_SW[m->I[cx]]) m->MPS[cx] = 1 - m->MPS[cx]; m->I[cx] = MQ_NLPS[m->I[cx]]; }
    m->A = qe;
  } else {
    m->C -= qe << 16;
    if (m->A & 0x8000) return m->MPS[cx];
    if (m->A < qe) { d = 1 - m->MPS[cx]; if (MQ_SW[m->I[cx]]) m->MPS[cx] = 1 - m->MPS[cx]; m->I[cx] = MQ_NLPS[m->I[cx]]; }
    else { d = m->MPS[cx]; m->I[cx] = MQ_NMPS[m->I[cx]]; }
  }
  do { if (m->CT == 0) mqd_bytein(m); m->A <<= 1; m->C <<= 1; m->CT--; } while ((m->A & 0x8000) == 0);
  return d;
}

/* one code-block: `npasses` coding passes of the segment -> magnitudes (with the bit-plane each sample was last coded in) */
static void t1_decode_block(const uint8_t *seg, int len, int npasses, int nbps, int w, int h, int orient, uint32_t *mag /*64x64*/,
                            uint8_t *lastp /*64x64*/, uint8_t *neg /*64x64*/) {
  static t1blk blk;
  t1blk *b = &blk;
  b->w = w; b->h = h;
  memset(b->fl, 0, sizeof b->fl);
  memset(mag, 0, sizeof(uint32_t) * 64 * 64); memset(lastp, 0, 64 * 64); memset(neg, 0, 64 * 64);
  if (nbps <= 0 || npasses <= 0) return;
  mqdec m;
  mqd_init(&m, seg, len);
  int np = 0;
#define DEC_SIGN(y, x) do { int xb_, cx_ = sc_ctx(b, (y), (x), &xb_); int s_ = mqd_decode(&m, cx_) ^ xb_; \
                            if (s_) { FL(b, (y), (x)) |= F_NEG; neg[(y) * 64 + (x)] = 1; } FL(b, (y), (x)) |= F_SIG; } while (0)
  for (int p = nbps - 1; p >= 0 && np < npasses; p--) {
    if (p != nbps - 1) {
      for (int y0 = 0; y0 < h; y0 += 4)
        for (int x = 0; x < w; x++)
          for (int y = y0; y < y0 + 4 && y < h; y++) {
            uint8_t f = FL(b, y, x);
            if ((f & F_SIG) || !any_sig_neighbour(b, y, x)) continue;
            int bit = mqd_decode(&m, zc_ctx(b, y, x, orient));
            if (bit) { mag[y * 64 + x] |= 1u << p; DEC_SIGN(y, x); }
            lastp[y * 64 + x] = (uint8_t)p;
            FL(b, y, x) |= F_VISIT;
          }
      if (++np >= npasses) break;
      for (int y0 = 0; y0 < h; y0 += 4)
        for (int x = 0; x < w; x++)
          for (int y = y0; y < y0 + 4 && y < h; y++) {
            uint8_t f = FL(b, y, x);
            if (!(f & F_SIG) || (f & F_VISIT)) continue;
            int cx = (f & F_REFINED) ? 16 : (any_sig_neighbour(b, y, x) ? 15 : 14);
            if (mqd_decode(&m, cx)) mag[y * 64 + x] |= 1u << p;
            lastp[y * 64 + x] = (uint8_t)p;
            FL(b, y, x) |= F_REFINED;
          }
      if (++np >= npasses) break;
    }
    for (int y0 = 0; y0 < h; y0 += 4)
      for (int x = 0; x < w; x++) {
        int y = y0;
        if (y0 + 3 < h) {
          int run = 1;
          for (int k = 0; k < 4; k++)
            if ((FL(b, y0 + k, x) & (F_SIG | F_VISIT)) || any_sig_neighbour(b, y0 + k, x)) { run = 0; break; }
          if (run) {
            if (!mqd_decode(&m, CTX_RUN)) { for (int k = 0; k < 4; k++) lastp[(y0 + k) * 64 + x] = (uint8_t)p; continue; }
            int first = mqd_decode(&m, CTX_UNI) << 1;
            first |= mqd_decode(&m, CTX_UNI);
            for (int k = 0; k < first; k++) lastp[(y0 + k) * 64 + x] = (uint8_t)p;
            y = y0 + first;
            mag[y * 64 + x] |= 1u << p; lastp[y * 64 + x] = (uint8_t)p;
            DEC_SIGN(y, x);
            y++;
          }
        }
        for (; y < y0 + 4 && y < h; y++) {
          uint8_t f = FL(b, y, x);
          if (f & (F_SIG | F_VISIT)) continue;
          int bit = mqd_decode(&m, zc_ctx(b, y, x, orient));
          if (bit) { mag[y * 64 + x] |= 1u << p; DEC_SIGN(y, x); }
          lastp[y * 64 + x] = (uint8_t)p;
        }
      }
    np++;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) FL(b, y, x) &= ~F_VISIT;
  }
#undef DEC_SIGN
}

/* inverse 1-D transforms on n samples of the Mallat layout (low half first), stride s */
static void idwt53_1d(int32_t *x, int n, ptrdiff_t s, int32_t *tmp) {
  if (n < 2) return;
  int nl = (n + 1) >> 1, nh = n >> 1;
  int32_t *lo = tmp, *hi = tmp + nl, *out = tmp + n;
  for (int i = 0; i < nl; i++) lo[i] = x[i * s];
  for (int i = 0; i < nh; i++) hi[i] = x[(nl + i) * s];
  for (int i = 0; i < nl; i++) {  /* X(2n) = Y(2n) - floor((Y(2n-1) + Y(2n+1) + 2) / 4) */
    int32_t a = (i > 0) ? hi[i - 1] : hi[0], b = (i < nh) ? hi[i] : hi[nh - 1];
    out[2 * i] = lo[i] - ((a + b + 2) >> 2);
  }
  for (int i = 0; i < nh; i++) {  /* X(2n+1) = Y(2n+1) + floor((X(2n) + X(2n+2)) / 2) */
    int32_t a = out[2 * i], b = (2 * i + 2 < n) ? out[2 * i + 2] : out[2 * i];
    out[2 * i + 1] = hi[i] + ((a + b) >> 1);
  }
  for (int i = 0; i < n; i++) x[i * s] = out[i];
}
static void idwt97_1d(float *x, int n, ptrdiff_t s, float *tmp) {
  if (n < 2) return;
  int nl = (n + 1) >> 1, nh = n >> 1;
  float *v = tmp;  /* interleaved working signal */
  const float invK = 1.0f / F97_K;
  for (int i = 0; i < nl; i++) v[2 * i] = x[i * s] * F97_K;          /* undo low * 1/K */
  for (int i = 0; i < nh; i++) v[2 * i + 1] = x[(nl + i) * s] * invK; /* undo high * K */
#define NB(i, par) /* neighbours of position i: the other one when outside */ \
  float lo_ = (i) - 1 >= 0 ? v[(i) - 1] : v[(i) + 1], hi_ = (i) + 1 < n ? v[(i) + 1] : v[(i) - 1]
  for (int i = 0; i < n; i += 2) { NB(i, 0); v[i] = v[i] - F97_DELTA * (lo_ + hi_); }
  for (int i = 1; i < n; i += 2) { NB(i, 1); v[i] = v[i] - F97_GAMMA * (lo_ + hi_); }
  for (int i = 0; i < n; i += 2) { NB(i, 0); v[i] = v[i] - F97_BETA * (lo_ + hi_); }
  for (int i = 1; i < n; i += 2) { NB(i, 1); v[i] = v[i] - F97_ALPHA * (lo_ + hi_); }
#undef NB
  for (int i = 0; i < n; i++) x[i * s] = v[i];
}

/* Decode a single-tile, single-component code-stream as this path writes them (LRCP, no precincts, style 0).
 * max_layers <= 0: all layers.  out: W*H uint8 (8-bit unsigned) or int16 (16-bit signed) samples.  Returns 0, or < 0. */
int qo_j2k_decode_plane(const uint8_t *cs, size_t n, int max_layers, void *out, int *Wo, int *Ho, int *preco, int *sgndo) {
  if (n < 4 || cs[0] != 0xFF || cs[1] != 0x4F) return -1;
  size_t p = 2;
  /* main header: segments inside the buffer and long enough for every field read; SIZ first; the first SIZ / COD / QCD count */
  const uint8_t *siz = NULL, *cod = NULL, *qcd = NULL;
  size_t lsiz = 0, lcod = 0, lqcd = 0;
  int sot = 0;
  while (p + 4 <= n) {
    if (cs[p] != 0xFF) return -2;
    int mk = (cs[p] << 8) | cs[p + 1];
    if (mk == 0xFF90) { sot = 1; break; }
    size_t ln = ((size_t)cs[p + 2] << 8) | cs[p + 3];
    if (ln < 2 || p + 2 + ln > n) return -2;
    if (mk == 0xFF51) { if (p != 2) return -2; if (!siz) { siz = cs + p + 4; lsiz = ln - 2; } }
    else if (mk == 0xFF52) { if (!cod) { cod = cs + p + 4; lcod = ln - 2; } }
    else if (mk == 0xFF5C) { if (!qcd) { qcd = cs + p + 4; lqcd = ln - 2; } }
    p += 2 + ln;
  }
  if (!siz || !cod || !qcd || !sot || p + 12 > n) return -2;
  if (lsiz < 39 || lcod < 10 || lqcd < 1) return -2;
#define U32(q) (((uint32_t)(q)[0] << 24) | ((uint32_t)(q)[1] << 16) | ((uint32_t)(q)[2] << 8) | (q)[3])
  int W = (int)U32(siz + 2), H = (int)U32(siz + 6);
  if (W < 1 || H < 1 || W > 16384 || H > 16384 || ((siz[34] << 8) | siz[35]) != 1) return -3;
  int prec = (siz[36] & 0x7F) + 1, sgnd = siz[36] >> 7;
  if (cod[6] > 4 || cod[7] > 4) return -3;
  int scod = cod[0], layers = (cod[2] << 8) | cod[3], L = cod[5], cbw = 1 << (cod[6] + 2), cbh = 1 << (cod[7] + 2), rev = cod[9] == 1;
  if (cod[1] != 0 || cod[8] != 0 || cbw > 64 || cbh > 64 || L > MAX_LEVELS || layers < 1) return -3;
  if (lqcd < 1 + (size_t)(3 * L + 1) * ((qcd[0] & 31) == 0 ? 1 : 2)) return -2;
  int guard = qcd[0] >> 5, qstyle = qcd[0] & 31;
  subband sb[3 * MAX_LEVELS + 1];
  int nsb = subbands_prec(W, H, L, 1, 1.0f, prec, sb);   /* geometry; eps / mant / delta / Mb from QCD below */
  for (int s = 0; s < nsb; s++) {
    int gain = sb[s].orient == 0 ? 0 : (sb[s].orient == 3 ? 2 : 1), Rb = prec + gain;
    if (qstyle == 0) { sb[s].eps = qcd[1 + s] >> 3; sb[s].mant = 0; sb[s].delta = 1.0f; }
    else if (qstyle == 2) {
      int v = (qcd[1 + 2 * s] << 8) | qcd[2 + 2 * s];
      sb[s].eps = v >> 11; sb[s].mant = v & 0x7FF;
      sb[s].delta = (float)(ldexp(1.0, Rb - sb[s].eps) * (1.0 + sb[s].mant / 2048.0));
    } else return -4;
    sb[s].Mb = guard + sb[s].eps - 1;
  }
  if (Wo) *Wo = W;
  if (Ho) *Ho = H;
  if (preco) *preco = prec;
  if (sgndo) *sgndo = sgnd;
  /* tile-part */
  uint32_t psot = U32(cs + p + 6);
  size_t end = psot ? p + psot : n - 2;
  p += 12;
  while (p + 2 <= n && !(cs[p] == 0xFF && cs[p + 1] == 0x93)) p += 2 + (size_t)((cs[p + 2] << 8) | cs[p + 3]);
  p += 2;
  /* code-blocks */
  int ncb = 0, first_cb[3 * MAX_LEVELS + 2];
  for (int s = 0; s < nsb; s++) { first_cb[s] = ncb; ncb += ((sb[s].w + cbw - 1) / cbw) * ((sb[s].h + cbh - 1) / cbh); }
  first_cb[nsb] = ncb;
  typedef struct { int zbp, npasses, len, lblock, incl; uint8_t *data; size_t cap; } dblk;
  dblk *cb = (dblk *)calloc(ncb ? ncb : 1, sizeof(dblk));
  for (int i = 0; i < ncb; i++) cb[i].lblock = 3;
  tagtree *incl = (tagtree *)calloc(nsb, sizeof(tagtree)), *zbpt = (tagtree *)calloc(nsb, sizeof(tagtree));
  for (int s = 0; s < nsb; s++) {
    int gx = (sb[s].w + cbw - 1) / cbw, gy = (sb[s].h + cbh - 1) / cbh;
    if (gx * gy > 0) { tt_init(&incl[s], gx, gy); tt_init(&zbpt[s], gx, gy); }
  }
  int use_layers = (max_layers > 0 && max_layers < layers) ? max_layers : layers, ret = 0;
  for (int l = 0; l < layers && ret == 0; l++)
    for (int r = 0; r <= L; r++) {
      if (p + 6 <= n && cs[p] == 0xFF && cs[p + 1] == 0x91) p += 6;  /* SOP */
      int s0 = r == 0 ? 0 : 1 + 3 * (r - 1), s1 = r == 0 ? 1 : s0 + 3;
      bitr br; br_init(&br, cs, n, p);
      int *plen = (int *)calloc(ncb ? ncb : 1, sizeof(int));
      if (br_bit(&br)) {
        for (int s = s0; s < s1; s++) {
          int gx = (sb[s].w + cbw - 1) / cbw;
          for (int i = first_cb[s]; i < first_cb[s + 1]; i++) {
            dblk *c = &cb[i];
            int bx = (i - first_cb[s]) % gx, by = (i - first_cb[s]) / gx;
            if (!c->incl) {
              if (!tt_decode(&incl[s], &br, bx, by, l + 1)) continue;
              int t = 1;
              while (!tt_decode(&zbpt[s], &br, bx, by, t)) t++;
              c->zbp = zbpt[s].val[zbpt[s].off[0] + by * zbpt[s].lw[0] + bx];
              c->incl = 1;
            } else if (!br_bit(&br)) continue;
            int np = get_npasses(&br);
            while (br_bit(&br)) c->lblock++;
            int ln = (int)br_bits(&br, c->lblock + ilog2f_(np));
            plen[i] = ln;
            if (l < use_layers) c->npasses += np;
            else plen[i] = -ln;  /* skipped layer: bytes are stepped over */
          }
        }
      }
      p = br_align(&br);
      if ((scod & 4) && p + 2 <= n && cs[p] == 0xFF && cs[p + 1] == 0x92) p += 2;  /* EPH */
      for (int s = s0; s < s1; s++)
        for (int i = first_cb[s]; i < first_cb[s + 1]; i++) {
          int ln = plen[i];
          if (ln > 0) {
            dblk *c = &cb[i];
            if ((size_t)(c->len + ln) > c->cap) { c->cap = (size_t)(c->len + ln) * 2 + 64; c->data = (uint8_t *)realloc(c->data, c->cap); }
            if (p + (size_t)ln > n) { ret = -5; break; }
            memcpy(c->data + c->len, cs + p, (size_t)ln); c->len += ln;
          }
          p += (size_t)(ln < 0 ? -ln : ln);
        }
      free(plen);
    }
  (void)end;
  /* Tier-1 + dequantisation into the Mallat plane */
  size_t N = (size_t)W * H;
  int32_t *ip = rev ? (int32_t *)calloc(N, sizeof(int32_t)) : NULL;
  float *fp = rev ? NULL : (float *)calloc(N, sizeof(float));
  uint32_t *mag = (uint32_t *)malloc(sizeof(uint32_t) * 64 * 64);
  uint8_t *lastp = (uint8_t *)malloc(64 * 64), *neg = (uint8_t *)malloc(64 * 64);
  for (int s = 0; s < nsb && ret == 0; s++) {
    int gx = (sb[s].w + cbw - 1) / cbw;
    for (int i = first_cb[s]; i < first_cb[s + 1]; i++) {
      dblk *c = &cb[i];
      if (!c->incl || c->npasses == 0) continue;
      int bx = (i - first_cb[s]) % gx, by = (i - first_cb[s]) / gx;
      int w = (bx + 1) * cbw <= sb[s].w ? cbw : sb[s].w - bx * cbw, h = (by + 1) * cbh <= sb[s].h ? cbh : sb[s].h - by * cbh;
      int nbps = sb[s].Mb - c->zbp;
      t1_decode_block(c->data, c->len, c->npasses, nbps, w, h, sb[s].orient, mag, lastp, neg);
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
          uint32_t m = mag[y * 64 + x];
          size_t o = (size_t)(sb[s].y0 + by * cbh + y) * W + sb[s].x0 + bx * cbw + x;
          if (!m) continue;
          int pl = lastp[y * 64 + x];
          if (rev) {  /* bits below the last decoded plane: midpoint (none left when every pass was decoded) */
            int32_t v = (int32_t)m + (pl > 0 ? (1 << (pl - 1)) : 0);
            ip[o] = neg[y * 64 + x] ? -v : v;
          } else {    /* E.1.1.2 with r = 1/2 */
            float v = ((float)m + (pl > 0 ? (float)(1 << (pl - 1)) : 0.5f)) * sb[s].delta;
            fp[o] = neg[y * 64 + x] ? -v : v;
          }
        }
    }
  }
  /* inverse DWT: levels L .. 1, horizontal then vertical (the reverse of F.4.2's order) */
  if (ret == 0) {
    int ws[MAX_LEVELS + 1], hs[MAX_LEVELS + 1];
    ws[0] = W; hs[0] = H;
    for (int d = 1; d <= L; d++) { ws[d] = (ws[d - 1] + 1) >> 1; hs[d] = (hs[d - 1] + 1) >> 1; }
    void *tmp = malloc(sizeof(float) * 3 * (size_t)((W > H ? W : H) + 8));
    for (int d = L; d >= 1; d--) {
      int w = ws[d - 1], h = hs[d - 1];
      if (rev) {
        for (int y = 0; y < h; y++) idwt53_1d(ip + (size_t)y * W, w, 1, (int32_t *)tmp);
        for (int x = 0; x < w; x++) idwt53_1d(ip + x, h, W, (int32_t *)tmp);
      } else {
        for (int y = 0; y < h; y++) idwt97_1d(fp + (size_t)y * W, w, 1, (float *)tmp);
        for (int x = 0; x < w; x++) idwt97_1d(fp + x, h, W, (float *)tmp);
      }
    }
    free(tmp);
    int shift = sgnd ? 0 : 1 << (prec - 1), lo = sgnd ? -(1 << (prec - 1)) : 0, hi = sgnd ? (1 << (prec - 1)) - 1 : (1 << prec) - 1;
    for (size_t i = 0; i < N; i++) {
      long v = rev ? (long)ip[i] : lrintf(fp[i]);
      v += shift;
      v = v < lo ? lo : (v > hi ? hi : v);
      if (prec <= 8) ((uint8_t *)out)[i] = (uint8_t)v; else ((int16_t *)out)[i] = (int16_t)v;
    }
  }
  for (int s = 0; s < nsb; s++) if (incl[s].val) { tt_free(&incl[s]); tt_free(&zbpt[s]); }
  for (int i = 0; i < ncb; i++) free(cb[i].data);
  free(incl); free(zbpt); free(cb); free(ip); free(fp); free(mag); free(lastp); free(neg);
  return ret;
#undef U32
}
