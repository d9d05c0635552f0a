/*
 * oracle/mctf_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product path.
 *
 * A single-threaded CPU restatement (our own code, plain C) of the MCTF half of the
 * reference's `mctf compress` hot path.  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check the HIP path bit-for-bit on machines where
 * /root/reference does not exist.  Nothing in qsvc_amd/ may import, link or call it.
 *
 * Parity pin: this file is validated byte-for-byte against the reference's own tools compiled
 * from /root/reference (oracle/_ref, see oracle/Makefile) by tests/test_oracle_vs_ref.py, and
 * against the committed fixtures those tools produced (tests/golden/, made by
 * tests/golden/make_goldens.py).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/trunk/src).  The reference keeps pictures as arrays of separately
 * malloc'ed rows with an optional border; here a picture is one flat int16 array with a
 * stride, and the reference's heap-layout quirks (SURVEY.md Q1,Q2,Q3,Q4,Q7) are reproduced
 * explicitly where they change results.
 *
 * Frame sequences are planar I420 u8, one frame = W*H + 2*(W/2)*(H/2) bytes
 * (split.cpp:55-56,229-253).  Motion fields: per B picture four int16 planes in the order
 * PREV-X, PREV-Y, NEXT-X, NEXT-Y, each BY x BX row-major (motion.cpp:9-15,93-101).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define QO_OK 0
#define QO_EINVAL (-1)

/* ------------------------------------------------------------------------------------------ */
/* 1-D truncating 5/3 lifting (5_3.cpp:39-115).  C integer division truncates toward zero,
 * exactly like the reference; operands are promoted to int, results stored to int16.        */

static void a53_even(const int16_t *s, int16_t *l, int16_t *h, int n) { /* 5_3.cpp:39-53 */
  int i;
  for (i = 0; i < n / 2 - 1; i++) h[i] = (int16_t)(s[2 * i + 1] - (s[2 * i] + s[2 * i + 2]) / 2);
  h[i] = (int16_t)(s[n - 1] - s[n - 2]);
  l[0] = (int16_t)(s[0] + h[0] / 2);
  for (i = 1; i < n / 2; i++) l[i] = (int16_t)(s[2 * i] + (h[i] + h[i - 1]) / 4);
}
static void a53_odd(const int16_t *s, int16_t *l, int16_t *h, int n) { /* 5_3.cpp:61-74 */
  int i;
  for (i = 0; i < n / 2; i++) h[i] = (int16_t)(s[2 * i + 1] - (s[2 * i] + s[2 * i + 2]) / 2);
  l[0] = (int16_t)(s[0] + h[0] / 2);
  for (i = 1; i < n / 2; i++) l[i] = (int16_t)(s[2 * i] + (h[i] + h[i - 1]) / 4);
  l[i] = (int16_t)(s[n - 1] + h[i - 1] / 2);
}
static void s53_even(int16_t *s, const int16_t *l, const int16_t *h, int n) { /* 5_3.cpp:82-95 */
  int i;
  s[0] = (int16_t)(l[0] - h[0] / 2);
  for (i = 1; i < n / 2; i++) s[2 * i] = (int16_t)(l[i] - (h[i] + h[i - 1]) / 4);
  for (i = 0; i < n / 2 - 1; i++) s[2 * i + 1] = (int16_t)(h[i] + (s[2 * i] + s[2 * i + 2]) / 2);
  s[n - 1] = (int16_t)(h[i] + s[n - 2]);
}
static void s53_odd(int16_t *s, const int16_t *l, const int16_t *h, int n) { /* 5_3.cpp:103-115 */
  int i;
  s[0] = (int16_t)(l[0] - h[0] / 2);
  for (i = 1; i < n / 2; i++) s[2 * i] = (int16_t)(l[i] - (h[i] + h[i - 1]) / 4);
  s[n - 1] = (int16_t)(l[i] - h[i - 1] / 2);
  for (i = 0; i < n / 2; i++) s[2 * i + 1] = (int16_t)(h[i] + (s[2 * i] + s[2 * i + 2]) / 2);
}
/* Haar (Haar.cpp:39-89) */
static void shaar_even(int16_t *s, const int16_t *l, const int16_t *h, int n) {
  int i, k;
  for (i = k = 0; k < n; i++, k += 2) { s[k] = (int16_t)(l[i] - h[i] / 2); s[k + 1] = (int16_t)(s[k] + h[i]); }
}
static void shaar_odd(int16_t *s, const int16_t *l, const int16_t *h, int n) {
  int i, k;
  for (i = k = 0; k < n - 1; i++, k += 2) { s[k] = (int16_t)(l[i] - h[i] / 2); s[k + 1] = (int16_t)(s[k] + h[i]); }
  s[k] = l[i];
}
static void ahaar_even(const int16_t *s, int16_t *l, int16_t *h, int n) {
  int i, k;
  for (i = k = 0; k < n; i++, k += 2) { h[i] = (int16_t)(s[k + 1] - s[k]); l[i] = (int16_t)(s[k] + h[i] / 2); }
}
static void ahaar_odd(const int16_t *s, int16_t *l, int16_t *h, int n) {
  int i, k;
  for (i = k = 0; k < n - 1; i++, k += 2) { h[i] = (int16_t)(s[k + 1] - s[k]); l[i] = (int16_t)(s[k] + h[i] / 2); }
  l[i] = s[k];
}

typedef void (*ana_fn)(const int16_t *, int16_t *, int16_t *, int);
typedef void (*syn_fn)(int16_t *, const int16_t *, const int16_t *, int);

/* 2-D in-place (Mallat layout) analysis, dwt2d.cpp:76-119: per level rows first, then columns;
 * next level on the floor-halved top-left region. */
static void dwt2d_analyze(int16_t *p, int stride, int y, int x, int levels, ana_fn aeven, ana_fn aodd) {
  int cap = (x > y ? x : y) + 8;
  int16_t *in = (int16_t *)malloc(sizeof(int16_t) * cap), *out = (int16_t *)malloc(sizeof(int16_t) * cap);
  for (int lv = 0; lv < levels; lv++) {
    int nx = x; x >>= 1;
    int ny = y; y >>= 1;
    if (y == 0) y = 1;
    if (x == 0) x = 1;
    for (int j = 0; j < ny; j++) {
      int16_t *row = p + (size_t)j * stride;
      memcpy(in, row, sizeof(int16_t) * nx);
      if (nx & 1) aodd(in, row, row + x + 1, nx); else aeven(in, row, row + x, nx);
    }
    for (int i = 0; i < nx; i++) {
      for (int j = 0; j < ny; j++) in[j] = p[(size_t)j * stride + i];
      if (ny & 1) aodd(in, out, out + y + 1, ny); else aeven(in, out, out + y, ny);
      for (int j = 0; j < ny; j++) p[(size_t)j * stride + i] = out[j];
    }
  }
  free(in); free(out);
}
/* 2-D synthesis, dwt2d.cpp:128-175: per level columns first, then rows. */
static void dwt2d_synthesize(int16_t *p, int stride, int y, int x, int levels, syn_fn seven, syn_fn sodd) {
  int cap = (x > y ? x : y) + 8;
  int16_t *in = (int16_t *)malloc(sizeof(int16_t) * cap), *out = (int16_t *)malloc(sizeof(int16_t) * cap);
  int nx = x >> levels, ny = y >> levels;
  for (int lv = levels - 1; lv >= 0; lv--) {
    int mx = nx, my = ny;
    nx = x >> lv; ny = y >> lv;
    if (nx == 0) nx = 1;
    if (ny == 0) ny = 1;
    for (int i = 0; i < nx; i++) {
      for (int j = 0; j < ny; j++) in[j] = p[(size_t)j * stride + i];
      if (ny & 1) sodd(out, in, in + my + 1, ny); else seven(out, in, in + my, ny);
      for (int j = 0; j < ny; j++) p[(size_t)j * stride + i] = out[j];
    }
    for (int j = 0; j < ny; j++) {
      int16_t *row = p + (size_t)j * stride;
      memcpy(in, row, sizeof(int16_t) * nx);
      if (nx & 1) sodd(row, in, in + mx + 1, nx); else seven(row, in, in + mx, nx);
    }
  }
  free(in); free(out);
}

void qo_dwt53t_analyze(int16_t *plane, int stride, int H, int W, int levels) {
  dwt2d_analyze(plane, stride, H, W, levels, a53_even, a53_odd);
}
void qo_dwt53t_synthesize(int16_t *plane, int stride, int H, int W, int levels) {
  dwt2d_synthesize(plane, stride, H, W, levels, s53_even, s53_odd);
}
void qo_haar_analyze(int16_t *plane, int stride, int H, int W, int levels) {
  dwt2d_analyze(plane, stride, H, W, levels, ahaar_even, ahaar_odd);
}
void qo_haar_synthesize(int16_t *plane, int stride, int H, int W, int levels) {
  dwt2d_synthesize(plane, stride, H, W, levels, shaar_even, shaar_odd);
}

/* ------------------------------------------------------------------------------------------ */
/* Bordered int16 picture (texture.cpp:34-113).                                              */

typedef struct {
  int H, W, b, stride;
  int16_t *base; /* (H+2b) x (W+2b) */
  int16_t *px;   /* -> interior (0,0) */
} bplane;

static int bplane_alloc(bplane *p, int H, int W, int b) {
  p->H = H; p->W = W; p->b = b; p->stride = W + 2 * b;
  p->base = (int16_t *)calloc((size_t)(H + 2 * b) * p->stride, sizeof(int16_t));
  if (!p->base) return -1;
  p->px = p->base + (size_t)b * p->stride + b;
  return 0;
}
static void bplane_free(bplane *p) { free(p->base); p->base = 0; }
#define BP(p, y, x) ((p)->px[(ptrdiff_t)(y) * (p)->stride + (x)])

/* texture::read (texture.cpp:122-129): u8 -> int16 widening into the interior. */
static void bplane_load_u8(bplane *p, const uint8_t *src, int h, int w) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) BP(p, y, x) = src[(size_t)y * w + x];
}

/* texture::fill_border (texture.cpp:55-113) as it behaves on planes made by
 * texture::alloc (texture.cpp:34-46) under glibc malloc:
 *  Q2: the lower-left corner is filled with the bottom-RIGHT pixel (texture.cpp:93-97).
 *  Q1: alloc() shifts only the first H of the H+2b row pointers, so logical rows >= H-b are
 *      addressed un-shifted and their left border lies before the row's own allocation.  For
 *      row r = H-b the cells x < -safe alias the right border of row r-1 (rows are malloc'ed
 *      back to back; a row of req = 2(W+2b) bytes occupies a chunk of align16(req+8) bytes),
 *      which Region 5 rewrites after Region 4, so they read back data[r-1][W-1].
 */
static void bplane_fill_border(bplane *p) {
  int H = p->H, W = p->W, b = p->b;
  if (b <= 0) return;
  for (int y = -b; y < 0; y++) {
    for (int x = -b; x < 0; x++) BP(p, y, x) = BP(p, 0, 0);
    for (int x = 0; x < W; x++) BP(p, y, x) = BP(p, 0, x);
    for (int x = W; x < W + b; x++) BP(p, y, x) = BP(p, 0, W - 1);
  }
  for (int y = 0; y < H; y++) {
    for (int x = -b; x < 0; x++) BP(p, y, x) = BP(p, y, 0);
    for (int x = W; x < W + b; x++) BP(p, y, x) = BP(p, y, W - 1);
  }
  for (int y = H; y < H + b; y++) {
    for (int x = -b; x < 0; x++) BP(p, y, x) = BP(p, H - 1, W - 1); /* Q2 */
    for (int x = 0; x < W; x++) BP(p, y, x) = BP(p, H - 1, x);
    for (int x = W; x < W + b; x++) BP(p, y, x) = BP(p, H - 1, W - 1);
  }
  /* Q1 */
  int r = H - b;
  if (r >= 1 && r < H) {
    long req = 2L * (W + 2 * b);
    long pad = (16 - ((req + 8) % 16)) % 16;
    int safe = (int)((8 + pad) / 2);
    for (int x = -b; x < -safe; x++) BP(p, r, x) = BP(p, r - 1, W - 1);
  }
}

int qo_q1_safe(int W, int b) { /* exported for the tests / host-side documentation */
  long req = 2L * (W + 2 * b);
  long pad = (16 - ((req + 8) % 16)) % 16;
  return (int)((8 + pad) / 2);
}

/* Fill a bordered plane from a u8 image and apply the border rules: exported so tests can
 * check the GPU's on-the-fly border addressing against it. out is (H+2b)x(W+2b) int16. */
int qo_bordered_from_u8(const uint8_t *src, int H, int W, int b, int16_t *out) {
  bplane p;
  if (bplane_alloc(&p, H, W, b)) return QO_EINVAL;
  bplane_load_u8(&p, src, H, W);
  bplane_fill_border(&p);
  memcpy(out, p.base, sizeof(int16_t) * (size_t)(H + 2 * b) * (W + 2 * b));
  bplane_free(&p);
  return QO_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Motion estimation (motion_estimate.cpp).                                                  */

static int desp(int x, int l) { /* motion_estimate.cpp:232-236 */
  for (int i = 0; i < l; i++) x = (x + 1) / 2;
  return x;
}

typedef struct { int BY, BX; int16_t *f[4]; } mvfield; /* f[0]=PREV-X f[1]=PREV-Y f[2]=NEXT-X f[3]=NEXT-Y */
#define MV(m, k, by, bx) ((m)->f[k][(size_t)(by) * (m)->BX + (bx)])

/* local_me_for_block (motion_estimate.cpp:70-184); `bord` = border_size: the SAD window grows by it on every side
 * (local_me_for_image, :213-217).  The predicted plane's own border is never filled by the reference (:792-796 read the
 * interior only): it reads as the zeros of a fresh heap there, which is what calloc gives here. */
static void me_block(mvfield *mv, const bplane *r0, const bplane *r1, const bplane *pr, int bs, int bord, int by, int bx) {
  static const int cand[9][2] = {{-1, -1}, {-1, 1}, {1, -1}, {1, 1}, {-1, 0}, {1, 0}, {0, 1}, {0, -1}, {0, 0}};
  int pvy = MV(mv, 1, by, bx), pvx = MV(mv, 0, by, bx), nvy = MV(mv, 3, by, bx), nvx = MV(mv, 2, by, bx);
  int best[2] = {0, 0}, vy[2] = {0, 0}, vx[2] = {0, 0};
  int y0 = by * bs, x0 = bx * bs;
  for (int c = 0; c < 9; c++) {
    int dy = cand[c][0], dx = cand[c][1];
    int16_t yy[2] = {(int16_t)(pvy + dy), (int16_t)(nvy - dy)};
    int16_t xx[2] = {(int16_t)(pvx + dx), (int16_t)(nvx - dx)};
    int err[2] = {0, 0};
    for (int py = y0 - bord; py < y0 + bs + bord; py++)
      for (int px = x0 - bord; px < x0 + bs + bord; px++) {
        int pv = BP(pr, py, px);
        err[0] += abs(pv - BP(r0, py + yy[0], px + xx[0]));
        err[1] += abs(pv - BP(r1, py + yy[1], px + xx[1]));
      }
    for (int d = 0; d < 2; d++)
      if (c == 0 || err[d] <= best[d]) { best[d] = err[d]; vy[d] = yy[d]; vx[d] = xx[d]; }
  }
  MV(mv, 1, by, bx) = (int16_t)vy[0]; MV(mv, 0, by, bx) = (int16_t)vx[0];
  MV(mv, 3, by, bx) = (int16_t)vy[1]; MV(mv, 2, by, bx) = (int16_t)vx[1];
}
static void me_search(mvfield *mv, const bplane *r0, const bplane *r1, const bplane *pr, int bs, int bord, int nby, int nbx) {
  for (int by = 0; by < nby; by++) /* local_me_for_image, motion_estimate.cpp:196-225 */
    for (int bx = 0; bx < nbx; bx++) me_block(mv, r0, r1, pr, bs, bord, by, bx);
}

static int ilog2_rint(int s) { return (int)rint(log((double)s) / log(2.0)); }

/* me_for_image (motion_estimate.cpp:260-412), FAST_SEARCH path, subpixel_accuracy = 0:
 * literal replay (analyse L levels, search, then per level synthesise / expand MVs / search). */
static void me_picture(mvfield *mv, bplane *r0, bplane *r1, bplane *pr, int H, int W, int bs, int S, int bord) {
  int L = ilog2_rint(S) - 1;
  int BY = mv->BY, BX = mv->BX;
  int La = L < 0 ? 0 : L;
  qo_dwt53t_analyze(r0->px, r0->stride, H, W, L);
  qo_dwt53t_analyze(r1->px, r1->stride, H, W, L);
  qo_dwt53t_analyze(pr->px, pr->stride, H, W, L);
  me_search(mv, r0, r1, pr, bs, bord, desp(BY, La), desp(BX, La));
  for (int l = L - 1; l >= 0; --l) {
    int Yl = desp(H, l), Xl = desp(W, l), byl = desp(BY, l), bxl = desp(BX, l);
    qo_dwt53t_synthesize(r0->px, r0->stride, Yl, Xl, 1);
    qo_dwt53t_synthesize(r1->px, r1->stride, Yl, Xl, 1);
    qo_dwt53t_synthesize(pr->px, pr->stride, Yl, Xl, 1);
    for (int k = 0; k < 4; k++) qo_haar_synthesize(mv->f[k], BX, byl, bxl, 1); /* :314-317 */
    for (int by = 0; by < byl; by++)
      for (int bx = 0; bx < bxl; bx++)
        for (int k = 0; k < 4; k++) { /* :321-348 */
          int16_t v = (int16_t)(MV(mv, k, by, bx) * 2);
          if (v > S) v = (int16_t)S;
          if (v < -S) v = (int16_t)-S;
          MV(mv, k, by, bx) = v;
        }
    me_search(mv, r0, r1, pr, bs, bord, byl, bxl);
  }
}

static size_t frame_bytes(int W, int H) { return (size_t)W * H + 2 * (size_t)(W / 2) * (H / 2); }

/* motion_estimate main loop (motion_estimate.cpp:714-907).  `low` holds P pictures; even[i] is
 * picture 2i, odd[i] picture 2i+1 (split.cpp:226-341).  mv_out: [P/2][4][BY][BX]. */
int qo_motion_estimate_b(const uint8_t *low, int P, int W, int H, int bs, int S, int bord, int16_t *mv_out) {
  if (bs <= 0 || S <= 0 || W < 2 || H < 2 || bord < 0) return QO_EINVAL;
  int BY = H / bs, BX = W / bs, nB = P / 2;
  size_t F = frame_bytes(W, H), fld = (size_t)BY * BX;
  bplane r[2], pr;
  if (bplane_alloc(&r[0], H, W, S + bord) || bplane_alloc(&r[1], H, W, S + bord) || bplane_alloc(&pr, H, W, S + bord)) return QO_EINVAL;
  mvfield mv; mv.BY = BY; mv.BX = BX;
  int a = 0;
  bplane_load_u8(&r[a], low, H, W);
  bplane_fill_border(&r[a]);
  for (int i = 0; i < nB; i++) {
    bplane_load_u8(&pr, low + (size_t)(2 * i + 1) * F, H, W);
    bplane_load_u8(&r[1 - a], low + (size_t)(2 * i + 2) * F, H, W);
    bplane_fill_border(&r[1 - a]);
    for (int k = 0; k < 4; k++) { mv.f[k] = mv_out + ((size_t)i * 4 + k) * fld; memset(mv.f[k], 0, fld * sizeof(int16_t)); }
    me_picture(&mv, &r[a], &r[1 - a], &pr, H, W, bs, S, bord);
    a = 1 - a; /* swap (:902-906): the new PREV reference keeps whatever state the replay left */
  }
  bplane_free(&r[0]); bplane_free(&r[1]); bplane_free(&pr);
  return QO_OK;
}

int qo_motion_estimate(const uint8_t *low, int P, int W, int H, int bs, int S, int16_t *mv_out) {
  return qo_motion_estimate_b(low, P, W, H, bs, S, 0, mv_out);
}

/* ------------------------------------------------------------------------------------------ */
/* entropy() (entropy.cpp:20-34): float accumulation, log in double, index order. */
float qo_entropy(const int *count, int n) {
  float e = 0.0f;
  int total = 0;
  for (int i = 0; i < n; i++) total += count[i];
  for (int i = 0; i < n; i++)
    if (count[i]) {
      float prob = (float)count[i] / total;
      e += prob * (float)(log(prob) / log(2.0));
    }
  return -e;
}

/* The I/B rule of decorrelate (decorrelate.cpp:798-819,945-979) from the three histograms.
 * Returns 'I' or 'B'.  Exported: the product's host code has its own copy of this rule; the
 * tests compare the two. */
int qo_frame_type(const int *hist_odd, const int *hist_res, const int *hist_mv, int H, int W, int BY, int BX,
                  int always_B, int *sizes /* nullable [3] */) {
  float motion_entropy = 0.0f, residue_entropy = 0.0f, predicted_entropy = 1.0f;
  if (!always_B) {
    motion_entropy = qo_entropy(hist_mv, 256);
    predicted_entropy = qo_entropy(hist_odd, 256);
    residue_entropy = qo_entropy(hist_res, 256);
  }
  int predicted_size = (int)(predicted_entropy * (float)H * (float)W);
  int residue_size = (int)(residue_entropy * (float)H * (float)W);
  int motion_size = (int)(motion_entropy * (float)BY * (float)BX);
  if (sizes) { sizes[0] = predicted_size; sizes[1] = residue_size; sizes[2] = motion_size; }
  return predicted_size <= residue_size + motion_size ? 'I' : 'B';
}

/* Chroma x2 interpolation (decorrelate.cpp:610-648, update.cpp:472-503): half-size plane in the
 * top-left quadrant of a zeroed HxW plane, one level of synthesis. */
static void chroma_up(int16_t *px, int stride, const uint8_t *src, int H, int W) {
  for (int y = 0; y < H; y++) memset(px + (size_t)y * stride, 0, sizeof(int16_t) * W);
  for (int y = 0; y < H / 2; y++)
    for (int x = 0; x < W / 2; x++) px[(size_t)y * stride + x] = src[(size_t)y * (W / 2) + x];
  qo_dwt53t_synthesize(px, stride, H, W, 1);
}

/* decorrelate main loop (decorrelate.cpp:583-1075) with predict() (decorrelate.cpp:69-189),
 * block_overlaping = 0, subpixel_accuracy = 0.
 *   high      [P/2] frames I420 u8            (out)
 *   types     [P/2] 'I'/'B'                   (out)
 *   mv_out    [P/2][4][BY][BX]                (out; zeros for 'I', decorrelate.cpp:1000)
 *   prediction[P/2] frames I420 u8, nullable  (out; the prediction_<even_fn> debug dump)
 *   hists     [P/2][3][256] int, nullable     (out; odd luma, residue+128 luma, mv+128)
 */
/* Shared by decorrelate and correlate (the reference compiles both from decorrelate.cpp, #if defined ANALYZE).
 * ev(k) = even picture k, od(i) = odd picture i (analysis only).  synth: high/types are INPUTS and odd_out is written
 * (decorrelate.cpp:1029-1068). */
static int decor_run(int synth, const uint8_t *ev_base, size_t ev_step, const uint8_t *od_base, size_t od_step, int P,
                     int W, int H, int bs, int S, int always_B, const int16_t *mv_in, uint8_t *high, char *types,
                     int16_t *mv_out, uint8_t *prediction, int *hists, uint8_t *odd_out, int ov) {
  if (bs <= 0 || S <= 0 || W < 4 || H < 4 || ov < 0) return QO_EINVAL;
  int BY = H / bs, BX = W / bs, nB = P / 2, b = 4 * S + ov; /* decorrelate.cpp:539 */
  /* --block_overlaping (predict(), decorrelate.cpp:84-172): every block's prediction is taken with a margin of `ov` samples,
   * analysed `levels` = rint(log2 ov) levels (truncating 5/3, in place), its subbands are copied without their margin into
   * the picture's Mallat layout, and the picture is synthesised back.  ov = 1 gives levels = 0: the plain block copy. */
  int levels = ov > 0 ? (int)rint(log((double)ov) / log(2.0)) : 0;
  int nb_ = bs + 2 * ov;
  int16_t *blk = (int16_t *)calloc((size_t)nb_ * nb_, sizeof(int16_t));
  int cw = W / 2, ch = H / 2;
  size_t F = frame_bytes(W, H), fld = (size_t)BY * BX;
  bplane ref[2][3];
  for (int i = 0; i < 2; i++)
    for (int c = 0; c < 3; c++)
      if (bplane_alloc(&ref[i][c], H, W, b)) return QO_EINVAL;
  /* prediction planes persist across pictures and are never cleared (Q7): rows/cols beyond
   * BY*bs / BX*bs keep whatever the previous picture's chroma analysis left there. */
  int16_t *pred[3];
  for (int c = 0; c < 3; c++) pred[c] = (int16_t *)calloc((size_t)H * W, sizeof(int16_t));
  int a = 0;
  const uint8_t *f0 = ev_base;
  bplane_load_u8(&ref[a][0], f0, H, W);
  chroma_up(ref[a][1].px, ref[a][1].stride, f0 + (size_t)W * H, H, W);
  chroma_up(ref[a][2].px, ref[a][2].stride, f0 + (size_t)W * H + (size_t)cw * ch, H, W);
  for (int c = 0; c < 3; c++) bplane_fill_border(&ref[a][c]);

  for (int i = 0; i < nB; i++) {
    const uint8_t *odd = synth ? NULL : od_base + (size_t)i * od_step;
    const uint8_t *ev = ev_base + (size_t)(i + 1) * ev_step;
    int n = 1 - a;
    bplane_load_u8(&ref[n][0], ev, H, W);
    chroma_up(ref[n][1].px, ref[n][1].stride, ev + (size_t)W * H, H, W);
    chroma_up(ref[n][2].px, ref[n][2].stride, ev + (size_t)W * H + (size_t)cw * ch, H, W);
    for (int c = 0; c < 3; c++) bplane_fill_border(&ref[n][c]);

    const int16_t *mvx0 = mv_in + ((size_t)i * 4 + 0) * fld, *mvy0 = mv_in + ((size_t)i * 4 + 1) * fld;
    const int16_t *mvx1 = mv_in + ((size_t)i * 4 + 2) * fld, *mvy1 = mv_in + ((size_t)i * 4 + 3) * fld;
    int hmv[256], hodd[256], hres[256];
    memset(hmv, 0, sizeof hmv); memset(hodd, 0, sizeof hodd); memset(hres, 0, sizeof hres);
    if (!always_B && !synth)
      for (size_t k = 0; k < fld; k++) { /* decorrelate.cpp:805-812 */
        hmv[(mvy0[k] + 128) & 255]++; hmv[(mvx0[k] + 128) & 255]++;
        hmv[(mvy1[k] + 128) & 255]++; hmv[(mvx1[k] + 128) & 255]++;
      }
    /* predict(): decorrelate.cpp:86-172 */
    for (int c = 0; c < 3; c++) {
      for (int by = 0; by < BY; by++)
        for (int bx = 0; bx < BX; bx++) {
          size_t k = (size_t)by * BX + bx;
          int y0 = mvy0[k] + by * bs, y1 = mvy1[k] + by * bs, x0 = mvx0[k] + bx * bs, x1 = mvx1[k] + bx * bs;
          for (int y = -ov; y < bs + ov; y++)
            for (int x = -ov; x < bs + ov; x++)
              blk[(size_t)(y + ov) * nb_ + x + ov] = (int16_t)((BP(&ref[a][c], y0 + y, x0 + x) + BP(&ref[n][c], y1 + y, x1 + x)) / 2);
          qo_dwt53t_analyze(blk, nb_, nb_, nb_, levels); /* :110-113 */
          for (int l = 1; l <= levels; l++) { /* :116-146: names as the reference's comments */
            int bsl = bs >> l, o1 = ov >> l, o3 = (bs + ov * 3) >> l;
            for (int y = 0; y < bsl; y++)
              for (int x = 0; x < bsl; x++) {
                pred[c][(size_t)(by * bsl + y) * W + (W >> l) + bx * bsl + x] = blk[(size_t)(o1 + y) * nb_ + o3 + x];
                pred[c][(size_t)((H >> l) + by * bsl + y) * W + bx * bsl + x] = blk[(size_t)(o3 + y) * nb_ + o1 + x];
                pred[c][(size_t)((H >> l) + by * bsl + y) * W + (W >> l) + bx * bsl + x] = blk[(size_t)(o3 + y) * nb_ + o3 + x];
              }
          }
          {
            int bsl = bs >> levels, o1 = ov >> levels;
            for (int y = 0; y < bsl; y++)
              for (int x = 0; x < bsl; x++) pred[c][(size_t)(by * bsl + y) * W + bx * bsl + x] = blk[(size_t)(o1 + y) * nb_ + o1 + x];
          }
        }
      qo_dwt53t_synthesize(pred[c], W, H, W, levels); /* :172 */
    }
    for (int c = 0; c < 3; c++) /* clip, decorrelate.cpp:841-848 */
      for (size_t k = 0; k < (size_t)H * W; k++) {
        if (pred[c][k] < 0) pred[c][k] = 0; else if (pred[c][k] > 255) pred[c][k] = 255;
      }
    qo_dwt53t_analyze(pred[1], W, H, W, 1); /* chroma /2, decorrelate.cpp:860-861 */
    qo_dwt53t_analyze(pred[2], W, H, W, 1);
    if (prediction) { /* texture::write: plain short -> u8 conversion (Q4), decorrelate.cpp:868-870 */
      uint8_t *o = prediction + (size_t)i * F;
      for (size_t k = 0; k < (size_t)H * W; k++) o[k] = (uint8_t)pred[0][k];
      o += (size_t)H * W;
      for (int c = 1; c < 3; c++, o += (size_t)cw * ch)
        for (int y = 0; y < ch; y++)
          for (int x = 0; x < cw; x++) o[(size_t)y * cw + x] = (uint8_t)pred[c][(size_t)y * W + x];
    }
    uint8_t *hi = high + (size_t)i * F;
    if (synth) { /* decorrelate.cpp:1029-1068: 'I' -> the stored picture, 'B' -> clip(residue + prediction) */
      uint8_t *o = odd_out + (size_t)i * F;
      const uint8_t *h = hi;
      for (int c = 0; c < 3; c++) {
        int hh = c ? ch : H, ww = c ? cw : W;
        for (int y = 0; y < hh; y++)
          for (int x = 0; x < ww; x++) {
            int val = h[(size_t)y * ww + x] - 128;
            if (types[i] == 'I') val += 128;
            else { val += pred[c][(size_t)y * W + x]; if (val < 0) val = 0; else if (val > 255) val = 255; }
            o[(size_t)y * ww + x] = (uint8_t)val;
          }
        o += (size_t)hh * ww; h += (size_t)hh * ww;
      }
      a = n;
      continue;
    }
    /* residue (decorrelate.cpp:920-929) + luma histograms (:945-950) */
    {
      const uint8_t *o = odd; uint8_t *h = hi;
      for (int c = 0; c < 3; c++) {
        int hh = c ? ch : H, ww = c ? cw : W;
        for (int y = 0; y < hh; y++)
          for (int x = 0; x < ww; x++) {
            int val = o[(size_t)y * ww + x] - pred[c][(size_t)y * W + x];
            if (val < -128) val = -128; else if (val > 127) val = 127;
            if (c == 0 && !always_B) { hodd[o[(size_t)y * ww + x]]++; hres[val + 128]++; }
            h[(size_t)y * ww + x] = (uint8_t)(val + 128); /* :1011-1019: +128, clip is a no-op */
          }
        o += (size_t)hh * ww; h += (size_t)hh * ww;
      }
    }
    int t = qo_frame_type(hodd, hres, hmv, H, W, BY, BX, always_B, 0);
    types[i] = (char)t;
    if (t == 'I') { /* decorrelate.cpp:982-1000 */
      memcpy(hi, odd, F);
      memset(mv_out + (size_t)i * 4 * fld, 0, 4 * fld * sizeof(int16_t));
    } else {
      memcpy(mv_out + (size_t)i * 4 * fld, mv_in + (size_t)i * 4 * fld, 4 * fld * sizeof(int16_t));
    }
    if (hists) {
      memcpy(hists + ((size_t)i * 3 + 0) * 256, hodd, sizeof hodd);
      memcpy(hists + ((size_t)i * 3 + 1) * 256, hres, sizeof hres);
      memcpy(hists + ((size_t)i * 3 + 2) * 256, hmv, sizeof hmv);
    }
    a = n;
  }
  for (int i = 0; i < 2; i++)
    for (int c = 0; c < 3; c++) bplane_free(&ref[i][c]);
  for (int c = 0; c < 3; c++) free(pred[c]);
  free(blk);
  return QO_OK;
}

int qo_decorrelate_ov(const uint8_t *low, int P, int W, int H, int bs, int S, int always_B, const int16_t *mv_in,
                      uint8_t *high, char *types, int16_t *mv_out, uint8_t *prediction, int *hists, int ov) {
  size_t F = frame_bytes(W, H);
  return decor_run(0, low, 2 * F, low + F, 2 * F, P, W, H, bs, S, always_B, mv_in, high, types, mv_out, prediction, hists, 0, ov);
}
int qo_decorrelate(const uint8_t *low, int P, int W, int H, int bs, int S, int always_B, const int16_t *mv_in,
                   uint8_t *high, char *types, int16_t *mv_out, uint8_t *prediction, int *hists) {
  return qo_decorrelate_ov(low, P, W, H, bs, S, always_B, mv_in, high, types, mv_out, prediction, hists, 0);
}
/* correlate (decorrelate.cpp without ANALYZE; synthesize_step.py:117-136): even [P/2+1] frames (what un_update wrote),
 * high/types/mv as decoded -> odd_out [P/2] frames. */
int qo_correlate_ov(const uint8_t *even, int P, int W, int H, int bs, int S, const int16_t *mv, const uint8_t *high,
                    const char *types, uint8_t *odd_out, uint8_t *prediction, int ov) {
  return decor_run(1, even, frame_bytes(W, H), 0, 0, P, W, H, bs, S, 1, mv, (uint8_t *)high, (char *)types, 0, prediction, 0,
                   odd_out, ov);
}
int qo_correlate(const uint8_t *even, int P, int W, int H, int bs, int S, const int16_t *mv, const uint8_t *high,
                 const char *types, uint8_t *odd_out, uint8_t *prediction) {
  return qo_correlate_ov(even, P, W, H, bs, S, mv, high, types, odd_out, prediction, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* update (update.cpp:50-148, main :446-679).                                                 */

static int clipi(int x, int dim) { return x < 0 ? 0 : (x >= dim ? dim - 1 : x); } /* update.cpp:50-54 */

static void upd_apply(int16_t *E, const int16_t *res, const int16_t *mvx, const int16_t *mvy, int H, int W, int bs,
                      int BY, int BX, float uf, int undo) {
  for (int by = 0; by < BY; by++)
    for (int bx = 0; bx < BX; bx++) {
      int dy = mvy[(size_t)by * BX + bx], dx = mvx[(size_t)by * BX + bx];
      for (int y = 0; y < bs; y++)
        for (int x = 0; x < bs; x++) {
          int16_t *t = &E[(size_t)clipi(by * bs + y + dy, H) * W + clipi(bx * bs + x + dx, W)];
          float aux = (float)*t;
          if (undo) aux -= res[(size_t)(by * bs + y) * W + bx * bs + x] * uf; /* update.cpp:100-104 */
          else aux += res[(size_t)(by * bs + y) * W + bx * bs + x] * uf;
          if (aux > 255) aux = 255; else if (aux < 0) aux = 0;
          *t = (int16_t)aux;
        }
    }
}

/* update / un_update share update.cpp (#ifdef ANALYZE at :100-104,:466-478,:625-641): src picture k at
 * src + k*src_step.  low_out: [P/2+1] frames.  high/mv/types are the decorrelate outputs (motion_filtered). */
static int upd_run(int undo, const uint8_t *low, size_t src_step, int P, int W, int H, int bs, const uint8_t *high,
                   const int16_t *mv, const char *types, float uf, uint8_t *low_out) {
  if (bs <= 0 || W < 4 || H < 4) return QO_EINVAL;
  int BY = H / bs, BX = W / bs, nB = P / 2, cw = W / 2, ch = H / 2;
  size_t F = frame_bytes(W, H), fld = (size_t)BY * BX, N = (size_t)H * W;
  int16_t *E[2][3], *res[3];
  for (int i = 0; i < 2; i++) for (int c = 0; c < 3; c++) E[i][c] = (int16_t *)calloc(N, sizeof(int16_t));
  /* Q3: chroma residue planes are full-size, only the top-left quadrant is ever written
   * (update.cpp:512-543 with UPDATE_STEP undefined); the rest is fresh-heap zero. */
  for (int c = 0; c < 3; c++) res[c] = (int16_t *)calloc(N, sizeof(int16_t));
#define LOAD_EVEN(dst, f)                                                         \
  do {                                                                            \
    const uint8_t *f_ = (f);                                                      \
    for (size_t k = 0; k < N; k++) (dst)[0][k] = f_[k];                           \
    chroma_up((dst)[1], W, f_ + N, H, W);                                         \
    chroma_up((dst)[2], W, f_ + N + (size_t)cw * ch, H, W);                       \
  } while (0)
#define STORE_LOW(src, o)                                                         \
  do {                                                                            \
    uint8_t *o_ = (o);                                                            \
    qo_dwt53t_analyze((src)[1], W, H, W, 1);                                      \
    qo_dwt53t_analyze((src)[2], W, H, W, 1);                                      \
    for (size_t k = 0; k < N; k++) o_[k] = (uint8_t)(src)[0][k];                  \
    o_ += N;                                                                      \
    for (int c = 1; c < 3; c++, o_ += (size_t)cw * ch)                            \
      for (int y = 0; y < ch; y++)                                                \
        for (int x = 0; x < cw; x++) o_[(size_t)y * cw + x] = (uint8_t)(src)[c][(size_t)y * W + x]; \
  } while (0)
  int a = 0;
  LOAD_EVEN(E[a], low);
  for (int i = 0; i < nB; i++) {
    const uint8_t *h = high + (size_t)i * F;
    for (size_t k = 0; k < N; k++) res[0][k] = (int16_t)(h[k] - 128);
    for (int c = 1; c < 3; c++) {
      const uint8_t *hc = h + N + (size_t)(c - 1) * cw * ch;
      for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) res[c][(size_t)y * W + x] = (int16_t)(hc[(size_t)y * cw + x] - 128);
    }
    LOAD_EVEN(E[1 - a], low + (size_t)(i + 1) * src_step);
    if (types[i] == 'B') {
      const int16_t *m = mv + (size_t)i * 4 * fld;
      /* update.cpp:85-146: per source pixel PREV target then NEXT target; the two targets are
       * different pictures, so per picture the order is simply (c, by, bx, y, x). */
      for (int c = 0; c < 3; c++) {
        upd_apply(E[a][c], res[c], m + 0 * fld, m + 1 * fld, H, W, bs, BY, BX, uf, undo);
        upd_apply(E[1 - a][c], res[c], m + 2 * fld, m + 3 * fld, H, W, bs, BY, BX, uf, undo);
      }
    }
    STORE_LOW(E[a], low_out + (size_t)i * F);
    a = 1 - a;
  }
  STORE_LOW(E[a], low_out + (size_t)nB * F);
  for (int i = 0; i < 2; i++) for (int c = 0; c < 3; c++) free(E[i][c]);
  for (int c = 0; c < 3; c++) free(res[c]);
  return QO_OK;
}

int qo_update(const uint8_t *low, int P, int W, int H, int bs, const uint8_t *high, const int16_t *mv,
              const char *types, float uf, uint8_t *low_out) {
  return upd_run(0, low, 2 * frame_bytes(W, H), P, W, H, bs, high, mv, types, uf, low_out);
}
/* un_update (synthesize_step.py:96-113): low_t [P/2+1] frames -> even_t [P/2+1] frames. */
int qo_un_update(const uint8_t *low_t, int P, int W, int H, int bs, const uint8_t *high, const int16_t *mv,
                 const char *types, float uf, uint8_t *even_out) {
  return upd_run(1, low_t, frame_bytes(W, H), P, W, H, bs, high, mv, types, uf, even_out);
}

/* One whole temporal level = split + motion_estimate + decorrelate + update
 * (analyze_step.py:117-232).  Convenience for the cpu_baseline timing leg. */
int qo_analyze_step(const uint8_t *low, int P, int W, int H, int bs, int S, int always_B, float uf, int16_t *mv_raw,
                    int16_t *mv_filtered, uint8_t *high, char *types, uint8_t *low_out) {
  int r = qo_motion_estimate(low, P, W, H, bs, S, mv_raw);
  if (r) return r;
  r = qo_decorrelate(low, P, W, H, bs, S, always_B, mv_raw, high, types, mv_filtered, 0, 0);
  if (r) return r;
  return qo_update(low, P, W, H, bs, high, mv_filtered, types, uf, low_out);
}
