/*
 * oracle/mctf_subpel_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product path.
 *
 * CPU restatement of the reference's `--subpixel_accuracy > 0` paths: the sub-pixel half of me_for_image
 * (motion_estimate.cpp:361-412 with the plane set-up of :716-735, :779-782, :817) and the decorrelate / correlate side
 * (decorrelate.cpp:534-566, :656-686, :760-788, :828-861).  Paths relative to /root/reference/trunk/src.
 *
 * Why this file looks different from mctf_oracle.c: with sub-pixel accuracy `a` the reference allocates its planes at the
 * up-sampled size (H<<a) x (W<<a) with border b<<a, but motion_estimate reads the picture and fills the border at the ORIGINAL
 * size, and texture::alloc (texture.cpp:34-46) shifts only the first (H<<a) of the (H<<a) + 2(b<<a) row pointers.  What the
 * search then reads outside the picture is partly the replicate border of the small picture, partly never-written heap, partly
 * -- left of column 0 on the rows that were not shifted -- the tail of the previous row's malloc chunk and glibc's chunk size
 * field.  Under default malloc all of that is deterministic (fresh heap pages are zero, the chunk header is a constant of the
 * row length; tests/golden/subpel_probe.json records that the outputs repeat, and that they change under MALLOC_PERTURB_),
 * so instead of closed-form rules this file keeps a LITERAL MODEL OF THE HEAP IMAGE of each plane and performs the reference's
 * operations on it in the reference's order:
 *
 *   a plane = (Hd + 2b) rows malloc'ed back to back, each a glibc chunk of align16(2 (Wd + 2b) + 8) bytes whose last 8 bytes
 *   before the next user pointer are that next chunk's size field (size | PREV_INUSE); data[y][x] of logical row y addresses
 *   user(y + b) + (y + b < Hd ? b : 0) + x, wherever that lands.
 *
 * Parity pin: tests/test_subpel_oracle.py compares every function here with the reference's own tools compiled into oracle/_ref
 * (default malloc) on fresh clips, and tests/golden/n4_subpel_*.npz hold what those tools produced.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define QO_OK 0
#define QO_EINVAL (-1)

void qo_dwt53t_analyze(int16_t *plane, int stride, int H, int W, int levels);
void qo_dwt53t_synthesize(int16_t *plane, int stride, int H, int W, int levels);
void qo_haar_synthesize(int16_t *plane, int stride, int H, int W, int levels);
int qo_frame_type(const int *hist_odd, const int *hist_res, const int *hist_mv, int H, int W, int BY, int BX, int always_B, int *sizes);

/* ---- heap image of a texture::alloc plane (texture.cpp:34-46) ------------------------------------------------------------ */
typedef struct {
  int Hd, Wd, b; /* dimensions it was allocated with */
  int rc;        /* shorts per chunk */
  int16_t *mem;  /* (Hd + 2b) chunks; chunk p: 4 shorts of size field, then the row's user memory */
} hplane;

static int hp_alloc(hplane *p, int Hd, int Wd, int b) {
  long req = 2L * (Wd + 2 * b);
  long chunk = (req + 8 + 15) & ~15L; /* request2size */
  if (chunk < 32) chunk = 32;
  p->Hd = Hd; p->Wd = Wd; p->b = b; p->rc = (int)(chunk / 2);
  size_t rows = (size_t)Hd + 2 * (size_t)b;
  p->mem = (int16_t *)calloc(rows * p->rc + 16, sizeof(int16_t)); /* fresh heap: zero pages */
  if (!p->mem) return -1;
  for (size_t r = 0; r < rows; r++) { /* chunk header: size | PREV_INUSE, little endian */
    p->mem[r * p->rc + 0] = (int16_t)((chunk | 1) & 0xFFFF);
    p->mem[r * p->rc + 1] = (int16_t)(((chunk | 1) >> 16) & 0xFFFF);
  }
  return 0;
}
static void hp_free(hplane *p) { free(p->mem); p->mem = 0; }
#define HP(p, y, x) ((p)->mem[(size_t)((y) + (p)->b) * (p)->rc + 4 + (((y) + (p)->b) < (p)->Hd ? (p)->b : 0) + (x)])

/* texture::read (texture.cpp:122-129) */
static void hp_read_u8(hplane *p, const uint8_t *src, int h, int w) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) HP(p, y, x) = src[(size_t)y * w + x];
}
/* texture::fill_border (texture.cpp:55-113), the eight regions in the reference's order */
static void hp_fill_border(hplane *p, int H, int W, int b) {
  for (int y = -b; y < 0; y++) for (int x = -b; x < 0; x++) HP(p, y, x) = HP(p, 0, 0);
  for (int y = -b; y < 0; y++) for (int x = 0; x < W; x++) HP(p, y, x) = HP(p, 0, x);
  for (int y = -b; y < 0; y++) for (int x = W; x < W + b; x++) HP(p, y, x) = HP(p, 0, W - 1);
  for (int y = 0; y < H; y++) for (int x = -b; x < 0; x++) HP(p, y, x) = HP(p, y, 0);
  for (int y = 0; y < H; y++) for (int x = W; x < W + b; x++) HP(p, y, x) = HP(p, y, W - 1);
  for (int y = H; y < H + b; y++) for (int x = -b; x < 0; x++) HP(p, y, x) = HP(p, H - 1, W - 1); /* Q2 */
  for (int y = H; y < H + b; y++) for (int x = 0; x < W; x++) HP(p, y, x) = HP(p, H - 1, x);
  for (int y = H; y < H + b; y++) for (int x = W; x < W + b; x++) HP(p, y, x) = HP(p, H - 1, W - 1);
}
/* dwt2d::analyze / synthesize (dwt2d.cpp:76-175) on the top-left ny x nx of a plane: through a flat copy */
static void hp_dwt(hplane *p, int ny, int nx, int levels, int inverse, int16_t *tmp) {
  if (levels <= 0) return;
  for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) tmp[(size_t)y * nx + x] = HP(p, y, x);
  if (inverse) qo_dwt53t_synthesize(tmp, nx, ny, nx, levels); else qo_dwt53t_analyze(tmp, nx, ny, nx, levels);
  for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) HP(p, y, x) = tmp[(size_t)y * nx + x];
}

static int desp(int x, int l) { for (int i = 0; i < l; i++) x = (x + 1) / 2; return x; }
static int ilog2_rint(int s) { return (int)rint(log((double)s) / log(2.0)); }

typedef struct { int BY, BX; int16_t *f[4]; } mvfield; /* PREV-X, PREV-Y, NEXT-X, NEXT-Y */
#define MV(m, k, by, bx) ((m)->f[k][(size_t)(by) * (m)->BX + (bx)])

/* local_me_for_block / local_me_for_image (motion_estimate.cpp:70-225) on heap-image planes */
static void me_search(mvfield *mv, const hplane *r0, const hplane *r1, const hplane *pr, int bs, int bord, int nby, int nbx) {
  static const int cand[9][2] = {{-1, -1}, {-1, 1}, {1, -1}, {1, 1}, {-1, 0}, {1, 0}, {0, 1}, {0, -1}, {0, 0}};
  for (int by = 0; by < nby; by++)
    for (int bx = 0; bx < nbx; bx++) {
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
            int pv = HP(pr, py, px);
            err[0] += abs(pv - HP(r0, py + yy[0], px + xx[0]));
            err[1] += abs(pv - HP(r1, py + yy[1], px + xx[1]));
          }
        for (int d = 0; d < 2; d++)
          if (c == 0 || err[d] <= best[d]) { best[d] = err[d]; vy[d] = yy[d]; vx[d] = xx[d]; }
      }
      MV(mv, 1, by, bx) = (int16_t)vy[0]; MV(mv, 0, by, bx) = (int16_t)vx[0];
      MV(mv, 3, by, bx) = (int16_t)vy[1]; MV(mv, 2, by, bx) = (int16_t)vx[1];
    }
}

/* me_for_image (motion_estimate.cpp:260-412), FAST_SEARCH path, any subpixel_accuracy */
static void me_picture(mvfield *mv, hplane *r0, hplane *r1, hplane *pr, int H, int W, int bs, int S, int bord, int a, int16_t *tmp) {
  int L = ilog2_rint(S) - 1, La = L < 0 ? 0 : L, BY = mv->BY, BX = mv->BX;
  hp_dwt(r0, H, W, L, 0, tmp); hp_dwt(r1, H, W, L, 0, tmp); hp_dwt(pr, H, W, L, 0, tmp);
  me_search(mv, r0, r1, pr, bs, bord, desp(BY, La), desp(BX, La));
  for (int l = L - 1; l >= 0; --l) {
    int Yl = desp(H, l), Xl = desp(W, l), byl = desp(BY, l), bxl = desp(BX, l);
    hp_dwt(r0, Yl, Xl, 1, 1, tmp); hp_dwt(r1, Yl, Xl, 1, 1, tmp); hp_dwt(pr, Yl, Xl, 1, 1, tmp);
    for (int k = 0; k < 4; k++) qo_haar_synthesize(mv->f[k], BX, byl, bxl, 1);
    for (int by = 0; by < byl; by++)
      for (int bx = 0; bx < bxl; bx++)
        for (int k = 0; k < 4; k++) {
          int16_t v = (int16_t)(MV(mv, k, by, bx) * 2);
          if (v > S) v = (int16_t)S;
          if (v < -S) v = (int16_t)-S;
          MV(mv, k, by, bx) = v;
        }
    me_search(mv, r0, r1, pr, bs, bord, byl, bxl);
  }
  for (int l = 1; l <= a; l++) { /* :361-406 */
    hp_dwt(r0, H << l, W << l, 1, 1, tmp); hp_dwt(r1, H << l, W << l, 1, 1, tmp); hp_dwt(pr, H << l, W << l, 1, 1, tmp);
    const int lim = S << a;
    for (int by = 0; by < BY; by++)
      for (int bx = 0; bx < BX; bx++)
        for (int k = 0; k < 4; k++) {
          int16_t v = (int16_t)(MV(mv, k, by, bx) * 2);
          if (v > lim) v = (int16_t)lim;
          if (v < -lim) v = (int16_t)-lim;
          MV(mv, k, by, bx) = v;
        }
    me_search(mv, r0, r1, pr, bs << l, bord >> l, BY, BX);
  }
  /* "the images as they were left to the next search" (:408-411) */
  hp_dwt(r0, H << a, W << a, a, 0, tmp); hp_dwt(r1, H << a, W << a, a, 0, tmp); hp_dwt(pr, H << a, W << a, a, 0, tmp);
}

static size_t frame_bytes(int W, int H) { return (size_t)W * H + 2 * (size_t)(W / 2) * (H / 2); }

/* motion_estimate main (motion_estimate.cpp:714-907) with subpixel_accuracy = a.  mv_out: [P/2][4][BY][BX]. */
int qo_motion_estimate_sub(const uint8_t *low, int P, int W, int H, int bs, int S, int bord, int a, int16_t *mv_out) {
  if (bs <= 0 || S <= 0 || W < 2 || H < 2 || bord < 0 || a < 0 || a > 4) return QO_EINVAL;
  int BY = H / bs, BX = W / bs, nB = P / 2, Ha = H << a, Wa = W << a, b = S + bord, ba = b << a;
  size_t F = frame_bytes(W, H), fld = (size_t)BY * BX;
  hplane r[2], pr;
  if (hp_alloc(&r[0], Ha, Wa, ba) || hp_alloc(&r[1], Ha, Wa, ba) || hp_alloc(&pr, Ha, Wa, ba)) return QO_EINVAL;
  int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)Ha * Wa);
  for (int i = 0; i < 2; i++) /* "this initialization seems not necessary" (:723-728, :737-741) */
    for (int y = 0; y < Ha; y++) for (int x = 0; x < Wa; x++) HP(&r[i], y, x) = 0;
  for (int y = 0; y < Ha; y++) for (int x = 0; x < Wa; x++) HP(&pr, y, x) = 0;
  mvfield mv; mv.BY = BY; mv.BX = BX;
  int cur = 0;
  hp_read_u8(&r[cur], low, H, W);
  hp_fill_border(&r[cur], H, W, b); /* ORIGINAL size and border (:779-782) */
  for (int i = 0; i < nB; i++) {
    hp_read_u8(&pr, low + (size_t)(2 * i + 1) * F, H, W);
    for (int y = 0; y < Ha; y++) for (int x = 0; x < Wa; x++) HP(&r[1 - cur], y, x) = 0; /* :803-807 */
    hp_read_u8(&r[1 - cur], low + (size_t)(2 * i + 2) * F, H, W);
    hp_fill_border(&r[1 - cur], H, W, b);
    for (int k = 0; k < 4; k++) { mv.f[k] = mv_out + ((size_t)i * 4 + k) * fld; memset(mv.f[k], 0, fld * sizeof(int16_t)); }
    me_picture(&mv, &r[cur], &r[1 - cur], &pr, H, W, bs, S, bord, a, tmp);
    cur = 1 - cur;
  }
  free(tmp); hp_free(&r[0]); hp_free(&r[1]); hp_free(&pr);
  return QO_OK;
}

/* ---- decorrelate / correlate with subpixel_accuracy = a (decorrelate.cpp:534-1075) --------------------------------------- */
/* reference[.][c] of one even picture: component at its stored size, chroma x2 (:610-648), then a interpolation steps with the
 * three other quadrants zeroed (:656-676), then the border at the up-sampled size (:679-684) */
static void load_reference(hplane ref[3], const uint8_t *frame, int W, int H, int a, int ba, int16_t *tmp) {
  int cw = W / 2, ch = H / 2;
  hp_read_u8(&ref[0], frame, H, W);
  for (int c = 1; c < 3; c++) {
    hp_read_u8(&ref[c], frame + (size_t)W * H + (size_t)(c - 1) * cw * ch, ch, cw);
    for (int y = 0; y < H / 2; y++) for (int x = W / 2; x < W / 2 + W / 2; x++) HP(&ref[c], y, x) = 0;
    for (int y = H / 2; y < H; y++) for (int x = 0; x < W; x++) HP(&ref[c], y, x) = 0;
    hp_dwt(&ref[c], H, W, 1, 1, tmp);
  }
  for (int c = 0; c < 3; c++) {
    for (int s = 1; s <= a; s++) {
      int hs = H << s, ws = W << s;
      for (int y = 0; y < hs / 2; y++) for (int x = ws / 2; x < ws / 2 + ws / 2; x++) HP(&ref[c], y, x) = 0;
      for (int y = hs / 2; y < hs; y++) for (int x = 0; x < ws; x++) HP(&ref[c], y, x) = 0;
      hp_dwt(&ref[c], hs, ws, 1, 1, tmp);
    }
    hp_fill_border(&ref[c], H << a, W << a, ba);
  }
}

static int decor_sub(int synth, const uint8_t *ev_base, size_t ev_step, const uint8_t *od_base, size_t od_step, int P, int W, int H,
                     int bs0, int S, int always_B, const int16_t *mv_in, uint8_t *high, char *types, int16_t *mv_out,
                     uint8_t *prediction, int *hists, uint8_t *odd_out, int ov0, int a) {
  if (bs0 <= 0 || S <= 0 || W < 4 || H < 4 || ov0 < 0 || a < 0 || a > 4) return QO_EINVAL;
  int BY = H / bs0, BX = W / bs0, nB = P / 2, Ha = H << a, Wa = W << a, ba = (4 * S + ov0) << a; /* :539, :549-551 */
  int ov = ov0 << a, bs = bs0 << a;                                                              /* :828-829 */
  int levels = ov > 0 ? (int)rint(log((double)ov) / log(2.0)) : 0;
  int nb_ = (H / BY + ov0 * 2) << a; /* prediction_block (:532-537) */
  if (nb_ < bs + 2 * ov) nb_ = bs + 2 * ov;
  int16_t *blk = (int16_t *)calloc((size_t)nb_ * nb_, sizeof(int16_t));
  int cw = W / 2, ch = H / 2;
  size_t F = frame_bytes(W, H), fld = (size_t)BY * BX;
  hplane ref[2][3];
  for (int i = 0; i < 2; i++) for (int c = 0; c < 3; c++) if (hp_alloc(&ref[i][c], Ha, Wa, ba)) return QO_EINVAL;
  /* prediction planes (:563-567): allocated once at the up-sampled size, never cleared (Q7) */
  int16_t *pred[3];
  for (int c = 0; c < 3; c++) pred[c] = (int16_t *)calloc((size_t)Ha * Wa, sizeof(int16_t));
  int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)Ha * Wa);
  int cur = 0;
  load_reference(ref[cur], ev_base, W, H, a, ba, tmp);
  for (int i = 0; i < nB; i++) {
    const uint8_t *odd = synth ? NULL : od_base + (size_t)i * od_step;
    int n = 1 - cur;
    load_reference(ref[n], ev_base + (size_t)(i + 1) * ev_step, W, H, a, ba, tmp);
    const int16_t *mvx0 = mv_in + ((size_t)i * 4 + 0) * fld, *mvy0 = mv_in + ((size_t)i * 4 + 1) * fld;
    const int16_t *mvx1 = mv_in + ((size_t)i * 4 + 2) * fld, *mvy1 = mv_in + ((size_t)i * 4 + 3) * fld;
    int hmv[256], hodd[256], hres[256];
    memset(hmv, 0, sizeof hmv); memset(hodd, 0, sizeof hodd); memset(hres, 0, sizeof hres);
    if (!always_B && !synth)
      for (size_t k = 0; k < fld; k++) {
        hmv[(mvy0[k] + 128) & 255]++; hmv[(mvx0[k] + 128) & 255]++;
        hmv[(mvy1[k] + 128) & 255]++; hmv[(mvx1[k] + 128) & 255]++;
      }
    for (int c = 0; c < 3; c++) { /* predict() at the up-sampled resolution (:828-839 -> :69-189) */
      for (int by = 0; by < BY; by++)
        for (int bx = 0; bx < BX; bx++) {
          size_t k = (size_t)by * BX + bx;
          int y0 = mvy0[k] + by * bs, y1 = mvy1[k] + by * bs, x0 = mvx0[k] + bx * bs, x1 = mvx1[k] + bx * bs;
          for (int y = -ov; y < bs + ov; y++)
            for (int x = -ov; x < bs + ov; x++)
              blk[(size_t)(y + ov) * nb_ + x + ov] = (int16_t)((HP(&ref[cur][c], y0 + y, x0 + x) + HP(&ref[n][c], y1 + y, x1 + x)) / 2);
          qo_dwt53t_analyze(blk, nb_, bs + 2 * ov, bs + 2 * ov, levels);
          for (int l = 1; l <= levels; l++) {
            int bsl = bs >> l, o1 = ov >> l, o3 = (bs + ov * 3) >> l;
            for (int y = 0; y < bsl; y++)
              for (int x = 0; x < bsl; x++) {
                pred[c][(size_t)(by * bsl + y) * Wa + (Wa >> l) + bx * bsl + x] = blk[(size_t)(o1 + y) * nb_ + o3 + x];
                pred[c][(size_t)((Ha >> l) + by * bsl + y) * Wa + bx * bsl + x] = blk[(size_t)(o3 + y) * nb_ + o1 + x];
                pred[c][(size_t)((Ha >> l) + by * bsl + y) * Wa + (Wa >> l) + bx * bsl + x] = blk[(size_t)(o3 + y) * nb_ + o3 + x];
              }
          }
          {
            int bsl = bs >> levels, o1 = ov >> levels;
            for (int y = 0; y < bsl; y++)
              for (int x = 0; x < bsl; x++) pred[c][(size_t)(by * bsl + y) * Wa + bx * bsl + x] = blk[(size_t)(o1 + y) * nb_ + o1 + x];
          }
        }
      qo_dwt53t_synthesize(pred[c], Wa, Ha, Wa, levels);
    }
    for (int c = 0; c < 3; c++) /* :841-848 */
      for (size_t k = 0; k < (size_t)Ha * Wa; k++) { if (pred[c][k] < 0) pred[c][k] = 0; else if (pred[c][k] > 255) pred[c][k] = 255; }
    for (int c = 0; c < 3; c++) qo_dwt53t_analyze(pred[c], Wa, Ha, Wa, a); /* :852-857 */
    qo_dwt53t_analyze(pred[1], Wa, H, W, 1);                                 /* :860-861 */
    qo_dwt53t_analyze(pred[2], Wa, H, W, 1);
    if (prediction) {
      uint8_t *o = prediction + (size_t)i * F;
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) o[(size_t)y * W + x] = (uint8_t)pred[0][(size_t)y * Wa + x];
      o += (size_t)H * W;
      for (int c = 1; c < 3; c++, o += (size_t)cw * ch)
        for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) o[(size_t)y * cw + x] = (uint8_t)pred[c][(size_t)y * Wa + x];
    }
    uint8_t *hi = high + (size_t)i * F;
    if (synth) { /* :1029-1068 */
      uint8_t *o = odd_out + (size_t)i * F;
      const uint8_t *h = hi;
      for (int c = 0; c < 3; c++) {
        int hh = c ? ch : H, ww = c ? cw : W;
        for (int y = 0; y < hh; y++)
          for (int x = 0; x < ww; x++) {
            int val = h[(size_t)y * ww + x] - 128;
            if (types[i] == 'I') val += 128;
            else { val += pred[c][(size_t)y * Wa + x]; if (val < 0) val = 0; else if (val > 255) val = 255; }
            o[(size_t)y * ww + x] = (uint8_t)val;
          }
        o += (size_t)hh * ww; h += (size_t)hh * ww;
      }
      cur = n;
      continue;
    }
    {
      const uint8_t *o = odd; uint8_t *h = hi;
      for (int c = 0; c < 3; c++) {
        int hh = c ? ch : H, ww = c ? cw : W;
        for (int y = 0; y < hh; y++)
          for (int x = 0; x < ww; x++) {
            int val = o[(size_t)y * ww + x] - pred[c][(size_t)y * Wa + x];
            if (val < -128) val = -128; else if (val > 127) val = 127;
            if (c == 0 && !always_B) { hodd[o[(size_t)y * ww + x]]++; hres[val + 128]++; }
            h[(size_t)y * ww + x] = (uint8_t)(val + 128);
          }
        o += (size_t)hh * ww; h += (size_t)hh * ww;
      }
    }
    int t = qo_frame_type(hodd, hres, hmv, H, W, BY, BX, always_B, 0);
    types[i] = (char)t;
    if (t == 'I') {
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
    cur = n;
  }
  for (int i = 0; i < 2; i++) for (int c = 0; c < 3; c++) hp_free(&ref[i][c]);
  for (int c = 0; c < 3; c++) free(pred[c]);
  free(tmp); free(blk);
  return QO_OK;
}

int qo_decorrelate_sub(const uint8_t *low, int P, int W, int H, int bs, int S, int always_B, const int16_t *mv_in, uint8_t *high,
                       char *types, int16_t *mv_out, uint8_t *prediction, int *hists, int ov, int a) {
  size_t F = frame_bytes(W, H);
  return decor_sub(0, low, 2 * F, low + F, 2 * F, P, W, H, bs, S, always_B, mv_in, high, types, mv_out, prediction, hists, 0, ov, a);
}
int qo_correlate_sub(const uint8_t *even, int P, int W, int H, int bs, int S, const int16_t *mv, const uint8_t *high, const char *types,
                     uint8_t *odd_out, uint8_t *prediction, int ov, int a) {
  return decor_sub(1, even, frame_bytes(W, H), 0, 0, P, W, H, bs, S, 1, mv, (uint8_t *)high, (char *)types, 0, prediction, 0, odd_out, ov, a);
}
