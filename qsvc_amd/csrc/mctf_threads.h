// qsvc_amd/csrc/mctf_threads.h -- one function per kernel "thread body".
//
// The HIP kernels (mctf_kernels.hip) derive (x, y, z) from blockIdx/threadIdx and call these;
// the host loop harness (tests/emul/emul.cpp) calls the same bodies from nested for-loops.
// Layouts (all device-resident, see DESIGN.md "Data layout in HBM"):
//   frames ....... [P][F] u8 planar I420 (Y WxH, U, V (W/2)x(H/2)), F = W*H*3/2
//   snapshots .... [P][H][W] int16, level-l in-place (Mallat) analysis of every luma plane
//   motion ....... [nB][4][BY][BX] int16 (PREV-X, PREV-Y, NEXT-X, NEXT-Y)
//   histograms ... [nB][3][256] u32 (odd luma, residue+128 luma, mv+128)
#pragma once
#include "mctf_core.h"

namespace qsvc {

struct LevelGeom {
  int W, H;      // luma size (both even)
  int bs;        // block size
  int BY, BX;    // H/bs, W/bs (H % bs == 0 and W % bs == 0 on this path)
  int S;         // search range of this temporal level
  int P;         // pictures in low_{t-1}
  int nB;        // P/2 B pictures
  size_t F;      // bytes per I420 frame
  int ov;        // --block_overlaping (decorrelate.cpp:539: the reference planes' border is 4*S + ov)
  int bord;      // --border_size of the motion search (motion_estimate.cpp:213-217,714)
  int sub = 0;   // --subpixel_accuracy: the search planes are allocated at (H << sub) x (W << sub), every row of the picture keeps its
                 // shifted pointer and Q1 does not apply to the whole-pixel levels (mctf_subpel.hip)
};
// geometry of the sub-pixel search planes (mctf_subpel.hip)
struct SpGeom {
  int H, W;      // the picture
  int a;         // subpixel_accuracy
  int Ha, Wa;    // H << a, W << a: the planes' interior
  int b, ba;     // S + border_size: the border the picture was filled with; b << a: the border the planes were allocated with
  int c0, c1;    // the low / high 16 bits of a row chunk's size field, glibc: align16(2 (Wa + 2 ba) + 8) | PREV_INUSE
};

QHD const uint8_t* frame_y(const uint8_t* frames, const LevelGeom& g, int pic) { return frames + (size_t)pic * g.F; }
QHD const uint8_t* frame_c(const uint8_t* frames, const LevelGeom& g, int pic, int c /*1|2*/) {
  return frames + (size_t)pic * g.F + (size_t)g.W * g.H + (size_t)(c - 1) * (g.W / 2) * (g.H / 2);
}

// ---- ME pyramid: snapshot level l (>=1) from level l-1 --------------------------------------
// src: level l-1 planes (u8 luma of the frames when l == 1, int16 snapshots otherwise).
template <class T>
QHD void pyr_thread(const T* src, size_t src_pic_stride, int src_row_stride, int16_t* dst, size_t dst_pic_stride,
                    int H, int W, int l, int R, int C, int x, int y, int pic) {
  if (x >= C || y >= R) return;
  const T* s = src + (size_t)pic * src_pic_stride;
  int ny = H >> (l - 1), nx = W >> (l - 1);
  int v;
  if (y < ny && x < nx) {
    auto P = [&](int yy, int xx) { return (int)s[(size_t)yy * src_row_stride + xx]; };
    if (x >= 1 && x <= (nx >> 1) - 2 && y >= 1 && y <= (ny >> 1) - 2) v = analyze2d_ll_inner(P, y, x);
    else v = analyze2d_coef(P, y, x, ny, nx);
  } else {
    v = s[(size_t)y * src_row_stride + x];
  }
  dst[(size_t)pic * dst_pic_stride + (size_t)y * W + x] = (int16_t)v;
}

// ---- ME search at level l: one MC block per wave; `lane`/`nlanes` split the block's pixels ----
struct MeArgs {
  LevelGeom g;
  const uint8_t* frames;   // low_{t-1}
  const void* snap;        // level-l snapshots (int16), or nullptr: the u8 luma of the frames is read directly (level 0)
  size_t snap_pic_stride;  // elements
  int pic_mul = 2;         // B picture i searches snapshot planes pic_mul*i (PREV), +1 (predicted), +2 (NEXT): 2 = one plane per
                           // picture of low_{t-1}; 3 = one plane per ROLE (the literal replay of Q5 geometries, mctf_replay.hip,
                           // where a picture's state depends on the role it plays)
  const int16_t* mv_prev;  // level l+1 result [nB][4][BY][BX] (stride BX) or nullptr at the coarsest level
  int16_t* mv_out;         // level l result, same layout
  int l;                   // DWT level of this search
  int nby, nbx;            // desp(BY,l), desp(BX,l)
  // Fast path split (mctf_me.hip): blocks whose displaced windows lie inside the picture ("inner", me_inner) are searched by the
  // lane-per-8x8 kernel; the others can only lie outside the rectangle [rby_lo, rby_hi] x [rbx_lo, rbx_hi] of the block grid.
  // ring != 0: the border-rule kernels enumerate only that ring and skip its inner blocks.
  int ring;
  int rby_lo, rby_hi, rbx_lo, rbx_hi;
  unsigned* zero_words;    // mctf_me.hip: the level-0 launch also zeroes this many words (the level's histograms), or nullptr
  int zero_count;
};
// every sample the nine candidates of both directions read lies inside the picture (no border rule needed)
QHD bool me_inner(int H, int W, int bs, int y0, int x0, int pvy, int pvx, int nvy, int nvx) {
  return y0 + pvy - 1 >= 0 && y0 + pvy + bs + 1 <= H && x0 + pvx - 1 >= 0 && x0 + pvx + bs + 1 <= W &&
         y0 + nvy - 1 >= 0 && y0 + nvy + bs + 1 <= H && x0 + nvx - 1 >= 0 && x0 + nvx + bs + 1 <= W;
}
// ring enumeration: idx -> (by, bx); false past the end.  Rows above the rectangle, then the left/right parts of the rectangle's
// rows, then the rows below.
QHD int me_ring_count(const MeArgs& a) {
  const int mid = a.rby_hi - a.rby_lo + 1, per = a.rbx_lo + (a.nbx - 1 - a.rbx_hi);
  return a.rby_lo * a.nbx + mid * per + (a.nby - 1 - a.rby_hi) * a.nbx;
}
QHD bool me_ring_block(const MeArgs& a, int idx, int& by, int& bx) {
  const int ntop = a.rby_lo * a.nbx;
  if (idx < ntop) { by = idx / a.nbx; bx = idx % a.nbx; return true; }
  idx -= ntop;
  const int mid = a.rby_hi - a.rby_lo + 1, per = a.rbx_lo + (a.nbx - 1 - a.rbx_hi);
  if (per > 0 && idx < mid * per) {
    const int k = idx % per;
    by = a.rby_lo + idx / per;
    bx = k < a.rbx_lo ? k : a.rbx_hi + 1 + (k - a.rbx_lo);
    return true;
  }
  idx -= mid * per;
  by = a.rby_hi + 1 + idx / a.nbx; bx = idx % a.nbx;
  return by < a.nby;
}
template <class T>
QHD void me_block_setup(const MeArgs& a, int i, MePlane<T>& r0, MePlane<T>& r1, MePlane<T>& pr) {
  BorderGeom bg = make_border(a.g.H, a.g.W, a.g.S + a.g.bord);  // motion_estimate.cpp:714
  if (a.g.sub) bg.safe = 1 << 30;  // planes of the up-sampled size: the rows of the picture all have shifted pointers (no Q1)
  const uint8_t* y0 = frame_y(a.frames, a.g, 2 * i);
  const uint8_t* y1 = frame_y(a.frames, a.g, 2 * i + 2);
  const uint8_t* yp = frame_y(a.frames, a.g, 2 * i + 1);
  if (!a.snap) {
    r0.snap = (const T*)y0; r1.snap = (const T*)y1; pr.snap = (const T*)yp;
  } else {
    const T* s = (const T*)a.snap;
    r0.snap = s + (size_t)(a.pic_mul * i) * a.snap_pic_stride;
    r1.snap = s + (size_t)(a.pic_mul * i + 2) * a.snap_pic_stride;
    pr.snap = s + (size_t)(a.pic_mul * i + 1) * a.snap_pic_stride;
  }
  r0.stride = r1.stride = pr.stride = a.g.W;
  r0.luma0 = y0; r1.luma0 = y1; pr.luma0 = yp;
  r0.g = r1.g = pr.g = bg;
}
QHD void me_block_init_mv(const MeArgs& a, int i, int by, int bx, int& pvx, int& pvy, int& nvx, int& nvy) {
  pvx = pvy = nvx = nvy = 0;
  if (a.mv_prev) {
    size_t fld = (size_t)a.g.BY * a.g.BX;
    const int16_t* m = a.mv_prev + (size_t)i * 4 * fld + (size_t)(by >> 1) * a.g.BX + (bx >> 1);
    pvx = me_expand(m[0 * fld], a.g.S); pvy = me_expand(m[1 * fld], a.g.S);
    nvx = me_expand(m[2 * fld], a.g.S); nvy = me_expand(m[3 * fld], a.g.S);
  }
}
QHD void me_block_store(const MeArgs& a, int i, int by, int bx, const int* sad, int pvx, int pvy, int nvx, int nvy) {
  size_t fld = (size_t)a.g.BY * a.g.BX;
  int16_t* m = a.mv_out + (size_t)i * 4 * fld + (size_t)by * a.g.BX + bx;
  me_pick(sad, pvy, pvx, nvy, nvx, &m[0 * fld], &m[1 * fld], &m[2 * fld], &m[3 * fld]);
}

// ---- predict + residue (decorrelate) -------------------------------------------------------
struct PredArgs {
  LevelGeom g;
  const uint8_t* frames;  // low_{t-1}
  const int16_t* mv;      // raw motion [nB][4][BY][BX]
  uint8_t* high;          // [nB][F]  residue + 128
  uint8_t* pred_dump;     // [nB][F] or nullptr (prediction_<even_fn> debug picture)
  unsigned* hist;         // [nB][3][256]
  int16_t* margin;        // Q7: [nB][2][MarginGeom::count()] chroma margin state AFTER each picture; nullptr when the
                          // block grid tiles the picture
  // correlate (decorrelate.cpp without ANALYZE, :1029-1068): `high` and `types` are inputs and the odd picture 2i+1 of
  // `odd_out` (= the frames buffer itself: un_update put the even pictures at 2i) is written instead.
  uint8_t* odd_out;       // nullptr = analysis
  const uint8_t* types;   // [nB] 'I' / 'B' (synthesis only)
  const uint8_t* cup;     // [nB+1][2][H][W] u8: chroma x2 interpolation of every even picture (launch_chroma_up), or nullptr
  int16_t* mv_copy;       // mctf_pred.hip only: motion_filtered_t starts as a copy of motion_t (decorrelate.cpp:1005-1025), or nullptr
  int hist_is_zero;       // mctf_pred.hip only: the histograms were zeroed by the motion search's last launch (no memset needed)
  const uint8_t* up;      // --subpixel_accuracy with --block_overlaping >= 2 (k_ovl_block on the up-sampled geometry): [nB+1][3][H][W]
                          // u8 references of all three components, or nullptr
};
// what correlate stores for one sample: the picture itself for 'I', clip(residue + prediction) for 'B'
QHD int correlate_sample(int h, int p, int type) { return type == 'I' ? h : iclamp(h - 128 + p, 0, 255); }
QHD MarginGeom margin_geom(const LevelGeom& g) { return {g.H, g.W, g.BY * g.bs, g.BX * g.bs}; }
// luma: returns the two histogram bins through odd_bin / res_bin
QHD void pred_luma_thread(const PredArgs& a, int x, int y, int i, int& odd_bin, int& res_bin) {
  const LevelGeom& g = a.g;
  BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);  // decorrelate.cpp:539
  MvField mv = {a.mv + (size_t)i * 4 * g.BY * g.BX, g.BY, g.BX};
  int p = 0;  // Q7: the unpredicted margin of the luma plane stays zero
  if (y < g.BY * g.bs && x < g.BX * g.bs)
    p = predict_luma(frame_y(a.frames, g, 2 * i), frame_y(a.frames, g, 2 * i + 2), bg, mv, g.bs, y, x);
  size_t off = (size_t)i * g.F + (size_t)y * g.W + x;
  if (a.odd_out) {
    a.odd_out[(size_t)(2 * i + 1) * g.F + (size_t)y * g.W + x] = (uint8_t)correlate_sample(a.high[off], p, a.types[i]);
    if (a.pred_dump) a.pred_dump[off] = (uint8_t)p;
    odd_bin = res_bin = 0;
    return;
  }
  int o = frame_y(a.frames, g, 2 * i + 1)[(size_t)y * g.W + x];
  int val = iclamp(o - p, -128, 127);  // decorrelate.cpp:920-929
  a.high[off] = (uint8_t)(val + 128);
  if (a.pred_dump) a.pred_dump[off] = (uint8_t)p;
  odd_bin = o; res_bin = val + 128;
}
// one sample of picture i's full-resolution chroma prediction plane as the chroma/2 analysis sees it (after the clip)
QHD int pred_chroma_sample(const PredArgs& a, const uint8_t* u0, const uint8_t* u1, const BorderGeom& bg, const MvField& mv,
                           int i, int c, int yy, int xx) {
  if (!a.margin) return predict_chroma_full(u0, u1, bg, mv, a.g.bs, yy, xx);
  MarginGeom mg = margin_geom(a.g);
  if (mg.inside(yy, xx)) return predict_chroma_full(u0, u1, bg, mv, a.g.bs, yy, xx);
  if (i == 0) return 0;  // fresh planes
  return iclamp(a.margin[((size_t)(i - 1) * 2 + (c - 1)) * mg.count() + mg.index(yy, xx)], 0, 255);
}
// Q7: margin position k of picture i's analysed plane = the state picture i+1 starts from.  Pictures in order.
QHD void pred_margin_thread(const PredArgs& a, size_t k, int i, int c /*1|2*/) {
  const LevelGeom& g = a.g;
  BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  MvField mv = {a.mv + (size_t)i * 4 * g.BY * g.BX, g.BY, g.BX};
  const uint8_t* u0 = frame_c(a.frames, g, 2 * i, c);
  const uint8_t* u1 = frame_c(a.frames, g, 2 * i + 2, c);
  MarginGeom mg = margin_geom(g);
  int y, x;
  mg.coords(k, y, x);
  auto P = [&](int yy, int xx) { return pred_chroma_sample(a, u0, u1, bg, mv, i, c, yy, xx); };
  a.margin[((size_t)i * 2 + (c - 1)) * mg.count() + k] = (int16_t)analyze2d_coef(P, y, x, g.H, g.W);
}
QHD void pred_chroma_thread(const PredArgs& a, int ci, int cj, int i, int c /*1|2*/) {
  const LevelGeom& g = a.g;
  BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  MvField mv = {a.mv + (size_t)i * 4 * g.BY * g.BX, g.BY, g.BX};
  const uint8_t* u0 = frame_c(a.frames, g, 2 * i, c);
  const uint8_t* u1 = frame_c(a.frames, g, 2 * i + 2, c);
  auto P = [&](int yy, int xx) { return pred_chroma_sample(a, u0, u1, bg, mv, i, c, yy, xx); };
  int ll = analyze_ll(P, cj, ci, g.H, g.W);  // chroma /2 of the full-resolution prediction
  int cw = g.W / 2;
  size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * (g.H / 2) + (size_t)cj * cw + ci;
  size_t off = (size_t)i * g.F + coff;
  if (a.odd_out) {
    a.odd_out[(size_t)(2 * i + 1) * g.F + coff] = (uint8_t)correlate_sample(a.high[off], ll, a.types[i]);
    if (a.pred_dump) a.pred_dump[off] = (uint8_t)ll;
    return;
  }
  int o = frame_c(a.frames, g, 2 * i + 1, c)[(size_t)cj * cw + ci];
  int val = iclamp(o - ll, -128, 127);
  a.high[off] = (uint8_t)(val + 128);
  if (a.pred_dump) a.pred_dump[off] = (uint8_t)ll;  // texture::write: plain short -> u8 (Q4)
}

// ---- update ---------------------------------------------------------------------------------
struct UpdArgs {
  LevelGeom g;
  const uint8_t* frames;   // low_{t-1}
  const uint8_t* high;     // [nB][F] as decorrelate wrote it ('I' pictures hold the odd picture)
  const int16_t* mv;       // motion_filtered [nB][4][BY][BX]
  const uint8_t* types;    // [nB] 'I' / 'B'
  // Open-GOP neighbours on other GPUs (nullptr when absent): the B picture left of even picture 0
  // and right of even picture nB.  Each halo = one I420 frame (high) + [4][BY][BX] motion + type.
  const uint8_t* halo_high[2];
  const int16_t* halo_mv[2];
  int halo_type[2];        // 0 = no halo, else 'I' / 'B'
  uint8_t* low;            // [nB+1][F]
  float uf;                // update_factor; un_update (update.cpp:100-104 without ANALYZE) passes -update_factor:
                           // a - r*uf and a + r*(-uf) are the same float
  int reach;               // >= max |mv component| over every field used
  int src_mul, dst_mul;    // even picture j is frames[src_mul*j] and goes to low[dst_mul*j]: (2,1) update low_{t-1} -> low_t,
                           // (1,2) un_update low_t -> the even pictures of low_{t-1}
  const uint8_t* cup;      // [nB+1][2][H][W] u8: chroma x2 interpolation of even picture j (launch_chroma_up), or nullptr
};
struct UpdSide {  // one contribution to even picture j
  const uint8_t* high;  // frame base of the B picture, or nullptr
  const int16_t* mvx;
  const int16_t* mvy;
};
// Contributions to even picture j in the reference's order: first the NEXT field of B picture j-1,
// then the PREV field of B picture j (update.cpp:605-614 then the swap at :649-653).
QHD void upd_sides(const UpdArgs& a, int j, UpdSide s[2]) {
  const LevelGeom& g = a.g;
  size_t fld = (size_t)g.BY * g.BX;
  s[0].high = s[1].high = nullptr;
  s[0].mvx = s[0].mvy = s[1].mvx = s[1].mvy = nullptr;
  if (j >= 1) {
    if (a.types[j - 1] == 'B') {
      const int16_t* m = a.mv + (size_t)(j - 1) * 4 * fld;
      s[0].high = a.high + (size_t)(j - 1) * g.F; s[0].mvx = m + 2 * fld; s[0].mvy = m + 3 * fld;
    }
  } else if (a.halo_type[0] == 'B') {
    s[0].high = a.halo_high[0]; s[0].mvx = a.halo_mv[0] + 2 * fld; s[0].mvy = a.halo_mv[0] + 3 * fld;
  }
  if (j < g.nB) {
    if (a.types[j] == 'B') {
      const int16_t* m = a.mv + (size_t)j * 4 * fld;
      s[1].high = a.high + (size_t)j * g.F; s[1].mvx = m; s[1].mvy = m + fld;
    }
  } else if (a.halo_type[1] == 'B') {
    s[1].high = a.halo_high[1]; s[1].mvx = a.halo_mv[1]; s[1].mvy = a.halo_mv[1] + fld;
  }
}
QHD void upd_luma_thread(const UpdArgs& a, int x, int y, int j) {
  const LevelGeom& g = a.g;
  UpdSide s[2];
  upd_sides(a, j, s);
  int v = frame_y(a.frames, g, a.src_mul * j)[(size_t)y * g.W + x];
  for (int k = 0; k < 2; k++)
    if (s[k].high) {
      const uint8_t* h = s[k].high;
      auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * g.W + sx] - 128; };  // update.cpp:512-520
      v = update_gather(v, R, s[k].mvx, s[k].mvy, g.BY, g.BX, g.bs, g.H, g.W, g.H, g.W, a.reach, y, x, a.uf);
    }
  a.low[(size_t)(a.dst_mul * j) * g.F + (size_t)y * g.W + x] = (uint8_t)v;
}
// chroma: low sample (cj,ci) = LL of the 1-level analysis of the updated full-resolution plane
// (update.cpp:632-643).  Only targets within `reach` of the top-left quadrant can change (Q3); for
// the rest analysis(synthesis(u)) == u exactly, so the stored chroma is copied through.
QHD void upd_chroma_thread(const UpdArgs& a, int ci, int cj, int j, int c /*1|2*/) {
  const LevelGeom& g = a.g;
  int cw = g.W / 2, chh = g.H / 2;
  const uint8_t* u = frame_c(a.frames, g, a.src_mul * j, c);
  size_t off = (size_t)(a.dst_mul * j) * g.F + (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh + (size_t)cj * cw + ci;
  UpdSide s[2];
  upd_sides(a, j, s);
  bool untouched = (!s[0].high && !s[1].high) || 2 * cj - 2 >= chh + a.reach || 2 * ci - 2 >= cw + a.reach;
  if (untouched) { a.low[off] = u[(size_t)cj * cw + ci]; return; }
  size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh;
  auto E = [&](int yy, int xx) {
    int v = chroma_up(u, yy, xx, g.H, g.W);
    for (int k = 0; k < 2; k++)
      if (s[k].high) {
        const uint8_t* h = s[k].high + coff;
        auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * cw + sx] - 128; };
        v = update_gather(v, R, s[k].mvx, s[k].mvy, g.BY, g.BX, g.bs, g.H, g.W, chh, cw, a.reach, yy, xx, a.uf);
      }
    return v;
  };
  a.low[off] = (uint8_t)analyze_ll(E, cj, ci, g.H, g.W);  // texture::write wraps mod 256 (Q4)
}

}  // namespace qsvc
