// qsvc_amd/csrc/mctf_subpel.hip -- --subpixel_accuracy = a > 0.
//
// MOTION SEARCH (me_for_image, motion_estimate.cpp:361-412; planes of :716-735, border of :779-782, :817).  After the whole-pixel
// pyramid the reference doubles the vectors and repeats its +-1 search a times on planes synthesised one level up each time.
// The planes are allocated at the final size (H << a) x (W << a) with border (S + border_size) << a, but the picture is read and
// its border filled at the ORIGINAL size, so for the reference planes the three other quadrants of the first synthesis hold the
// replicate bands of width b = S + border_size (the "border" of the small picture lies inside the big plane) -- they act as
// high-pass coefficients -- and zeros; the predicted plane is never border-filled: its up-sampling is a pure interpolation.
// What a displaced window reads outside the up-sampled picture is whatever the plane holds there: zeros of a fresh heap, the
// band of the small picture's top / left border, and -- left of column 0 on the rows whose pointer texture::alloc
// (texture.cpp:34-46) does not shift, rows >= (H << a) - (b << a) -- the four shorts of glibc's chunk size field (SpGeom::c0, c1)
// with zeros before them.  sp_at() is that rule (derived from, and checked against, the literal heap model of
// oracle/mctf_subpel_oracle.c, which equals the reference's tools under default malloc).
//
// PREDICTION (decorrelate.cpp:534-566, 656-686, 828-861; also correlate).  Clean: every component is interpolated a times with zero
// high bands, the border is filled at the up-sampled size (Q1 / Q2 of texture.cpp at that size: border_map of the big geometry),
// predict() copies blocks of (bs << a) with the vectors as they are, and the prediction comes back down through a analysis
// levels (+ one for chroma).  Only the low-low band of each level is ever needed.
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"
#include "mctf_me_ring.h"

namespace qsvc {

// ---- motion search -----------------------------------------------------------------------------------------------------------
struct SpPlane {
  const int16_t* p;      // [Ha][Wa] interior
  const uint8_t* luma0;  // the picture the border was filled from, or nullptr: a predicted plane (never filled)
};
// one cell of a plane of the sub-pixel search, inside or outside the interior
__device__ __forceinline__ int sp_at(const SpGeom& g, const SpPlane& pl, int y, int x) {
  if ((unsigned)y < (unsigned)g.Ha && (unsigned)x < (unsigned)g.Wa) return pl.p[(size_t)y * g.Wa + x];
  if (x < 0 && y >= g.Ha - g.ba) {  // unshifted rows: left of the row's own memory
    if (x >= -4) return x == -4 ? g.c0 : (x == -3 ? g.c1 : 0);
    return 0;                       // slack, and cells of the previous row nothing ever wrote
  }
  if (!pl.luma0) return 0;
  // what texture::fill_border(H, W, b) wrote outside the big interior: the top band and the left band of the small picture
  if (y < 0) {
    if (y < -g.b || x < -g.b || x >= g.W + g.b) return 0;
    return pl.luma0[x < 0 ? 0 : (x >= g.W ? g.W - 1 : x)];
  }
  if (x < 0) {
    if (x < -g.b || y >= g.H + g.b) return 0;
    return y < g.H ? pl.luma0[(size_t)y * g.W] : pl.luma0[(size_t)(g.H - 1) * g.W + g.W - 1];  // Q2: the corner holds the bottom-RIGHT pixel
  }
  return 0;  // right of / below the interior: never written
}

// picture (and, for reference planes, the bands fill_border leaves inside the big interior: regions 5, 7, 8) into zeroed planes.
// src16: the level-0 state of the literal replay (Q5 geometries, mctf_replay.hip), same slot numbering, or nullptr: the u8 luma.
__global__ void __launch_bounds__(256) k_sp_load(const uint8_t* frames, size_t F, SpGeom g, int16_t* dst, const int16_t* src16, int slot0,
                                                  int slot_step, int pic0, int pic_step, int is_ref) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  const int ext = is_ref ? g.b : 0;
  if (x >= g.W + ext || y >= g.H + ext) return;
  const uint8_t* s = frames + (size_t)(pic0 + k * pic_step) * F;
  const int slot = slot0 + k * slot_step;
  int v;
  if (y < g.H && x < g.W) v = src16 ? src16[(size_t)slot * g.H * g.W + (size_t)y * g.W + x] : s[(size_t)y * g.W + x];
  else v = s[(size_t)imin(y, g.H - 1) * g.W + imin(x, g.W - 1)];
  dst[(size_t)slot * g.Ha * g.Wa + (size_t)y * g.Wa + x] = (int16_t)v;
}
void launch_sp_load(hipStream_t st, const uint8_t* frames, size_t F, const SpGeom& g, int16_t* dst, const int16_t* src16, int slot0,
                    int slot_step, int pic0, int pic_step, int count, int is_ref) {
  if (count <= 0) return;
  const int ext = is_ref ? g.b : 0;
  hipLaunchKernelGGL(k_sp_load, dim3((g.W + ext + 63) / 64, (g.H + ext + 3) / 4, count), dim3(64, 4), 0, st, frames, F, g, dst, src16, slot0,
                     slot_step, pic0, pic_step, is_ref);
}

// One sub-pixel search level l: vectors doubled and clamped to +-(S << a) (motion_estimate.cpp:375-398), then local_me_for_image
// with blocks of bs << l and border_size >> l on the planes synthesised to (H << l) x (W << l).  One wave per block.
__global__ void __launch_bounds__(64) k_sp_search(SpGeom g, const uint8_t* frames, size_t F, const int16_t* planes, int BY, int BX, int bsl,
                                                   int bordl, int lim, const int16_t* mv_in, int16_t* mv_out) {
  const int bx = blockIdx.x, by = blockIdx.y, i = blockIdx.z, lane = threadIdx.x;
  const size_t N = (size_t)g.Ha * g.Wa, fld = (size_t)BY * BX;
  SpPlane r0 = {planes + (size_t)(3 * i) * N, frames + (size_t)(2 * i) * F};
  SpPlane pr = {planes + (size_t)(3 * i + 1) * N, nullptr};
  SpPlane r1 = {planes + (size_t)(3 * i + 2) * N, frames + (size_t)(2 * i + 2) * F};
  const int16_t* m = mv_in + (size_t)i * 4 * fld + (size_t)by * BX + bx;
  const int pvx = iclamp((int16_t)(m[0] * 2), -lim, lim), pvy = iclamp((int16_t)(m[fld] * 2), -lim, lim);
  const int nvx = iclamp((int16_t)(m[2 * fld] * 2), -lim, lim), nvy = iclamp((int16_t)(m[3 * fld] * 2), -lim, lim);
  const int cand[9][2] = QSVC_ME_CANDS;
  int sad[18];
#pragma unroll
  for (int k = 0; k < 18; k++) sad[k] = 0;
  const int win = bsl + 2 * bordl, y0 = by * bsl - bordl, x0 = bx * bsl - bordl;
  const bool inner = y0 >= 0 && x0 >= 0 && y0 + win <= g.Ha && x0 + win <= g.Wa &&
                     y0 + pvy - 1 >= 0 && y0 + pvy + win + 1 <= g.Ha && x0 + pvx - 1 >= 0 && x0 + pvx + win + 1 <= g.Wa &&
                     y0 + nvy - 1 >= 0 && y0 + nvy + win + 1 <= g.Ha && x0 + nvx - 1 >= 0 && x0 + nvx + win + 1 <= g.Wa;
  for (int idx = lane; idx < win * win; idx += 64) {
    const int py = y0 + idx / win, px = x0 + idx % win;
    if (inner) {
      const int pv = pr.p[(size_t)py * g.Wa + px];
#pragma unroll
      for (int c = 0; c < 9; c++) {
        const int d0 = pv - r0.p[(size_t)(py + pvy + cand[c][0]) * g.Wa + px + pvx + cand[c][1]];
        const int d1 = pv - r1.p[(size_t)(py + nvy - cand[c][0]) * g.Wa + px + nvx - cand[c][1]];
        sad[2 * c] += d0 < 0 ? -d0 : d0;
        sad[2 * c + 1] += d1 < 0 ? -d1 : d1;
      }
    } else {
      const int pv = sp_at(g, pr, py, px);
      for (int c = 0; c < 9; c++) {
        const int d0 = pv - sp_at(g, r0, py + pvy + cand[c][0], px + pvx + cand[c][1]);
        const int d1 = pv - sp_at(g, r1, py + nvy - cand[c][0], px + nvx - cand[c][1]);
        sad[2 * c] += d0 < 0 ? -d0 : d0;
        sad[2 * c + 1] += d1 < 0 ? -d1 : d1;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 18; k++) {
    int v = sad[k];
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    sad[k] = v;
  }
  if (lane == 0) {
    int16_t* o = mv_out + (size_t)i * 4 * fld + (size_t)by * BX + bx;
    me_pick(sad, pvy, pvx, nvy, nvx, &o[0], &o[fld], &o[2 * fld], &o[3 * fld]);
  }
}
void launch_sp_search(hipStream_t st, const SpGeom& g, const uint8_t* frames, size_t F, const int16_t* planes, int nB, int BY, int BX, int bsl,
                      int bordl, int lim, const int16_t* mv_in, int16_t* mv_out) {
  if (nB <= 0) return;
  hipLaunchKernelGGL(k_sp_search, dim3(BX, BY, nB), dim3(64), 0, st, g, frames, F, planes, BY, BX, bsl, bordl, lim, mv_in, mv_out);
}

// ---- prediction --------------------------------------------------------------------------------------------------------------
// level 0 of the up-sampled references: [n][3][H][W] u8 = luma as stored, chroma x2 (decorrelate.cpp:610-648)
__global__ void __launch_bounds__(256) k_sp_refs0(const uint8_t* frames, LevelGeom g, int mul, uint8_t* dst) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, pic = blockIdx.z / 3, c = blockIdx.z % 3;
  if (x >= g.W || y >= g.H) return;
  const int v = c == 0 ? frame_y(frames, g, mul * pic)[(size_t)y * g.W + x] : chroma_up(frame_c(frames, g, mul * pic, c), y, x, g.H, g.W);
  dst[((size_t)blockIdx.z * g.H + y) * g.W + x] = (uint8_t)v;
}
// one interpolation step (decorrelate.cpp:658-676): the plane as the LL band of a (2h) x (2w) synthesis with zero high bands
__global__ void __launch_bounds__(256) k_sp_up2(const uint8_t* src, uint8_t* dst, int h, int w) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= 2 * w || y >= 2 * h) return;
  dst[((size_t)blockIdx.z * 2 * h + y) * 2 * w + x] = (uint8_t)chroma_up(src + (size_t)blockIdx.z * h * w, y, x, 2 * h, 2 * w);
}
// up: [count][3][H << a][W << a]; tmp: as large.  Returns the buffer that holds the result.
uint8_t* launch_sp_refs(hipStream_t st, const uint8_t* frames, const LevelGeom& g, int mul, int count, int a, uint8_t* up, uint8_t* tmp) {
  uint8_t* cur = (a & 1) ? tmp : up;  // after a ping-pong steps the result lands in `up`
  const dim3 block(64, 4);
  hipLaunchKernelGGL(k_sp_refs0, dim3((g.W + 63) / 64, (g.H + 3) / 4, count * 3), block, 0, st, frames, g, mul, cur);
  for (int s = 0; s < a; s++) {
    uint8_t* nxt = cur == up ? tmp : up;
    const int h = g.H << s, w = g.W << s;
    hipLaunchKernelGGL(k_sp_up2, dim3((2 * w + 63) / 64, (2 * h + 3) / 4, count * 3), block, 0, st, cur, nxt, h, w);
    cur = nxt;
  }
  return cur;
}
// predict() without overlap at the up-sampled resolution (decorrelate.cpp:93-108,153-167): ge = the up-sampled geometry
// (H << a, W << a, bs << a, S << a, ov << a); up = [nB+1][3][ge.H][ge.W]; ppl = [nB][3][ge.H][ge.W] int16
// one: only B picture `one` (>= 0), into the three planes at ppl, and what predict() does not write (rows / columns beyond the block
// grid, Q7) is LEFT AS IT IS -- the persistent prediction planes of a ragged picture (launch_sp_predict_one); one < 0: all pictures,
// plane (i, c) at ppl + (3 i + c) planes.
__global__ void __launch_bounds__(256) k_sp_predict(LevelGeom ge, const uint8_t* up, const int16_t* mv, int16_t* ppl, int one) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, i = one >= 0 ? one : blockIdx.z / 3, c = blockIdx.z % 3;
  if (x >= ge.W || y >= ge.H) return;
  const size_t plane = (size_t)ge.H * ge.W, fld = (size_t)ge.BY * ge.BX;
  const int by = y / ge.bs, bx = x / ge.bs;
  int p = 0;
  if (by < ge.BY && bx < ge.BX) {
    const BorderGeom bg = make_border(ge.H, ge.W, 4 * ge.S + ge.ov);
    const int16_t* m = mv + (size_t)i * 4 * fld + (size_t)by * ge.BX + bx;
    const uint8_t* r0 = up + ((size_t)i * 3 + c) * plane;
    const uint8_t* r1 = up + ((size_t)(i + 1) * 3 + c) * plane;
    p = (luma_b(r0, bg, y + m[fld], x + m[0]) + luma_b(r1, bg, y + m[3 * fld], x + m[2 * fld])) >> 1;
  } else if (one >= 0) return;
  ppl[((size_t)(one >= 0 ? 0 : i) * 3 + c) * plane + (size_t)y * ge.W + x] = (int16_t)p;
}
void launch_sp_predict(hipStream_t st, const LevelGeom& ge, const uint8_t* up, const int16_t* mv, int16_t* ppl) {
  hipLaunchKernelGGL(k_sp_predict, dim3((ge.W + 63) / 64, (ge.H + 3) / 4, ge.nB * 3), dim3(64, 4), 0, st, ge, up, mv, ppl, -1);
}
// Ragged pictures (Q7 at the up-sampled size): the reference's three prediction planes are allocated once per level and never
// cleared; predict() writes the block grid's area, the whole plane is clipped, analysed IN PLACE (a levels, chroma one more), and
// what lies beyond the block grid keeps that state for the next picture.  Literal, one B picture after the other:
//   pp   [3][Ha][Wa] int16, zero before the level's first picture (a fresh heap), carried from picture to picture
//   out  [3][H][W] compact: the planes' top-left H x W after the a levels (chroma before its extra level: launch_pred_finish takes it)
__global__ void __launch_bounds__(256) k_sp_clip(int16_t* p, size_t n) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n) { const int v = p[k]; p[k] = (int16_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
}
__global__ void __launch_bounds__(256) k_sp_extract(const int16_t* pp, int16_t* out, int H, int W, int Ha, int Wa) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, c = blockIdx.z;
  if (x < W && y < H) out[((size_t)c * H + y) * W + x] = pp[((size_t)c * Ha + y) * Wa + x];
}
void launch_sp_predict_one(hipStream_t st, const LevelGeom& ge, const uint8_t* up, const int16_t* mv, int i, int16_t* pp, int16_t* tmp, int H, int W,
                           int a, int16_t* out) {
  const size_t N = (size_t)ge.H * ge.W;
  hipLaunchKernelGGL(k_sp_predict, dim3((ge.W + 63) / 64, (ge.H + 3) / 4, 3), dim3(64, 4), 0, st, ge, up, mv, pp, i);
  hipLaunchKernelGGL(k_sp_clip, dim3((unsigned)((3 * N + 255) / 256)), dim3(256), 0, st, pp, 3 * N);                        // decorrelate.cpp:841-848
  for (int k = 0; k < a; k++) launch_rp_dwt(st, pp, tmp, N, 3, ge.W, ge.H >> k, ge.W >> k, 0);                                // :852-857
  hipLaunchKernelGGL(k_sp_extract, dim3((W + 63) / 64, (H + 3) / 4, 3), dim3(64, 4), 0, st, pp, out, H, W, ge.H, ge.W);
  launch_rp_dwt(st, pp + N, tmp + N, N, 2, ge.W, H, W, 0);                                                                   // :860-861 (the state the next picture finds)
}
// the low-low band of one analysis level (decorrelate.cpp:852-857) of `count` compact ny x nx planes -> compact (ny/2) x (nx/2)
// planes; clip: the planes are clipped to 0..255 first (decorrelate.cpp:841-848, before the first level only)
__global__ void __launch_bounds__(256) k_sp_ll(const int16_t* src, int16_t* dst, int ny, int nx, int clip) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (i >= nx / 2 || j >= ny / 2) return;
  const int16_t* s = src + (size_t)blockIdx.z * ny * nx;
  auto P = [&](int yy, int xx) { const int v = s[(size_t)yy * nx + xx]; return clip ? iclamp(v, 0, 255) : v; };
  dst[((size_t)blockIdx.z * (ny / 2) + j) * (nx / 2) + i] = (int16_t)analyze_ll(P, j, i, ny, nx);
}
// a levels: ppl ([count][H << a][W << a]) -> [count][H][W]; the two buffers are used alternately.  Returns the result's buffer.
int16_t* launch_sp_down(hipStream_t st, int16_t* ppl, int16_t* tmp, int count, int H, int W, int a) {
  int16_t* cur = ppl;
  for (int s = a; s >= 1; s--) {
    int16_t* nxt = cur == ppl ? tmp : ppl;
    const int ny = H << s, nx = W << s;
    hipLaunchKernelGGL(k_sp_ll, dim3((nx / 2 + 63) / 64, (ny / 2 + 3) / 4, count), dim3(64, 4), 0, st, cur, nxt, ny, nx, s == a ? 1 : 0);
    cur = nxt;
  }
  return cur;
}

}  // namespace qsvc
