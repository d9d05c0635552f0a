// qsvc_amd/csrc/mctf_kernels.hip -- gfx950 kernels of the MCTF half of `mctf compress`.
//
// Each kernel covers a WHOLE temporal level (all B pictures / all even pictures) in one launch;
// the reference runs one process per stage and a sequential 3-picture window inside it
// (motion_estimate.cpp:784-907, decorrelate.cpp:690-1075, update.cpp:506-656).  The per-sample
// arithmetic lives in mctf_core.h / mctf_threads.h (shared with the host loop harness).
//
// Wave = 64 lanes everywhere.  No MFMA: this is byte/integer streaming and gather work.
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"
#include "mctf_me_ring.h"
#include <stdlib.h>

namespace qsvc {


// ---------------------------------------------------------------------------------------------
// ME pyramid level: thread per coefficient, 64x4 tile per workgroup (row-contiguous stores).
template <class T>
__global__ void __launch_bounds__(256) k_pyr_level(const T* src, size_t src_pic_stride, int src_row_stride, int16_t* dst,
                                                    size_t dst_pic_stride, int H, int W, int l, int R, int C, int ox, int oy) {
  int x = ox + blockIdx.x * 64 + threadIdx.x;
  int y = oy + blockIdx.y * 4 + threadIdx.y;
  pyr_thread<T>(src, src_pic_stride, src_row_stride, dst, dst_pic_stride, H, W, l, R, C, x, y, blockIdx.z);
}
// What a block window reads beyond the LL quadrant the tile kernels (mctf_pyr.hip) write: rows [R, Rf) of all Cf columns, then columns
// [C, Cf) of the first R rows, one launch.  (Kept apart from the tile kernels: the general coefficient code is ~50 KB of instructions and
// evicts the tile loop from the instruction cache when both run on one CU.)
template <class T>
__global__ void __launch_bounds__(256) k_pyr_margin(const T* src, size_t src_pic_stride, int src_row_stride, int16_t* dst, size_t dst_pic_stride, int H, int W,
                                                     int l, int R, int C, int Rf, int Cf, int gxR, int nR, int gxC) {
  // (one linear block index over both roles: an empty workgroup still costs its dispatch, a common 2-D grid would launch 16 x as many)
  const int b = blockIdx.x;
  const bool rows = b < nR;
  const int q = rows ? b : b - nR, gx = rows ? gxR : gxC, by = q / gx, bx = q - by * gx;
  const int x = (rows ? 0 : C) + bx * 64 + threadIdx.x;
  const int y = (rows ? R : 0) + by * 4 + threadIdx.y;
  pyr_thread<T>(src, src_pic_stride, src_row_stride, dst, dst_pic_stride, H, W, l, rows ? Rf : R, Cf, x, y, blockIdx.z);
}
template <class T>
static void launch_pyr_margin(hipStream_t st, const T* src, size_t src_pic_stride, int16_t* dst, int W, int H, int P, int l, int R, int C, int Rf, int Cf) {
  const int gyR = Rf > R ? (Rf - R + 3) / 4 : 0, gxR = (Cf + 63) / 64, gyC = Cf > C ? (R + 3) / 4 : 0, gxC = Cf > C ? (Cf - C + 63) / 64 : 1;
  const int nR = gyR * gxR, nC = gyC * gxC;
  if (nR + nC == 0) return;
  hipLaunchKernelGGL(k_pyr_margin<T>, dim3(nR + nC, 1, P), dim3(64, 4), 0, st, src, src_pic_stride, W, dst, (size_t)H * W, H, W, l, R, C, Rf, Cf, gxR, nR, gxC);
}
void launch_pyr_level_u8(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int P, int R, int C) {
  dim3 block(64, 4);
  if (pyr_tile_applies(W, H) && tuning().pyr_tile) {
    // the LL quadrant out of LDS tiles (mctf_pyr.hip); what a block window reads beyond it, coefficient by coefficient
    const int Rl = imin(R, H / 2), Cl = imin(C, W / 2);
    launch_pyr_ll_tile(st, frames, F, W, H, dst, P, Rl, Cl);
    launch_pyr_margin<uint8_t>(st, frames, F, dst, W, H, P, 1, Rl, Cl, R, C);
    return;
  }
  dim3 grid((C + 63) / 64, (R + 3) / 4, P);
  hipLaunchKernelGGL(k_pyr_level<uint8_t>, grid, block, 0, st, frames, F, W, dst, (size_t)H * W, H, W, 1, R, C, 0, 0);
}
void launch_pyr_level_s16(hipStream_t st, const int16_t* src, int16_t* dst, int W, int H, int P, int l, int R, int C) {
  dim3 block(64, 4);
  const int ny = H >> (l - 1), nx = W >> (l - 1);
  if (pyr_tile16_applies(W, ny, nx) && tuning().pyr_tile) {
    // the LL quadrant out of LDS tiles (mctf_pyr.hip); what a block window reads beyond it (bands of this and of earlier levels), per coefficient
    const int Rl = imin(R, ny / 2), Cl = imin(C, nx / 2);
    launch_pyr_ll_tile16(st, src, dst, W, H, P, ny, nx, Rl, Cl);
    launch_pyr_margin<int16_t>(st, src, (size_t)H * W, dst, W, H, P, l, Rl, Cl, R, C);
    return;
  }
  dim3 grid((C + 63) / 64, (R + 3) / 4, P);
  hipLaunchKernelGGL(k_pyr_level<int16_t>, grid, block, 0, st, src, (size_t)H * W, W, dst, (size_t)H * W, H, W, l, R, C, 0, 0);
}

// stand-alone kernels over the border-rule bodies (mctf_me_ring.h): every case the piece kernel does not take
template <class T, int LPB>
__global__ void __launch_bounds__(64) k_me_search(MeArgs a) { me_search_body<T, LPB>(a, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x); }
template <class T>
__global__ void __launch_bounds__(64) k_me_search8(MeArgs a) { me_search8_body<T>(a, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x); }
// mv_bound: no component of the vectors this level starts from exceeds it (0 at the coarsest level)
void launch_me_search(hipStream_t st, const MeArgs& a0, int mv_bound) {
  MeArgs a = a0;
  a.ring = 0;
  if (!(me_piece_applies(a) && tuning().me_piece)) a.zero_words = nullptr;  // only the piece launch zeroes
  a.rby_lo = a.rby_hi = a.rbx_lo = a.rbx_hi = 0;
  if (me_piece_applies(a) && tuning().me_piece) {
    // one launch: inner blocks (one lane per 8x8 samples, no border rule) and, as a second role of the same grid, the ring of the
    // block grid where a displaced window can leave the picture (border-rule bodies)
    const int bs = a.g.bs, b1 = mv_bound + 1;
    a.ring = 1;
    a.rby_lo = (b1 + bs - 1) / bs; a.rby_hi = imin(a.nby - 1, (a.g.H - b1) / bs - 1);
    a.rbx_lo = (b1 + bs - 1) / bs; a.rbx_hi = imin(a.nbx - 1, (a.g.W - b1) / bs - 1);
    if (a.rby_hi < a.rby_lo || a.rbx_hi < a.rbx_lo) { a.rby_lo = a.nby; a.rby_hi = a.nby - 1; a.rbx_lo = 0; a.rbx_hi = a.nbx - 1; }  // all ring
    launch_me_piece(st, a, me_ring_count(a));
    return;
  }
  dim3 block(64);
  if (a.g.bs == 8 && a.g.bord == 0) {
    dim3 grid((a.nbx + 7) / 8, a.nby, a.g.nB);
    if (!a.snap) hipLaunchKernelGGL(k_me_search8<uint8_t>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_me_search8<int16_t>, grid, block, 0, st, a);
  } else if (a.g.bs * a.g.bs <= 64) {
    dim3 grid((a.nbx + 3) / 4, a.nby, a.g.nB);
    if (!a.snap) hipLaunchKernelGGL((k_me_search<uint8_t, 16>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_me_search<int16_t, 16>), grid, block, 0, st, a);
  } else {
    dim3 grid(a.nbx, a.nby, a.g.nB);
    if (!a.snap) hipLaunchKernelGGL((k_me_search<uint8_t, 64>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_me_search<int16_t, 64>), grid, block, 0, st, a);
  }
}

// ---------------------------------------------------------------------------------------------
// predict + residue, luma.  Fast form (block size a multiple of 8): a thread owns 8 consecutive pixels of one row --
// they share one MC block, so the two displaced reference runs are two unaligned 8-byte loads when they lie inside
// the picture (the common case); runs that touch the replicate border fall back to the per-pixel border rule
// (Q1/Q2).  The odd picture is read and the high picture written 8 bytes at a time.  LDS histograms as before.
__global__ void __launch_bounds__(256) k_pred_luma8(PredArgs a) {
  __shared__ unsigned h[2][256];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  h[0][tid] = 0; h[1][tid] = 0;
  __syncthreads();
  const LevelGeom& g = a.g;
  const int x = (blockIdx.x * 64 + threadIdx.x) * 8, y = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (x < g.W && y < g.H) {
    const uint8_t* r0 = frame_y(a.frames, g, 2 * i);
    const uint8_t* r1 = frame_y(a.frames, g, 2 * i + 2);
    const uint8_t* od = frame_y(a.frames, g, 2 * i + 1) + (size_t)y * g.W + x;
    const int by = y / g.bs, bx = x / g.bs;
    const size_t fld = (size_t)g.BY * g.BX;
    const bool predicted = by < g.BY && bx < g.BX;  // Q7: the margin of the luma prediction plane stays zero
    const int16_t* m = a.mv + (size_t)i * 4 * fld + (predicted ? (size_t)by * g.BX + bx : 0);
    const int x0 = x + m[0], y0 = y + m[fld], x1 = x + m[2 * fld], y1 = y + m[3 * fld];
    const int n = g.W - x < 8 ? g.W - x : 8;
    uint8_t p0[8], p1[8], o[8];
    if (!predicted) {
      for (int k = 0; k < 8; k++) p0[k] = 0;
    } else if (n == 8 && y0 >= 0 && y0 < g.H && x0 >= 0 && x0 + 8 <= g.W) {
      uint64_t v = ld8u(r0 + (size_t)y0 * g.W + x0);
      __builtin_memcpy(p0, &v, 8);
    } else {
      BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
      for (int k = 0; k < 8; k++) p0[k] = (uint8_t)luma_b(r0, bg, y0, x0 + k);
    }
    if (!predicted) {
      for (int k = 0; k < 8; k++) p1[k] = 0;
    } else if (n == 8 && y1 >= 0 && y1 < g.H && x1 >= 0 && x1 + 8 <= g.W) {
      uint64_t v = ld8u(r1 + (size_t)y1 * g.W + x1);
      __builtin_memcpy(p1, &v, 8);
    } else {
      BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
      for (int k = 0; k < 8; k++) p1[k] = (uint8_t)luma_b(r1, bg, y1, x1 + k);
    }
    const size_t off = (size_t)i * g.F + (size_t)y * g.W + x;
    if (a.odd_out) {  // correlate: high is the input, the odd picture the output
      const int ty = a.types[i];
      uint8_t* dst = a.odd_out + (size_t)(2 * i + 1) * g.F + (size_t)y * g.W + x;
      for (int k = 0; k < n; k++) {
        int p = ((int)p0[k] + (int)p1[k]) >> 1;
        dst[k] = (uint8_t)correlate_sample(a.high[off + k], p, ty);
        if (a.pred_dump) a.pred_dump[off + k] = (uint8_t)p;
      }
    } else {
    if (n == 8) { uint64_t v = ld8u(od); __builtin_memcpy(o, &v, 8); }
    else for (int k = 0; k < n; k++) o[k] = od[k];
    uint8_t hi[8], pr[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      int p = ((int)p0[k] + (int)p1[k]) >> 1;           // decorrelate.cpp:99-107 (both operands 0..255)
      int val = iclamp((int)o[k] - p, -128, 127);        // decorrelate.cpp:920-929
      hi[k] = (uint8_t)(val + 128); pr[k] = (uint8_t)p;
      if (k < n) { atomicAdd(&h[0][o[k]], 1u); atomicAdd(&h[1][val + 128], 1u); }
    }
    if (n == 8) {
      uint64_t v;
      __builtin_memcpy(&v, hi, 8); __builtin_memcpy(a.high + off, &v, 8);
      if (a.pred_dump) { __builtin_memcpy(&v, pr, 8); __builtin_memcpy(a.pred_dump + off, &v, 8); }
    } else {
      for (int k = 0; k < n; k++) { a.high[off + k] = hi[k]; if (a.pred_dump) a.pred_dump[off + k] = pr[k]; }
    }
    }
  }
  if (a.odd_out) return;  // correlate: no histograms (uniform over the launch)
  __syncthreads();
  unsigned* gh = a.hist + (size_t)i * 3 * 256;
  if (h[0][tid]) atomicAdd(&gh[tid], h[0][tid]);
  if (h[1][tid]) atomicAdd(&gh[256 + tid], h[1][tid]);
}
// predict + residue, luma, general form: thread per pixel, LDS histograms (odd picture, residue+128).
__global__ void __launch_bounds__(256) k_pred_luma(PredArgs a) {
  __shared__ unsigned h[2][256];
  int tid = threadIdx.y * 64 + threadIdx.x;
  h[0][tid] = 0; h[1][tid] = 0;
  __syncthreads();
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (x < a.g.W && y < a.g.H) {
    int ob, rb;
    pred_luma_thread(a, x, y, i, ob, rb);
    if (!a.odd_out) { atomicAdd(&h[0][ob], 1u); atomicAdd(&h[1][rb], 1u); }
  }
  if (a.odd_out) return;  // correlate: no histograms (uniform over the launch)
  __syncthreads();
  unsigned* gh = a.hist + (size_t)i * 3 * 256;
  if (h[0][tid]) atomicAdd(&gh[tid], h[0][tid]);
  if (h[1][tid]) atomicAdd(&gh[256 + tid], h[1][tid]);
}
// Chroma: a 32x8 tile of half-resolution outputs per workgroup.  The full-resolution prediction the tile's
// chroma/2 analysis needs (rows 2j-2..2j+2, columns 2i-2..2i+2: 20 x 68 samples) is evaluated ONCE into LDS,
// then the row pass (20 x 32 lows) and the column pass run out of LDS.  Same per-sample functions as
// pred_chroma_thread (mctf_threads.h), 5x fewer interpolated-chroma evaluations per output.
#define PC_TW 32
#define PC_TH 8
#define PC_FW (2 * PC_TW + 4)
#define PC_FH (2 * PC_TH + 4)
__global__ void __launch_bounds__(256) k_pred_chroma(PredArgs a) {
  __shared__ int16_t sP[PC_FH][PC_FW];
  __shared__ int16_t sL[PC_FH][PC_TW];
  const LevelGeom& g = a.g;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int i = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  const int i0 = blockIdx.x * PC_TW, j0 = blockIdx.y * PC_TH;
  const int cw = g.W / 2, chh = g.H / 2;
  const int xb = 2 * i0 - 2, yb = 2 * j0 - 2;  // full-resolution origin of the footprint (may be -2)
  BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  MvField mv = {a.mv + (size_t)i * 4 * g.BY * g.BX, g.BY, g.BX};
  const uint8_t* u0 = frame_c(a.frames, g, 2 * i, c);
  const uint8_t* u1 = frame_c(a.frames, g, 2 * i + 2, c);
  for (int k = tid; k < PC_FH * PC_FW; k += 256) {
    int yy = yb + k / PC_FW, xx = xb + k % PC_FW;
    int v = 0;
    if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) v = pred_chroma_sample(a, u0, u1, bg, mv, i, c, yy, xx);
    sP[k / PC_FW][k % PC_FW] = (int16_t)v;
  }
  __syncthreads();
  for (int k = tid; k < PC_FH * PC_TW; k += 256) {
    int ry = k / PC_TW, ci = i0 + k % PC_TW;
    int yy = yb + ry;
    int v = 0;
    if (yy >= 0 && yy < g.H && ci < cw) {
      auto row = [&](int xx) { return (int)sP[ry][xx - xb]; };
      v = a53_low(row, ci, g.W);
    }
    sL[ry][k % PC_TW] = (int16_t)v;
  }
  __syncthreads();
  const int ci = i0 + (tid & (PC_TW - 1)), cj = j0 + tid / PC_TW;
  if (ci < cw && cj < chh) {
    const int lx = tid & (PC_TW - 1);
    auto col = [&](int yy) { return (int)sL[yy - yb][lx]; };
    int ll = (int16_t)a53_low(col, cj, g.H);
    size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh + (size_t)cj * cw + ci;
    size_t off = (size_t)i * g.F + coff;
    if (a.odd_out) {
      a.odd_out[(size_t)(2 * i + 1) * g.F + coff] = (uint8_t)correlate_sample(a.high[off], ll, a.types[i]);
    } else {
      int o = frame_c(a.frames, g, 2 * i + 1, c)[(size_t)cj * cw + ci];
      int val = iclamp(o - ll, -128, 127);
      a.high[off] = (uint8_t)(val + 128);
    }
    if (a.pred_dump) a.pred_dump[off] = (uint8_t)ll;
  }
}
// ---------------------------------------------------------------------------------------------
// Chroma x2 interpolation of whole pictures (decorrelate.cpp:610-648 = one level of truncating 5/3 synthesis with zero high
// bands): 4 output samples per thread, u8 (averages of u8 stay in 0..255).  The prediction / update tiles below then read the
// up-sampled chroma like luma instead of re-interpolating it in every footprint.
__global__ void __launch_bounds__(256) k_chroma_up(const uint8_t* frames, LevelGeom g, int mul, uint8_t* cup) {
  const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
  const int pic = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  if (x0 >= g.W || y >= g.H) return;
  const uint8_t* u = frame_c(frames, g, mul * pic, c);
  const int cw = g.W / 2, k2 = x0 >> 1;
  const int c0 = chroma_up_col(u, cw, y, k2, g.H);
  const int c1 = k2 + 1 < cw ? chroma_up_col(u, cw, y, k2 + 1, g.H) : c0;
  const int c2 = k2 + 2 < cw ? chroma_up_col(u, cw, y, k2 + 2, g.H) : c1;
  const int o0 = c0, o1 = x0 + 2 < g.W ? (c0 + c1) >> 1 : c0, o2 = c1, o3 = x0 + 4 < g.W ? (c1 + c2) >> 1 : c1;
  uint8_t* d = cup + ((size_t)blockIdx.z * g.H + y) * g.W + x0;
  if ((g.W & 3) == 0) {
    *(uint32_t*)d = (uint32_t)o0 | ((uint32_t)o1 << 8) | ((uint32_t)o2 << 16) | ((uint32_t)o3 << 24);
  } else {
    d[0] = (uint8_t)o0; d[1] = (uint8_t)o1;
    if (x0 + 2 < g.W) { d[2] = (uint8_t)o2; d[3] = (uint8_t)o3; }
  }
}
void launch_chroma_up(hipStream_t st, const uint8_t* frames, const LevelGeom& g, int mul, int count, uint8_t* cup) {
  if (count > 0)
    hipLaunchKernelGGL(k_chroma_up, dim3((g.W / 4 + 64) / 64, (g.H + 3) / 4, count * 2), dim3(64, 4), 0, st, frames, g, mul, cup);
}

// 8 consecutive samples of one row of a bordered u8 plane displaced by a motion vector: one unaligned 8-byte load when the run
// lies inside the picture, the per-sample border rule (Q1/Q2) otherwise
__device__ __forceinline__ uint64_t ref_run8(const uint8_t* r, const LevelGeom& g, int y, int x) {
  if (y >= 0 && y < g.H && x >= 0 && x + 8 <= g.W) return ld8u(r + (size_t)y * g.W + x);
  const BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  uint64_t v = 0;
  for (int k = 0; k < 8; k++) v |= (uint64_t)luma_b(r, bg, y, x + k) << (8 * k);
  return v;
}
// Chroma prediction from the up-sampled planes, block size a multiple of 8, picture tiled by the block grid: the 20-row footprint
// of a 32x8 tile is cut into runs of 8 samples that share one MC block (80 columns from 2*i0 - 8: one run per thread).
#define PC8_FW 80
__global__ void __launch_bounds__(256) k_pred_chroma8(PredArgs a) {
  __shared__ int16_t sP[PC_FH][PC8_FW];
  __shared__ int16_t sL[PC_FH][PC_TW];
  const LevelGeom& g = a.g;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int i = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  const int i0 = blockIdx.x * PC_TW, j0 = blockIdx.y * PC_TH;
  const int cw = g.W / 2, chh = g.H / 2;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  if (tid < PC_FH * (PC8_FW / 8)) {
    const int ry = tid / (PC8_FW / 8), run = tid % (PC8_FW / 8), yy = yb + ry, xx = xb + 8 * run;
    uint64_t p = 0;  // 8 prediction samples, one byte each (0..255)
    if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) {
      const size_t plane = (size_t)g.H * g.W, fld = (size_t)g.BY * g.BX;
      const uint8_t* r0 = a.cup + ((size_t)i * 2 + (c - 1)) * plane;
      const uint8_t* r1 = a.cup + ((size_t)(i + 1) * 2 + (c - 1)) * plane;
      const int16_t* m = a.mv + (size_t)i * 4 * fld + (size_t)(yy / g.bs) * g.BX + xx / g.bs;
      const uint64_t p0 = ref_run8(r0, g, yy + m[fld], xx + m[0]);
      const uint64_t p1 = ref_run8(r1, g, yy + m[3 * fld], xx + m[2 * fld]);
      // per byte (a + b) >> 1 without carries between bytes (decorrelate.cpp:99-107)
      p = (p0 & p1) + (((p0 ^ p1) & 0xFEFEFEFEFEFEFEFEull) >> 1);
    }
    int16_t* d = &sP[ry][8 * run];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = (int16_t)((p >> (8 * k)) & 255);
  }
  __syncthreads();
  for (int k = tid; k < PC_FH * PC_TW; k += 256) {
    int ry = k / PC_TW, ci = i0 + k % PC_TW;
    int yy = yb + ry;
    int v = 0;
    if (yy >= 0 && yy < g.H && ci < cw) {
      auto row = [&](int xx) { return (int)sP[ry][xx - xb]; };
      v = a53_low(row, ci, g.W);
    }
    sL[ry][k % PC_TW] = (int16_t)v;
  }
  __syncthreads();
  const int ci = i0 + (tid & (PC_TW - 1)), cj = j0 + tid / PC_TW;
  if (ci < cw && cj < chh) {
    const int lx = tid & (PC_TW - 1);
    auto col = [&](int yy) { return (int)sL[yy - yb][lx]; };
    int ll = (int16_t)a53_low(col, cj, g.H);
    size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh + (size_t)cj * cw + ci;
    size_t off = (size_t)i * g.F + coff;
    if (a.odd_out) {
      a.odd_out[(size_t)(2 * i + 1) * g.F + coff] = (uint8_t)correlate_sample(a.high[off], ll, a.types[i]);
    } else {
      int o = frame_c(a.frames, g, 2 * i + 1, c)[(size_t)cj * cw + ci];
      int val = iclamp(o - ll, -128, 127);
      a.high[off] = (uint8_t)(val + 128);
    }
    if (a.pred_dump) a.pred_dump[off] = (uint8_t)ll;
  }
}
// histogram of the motion components + 128 (decorrelate.cpp:805-812)
__global__ void __launch_bounds__(256) k_mv_hist(const int16_t* mv, size_t per_pic, unsigned* hist) {
  __shared__ unsigned h[256];
  int i = blockIdx.y;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int16_t* m = mv + (size_t)i * per_pic;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < per_pic; k += (size_t)gridDim.x * 256)
    atomicAdd(&h[(m[k] + 128) & 255], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[((size_t)i * 3 + 2) * 256 + threadIdx.x], h[threadIdx.x]);
}
// ---------------------------------------------------------------------------------------------
// --block_overlaping >= 2 (predict(), decorrelate.cpp:84-172): every block's bidirectional prediction is taken with a margin of
// `ov` samples, analysed `levels` = log2(ov) levels in place (truncating 5/3), its subbands -- without the margin -- are copied
// into the picture's Mallat layout, and the picture is synthesised back (the blocks blend through the synthesis filters).
// One workgroup per (block, component, B picture): the (bs + 2 ov)^2 block lives in LDS.  Chroma from the up-sampled planes.
#define OVL_MAXN 96
__global__ void __launch_bounds__(256) k_ovl_block(PredArgs a, int levels, int16_t* ppl) {
  __shared__ int16_t sA[OVL_MAXN * OVL_MAXN];
  __shared__ int16_t sB[OVL_MAXN * OVL_MAXN];
  const LevelGeom& g = a.g;
  const int tid = threadIdx.x, bx = blockIdx.x, by = blockIdx.y, i = blockIdx.z / 3, c = blockIdx.z % 3;
  const int ov = g.ov, bs = g.bs, n = bs + 2 * ov;
  const size_t plane = (size_t)g.H * g.W, fld = (size_t)g.BY * g.BX;
  const uint8_t* r0 = a.up ? a.up + ((size_t)i * 3 + c) * plane : (c == 0 ? frame_y(a.frames, g, 2 * i) : a.cup + ((size_t)i * 2 + (c - 1)) * plane);
  const uint8_t* r1 = a.up ? a.up + ((size_t)(i + 1) * 3 + c) * plane
                           : (c == 0 ? frame_y(a.frames, g, 2 * i + 2) : a.cup + ((size_t)(i + 1) * 2 + (c - 1)) * plane);
  const int16_t* m = a.mv + (size_t)i * 4 * fld + (size_t)by * g.BX + bx;
  const int y0 = by * bs + m[fld], x0 = bx * bs + m[0], y1 = by * bs + m[3 * fld], x1 = bx * bs + m[2 * fld];
  const BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  for (int k = tid; k < n * n; k += 256) {
    const int y = k / n - ov, x = k % n - ov;
    sA[k] = (int16_t)((luma_b(r0, bg, y0 + y, x0 + x) + luma_b(r1, bg, y1 + y, x1 + x)) / 2);  // decorrelate.cpp:99-107
  }
  __syncthreads();
  for (int l = 0; l < levels; l++) {  // dwt2d::analyze on the block (dwt2d.cpp:76-119): region (n >> l)^2
    const int mreg = n >> l;
    auto P = [&](int yy, int xx) { return (int)sA[yy * n + xx]; };
    for (int k = tid; k < mreg * mreg; k += 256) sB[(k / mreg) * n + k % mreg] = (int16_t)analyze2d_coef(P, k / mreg, k % mreg, mreg, mreg);
    __syncthreads();
    for (int k = tid; k < mreg * mreg; k += 256) sA[(k / mreg) * n + k % mreg] = sB[(k / mreg) * n + k % mreg];
    __syncthreads();
  }
  int16_t* out = ppl + ((size_t)i * 3 + c) * plane;
  for (int l = 1; l <= levels; l++) {  // decorrelate.cpp:116-146
    const int bsl = bs >> l, o1 = ov >> l, o3 = (bs + 3 * ov) >> l;
    for (int k = tid; k < bsl * bsl; k += 256) {
      const int y = k / bsl, x = k % bsl;
      out[(size_t)(by * bsl + y) * g.W + (g.W >> l) + bx * bsl + x] = sA[(o1 + y) * n + o3 + x];
      out[(size_t)((g.H >> l) + by * bsl + y) * g.W + bx * bsl + x] = sA[(o3 + y) * n + o1 + x];
      out[(size_t)((g.H >> l) + by * bsl + y) * g.W + (g.W >> l) + bx * bsl + x] = sA[(o3 + y) * n + o3 + x];
    }
  }
  {
    const int bsl = bs >> levels, o1 = ov >> levels;
    for (int k = tid; k < bsl * bsl; k += 256) out[(size_t)(by * bsl + k / bsl) * g.W + bx * bsl + k % bsl] = sA[(o1 + k / bsl) * n + o1 + k % bsl];
  }
}
// one synthesis level of the truncating 5/3 over the top-left ny x nx of every plane: src -> dst
__global__ void __launch_bounds__(256) k_dwt53t_synth_planes(const int16_t* src, int16_t* dst, size_t plane, int stride, int ny, int nx) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= nx || y >= ny) return;
  const int16_t* s = src + (size_t)blockIdx.z * plane;
  auto P = [&](int yy, int xx) { return (int)s[(size_t)yy * stride + xx]; };
  dst[(size_t)blockIdx.z * plane + (size_t)y * stride + x] = (int16_t)synthesize2d_sample(P, y, x, ny, nx);
}
__global__ void __launch_bounds__(256) k_copy_region_planes(const int16_t* src, int16_t* dst, size_t plane, int stride, int ny, int nx) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x < nx && y < ny) dst[(size_t)blockIdx.z * plane + (size_t)y * stride + x] = src[(size_t)blockIdx.z * plane + (size_t)y * stride + x];
}
// residue / correlate from the synthesised prediction planes: luma (thread per sample) ...
__global__ void __launch_bounds__(256) k_ovl_finish_luma(PredArgs a, const int16_t* ppl, int clip) {
  __shared__ unsigned h[2][256];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  h[0][tid] = 0; h[1][tid] = 0;
  __syncthreads();
  const LevelGeom& g = a.g;
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z;
  if (x < g.W && y < g.H) {
    int p = ppl[((size_t)i * 3) * g.H * g.W + (size_t)y * g.W + x];
    if (clip) p = iclamp(p, 0, 255);  // decorrelate.cpp:841-848 (with --subpixel_accuracy the clip was done before the planes came down)
    const size_t off = (size_t)i * g.F + (size_t)y * g.W + x;
    if (a.odd_out) a.odd_out[(size_t)(2 * i + 1) * g.F + (size_t)y * g.W + x] = (uint8_t)correlate_sample(a.high[off], p, a.types[i]);
    else {
      const int o = frame_y(a.frames, g, 2 * i + 1)[(size_t)y * g.W + x];
      const int val = iclamp(o - p, -128, 127);
      a.high[off] = (uint8_t)(val + 128);
      atomicAdd(&h[0][o], 1u); atomicAdd(&h[1][val + 128], 1u);
    }
    if (a.pred_dump) a.pred_dump[off] = (uint8_t)p;
  }
  if (a.odd_out) return;
  __syncthreads();
  unsigned* gh = a.hist + (size_t)i * 3 * 256;
  if (h[0][tid]) atomicAdd(&gh[tid], h[0][tid]);
  if (h[1][tid]) atomicAdd(&gh[256 + tid], h[1][tid]);
}
// ... and chroma: clip, chroma/2 (LL of a 1-level analysis, decorrelate.cpp:860-861), residue
__global__ void __launch_bounds__(256) k_ovl_finish_chroma(PredArgs a, const int16_t* ppl, int clip) {
  const LevelGeom& g = a.g;
  const int cw = g.W / 2, chh = g.H / 2;
  const int ci = blockIdx.x * 64 + threadIdx.x, cj = blockIdx.y * 4 + threadIdx.y, i = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  if (ci >= cw || cj >= chh) return;
  const int16_t* pp = ppl + ((size_t)i * 3 + c) * g.H * g.W;
  auto P = [&](int yy, int xx) { const int v = pp[(size_t)yy * g.W + xx]; return clip ? iclamp(v, 0, 255) : v; };
  const int ll = analyze_ll(P, cj, ci, g.H, g.W);
  const size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh + (size_t)cj * cw + ci, off = (size_t)i * g.F + coff;
  if (a.odd_out) a.odd_out[(size_t)(2 * i + 1) * g.F + coff] = (uint8_t)correlate_sample(a.high[off], ll, a.types[i]);
  else {
    const int o = frame_c(a.frames, g, 2 * i + 1, c)[(size_t)cj * cw + ci];
    a.high[off] = (uint8_t)(iclamp(o - ll, -128, 127) + 128);
  }
  if (a.pred_dump) a.pred_dump[off] = (uint8_t)ll;
}
// ppl0 / ppl1: two [nB][3][H][W] int16 buffers.  Returns false when the geometry is outside this path.
bool launch_ovl_blocks(hipStream_t st, const PredArgs& a, int16_t* ppl0, int16_t* ppl1) {
  const LevelGeom& g = a.g;
  int levels = 0;
  while ((1 << (levels + 1)) <= g.ov) levels++;
  if ((1 << levels) != g.ov || levels < 1 || g.bs + 2 * g.ov > OVL_MAXN || (g.bs & (g.ov - 1)) || (!a.cup && !a.up) || a.margin) return false;
  const size_t plane = (size_t)g.H * g.W;
  hipLaunchKernelGGL(k_ovl_block, dim3(g.BX, g.BY, g.nB * 3), dim3(256), 0, st, a, levels, ppl0);
  for (int lv = levels - 1; lv >= 0; lv--) {  // dwt2d::synthesize (dwt2d.cpp:128-175)
    const int ny = g.H >> lv, nx = g.W >> lv;
    dim3 grid((nx + 63) / 64, (ny + 3) / 4, g.nB * 3), block(64, 4);
    hipLaunchKernelGGL(k_dwt53t_synth_planes, grid, block, 0, st, ppl0, ppl1, plane, g.W, ny, nx);
    hipLaunchKernelGGL(k_copy_region_planes, grid, block, 0, st, ppl1, ppl0, plane, g.W, ny, nx);
  }
  return true;
}
void launch_pred_finish(hipStream_t st, const PredArgs& a, const int16_t* ppl, int clip) {
  const LevelGeom& g = a.g;
  if (!a.odd_out) hipMemsetAsync(a.hist, 0, sizeof(unsigned) * 3 * 256 * g.nB, st);
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_ovl_finish_luma, dim3((g.W + 63) / 64, (g.H + 3) / 4, g.nB), block, 0, st, a, ppl, clip);
  hipLaunchKernelGGL(k_ovl_finish_chroma, dim3((g.W / 2 + 63) / 64, (g.H / 2 + 3) / 4, g.nB * 2), block, 0, st, a, ppl, clip);
  if (!a.odd_out) {
    size_t per_pic = (size_t)4 * g.BY * g.BX;
    int nblk = (int)((per_pic + 255) / 256);
    if (nblk > 64) nblk = 64;
    hipLaunchKernelGGL(k_mv_hist, dim3(nblk, g.nB), dim3(256), 0, st, a.mv, per_pic, a.hist);
  }
}
bool launch_predict_overlap(hipStream_t st, const PredArgs& a, int16_t* ppl0, int16_t* ppl1) {
  if (!launch_ovl_blocks(st, a, ppl0, ppl1)) return false;
  launch_pred_finish(st, a, ppl0, 1);
  return true;
}

bool pred_fast_path(const PredArgs& a) { return pred_run_applies(a) && tuning().pred_run; }
void launch_predict(hipStream_t st, const PredArgs& a) {
  const LevelGeom& g = a.g;
  dim3 block(64, 4);
  if (pred_fast_path(a)) {  // mctf_pred.hip: one launch (chroma tiles, luma strips, motion histogram + motion_filtered copy)
    launch_predict_run(st, a);
    return;
  }
  if (!a.odd_out) hipMemsetAsync(a.hist, 0, sizeof(unsigned) * 3 * 256 * g.nB, st);
  if (a.margin) launch_pred_margin(st, a);  // mctf_margin.hip
  if (g.bs % 8 == 0) hipLaunchKernelGGL(k_pred_luma8, dim3((g.W + 511) / 512, (g.H + 3) / 4, g.nB), block, 0, st, a);
  else hipLaunchKernelGGL(k_pred_luma, dim3((g.W + 63) / 64, (g.H + 3) / 4, g.nB), block, 0, st, a);
  const dim3 cgrid((g.W / 2 + PC_TW - 1) / PC_TW, (g.H / 2 + PC_TH - 1) / PC_TH, g.nB * 2);
  if (a.cup && !a.margin && g.bs % 8 == 0 && g.W % 8 == 0) hipLaunchKernelGGL(k_pred_chroma8, cgrid, block, 0, st, a);
  else hipLaunchKernelGGL(k_pred_chroma, cgrid, block, 0, st, a);
  if (a.odd_out) return;
  size_t per_pic = (size_t)4 * g.BY * g.BX;
  int nblk = (int)((per_pic + 255) / 256);
  if (nblk > 64) nblk = 64;
  hipLaunchKernelGGL(k_mv_hist, dim3(nblk, g.nB), dim3(256), 0, st, a.mv, per_pic, a.hist);
}

// ---------------------------------------------------------------------------------------------
// update.  General chroma form (any geometry): thread per target sample of the footprint, ordered gather.
// Chroma: same 32x8 tiling as k_pred_chroma.  Tiles out of reach of the top-left quadrant (Q3) or without any
// contributing B picture are a plain copy; otherwise the updated full-resolution samples of the footprint are
// gathered once into LDS and the chroma/2 analysis runs out of LDS.
__global__ void __launch_bounds__(256) k_upd_chroma(UpdArgs a) {
  __shared__ int16_t sP[PC_FH][PC_FW];
  __shared__ int16_t sL[PC_FH][PC_TW];
  const LevelGeom& g = a.g;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int j = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  const int i0 = blockIdx.x * PC_TW, j0 = blockIdx.y * PC_TH;
  const int cw = g.W / 2, chh = g.H / 2;
  const int xb = 2 * i0 - 2, yb = 2 * j0 - 2;
  const uint8_t* u = frame_c(a.frames, g, a.src_mul * j, c);
  const size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh;
  UpdSide s[2];
  upd_sides(a, j, s);
  const int ci = i0 + (tid & (PC_TW - 1)), cj = j0 + tid / PC_TW;
  const bool untouched = (!s[0].high && !s[1].high) || yb >= chh + a.reach || xb >= cw + a.reach;  // wave-uniform
  if (untouched) {
    if (ci < cw && cj < chh) a.low[(size_t)(a.dst_mul * j) * g.F + coff + (size_t)cj * cw + ci] = u[(size_t)cj * cw + ci];
    return;
  }
  for (int k = tid; k < PC_FH * PC_FW; k += 256) {
    int yy = yb + k / PC_FW, xx = xb + k % PC_FW;
    int v = 0;
    if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) {
      v = chroma_up(u, yy, xx, g.H, g.W);
      for (int q = 0; q < 2; q++)
        if (s[q].high) {
          const uint8_t* h = s[q].high + coff;
          auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * cw + sx] - 128; };
          v = update_gather(v, R, s[q].mvx, s[q].mvy, g.BY, g.BX, g.bs, g.H, g.W, chh, cw, a.reach, yy, xx, a.uf);
        }
    }
    sP[k / PC_FW][k % PC_FW] = (int16_t)v;
  }
  __syncthreads();
  for (int k = tid; k < PC_FH * PC_TW; k += 256) {
    int ry = k / PC_TW, cx = i0 + k % PC_TW;
    int yy = yb + ry;
    int v = 0;
    if (yy >= 0 && yy < g.H && cx < cw) {
      auto row = [&](int xx) { return (int)sP[ry][xx - xb]; };
      v = a53_low(row, cx, g.W);
    }
    sL[ry][k % PC_TW] = (int16_t)v;
  }
  __syncthreads();
  if (ci < cw && cj < chh) {
    const int lx = tid & (PC_TW - 1);
    auto col = [&](int yy) { return (int)sL[yy - yb][lx]; };
    a.low[(size_t)(a.dst_mul * j) * g.F + coff + (size_t)cj * cw + ci] = (uint8_t)(int16_t)a53_low(col, cj, g.H);
  }
}
// ---------------------------------------------------------------------------------------------
// update, tiled: the reference's in-place scatter (update.cpp:85-146) done as a scatter INTO AN LDS TILE OF TARGETS.
// One wave owns a tile of the even picture; it walks the source blocks of the B picture in the reference's raster order
// (by, bx) and, for every block whose displaced footprint meets the tile, its 64 lanes apply that block's pixels to the tile
// (read-modify-write in LDS with update.cpp:93-115's float operations).  Within one block the targets are distinct (one
// vector per block), and the LDS operations of a wave are performed in program order, so every target sees its sources in
// exactly the reference's order -- which is what the saturation needs -- while the candidate scan costs a few
// scalar operations per block instead of a gather per target.  Targets ON the picture border collect everything that the
// reference's clip() folds onto them, several per block and in (y, x) order: those few cells are left out of the scatter and
// evaluated by the ordered gather (update_gather, mctf_core.h) afterwards.
#define UPD_STAGE_REACH 16   // largest |mv| for which the source region of a workgroup's targets is staged in LDS
// One side (one B picture's PREV or NEXT field) of the update of a wave's target tile [ty0, ty0+TH) x [tx0, tx0+TW).
template <int TW, int TH>
struct UpdScatter {
  int by0, bx0, nbx, n;   // candidate source blocks: those within `reach` of the tile
  int dy[2], dx[2];       // this lane's candidates' vectors (candidate k = 64 r + lane), fetched before anything waits
  // The candidates' vectors are requested up front: the latency overlaps with the staging of the tile and of the residues.
  __device__ __forceinline__ void plan(int ty0, int tx0, const int16_t* mvx, const int16_t* mvy, int BY, int BX, int bs, int Hs, int Ws,
                                       int reach, int lane) {
    const int Hl = imin(Hs, BY * bs), Wl = imin(Ws, BX * bs);
    const int sy_lo = imax(ty0 - reach, 0), sy_hi = imin(ty0 + TH - 1 + reach, Hl - 1);
    const int sx_lo = imax(tx0 - reach, 0), sx_hi = imin(tx0 + TW - 1 + reach, Wl - 1);
    n = 0; by0 = bx0 = 0; nbx = 1;
    dy[0] = dy[1] = dx[0] = dx[1] = 0;
    if (sy_hi < sy_lo || sx_hi < sx_lo) return;
    by0 = sy_lo / bs; bx0 = sx_lo / bs; nbx = sx_hi / bs - bx0 + 1; n = (sy_hi / bs - by0 + 1) * nbx;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int k = 64 * r + lane;
      if (k < n) {
        const size_t o = (size_t)(by0 + k / nbx) * BX + bx0 + k % nbx;
        dy[r] = mvy[o]; dx[r] = mvx[o];
      }
    }
  }
  // reg: the residues of rows [ry0, ..) x columns [rx0, ..), row stride RW, or nullptr (then `h`, the B picture's plane, is read).
  // TGT_IN (wave-uniform): every target of the tile lies strictly inside the picture, so only the tile's own bounds decide a hit
  // pixel; SRC_IN: every pixel of every candidate block can carry a residue (Hs, Ws are multiples of the block size).
  template <bool TGT_IN, bool SRC_IN>
  __device__ __forceinline__ void run_t(uint8_t* tile, const uint8_t* reg, int ry0, int rx0, int RW, int ty0, int tx0, const uint8_t* h,
                                        int hstride, const int16_t* mvx, const int16_t* mvy, int BX, int bs, int Hs, int Ws, int H, int W,
                                        float uf, int lane) {
    const int lbs = 31 - __builtin_clz(bs);
    const bool pow2 = (1 << lbs) == bs;
    // 64 candidates at a time are tested against the tile in parallel; the hits are visited in ascending order = the reference's
    // raster order of blocks, all 64 lanes applying one block's pixels (distinct targets) to the tile.
    for (int base = 0; base < n; base += 64) {
      const int k = base + lane;
      const bool valid = k < n;
      const int by = by0 + (valid ? k / nbx : 0), bx = bx0 + (valid ? k % nbx : 0);
      int cdy, cdx;
      if (base == 0) { cdy = dy[0]; cdx = dx[0]; }
      else if (base == 64) { cdy = dy[1]; cdx = dx[1]; }
      else { cdy = valid ? (int)mvy[(size_t)by * BX + bx] : 0; cdx = valid ? (int)mvx[(size_t)by * BX + bx] : 0; }
      const int y0 = by * bs, x0 = bx * bs;
      const bool hit = valid && !(y0 + cdy > ty0 + TH - 1 || y0 + bs - 1 + cdy < ty0 || x0 + cdx > tx0 + TW - 1 || x0 + bs - 1 + cdx < tx0);
      unsigned long long m = __ballot(hit);
      while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        const int hdy = __builtin_amdgcn_readlane(cdy, l), hdx = __builtin_amdgcn_readlane(cdx, l);
        const int hy0 = __builtin_amdgcn_readlane(y0, l), hx0 = __builtin_amdgcn_readlane(x0, l);
        for (int p = lane; p < bs * bs; p += 64) {
          const int py = pow2 ? p >> lbs : p / bs, px = pow2 ? p & (bs - 1) : p % bs;
          const int sy = hy0 + py, sx = hx0 + px;
          const int t_y = sy + hdy, t_x = sx + hdx, ly = t_y - ty0, lx = t_x - tx0;
          bool ok = (unsigned)ly < (unsigned)TH && (unsigned)lx < (unsigned)TW;
          if (!SRC_IN) ok = ok && sy < Hs && sx < Ws;
          if (!TGT_IN) ok = ok && t_y > 0 && t_y < H - 1 && t_x > 0 && t_x < W - 1;
          if (ok) {
            const int r = (reg ? (int)reg[(sy - ry0) * RW + (sx - rx0)] : (int)h[(size_t)sy * hstride + sx]) - 128;
            uint8_t* t = tile + ly * TW + lx;
            *t = (uint8_t)upd_one((int)*t, r, uf);
          }
        }
      }
    }
  }
  __device__ __forceinline__ void run(uint8_t* tile, const uint8_t* reg, int ry0, int rx0, int RW, int ty0, int tx0, const uint8_t* h,
                                      int hstride, const int16_t* mvx, const int16_t* mvy, int BX, int bs, int Hs, int Ws, int H, int W,
                                      float uf, int lane) {
    const bool tgt_in = ty0 >= 1 && ty0 + TH <= H - 1 && tx0 >= 1 && tx0 + TW <= W - 1;   // wave-uniform
    const bool src_in = Hs % bs == 0 && Ws % bs == 0;
    if (tgt_in && src_in) run_t<true, true>(tile, reg, ry0, rx0, RW, ty0, tx0, h, hstride, mvx, mvy, BX, bs, Hs, Ws, H, W, uf, lane);
    else if (tgt_in) run_t<true, false>(tile, reg, ry0, rx0, RW, ty0, tx0, h, hstride, mvx, mvy, BX, bs, Hs, Ws, H, W, uf, lane);
    else run_t<false, false>(tile, reg, ry0, rx0, RW, ty0, tx0, h, hstride, mvx, mvy, BX, bs, Hs, Ws, H, W, uf, lane);
  }
};
// residues of rows [ry0, ry0+RH) x columns [rx0, rx0+RW) of a B picture's plane into LDS, 4 bytes per thread-load (all threads)
template <int RW, int RH>
__device__ __forceinline__ void upd_stage_region(uint8_t* reg, int ry0, int rx0, const uint8_t* h, int hstride, int Hs, int tid) {
  constexpr int NQ = RH * (RW / 4), NIT = (NQ + 255) / 256;
  uint32_t v[NIT];
#pragma unroll
  for (int it = 0; it < NIT; it++) {  // every load of the thread is issued before the first store waits for one
    const int q = tid + it * 256, ly = q / (RW / 4), lx = 4 * (q % (RW / 4)), y = ry0 + ly, x = rx0 + lx;
    v[it] = 0x80808080u;
    if (q < NQ && y >= 0 && y < Hs && x >= 0 && x + 3 < hstride) v[it] = *(const uint32_t*)(h + (size_t)y * hstride + x);
  }
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int q = tid + it * 256, ly = q / (RW / 4), lx = 4 * (q % (RW / 4));
    if (q < NQ) *(uint32_t*)(reg + ly * RW + lx) = v[it];
  }
}
#define UL_TW 64
#define UL_TH 16   // per wave; a workgroup of 4 waves covers 64 x 64 luma targets
__global__ void __launch_bounds__(256) k_upd_luma_tile(UpdArgs a) {
  constexpr int RW = UL_TW + 2 * UPD_STAGE_REACH, RH = 4 * UL_TH + 2 * UPD_STAGE_REACH;
  __shared__ __attribute__((aligned(16))) uint8_t sT[4][UL_TH * UL_TW];
  __shared__ __attribute__((aligned(16))) uint8_t sR[2][RH * RW];
  const LevelGeom& g = a.g;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, j = blockIdx.z;
  const int tx0 = blockIdx.x * UL_TW, wy0 = blockIdx.y * (4 * UL_TH), ty0 = wy0 + wv * UL_TH;
  uint8_t* tile = sT[wv];
  UpdSide s[2];
  upd_sides(a, j, s);
  const uint8_t* src = frame_y(a.frames, g, a.src_mul * j);
  uint8_t* dst = a.low + (size_t)(a.dst_mul * j) * g.F;
  const bool wide = (g.W & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 3) == 0;
  const bool live = ty0 < g.H;  // wave-uniform
  UpdScatter<UL_TW, UL_TH> sc[2];
  for (int q = 0; q < 2; q++) {
    sc[q].n = 0;
    if (live && s[q].high) sc[q].plan(ty0, tx0, s[q].mvx, s[q].mvy, g.BY, g.BX, g.bs, g.H, g.W, a.reach, lane);
  }
  // the tile: 4 samples per lane-load (16 lanes per row)
  if (live)
    for (int q = lane; q < UL_TH * (UL_TW / 4); q += 64) {
      const int ly = q / (UL_TW / 4), lx = 4 * (q % (UL_TW / 4)), y = ty0 + ly, x = tx0 + lx;
      uint32_t v = 0;
      if (y < g.H && x < g.W) {
        if (wide) v = *(const uint32_t*)(src + (size_t)y * g.W + x);
        else for (int k = 0; k < 4 && x + k < g.W; k++) v |= (uint32_t)src[(size_t)y * g.W + x + k] << (8 * k);
      }
      *(uint32_t*)(tile + ly * UL_TW + lx) = v;
    }
  // both sides' source regions for the whole workgroup (block-uniform conditions)
  const bool staged = a.reach <= UPD_STAGE_REACH && (g.W & 3) == 0;
  const int ry0 = wy0 - UPD_STAGE_REACH, rx0 = tx0 - UPD_STAGE_REACH;
  if (staged) {
    for (int q = 0; q < 2; q++)
      if (s[q].high && (((uintptr_t)s[q].high) & 3) == 0) upd_stage_region<RW, RH>(sR[q], ry0, rx0, s[q].high, g.W, g.H, tid);
    __syncthreads();
  }
  if (!live) return;
  if (s[0].high || s[1].high) {
    for (int q = 0; q < 2; q++)
      if (s[q].high) {
        const uint8_t* reg = staged && (((uintptr_t)s[q].high) & 3) == 0 ? sR[q] : nullptr;
        sc[q].run(tile, reg, ry0, rx0, RW, ty0, tx0, s[q].high, g.W, s[q].mvx, s[q].mvy, g.BX, g.bs, g.H, g.W, g.H, g.W, a.uf, lane);
      }
    // cells on the picture border: ordered gather, the border cells of the tile spread over the lanes (a picture row: one
    // cell per lane; a picture column: the tile's rows over the first lanes)
    auto border_cell = [&](int y, int x) {
      int v = src[(size_t)y * g.W + x];
      for (int k = 0; k < 2; k++)
        if (s[k].high) {
          const uint8_t* h = s[k].high;
          auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * g.W + sx] - 128; };
          v = update_gather(v, R, s[k].mvx, s[k].mvy, g.BY, g.BX, g.bs, g.H, g.W, g.H, g.W, a.reach, y, x, a.uf);
        }
      tile[(y - ty0) * UL_TW + (x - tx0)] = (uint8_t)v;
    };
    for (int e = 0; e < 2; e++) {  // picture rows 0 and H-1
      const int y = e ? g.H - 1 : 0, x = tx0 + lane;
      if (y >= ty0 && y < ty0 + UL_TH && x < g.W) border_cell(y, x);
    }
    for (int e = 0; e < 2; e++) {  // picture columns 0 and W-1 (corners are done)
      const int x = e ? g.W - 1 : 0, y = ty0 + lane;
      if (x >= tx0 && x < tx0 + UL_TW && lane < UL_TH && y > 0 && y < g.H - 1) border_cell(y, x);
    }
  }
  for (int q = lane; q < UL_TH * (UL_TW / 4); q += 64) {
    const int ly = q / (UL_TW / 4), lx = 4 * (q % (UL_TW / 4)), y = ty0 + ly, x = tx0 + lx;
    if (y >= g.H || x >= g.W) continue;
    const uint32_t v = *(const uint32_t*)(tile + ly * UL_TW + lx);
    if (wide) *(uint32_t*)(dst + (size_t)y * g.W + x) = v;
    else for (int k = 0; k < 4 && x + k < g.W; k++) dst[(size_t)y * g.W + x + k] = (uint8_t)(v >> (8 * k));
  }
}
// both chroma planes of every even picture, as stored: 16 bytes per thread (the tile kernel then rewrites what an update reaches)
__global__ void __launch_bounds__(256) k_upd_chroma_copy(UpdArgs a) {
  const LevelGeom& g = a.g;
  const size_t n = (size_t)2 * (g.W / 2) * (g.H / 2), k = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (k >= n) return;
  const uint8_t* src = a.frames + (size_t)(a.src_mul * blockIdx.y) * g.F + (size_t)g.W * g.H + k;
  uint8_t* dst = a.low + (size_t)(a.dst_mul * blockIdx.y) * g.F + (size_t)g.W * g.H + k;
  if (k + 16 <= n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) *(uint4*)dst = *(const uint4*)src;
  else for (size_t q = 0; q < 16 && k + q < n; q++) dst[q] = src[q];
}
// Chroma: the targets are the up-sampled planes (update.cpp:472-503), the sources the half-resolution residue in the top-left
// quadrant (Q3), and the stored picture is the chroma/2 analysis of the result (update.cpp:632-643).  The 20 x 80 footprint of a
// 32x8 output tile is the LDS target tile; each of the four waves owns five of its rows for the scatter.
__global__ void __launch_bounds__(256) k_upd_chroma_tile(UpdArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t sP[PC_FH * PC8_FW];
  constexpr int RW = PC8_FW + 2 * UPD_STAGE_REACH, RH = PC_FH + 2 * UPD_STAGE_REACH;
  __shared__ __attribute__((aligned(16))) uint8_t sR[2][RH * RW];
  __shared__ int16_t sL[PC_FH][PC_TW];
  const LevelGeom& g = a.g;
  const int tid = threadIdx.y * 64 + threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int j = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  const int i0 = blockIdx.x * PC_TW, j0 = blockIdx.y * PC_TH;
  const int cw = g.W / 2, chh = g.H / 2;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  const size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh;
  UpdSide s[2];
  upd_sides(a, j, s);
  const int ci = i0 + (tid & (PC_TW - 1)), cj = j0 + tid / PC_TW;
  // tiles out of reach of the top-left quadrant (Q3), and pictures without a contributing B picture, keep the stored chroma:
  // k_upd_chroma_copy has already written it (analysis(synthesis(u)) == u exactly)
  if ((!s[0].high && !s[1].high) || yb >= chh + a.reach || 2 * i0 - 2 >= cw + a.reach) return;  // workgroup-uniform
  constexpr int SR = PC_FH / 4;  // footprint rows per wave
  uint8_t* tile = sP + wv * SR * PC8_FW;
  const int ty0 = yb + wv * SR;
  const uint8_t* up = a.cup + ((size_t)j * 2 + (c - 1)) * g.H * g.W;
  UpdScatter<PC8_FW, SR> sc[2];
  for (int q = 0; q < 2; q++) {
    sc[q].n = 0;
    if (s[q].high) sc[q].plan(ty0, xb, s[q].mvx, s[q].mvy, g.BY, g.BX, g.bs, chh, cw, a.reach, lane);
  }
  for (int q = lane; q < SR * (PC8_FW / 8); q += 64) {   // W % 8 == 0 on this path: runs are inside or outside as a whole
    const int ly = q / (PC8_FW / 8), lx = 8 * (q % (PC8_FW / 8)), y = ty0 + ly, x = xb + lx;
    uint64_t v = 0;
    if (y >= 0 && y < g.H && x >= 0 && x < g.W) v = *(const uint64_t*)(up + (size_t)y * g.W + x);
    *(uint64_t*)(tile + ly * PC8_FW + lx) = v;
  }
  const bool staged = a.reach <= UPD_STAGE_REACH;  // cw % 4 == 0 and the chroma planes are 4-byte aligned on this path
  const int ry0 = yb - UPD_STAGE_REACH, rx0 = xb - UPD_STAGE_REACH;
  if (staged) {
    for (int q = 0; q < 2; q++)
      if (s[q].high && (((uintptr_t)(s[q].high + coff)) & 3) == 0) upd_stage_region<RW, RH>(sR[q], ry0, rx0, s[q].high + coff, cw, chh, tid);
    __syncthreads();
  }
  for (int q = 0; q < 2; q++)
    if (s[q].high) {
      const uint8_t* reg = staged && (((uintptr_t)(s[q].high + coff)) & 3) == 0 ? sR[q] : nullptr;
      sc[q].run(tile, reg, ry0, rx0, RW, ty0, xb, s[q].high + coff, cw, s[q].mvx, s[q].mvy, g.BX, g.bs, chh, cw, g.H, g.W, a.uf, lane);
    }
  {
    auto border_cell = [&](int y, int x) {
      int v = up[(size_t)y * g.W + x];
      for (int k = 0; k < 2; k++)
        if (s[k].high) {
          const uint8_t* h = s[k].high + coff;
          auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * cw + sx] - 128; };
          v = update_gather(v, R, s[k].mvx, s[k].mvy, g.BY, g.BX, g.bs, g.H, g.W, chh, cw, a.reach, y, x, a.uf);
        }
      tile[(y - ty0) * PC8_FW + (x - xb)] = (uint8_t)v;
    };
    for (int e = 0; e < 2; e++) {  // picture rows 0 and H-1: 80 cells, two rounds of lanes
      const int y = e ? g.H - 1 : 0;
      if (y < ty0 || y >= ty0 + SR) continue;  // uniform
      for (int x = xb + lane; x < xb + PC8_FW; x += 64)
        if (x >= 0 && x < g.W) border_cell(y, x);
    }
    for (int e = 0; e < 2; e++) {  // picture columns 0 and W-1
      const int x = e ? g.W - 1 : 0, y = ty0 + lane;
      if (x >= xb && x < xb + PC8_FW && lane < SR && y > 0 && y < g.H - 1) border_cell(y, x);
    }
  }
  __syncthreads();
  for (int k = tid; k < PC_FH * PC_TW; k += 256) {
    int ry = k / PC_TW, cx = i0 + k % PC_TW;
    int yy = yb + ry;
    int v = 0;
    if (yy >= 0 && yy < g.H && cx < cw) {
      auto row = [&](int xx) { return (int)sP[ry * PC8_FW + xx - xb]; };
      v = a53_low(row, cx, g.W);
    }
    sL[ry][k % PC_TW] = (int16_t)v;
  }
  __syncthreads();
  if (ci < cw && cj < chh) {
    const int lx = tid & (PC_TW - 1);
    auto col = [&](int yy) { return (int)sL[yy - yb][lx]; };
    a.low[(size_t)(a.dst_mul * j) * g.F + coff + (size_t)cj * cw + ci] = (uint8_t)(int16_t)a53_low(col, cj, g.H);
  }
}
bool upd_fast_path(const UpdArgs& a) { return upd_run_applies(a) && tuning().upd_run; }
void launch_update(hipStream_t st, const UpdArgs& a) {
  const LevelGeom& g = a.g;
  dim3 block(64, 4);
  if (upd_fast_path(a)) {  // mctf_upd.hip: ordered gather by runs of 8 targets, chroma interpolated inside its tiles; one launch
    launch_update_run(st, a);
    return;
  }
  const dim3 cgrid((g.W / 2 + PC_TW - 1) / PC_TW, (g.H / 2 + PC_TH - 1) / PC_TH, (g.nB + 1) * 2);
  hipLaunchKernelGGL(k_upd_luma_tile, dim3((g.W + UL_TW - 1) / UL_TW, (g.H + 4 * UL_TH - 1) / (4 * UL_TH), g.nB + 1), dim3(256), 0, st, a);
  if (a.cup && g.W % 8 == 0) {
    const size_t n = (size_t)2 * (g.W / 2) * (g.H / 2);
    hipLaunchKernelGGL(k_upd_chroma_copy, dim3((unsigned)((n / 16 + 256) / 256), g.nB + 1), dim3(256), 0, st, a);
    // only the tiles an update can reach: output columns < (cw + reach + 2) / 2, rows < (chh + reach + 2) / 2 (Q3)
    const int cw = g.W / 2, chh = g.H / 2;
    const int tx = imin((int)cgrid.x, ((cw + a.reach + 2) / 2 + PC_TW - 1) / PC_TW + 1), ty = imin((int)cgrid.y, ((chh + a.reach + 2) / 2 + PC_TH - 1) / PC_TH + 1);
    hipLaunchKernelGGL(k_upd_chroma_tile, dim3(tx, ty, cgrid.z), block, 0, st, a);
  } else hipLaunchKernelGGL(k_upd_chroma, cgrid, block, 0, st, a);
}

// max |component| of a motion buffer -> *out (int, pre-zeroed)
__global__ void __launch_bounds__(256) k_mv_maxabs(const int16_t* mv, size_t n, int* out) {
  int m = 0;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) {
    int v = mv[k];
    v = v < 0 ? -v : v;
    m = v > m ? v : m;
  }
  for (int off = 32; off >= 1; off >>= 1) { int o = __shfl_xor(m, off, 64); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
void launch_mv_maxabs(hipStream_t st, const int16_t* mv, size_t n, int* out) {
  int nblk = (int)((n + 255) / 256);
  if (nblk > 256) nblk = 256;
  if (nblk < 1) nblk = 1;
  hipLaunchKernelGGL(k_mv_maxabs, dim3(nblk), dim3(256), 0, st, mv, n, out);
}

// ---------------------------------------------------------------------------------------------
// Generic truncating 5/3 level (API qsvc_dwt53_trunc): out-of-place, thread per output.
__global__ void __launch_bounds__(256) k_dwt53t_level(const int16_t* src, int16_t* dst, int stride, int ny, int nx, int inverse) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= nx || y >= ny) return;
  auto P = [&](int yy, int xx) { return (int)src[(size_t)yy * stride + xx]; };
  dst[(size_t)y * stride + x] = (int16_t)(inverse ? synthesize2d_sample(P, y, x, ny, nx) : analyze2d_coef(P, y, x, ny, nx));
}
void launch_dwt53t_level(hipStream_t st, const int16_t* src, int16_t* dst, int stride, int ny, int nx, int inverse) {
  hipLaunchKernelGGL(k_dwt53t_level, dim3((nx + 63) / 64, (ny + 3) / 4), dim3(64, 4), 0, st, src, dst, stride, ny, nx, inverse);
}

}  // namespace qsvc
