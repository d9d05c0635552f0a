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

namespace qsvc {

// ---------------------------------------------------------------------------------------------
// ME pyramid level: thread per coefficient, 64x4 tile per workgroup (row-contiguous stores).
template <class T>
__global__ void __launch_bounds__(256) k_pyr_level(const T* src, size_t src_pic_stride, int src_row_stride, int16_t* dst,
                                                    size_t dst_pic_stride, int H, int W, int l, int R, int C) {
  int x = blockIdx.x * 64 + threadIdx.x;
  int y = blockIdx.y * 4 + threadIdx.y;
  pyr_thread<T>(src, src_pic_stride, src_row_stride, dst, dst_pic_stride, H, W, l, R, C, x, y, blockIdx.z);
}
void launch_pyr_level_u8(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int P, int R, int C) {
  dim3 grid((C + 63) / 64, (R + 3) / 4, P), block(64, 4);
  hipLaunchKernelGGL(k_pyr_level<uint8_t>, grid, block, 0, st, frames, F, W, dst, (size_t)H * W, H, W, 1, R, C);
}
void launch_pyr_level_s16(hipStream_t st, const int16_t* src, int16_t* dst, int W, int H, int P, int l, int R, int C) {
  dim3 grid((C + 63) / 64, (R + 3) / 4, P), block(64, 4);
  hipLaunchKernelGGL(k_pyr_level<int16_t>, grid, block, 0, st, src, (size_t)H * W, W, dst, (size_t)H * W, H, W, l, R, C);
}

// ---------------------------------------------------------------------------------------------
// ME search: one wave per MC block; lanes split the block's pixels, 18 SADs reduced across the
// wave with DPP-free xor shuffles (64 lanes).
// LPB = lanes per MC block: 64 (one block per wave) or 16 (four 8x8 blocks per wave, 4 pixels per lane: the
// xor-shuffle reduction then needs 4 steps instead of 6 and serves four blocks at once).
template <class T, int LPB>
__global__ void __launch_bounds__(64) k_me_search(MeArgs a) {
  const int sub = threadIdx.x / LPB, lane = threadIdx.x % LPB;
  const int bx = blockIdx.x * (64 / LPB) + sub, by = blockIdx.y, i = blockIdx.z;
  const bool live = bx < a.nbx;
  MePlane<T> r0, r1, pr;
  me_block_setup<T>(a, i, r0, r1, pr);
  int pvx = 0, pvy = 0, nvx = 0, nvy = 0;
  int sad[18];
  if (live) {
    me_block_init_mv(a, i, by, bx, pvx, pvy, nvx, nvy);
    me_block_sads<T>(r0, r1, pr, a.g.bs, by, bx, pvy, pvx, nvy, nvx, lane, LPB, sad);
  } else {
#pragma unroll
    for (int k = 0; k < 18; k++) sad[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < 18; k++) {
    int v = sad[k];
#pragma unroll
    for (int off = LPB / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    sad[k] = v;
  }
  if (live && lane == 0) me_block_store(a, i, by, bx, sad, pvx, pvy, nvx, nvy);
}
void launch_me_search(hipStream_t st, const MeArgs& a) {
  dim3 block(64);
  if (a.g.bs * a.g.bs <= 64) {
    dim3 grid((a.nbx + 3) / 4, a.nby, a.g.nB);
    if (a.l == 0) hipLaunchKernelGGL((k_me_search<uint8_t, 16>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_me_search<int16_t, 16>), grid, block, 0, st, a);
  } else {
    dim3 grid(a.nbx, a.nby, a.g.nB);
    if (a.l == 0) hipLaunchKernelGGL((k_me_search<uint8_t, 64>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_me_search<int16_t, 64>), grid, block, 0, st, a);
  }
}

// ---------------------------------------------------------------------------------------------
// predict + residue, luma.  Fast form (block size a multiple of 8): a thread owns 8 consecutive pixels of one row --
// they share one MC block, so the two displaced reference runs are two unaligned 8-byte loads when they lie inside
// the picture (the common case); runs that touch the replicate border fall back to the per-pixel border rule
// (Q1/Q2).  The odd picture is read and the high picture written 8 bytes at a time.  LDS histograms as before.
__device__ __forceinline__ uint64_t ld8u(const uint8_t* p) {  // unaligned 8-byte load
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
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
      BorderGeom bg = make_border(g.H, g.W, 4 * g.S);
      for (int k = 0; k < 8; k++) p0[k] = (uint8_t)luma_b(r0, bg, y0, x0 + k);
    }
    if (!predicted) {
      for (int k = 0; k < 8; k++) p1[k] = 0;
    } else if (n == 8 && y1 >= 0 && y1 < g.H && x1 >= 0 && x1 + 8 <= g.W) {
      uint64_t v = ld8u(r1 + (size_t)y1 * g.W + x1);
      __builtin_memcpy(p1, &v, 8);
    } else {
      BorderGeom bg = make_border(g.H, g.W, 4 * g.S);
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
  BorderGeom bg = make_border(g.H, g.W, 4 * g.S);
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
// Q7: the chroma margin state picture i leaves behind (one launch per picture, in order)
__global__ void __launch_bounds__(256) k_pred_margin(PredArgs a, int i) {
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < margin_geom(a.g).count()) pred_margin_thread(a, k, i, 1 + (int)blockIdx.y);
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
void launch_predict(hipStream_t st, const PredArgs& a) {
  const LevelGeom& g = a.g;
  if (!a.odd_out) hipMemsetAsync(a.hist, 0, sizeof(unsigned) * 3 * 256 * g.nB, st);
  dim3 block(64, 4);
  if (a.margin) {
    size_t cnt = margin_geom(g).count();
    for (int i = 0; i + 1 < g.nB; i++)  // the last picture's state has no reader
      hipLaunchKernelGGL(k_pred_margin, dim3((unsigned)((cnt + 255) / 256), 2), dim3(256), 0, st, a, i);
  }
  if (g.bs % 8 == 0) hipLaunchKernelGGL(k_pred_luma8, dim3((g.W + 511) / 512, (g.H + 3) / 4, g.nB), block, 0, st, a);
  else hipLaunchKernelGGL(k_pred_luma, dim3((g.W + 63) / 64, (g.H + 3) / 4, g.nB), block, 0, st, a);
  hipLaunchKernelGGL(k_pred_chroma, dim3((g.W / 2 + PC_TW - 1) / PC_TW, (g.H / 2 + PC_TH - 1) / PC_TH, g.nB * 2), block, 0, st, a);
  if (a.odd_out) return;
  size_t per_pic = (size_t)4 * g.BY * g.BX;
  int nblk = (int)((per_pic + 255) / 256);
  if (nblk > 64) nblk = 64;
  hipLaunchKernelGGL(k_mv_hist, dim3(nblk, g.nB), dim3(256), 0, st, a.mv, per_pic, a.hist);
}

// ---------------------------------------------------------------------------------------------
// update: thread per target sample (ordered gather), all even pictures of the level at once.
__global__ void __launch_bounds__(256) k_upd_luma(UpdArgs a) {
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, j = blockIdx.z;
  if (x < a.g.W && y < a.g.H) upd_luma_thread(a, x, y, j);
}
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
void launch_update(hipStream_t st, const UpdArgs& a) {
  const LevelGeom& g = a.g;
  dim3 block(64, 4);
  hipLaunchKernelGGL(k_upd_luma, dim3((g.W + 63) / 64, (g.H + 3) / 4, g.nB + 1), block, 0, st, a);
  hipLaunchKernelGGL(k_upd_chroma, dim3((g.W / 2 + PC_TW - 1) / PC_TW, (g.H / 2 + PC_TH - 1) / PC_TH, (g.nB + 1) * 2), block, 0, st, a);
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
