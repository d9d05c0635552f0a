// qsvc_amd/csrc/mctf_pyr.hip -- first level of the motion search's pyramid (dwt2d<short,_5_3>::analyze on the three luma planes,
// motion_estimate.cpp:283-285): the LL band of one truncating-5/3 analysis level of every u8 luma plane, out of an LDS tile
// (coalesced 8-byte row loads, packed 16-bit lifting, the picture's ends by symmetric extension -- mctf_runs.h).  The few
// coefficients the search reads OUTSIDE the LL quadrant (Q6/H3: the margin of the Mallat layout a block window can reach) and
// the deeper levels stay with the per-coefficient kernel of mctf_kernels.hip.
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"
#include "mctf_runs.h"

namespace qsvc {

__global__ void __launch_bounds__(PCT_T) k_pyr_ll_tile(const uint8_t* frames, size_t F, int W, int H, int16_t* dst, size_t dst_pic_stride, int R, int C) {
  __shared__ __attribute__((aligned(16))) uint8_t sP[PCT_FH][PCT_FW];
  __shared__ __attribute__((aligned(16))) uint32_t sL[PCT_FH][PCT_W / 2];
  const int tid = threadIdx.x, pic = blockIdx.z;
  const int i0 = blockIdx.x * PCT_W, j0 = blockIdx.y * PCT_H;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  const uint8_t* src = frames + (size_t)pic * F;
  for (int k = tid; k < PCT_FH * (PCT_FW / 8); k += PCT_T) {
    const int ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8), yy = yb + ry, xx = xb + 8 * run;
    uint64_t v = 0;
    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = *(const uint64_t*)(src + (size_t)yy * W + xx);
    *(uint64_t*)&sP[ry][8 * run] = v;
  }
  __syncthreads();
  LevelGeom g;
  g.W = W; g.H = H;
  chroma_tile_lows(sP, sL, g, i0, xb, yb, tid);
  if (tid < PCT_H * (PCT_W / 4)) {
    const int jr = tid / (PCT_W / 4), gq = tid % (PCT_W / 4), cj = j0 + jr, ci = i0 + 4 * gq;
    if (cj < R && ci < C) {  // R <= H/2, C <= W/2, multiples of 4 columns
      uint32_t ll[2];
      chroma_tile_ll(sL, jr, gq, ll);
      *(uint2*)(dst + (size_t)pic * dst_pic_stride + (size_t)cj * W + ci) = make_uint2(ll[0], ll[1]);
    }
  }
}
// Deeper levels: the same tile from the int16 LL quadrant of the level above (the top-left ny x nx of its snapshot).  The footprint
// is staged as pairs of samples with the region's ends mirrored (so neither pass has a boundary case), the row pass de-interleaves
// even / odd samples with v_perm and uses the truncating divisions of 5_3.cpp on both 16-bit lanes.
__device__ __forceinline__ int pyr_reflect(int i, int n) {  // whole-sample symmetric extension into [0, n), n >= 4, |overshoot| <= 8
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
__global__ void __launch_bounds__(PCT_T) k_pyr_ll_tile16(const int16_t* src, int16_t* dst, size_t pic_stride, int W, int ny, int nx, int R, int C) {
  __shared__ __attribute__((aligned(16))) uint32_t sX[PCT_FH][PCT_FW / 2];
  __shared__ __attribute__((aligned(16))) uint32_t sL[PCT_FH][PCT_W / 2];
  const int tid = threadIdx.x, pic = blockIdx.z;
  const int i0 = blockIdx.x * PCT_W, j0 = blockIdx.y * PCT_H;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  const int16_t* s = src + (size_t)pic * pic_stride;
  for (int k = tid; k < PCT_FH * (PCT_FW / 2); k += PCT_T) {
    const int ry = k / (PCT_FW / 2), q = k % (PCT_FW / 2), yy = yb + ry, xx = xb + 2 * q;
    uint32_t v;
    if ((unsigned)yy < (unsigned)ny && xx >= 0 && xx + 1 < nx) v = *(const uint32_t*)(s + (size_t)yy * W + xx);
    else {
      const int16_t* row = s + (size_t)pyr_reflect(yy, ny) * W;
      v = (uint32_t)(uint16_t)row[pyr_reflect(xx, nx)] | ((uint32_t)(uint16_t)row[pyr_reflect(xx + 1, nx)] << 16);
    }
    sX[ry][q] = v;
  }
  __syncthreads();
  for (int k = tid; k < PCT_FH * (PCT_W / 4); k += PCT_T) {
    const int ry = k / (PCT_W / 4), gq = k % (PCT_W / 4);
    const uint32_t* d = &sX[ry][4 * gq + 3];   // pairs 4 gq + 3 .. 4 gq + 8: samples 8 gq + 6 .. 8 gq + 17 of the footprint row
    const uint32_t E0 = __builtin_amdgcn_perm(d[1], d[0], 0x05040100u), E1 = __builtin_amdgcn_perm(d[3], d[2], 0x05040100u),
                   E2 = __builtin_amdgcn_perm(d[5], d[4], 0x05040100u);
    const uint32_t O0 = __builtin_amdgcn_perm(d[1], d[0], 0x07060302u), O1 = __builtin_amdgcn_perm(d[3], d[2], 0x07060302u),
                   O2 = __builtin_amdgcn_perm(d[5], d[4], 0x07060302u);
    const uint32_t Es0 = __builtin_amdgcn_alignbit(E1, E0, 16), Es1 = __builtin_amdgcn_alignbit(E2, E1, 16), Es2 = __builtin_amdgcn_alignbit(0u, E2, 16);
    const uint32_t H0 = pk_sub(O0, pk_tdiv2(pk_add(E0, Es0))), H1 = pk_sub(O1, pk_tdiv2(pk_add(E1, Es1))), H2 = pk_sub(O2, pk_tdiv2(pk_add(E2, Es2)));
    const uint32_t L01 = pk_add(Es0, pk_tdiv4(pk_add(__builtin_amdgcn_alignbit(H1, H0, 16), H0)));
    const uint32_t L23 = pk_add(Es1, pk_tdiv4(pk_add(__builtin_amdgcn_alignbit(H2, H1, 16), H1)));
    *(uint2*)&sL[ry][2 * gq] = make_uint2(L01, L23);
  }
  __syncthreads();
  if (tid < PCT_H * (PCT_W / 4)) {
    const int jr = tid / (PCT_W / 4), gq = tid % (PCT_W / 4), cj = j0 + jr, ci = i0 + 4 * gq;
    if (cj < R && ci < C) {
      uint32_t ll[2];
      chroma_tile_ll(sL, jr, gq, ll);
      *(uint2*)(dst + (size_t)pic * pic_stride + (size_t)cj * W + ci) = make_uint2(ll[0], ll[1]);
    }
  }
}
bool pyr_tile16_applies(int W, int ny, int nx) { return W % 8 == 0 && ny % 2 == 0 && nx % 8 == 0 && nx >= 16 && ny >= 8; }
void launch_pyr_ll_tile16(hipStream_t st, const int16_t* src, int16_t* dst, int W, int H, int P, int ny, int nx, int R, int C) {
  hipLaunchKernelGGL(k_pyr_ll_tile16, dim3((C + PCT_W - 1) / PCT_W, (R + PCT_H - 1) / PCT_H, P), dim3(PCT_T), 0, st, src, dst, (size_t)H * W, W, ny, nx, R, C);
}
bool pyr_tile_applies(int W, int H) { return W % 8 == 0 && H % 2 == 0 && W >= 16 && H >= 8; }
// LL coefficients (rows < R, columns < C; R <= H/2, C <= W/2 rounded up to 4 by the caller within W/2) of level 1 of P pictures
void launch_pyr_ll_tile(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int P, int R, int C) {
  hipLaunchKernelGGL(k_pyr_ll_tile, dim3((C + PCT_W - 1) / PCT_W, (R + PCT_H - 1) / PCT_H, P), dim3(PCT_T), 0, st, frames, F, W, H, dst,
                     (size_t)H * W, R, C);
}

}  // namespace qsvc
