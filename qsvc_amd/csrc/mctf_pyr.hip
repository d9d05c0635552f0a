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
bool pyr_tile_applies(int W, int H) { return W % 8 == 0 && H % 2 == 0 && W >= 16 && H >= 8; }
// LL coefficients (rows < R, columns < C; R <= H/2, C <= W/2 rounded up to 4 by the caller within W/2) of level 1 of P pictures
void launch_pyr_ll_tile(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int P, int R, int C) {
  hipLaunchKernelGGL(k_pyr_ll_tile, dim3((C + PCT_W - 1) / PCT_W, (R + PCT_H - 1) / PCT_H, P), dim3(PCT_T), 0, st, frames, F, W, H, dst,
                     (size_t)H * W, R, C);
}

}  // namespace qsvc
