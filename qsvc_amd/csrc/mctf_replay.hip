// qsvc_amd/csrc/mctf_replay.hip -- the search pyramid of Q5 geometries by LITERAL REPLAY (me_for_image, motion_estimate.cpp:260-358).
//
// The reference analyses its three luma planes L levels in place with FLOOR-halved sizes (dwt2d.cpp:78-79: x >>= 1) and walks
// back up by one-level syntheses at CEIL-halved sizes (motion_estimate.cpp:307-309: desp()).  While every intermediate size is even
// the two agree, the walk is perfect reconstruction and the state at level l is the l-level analysis of the picture (the
// snapshot pyramid of mctf_pyr.hip).  When a size along the way is odd (1080 rows at search range 32, every CIF size at 128) they
// differ: the synthesis reads one sample the analysis never wrote, the walk does not reconstruct, and
//   * the state searched at level l is whatever the replay left -- including level 0, which is no longer the picture, and
//   * the plane that was NEXT for B picture i-1 is PREV for B picture i in the state the first replay left it in
//     (the swap of motion_estimate.cpp:902-906), so an even picture's state depends on the ROLE it plays.
// So here the replay is done as the reference does it, on one int16 plane per (B picture, role): slot 3i = PREV, 3i+1 = predicted,
// 3i+2 = NEXT (MeArgs::pic_mul = 3).  PREV of picture i >= 1 starts from T(E_i), the even picture after one whole
// analyse / synthesise walk, which a pre-pass computes for all of them at once.  The border cells of a plane keep the pixel-domain
// values they were filled with before the first analysis (border_map over the u8 luma, as on the snapshot path).
// Thread per coefficient: this is the rare path (a handful of pictures at the deepest temporal levels), exactness first.
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"

namespace qsvc {

__global__ void __launch_bounds__(256) k_rp_load(const uint8_t* frames, size_t F, int W, int H, int16_t* dst, size_t N, int slot0,
                                                  int slot_step, int pic0, int pic_step) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (x >= W || y >= H) return;
  const uint8_t* s = frames + (size_t)(pic0 + k * pic_step) * F;
  dst[(size_t)(slot0 + k * slot_step) * N + (size_t)y * W + x] = s[(size_t)y * W + x];   // texture::read (texture.cpp:122-129)
}
void launch_rp_load(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int slot0, int slot_step, int pic0,
                    int pic_step, int count) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_rp_load, dim3((W + 63) / 64, (H + 3) / 4, count), dim3(64, 4), 0, st, frames, F, W, H, dst, (size_t)H * W, slot0,
                     slot_step, pic0, pic_step);
}

// one level of dwt2d::analyze (dwt2d.cpp:84-117) / dwt2d::synthesize (:139-173) over the top-left ny x nx of `count` planes
// (plane k at src + k * plane_stride, row stride W): src -> tmp, then the region back (the transform is in place in the reference)
__global__ void __launch_bounds__(256) k_rp_dwt(const int16_t* src, int16_t* dst, size_t plane_stride, int W, int ny, int nx, int inverse) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= nx || y >= ny) return;
  const int16_t* s = src + (size_t)blockIdx.z * plane_stride;
  auto P = [&](int yy, int xx) { return (int)s[(size_t)yy * W + xx]; };
  dst[(size_t)blockIdx.z * plane_stride + (size_t)y * W + x] = (int16_t)(inverse ? synthesize2d_sample(P, y, x, ny, nx) : analyze2d_coef(P, y, x, ny, nx));
}
__global__ void __launch_bounds__(256) k_rp_copy(const int16_t* src, int16_t* dst, size_t plane_stride, int W, int ny, int nx) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x < nx && y < ny) dst[(size_t)blockIdx.z * plane_stride + (size_t)y * W + x] = src[(size_t)blockIdx.z * plane_stride + (size_t)y * W + x];
}
void launch_rp_dwt(hipStream_t st, int16_t* planes, int16_t* tmp, size_t plane_stride, int count, int W, int ny, int nx, int inverse) {
  if (count <= 0 || ny < 1 || nx < 1) return;
  if (ny < 2 && nx < 2) return;  // a 1 x 1 region: both passes copy (5_3.cpp needs two samples)
  const dim3 grid((nx + 63) / 64, (ny + 3) / 4, count), block(64, 4);
  hipLaunchKernelGGL(k_rp_dwt, grid, block, 0, st, planes, tmp, plane_stride, W, ny, nx, inverse);
  hipLaunchKernelGGL(k_rp_copy, grid, block, 0, st, tmp, planes, plane_stride, W, ny, nx);
}

}  // namespace qsvc
