// qsvc_amd/csrc/mctf_pred.hip -- predict + residue of a whole temporal level (decorrelate.cpp:69-189, 841-861, 920-950), the
// form used when the block size is a multiple of 8, the picture width a multiple of 8 and the block grid tiles the picture
// (analysis direction; the general kernels of mctf_kernels.hip cover everything else).
//
// Everything works on RUNS OF 8 SAMPLES of one row that share one MC block, as packed bytes / packed 16-bit pairs:
//   * a displaced reference run is one unaligned 8-byte load; per-byte (a + b) >> 1 without carries between bytes;
//   * the residue clamp(odd - pred, -128, 127) + 128 on 16-bit pairs (v_pk_sub_i16, v_pk_max/min_i16);
//   * the luma histograms go to 8 interleaved copies per bin in LDS (the residue histogram is one narrow peak: with a single
//     copy the 64 lanes of a wave queue up on one address);
//   * chroma is predicted at full resolution straight from the HALF-resolution planes -- the x2 interpolation of
//     decorrelate.cpp:610-648 (one level of truncating 5/3 synthesis with zero high bands = averages of neighbours, columns first)
//     of a displaced run needs 5 samples of one or two stored rows: no interpolated plane is ever written to memory --
//     and taken back to half resolution (decorrelate.cpp:860-861: the LL band of one analysis level, rows then columns) out of an
//     LDS tile with packed 16-bit lifting.  The picture's ends follow 5_3.cpp's rules, which for even lengths are exactly
//     whole-sample symmetric extension (h[-1] = h[0] makes tdiv4(2 h0) = tdiv2(h0); s[n] = s[n-2] makes the last high
//     s[n-1] - s[n-2]), so the tile mirrors two cells and runs the interior expression everywhere.
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"
#include "mctf_runs.h"

namespace qsvc {

// ---------------------------------------------------------------------------------------------
// luma: a thread owns one run of 8 samples in each of 4 rows; LDS histograms of the odd picture and of residue + 128
#ifndef PL_K
#define PL_K 8    // histogram copies per bin: 16 KB of LDS per workgroup = 8 waves per SIMD (round 4: with 16 copies and 32 KB the kernel ran
#endif            // at 5 waves per SIMD and waited half of its cycles on memory; 8 copies: 0.187 -> 0.158 ms per 1080p step, 4K 1.30 -> 1.04 ms)
__device__ __forceinline__ void pred_luma_body(const PredArgs& a, uint8_t* smem, const int bx_, const int by_, const int bz_, const int tx_, const int ty_) {
  unsigned (*hs)[256 * PL_K] = (unsigned (*)[256 * PL_K])smem;
  const int tid = ty_ * 64 + tx_;
  for (int k = tid; k < 2 * 256 * PL_K; k += 256) (&hs[0][0])[k] = 0;
  __syncthreads();
  const LevelGeom& g = a.g;
  const int x = (bx_ * 64 + tx_) * 8, i = bz_;
  const unsigned cp = tx_ & (PL_K - 1);
  if (x < g.W) {
    const uint8_t* r0 = frame_y(a.frames, g, 2 * i);
    const uint8_t* r1 = frame_y(a.frames, g, 2 * i + 2);
    const uint8_t* od = frame_y(a.frames, g, 2 * i + 1);
    const size_t fld = (size_t)g.BY * g.BX;
    const int lbs = 31 - __builtin_clz(g.bs);  // block size is a power of two on this path
    const int bx = x >> lbs;
    // the four rows' requests are issued together: vectors, then the three runs of every row, then the arithmetic
    int yv[4], mv4[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      yv[rr] = imin(by_ * 16 + rr * 4 + ty_, g.H - 1);
      const int16_t* m = a.mv + (size_t)i * 4 * fld + (size_t)(yv[rr] >> lbs) * g.BX + bx;
      mv4[rr][0] = m[0]; mv4[rr][1] = m[fld]; mv4[rr][2] = m[2 * fld]; mv4[rr][3] = m[3 * fld];
    }
    uint64_t l0[4], l1[4], ov[4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      l0[rr] = luma_run8_ld(r0, g, yv[rr] + mv4[rr][1], x + mv4[rr][0]);
      l1[rr] = luma_run8_ld(r1, g, yv[rr] + mv4[rr][3], x + mv4[rr][2]);
      ov[rr] = ld8(od + (size_t)yv[rr] * g.W + x);
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int y = by_ * 16 + rr * 4 + ty_;
      if (y >= g.H) break;
      const uint64_t p0 = luma_run8_fin(l0[rr], r0, g, y + mv4[rr][1], x + mv4[rr][0]);
      const uint64_t p1 = luma_run8_fin(l1[rr], r1, g, y + mv4[rr][3], x + mv4[rr][2]);
      const uint64_t p = avg8(p0, p1);                                   // decorrelate.cpp:99-107
      const uint64_t o = ov[rr];
      const uint32_t h0 = residue4((uint32_t)o, (uint32_t)p), h1 = residue4((uint32_t)(o >> 32), (uint32_t)(p >> 32));
      const size_t off = (size_t)i * g.F + (size_t)y * g.W + x;
      *(uint2*)(a.high + off) = make_uint2(h0, h1);
      if (a.pred_dump) *(uint2*)(a.pred_dump + off) = make_uint2((uint32_t)p, (uint32_t)(p >> 32));
      const uint32_t ow[2] = {(uint32_t)o, (uint32_t)(o >> 32)}, hw[2] = {h0, h1};
#pragma unroll
      for (int w = 0; w < 2; w++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          atomicAdd(&hs[0][((ow[w] >> (8 * k)) & 255u) * PL_K + cp], 1u);
          atomicAdd(&hs[1][((hw[w] >> (8 * k)) & 255u) * PL_K + cp], 1u);
        }
    }
  }
  __syncthreads();
  unsigned s0 = 0, s1 = 0;
#pragma unroll
  for (int k = 0; k < PL_K; k++) { s0 += hs[0][tid * PL_K + ((k + tid) & (PL_K - 1))]; s1 += hs[1][tid * PL_K + ((k + tid) & (PL_K - 1))]; }
  unsigned* gh = a.hist + (size_t)i * 3 * 256;
  if (s0) atomicAdd(&gh[tid], s0);
  if (s1) atomicAdd(&gh[256 + tid], s1);
}

// ---------------------------------------------------------------------------------------------
// chroma: a workgroup produces a 64 x 14 tile of one half-resolution component.
__device__ __forceinline__ void pred_chroma_body(const PredArgs& a, uint8_t* smem, const int bx_, const int by_, const int bz_, const int tid) {
  uint8_t (*sP)[PCT_FW] = (uint8_t (*)[PCT_FW])smem;
  uint32_t (*sL)[PCT_W / 2] = (uint32_t (*)[PCT_W / 2])(smem + PCT_SP_BYTES);  // row-pass lows, two columns per word
  const LevelGeom& g = a.g;
  const int i = bz_ >> 1, c = 1 + (bz_ & 1);
  const int i0 = bx_ * PCT_W, j0 = by_ * PCT_H;
  const int cw = g.W >> 1, chh = g.H >> 1;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  const size_t fld = (size_t)g.BY * g.BX;
  const int lbs = 31 - __builtin_clz(g.bs);
  const uint8_t* u0 = frame_c(a.frames, g, 2 * i, c);
  const uint8_t* u1 = frame_c(a.frames, g, 2 * i + 2, c);
  // A. full-resolution prediction of the footprint, one run per task; a thread's (up to three) tasks have their requests in
  // flight together: vectors, then the two stored rows of both references, then the arithmetic
  constexpr int NT = (PCT_FH * (PCT_FW / 8) + PCT_T - 1) / PCT_T;
  int ty_[NT], tx_[NT], mv4[NT][4];
  bool in_[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int k = tid + PCT_T * t, ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8);
    ty_[t] = yb + ry; tx_[t] = xb + 8 * run;
    in_[t] = k < PCT_FH * (PCT_FW / 8) && (unsigned)ty_[t] < (unsigned)g.H && (unsigned)tx_[t] < (unsigned)g.W;
    const int16_t* m = a.mv + (size_t)i * 4 * fld + (in_[t] ? (size_t)(ty_[t] >> lbs) * g.BX + (tx_[t] >> lbs) : (size_t)0);
    mv4[t][0] = m[0]; mv4[t][1] = m[fld]; mv4[t][2] = m[2 * fld]; mv4[t][3] = m[3 * fld];
  }
  ChromaLd c0[NT], c1[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    c0[t] = chroma_run8_ld(u0, g, ty_[t] + mv4[t][1], tx_[t] + mv4[t][0]);
    c1[t] = chroma_run8_ld(u1, g, ty_[t] + mv4[t][3], tx_[t] + mv4[t][2]);
  }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int k = tid + PCT_T * t;
    if (k >= PCT_FH * (PCT_FW / 8)) break;
    uint64_t p = 0;
    if (in_[t]) {
      const uint64_t p0 = chroma_run8_fin(c0[t], u0, g, ty_[t] + mv4[t][1], tx_[t] + mv4[t][0]);
      const uint64_t p1 = chroma_run8_fin(c1[t], u1, g, ty_[t] + mv4[t][3], tx_[t] + mv4[t][2]);
      p = avg8(p0, p1);
    }
    *(uint64_t*)&sP[k / (PCT_FW / 8)][8 * (k % (PCT_FW / 8))] = p;
  }
  __syncthreads();
  chroma_tile_lows(sP, sL, g, i0, xb, yb, tid);
  // C. column pass + residue: 4 columns x 1 row per task
  if (tid < PCT_H * (PCT_W / 4)) {
    const int jr = tid / (PCT_W / 4), gq = tid % (PCT_W / 4), cj = j0 + jr, ci = i0 + 4 * gq;
    if (cj < chh && ci < cw) {
      uint32_t ll[2];
      chroma_tile_ll(sL, jr, gq, ll);
      const size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh + (size_t)cj * cw + ci;
      const size_t off = (size_t)i * g.F + coff;
      const uint32_t o = *(const uint32_t*)(frame_c(a.frames, g, 2 * i + 1, c) + (size_t)cj * cw + ci);
      const uint32_t C = 0x00800080u;
      const uint32_t t01 = pk_clamp255(pk_sub(__builtin_amdgcn_perm(0u, o, 0x0c010c00u) + C, ll[0]));
      const uint32_t t23 = pk_clamp255(pk_sub(__builtin_amdgcn_perm(0u, o, 0x0c030c02u) + C, ll[1]));
      *(uint32_t*)(a.high + off) = __builtin_amdgcn_perm(t23, t01, 0x06040200u);
      if (a.pred_dump) *(uint32_t*)(a.pred_dump + off) = __builtin_amdgcn_perm(ll[1], ll[0], 0x06040200u);  // short -> u8 wraps (Q4)
    }
  }
}

bool pred_run_applies(const PredArgs& a) {
  const LevelGeom& g = a.g;
  return !a.odd_out && !a.margin && g.ov < 2 && g.bs % 8 == 0 && (g.bs & (g.bs - 1)) == 0 && g.W % 8 == 0 && g.H % 2 == 0 && g.H % g.bs == 0 && g.W % g.bs == 0 &&
         g.H >= 8 && g.W >= 16;
}
// One launch per level: the chroma tiles (the longer workgroups) first, then the luma strips; a workgroup's role follows from
// its index.  256 threads, 32 KB of LDS (the luma histograms; a chroma tile needs 8.4 KB of it).
__global__ void __launch_bounds__(256) k_pred_level(PredArgs a, int ncx, int ncy, int nlx, int nly, int nmh) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[2 * 256 * PL_K * 4];
  static_assert(sizeof(smem) >= PCT_SMEM_BYTES, "chroma tile fits");
  int b = blockIdx.x;
  const int nchroma = ncx * ncy * a.g.nB * 2, nluma = nlx * nly * a.g.nB;
  if (b < nchroma) { pred_chroma_body(a, smem, b % ncx, (b / ncx) % ncy, b / (ncx * ncy), threadIdx.x); return; }
  b -= nchroma;
  if (b < nluma) { pred_luma_body(a, smem, b % nlx, (b / nlx) % nly, b / (nlx * nly), threadIdx.x & 63, threadIdx.x >> 6); return; }
  b -= nluma;
  // last role: the histogram of the motion components + 128 (decorrelate.cpp:805-812), nmh workgroups per picture, which also
  // copy their share of motion_t into motion_filtered_t
  const int i = b / nmh, part = b % nmh, tid = threadIdx.x;
  unsigned* h = (unsigned*)smem;
  h[tid] = 0;
  __syncthreads();
  const size_t per_pic = (size_t)4 * a.g.BY * a.g.BX;
  const int16_t* m = a.mv + (size_t)i * per_pic;
  for (size_t k = (size_t)part * 256 + tid; k < per_pic; k += (size_t)nmh * 256) {
    const int16_t v = m[k];
    atomicAdd(&h[(v + 128) & 255], 1u);
    if (a.mv_copy) a.mv_copy[(size_t)i * per_pic + k] = v;
  }
  __syncthreads();
  if (h[tid]) atomicAdd(&a.hist[((size_t)i * 3 + 2) * 256 + tid], h[tid]);
}
void launch_predict_run(hipStream_t st, const PredArgs& a) {
  const LevelGeom& g = a.g;
  const int nlx = (g.W + 511) / 512, nly = (g.H + 15) / 16, ncx = (g.W / 2 + PCT_W - 1) / PCT_W, ncy = (g.H / 2 + PCT_H - 1) / PCT_H;
  int nmh = (int)(((size_t)4 * g.BY * g.BX + 255) / 256);
  if (nmh > 64) nmh = 64;
  if (!a.hist_is_zero) hipMemsetAsync(a.hist, 0, sizeof(unsigned) * 3 * 256 * g.nB, st);
  hipLaunchKernelGGL(k_pred_level, dim3((unsigned)(ncx * ncy * g.nB * 2 + nlx * nly * g.nB + nmh * g.nB)), dim3(256), 0, st, a, ncx, ncy, nlx, nly, nmh);
}

}  // namespace qsvc
