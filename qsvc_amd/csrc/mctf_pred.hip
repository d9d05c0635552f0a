// qsvc_amd/csrc/mctf_pred.hip -- predict + residue of a whole temporal level (decorrelate.cpp:69-189, 841-861, 920-950), the
// form used when the block size is a multiple of 8, the picture width a multiple of 8 and the block grid tiles the picture
// (analysis direction; the general kernels of mctf_kernels.hip cover everything else).
//
// Everything works on RUNS OF 8 SAMPLES of one row that share one MC block, as packed bytes / packed 16-bit pairs:
//   * a displaced reference run is one unaligned 8-byte load; per-byte (a + b) >> 1 without carries between bytes;
//   * the residue clamp(odd - pred, -128, 127) + 128 on 16-bit pairs (v_pk_sub_i16, v_pk_max/min_i16);
//   * the luma histograms go to 16 interleaved copies per bin in LDS (the residue histogram is one narrow peak: with a single
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

namespace qsvc {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 S2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t U(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return U(S2(a) + S2(b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return U(S2(a) - S2(b)); }
__device__ __forceinline__ uint32_t pk_clamp255(uint32_t a) {
  s16x2 v = S2(a), z = {0, 0}, m = {255, 255};
  v = __builtin_elementwise_max(v, z);
  v = __builtin_elementwise_min(v, m);
  return U(v);
}
// C's truncating division of both 16-bit lanes (5_3.cpp divides shorts promoted to int)
__device__ __forceinline__ uint32_t pk_tdiv2(uint32_t a) {
  const u16x2 sign = __builtin_bit_cast(u16x2, a) >> 15;
  return U((S2(a) + __builtin_bit_cast(s16x2, sign)) >> 1);
}
__device__ __forceinline__ uint32_t pk_tdiv4(uint32_t a) {
  const uint32_t bias = U(S2(a) >> 15) & 0x00030003u;
  return U((S2(a) + S2(bias)) >> 2);
}
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return (a & b) + (((a ^ b) & 0xFEFEFEFEu) >> 1); }  // per byte (a + b) >> 1
__device__ __forceinline__ uint64_t avg8(uint64_t a, uint64_t b) { return (a & b) + (((a ^ b) & 0xFEFEFEFEFEFEFEFEull) >> 1); }

// 8 samples of a row of the luma plane displaced by a vector, replicate border (Q1/Q2) when the run leaves the picture
__device__ __forceinline__ uint64_t luma_run8(const uint8_t* r, const LevelGeom& g, int y, int x) {
  if ((unsigned)y < (unsigned)g.H && x >= 0 && x + 8 <= g.W) return ld8(r + (size_t)y * g.W + x);
  const BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  uint64_t v = 0;
  for (int k = 0; k < 8; k++) v |= (uint64_t)luma_b(r, bg, y, x + k) << (8 * k);
  return v;
}
// 8 samples (y, x .. x+7) of the x2-interpolated chroma plane, computed from the stored half-resolution plane u
__device__ __forceinline__ uint64_t chroma_run8(const uint8_t* u, const LevelGeom& g, int y, int x) {
  const int cw = g.W >> 1, chh = g.H >> 1;
  if ((unsigned)y < (unsigned)g.H && x >= 0 && x + 8 < g.W) {
    const int ja = y >> 1, jb = imin((y + 1) >> 1, chh - 1), k0 = x >> 1;  // odd rows: mean of two stored rows (the last row repeats)
    const uint64_t a = ld8(u + (size_t)ja * cw + k0), b = ld8(u + (size_t)jb * cw + k0);
    const uint64_t c = avg8(a, b);                                         // columns first (decorrelate.cpp:610-648)
    const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
    const uint32_t cs = __builtin_amdgcn_alignbyte(c1, c0, 1);             // c[1..4]
    const uint32_t m = avg4(c0, cs);                                       // m[k] = (c[k] + c[k+1]) >> 1
    const bool odd = x & 1;
    const uint32_t A = odd ? m : c0, B = odd ? cs : m;                     // even x: c0 m0 c1 m1 ..., odd x: m0 c1 m1 c2 ...
    const uint32_t lo = __builtin_amdgcn_perm(B, A, 0x05010400u), hi = __builtin_amdgcn_perm(B, A, 0x07030602u);
    return (uint64_t)lo | ((uint64_t)hi << 32);
  }
  const BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  uint64_t v = 0;
  for (int k = 0; k < 8; k++) v |= (uint64_t)chroma_up_b(u, bg, y, x + k) << (8 * k);
  return v;
}
// clamp(o - p, -128, 127) + 128 of four samples
__device__ __forceinline__ uint32_t residue4(uint32_t o, uint32_t p) {
  const uint32_t M = 0x00FF00FFu, C = 0x00800080u;
  const uint32_t te = pk_clamp255(pk_sub((o & M) + C, p & M));
  const uint32_t to = pk_clamp255(pk_sub(((o >> 8) & M) + C, (p >> 8) & M));
  return te | (to << 8);
}

// ---------------------------------------------------------------------------------------------
// luma: a thread owns one run of 8 samples in each of 4 rows; LDS histograms of the odd picture and of residue + 128
#define PL_K 16   // histogram copies per bin
__global__ void __launch_bounds__(256) k_pred_luma_run(PredArgs a) {
  __shared__ unsigned hs[2][256 * PL_K];
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int k = tid; k < 2 * 256 * PL_K; k += 256) (&hs[0][0])[k] = 0;
  __syncthreads();
  const LevelGeom& g = a.g;
  const int x = (blockIdx.x * 64 + threadIdx.x) * 8, i = blockIdx.z;
  const unsigned cp = threadIdx.x & (PL_K - 1);
  if (x < g.W) {
    const uint8_t* r0 = frame_y(a.frames, g, 2 * i);
    const uint8_t* r1 = frame_y(a.frames, g, 2 * i + 2);
    const uint8_t* od = frame_y(a.frames, g, 2 * i + 1);
    const size_t fld = (size_t)g.BY * g.BX;
    const int lbs = 31 - __builtin_clz(g.bs);  // block size is a power of two on this path
    const int bx = x >> lbs;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int y = blockIdx.y * 16 + rr * 4 + threadIdx.y;
      if (y >= g.H) break;
      const int16_t* m = a.mv + (size_t)i * 4 * fld + (size_t)(y >> lbs) * g.BX + bx;
      const uint64_t p0 = luma_run8(r0, g, y + m[fld], x + m[0]);
      const uint64_t p1 = luma_run8(r1, g, y + m[3 * fld], x + m[2 * fld]);
      const uint64_t p = avg8(p0, p1);                                   // decorrelate.cpp:99-107
      const uint64_t o = ld8(od + (size_t)y * g.W + x);
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
#define PCT_W 64                      // outputs per tile row
#define PCT_H 14                      // output rows per tile
#define PCT_FH (2 * PCT_H + 3)        // full-resolution rows 2 j0 - 2 .. 2 j0 + 2 PCT_H
#define PCT_FW (2 * PCT_W + 16)       // full-resolution columns 2 i0 - 8 .. 2 i0 + 2 PCT_W + 7, in runs of 8
__global__ void __launch_bounds__(256) k_pred_chroma_run(PredArgs a) {
  __shared__ __attribute__((aligned(16))) uint8_t sP[PCT_FH][PCT_FW];
  __shared__ __attribute__((aligned(16))) uint32_t sL[PCT_FH][PCT_W / 2];  // row-pass lows, two columns per word
  const LevelGeom& g = a.g;
  const int tid = threadIdx.x;
  const int i = blockIdx.z >> 1, c = 1 + (blockIdx.z & 1);
  const int i0 = blockIdx.x * PCT_W, j0 = blockIdx.y * PCT_H;
  const int cw = g.W >> 1, chh = g.H >> 1;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  const size_t fld = (size_t)g.BY * g.BX;
  const int lbs = 31 - __builtin_clz(g.bs);
  const uint8_t* u0 = frame_c(a.frames, g, 2 * i, c);
  const uint8_t* u1 = frame_c(a.frames, g, 2 * i + 2, c);
  // A. full-resolution prediction of the footprint, one run per task
  for (int k = tid; k < PCT_FH * (PCT_FW / 8); k += 256) {
    const int ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8), yy = yb + ry, xx = xb + 8 * run;
    uint64_t p = 0;
    if ((unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) {
      const int16_t* m = a.mv + (size_t)i * 4 * fld + (size_t)(yy >> lbs) * g.BX + (xx >> lbs);
      const uint64_t p0 = chroma_run8(u0, g, yy + m[fld], xx + m[0]);
      const uint64_t p1 = chroma_run8(u1, g, yy + m[3 * fld], xx + m[2 * fld]);
      p = avg8(p0, p1);
    }
    *(uint64_t*)&sP[ry][8 * run] = p;
  }
  __syncthreads();
  // the picture's left / right ends: whole-sample symmetric extension (see the file header)
  if (xb < 0 && tid < PCT_FH) { sP[tid][6] = sP[tid][10]; sP[tid][7] = sP[tid][9]; }
  if (xb + PCT_FW > g.W && g.W - xb >= 0 && tid >= 64 && tid < 64 + PCT_FH) sP[tid - 64][g.W - xb] = sP[tid - 64][g.W - xb - 2];
  __syncthreads();
  // B. row pass: lows of 4 outputs per task out of 4 words of the row
  for (int k = tid; k < PCT_FH * (PCT_W / 4); k += 256) {
    const int ry = k / (PCT_W / 4), gq = k % (PCT_W / 4), yy = yb + ry;
    if ((unsigned)yy >= (unsigned)g.H || i0 + 4 * gq >= cw) continue;
    const uint32_t* row = (const uint32_t*)&sP[ry][8 * gq + 4];
    const uint32_t D0 = row[0], D1 = row[1], D2 = row[2], D3 = row[3];   // columns 4..19 of the run pair (centres 8, 10, 12, 14)
    const uint32_t M = 0x00FF00FFu;
    const uint32_t E0 = D0 & M, E1 = D1 & M, E2 = D2 & M, E3 = D3 & M;   // {4,6} {8,10} {12,14} {16,18}
    const uint32_t O0 = (D0 >> 8) & M, O1 = (D1 >> 8) & M, O2 = (D2 >> 8) & M;  // {5,7} {9,11} {13,15}
    const uint32_t e68 = __builtin_amdgcn_alignbit(E1, E0, 16), e1012 = __builtin_amdgcn_alignbit(E2, E1, 16), e1416 = __builtin_amdgcn_alignbit(E3, E2, 16);
    const uint32_t o79 = __builtin_amdgcn_alignbit(O1, O0, 16), o1113 = __builtin_amdgcn_alignbit(O2, O1, 16), o15 = O2 >> 16;
    const uint32_t HA = pk_sub(o79, (pk_add(e68, E1) >> 1) & 0x7FFF7FFFu);      // {h7, h9}: sums of two samples are >= 0
    const uint32_t HB = pk_sub(o1113, (pk_add(e1012, E2) >> 1) & 0x7FFF7FFFu);  // {h11, h13}
    const uint32_t HC = pk_sub(o15, (pk_add(e1416, E3) >> 1) & 0x7FFF7FFFu);    // {h15, -}
    const uint32_t L01 = pk_add(E1, pk_tdiv4(pk_add(__builtin_amdgcn_alignbit(HB, HA, 16), HA)));
    const uint32_t L23 = pk_add(E2, pk_tdiv4(pk_add(__builtin_amdgcn_alignbit(HC, HB, 16), HB)));
    *(uint2*)&sL[ry][2 * gq] = make_uint2(L01, L23);
  }
  __syncthreads();
  // the picture's top / bottom ends: mirror rows of the row-pass output
  if (yb < 0 && tid < PCT_W / 2) { sL[0][tid] = sL[4][tid]; sL[1][tid] = sL[3][tid]; }
  if (yb + PCT_FH > g.H && g.H - yb >= 0 && g.H - yb < PCT_FH && tid >= 64 && tid < 64 + PCT_W / 2) sL[g.H - yb][tid - 64] = sL[g.H - yb - 2][tid - 64];
  __syncthreads();
  // C. column pass + residue: 4 columns x 1 row per task
  if (tid < PCT_H * (PCT_W / 4)) {
    const int jr = tid / (PCT_W / 4), gq = tid % (PCT_W / 4), cj = j0 + jr, ci = i0 + 4 * gq;
    if (cj < chh && ci < cw) {
      uint32_t ll[2];
#pragma unroll
      for (int w = 0; w < 2; w++) {
        const uint32_t r0 = sL[2 * jr][2 * gq + w], r1 = sL[2 * jr + 1][2 * gq + w], r2 = sL[2 * jr + 2][2 * gq + w],
                       r3 = sL[2 * jr + 3][2 * gq + w], r4 = sL[2 * jr + 4][2 * gq + w];
        const uint32_t hA = pk_sub(r1, pk_tdiv2(pk_add(r0, r2))), hB = pk_sub(r3, pk_tdiv2(pk_add(r2, r4)));
        ll[w] = pk_add(r2, pk_tdiv4(pk_add(hA, hB)));
      }
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
void launch_predict_run(hipStream_t st, const PredArgs& a) {
  const LevelGeom& g = a.g;
  hipLaunchKernelGGL(k_pred_luma_run, dim3((g.W + 511) / 512, (g.H + 15) / 16, g.nB), dim3(64, 4), 0, st, a);
  hipLaunchKernelGGL(k_pred_chroma_run, dim3((g.W / 2 + PCT_W - 1) / PCT_W, (g.H / 2 + PCT_H - 1) / PCT_H, g.nB * 2), dim3(256), 0, st, a);
}

}  // namespace qsvc
