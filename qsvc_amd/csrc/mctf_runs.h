// qsvc_amd/csrc/mctf_runs.h -- device helpers shared by mctf_pred.hip and mctf_upd.hip: packed 16-bit lifting, runs of 8 samples,
// the chroma tile (full-resolution footprint in LDS -> half-resolution LL band, decorrelate.cpp:860-861 / update.cpp:632-643).
#pragma once
#include <hip/hip_runtime.h>
#include "mctf_threads.h"

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

// 8 samples of a row of the luma plane displaced by a vector, replicate border (Q1/Q2) when the run leaves the picture.
// Two steps so that a caller can have the loads of several runs in flight: *_ld issues the load at a clamped (always valid)
// position, *_fin returns it when the run lies inside the picture and walks the border rule otherwise.
__device__ __forceinline__ bool luma_run8_in(const LevelGeom& g, int y, int x) { return (unsigned)y < (unsigned)g.H && x >= 0 && x + 8 <= g.W; }
__device__ __forceinline__ uint64_t luma_run8_ld(const uint8_t* r, const LevelGeom& g, int y, int x) {
  return ld8(r + (size_t)iclamp(y, 0, g.H - 1) * g.W + iclamp(x, 0, g.W - 8));
}
__device__ __forceinline__ uint64_t luma_run8_fin(uint64_t ld, const uint8_t* r, const LevelGeom& g, int y, int x) {
  if (luma_run8_in(g, y, x)) return ld;
  const BorderGeom bg = make_border(g.H, g.W, 4 * g.S + g.ov);
  uint64_t v = 0;
  for (int k = 0; k < 8; k++) v |= (uint64_t)luma_b(r, bg, y, x + k) << (8 * k);
  return v;
}
__device__ __forceinline__ uint64_t luma_run8(const uint8_t* r, const LevelGeom& g, int y, int x) {
  return luma_run8_fin(luma_run8_ld(r, g, y, x), r, g, y, x);
}
// 8 samples (y, x .. x+7) of the x2-interpolated chroma plane, computed from the stored half-resolution plane u: the two stored
// rows an output row is the mean of (odd rows; the last row repeats) -- same two-step form
struct ChromaLd { uint64_t a, b; };
__device__ __forceinline__ bool chroma_run8_in(const LevelGeom& g, int y, int x) { return (unsigned)y < (unsigned)g.H && x >= 0 && x + 8 < g.W; }
__device__ __forceinline__ ChromaLd chroma_run8_ld(const uint8_t* u, const LevelGeom& g, int y, int x) {
  const int cw = g.W >> 1, chh = g.H >> 1;
  const int yc = iclamp(y, 0, g.H - 1), k0 = iclamp(x, 0, g.W - 10) >> 1;
  const int ja = yc >> 1, jb = imin((yc + 1) >> 1, chh - 1);
  ChromaLd l;
  l.a = ld8(u + (size_t)ja * cw + k0); l.b = ld8(u + (size_t)jb * cw + k0);
  return l;
}
__device__ __forceinline__ uint64_t chroma_run8_fin(ChromaLd l, const uint8_t* u, const LevelGeom& g, int y, int x) {
  if (chroma_run8_in(g, y, x)) {
    const uint64_t c = avg8(l.a, l.b);                                     // columns first (decorrelate.cpp:610-648)
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
__device__ __forceinline__ uint64_t chroma_run8(const uint8_t* u, const LevelGeom& g, int y, int x) {
  return chroma_run8_fin(chroma_run8_ld(u, g, y, x), u, g, y, x);
}
// clamp(o - p, -128, 127) + 128 of four samples
__device__ __forceinline__ uint32_t residue4(uint32_t o, uint32_t p) {
  const uint32_t M = 0x00FF00FFu, C = 0x00800080u;
  const uint32_t te = pk_clamp255(pk_sub((o & M) + C, p & M));
  const uint32_t to = pk_clamp255(pk_sub(((o >> 8) & M) + C, (p >> 8) & M));
  return te | (to << 8);
}


// ---------------------------------------------------------------------------------------------
// chroma tile: a workgroup of PCT_T threads produces 64 x 16 half-resolution samples out of their full-resolution footprint
#define PCT_W 64                      // outputs per tile row
#ifndef PCT_H
#define PCT_H 16                      // output rows per tile (14 until round 4: 1080p chroma then needed 585 tiles per plane, 9.1 per CU at the deepest
                                      // level where 8 fit; 510 tiles are one round: predict -4 %, at 4K predict -7 %, update -5 %).  At most 16: a thread per 4 outputs
#endif
static_assert(PCT_H * 16 <= 256 && (2 * PCT_H) % 4 == 0, "a thread per 4 outputs of the tile; the footprint starts two rows above a multiple of 4");
#define PCT_FH (2 * PCT_H + 3)        // full-resolution rows 2 j0 - 2 .. 2 j0 + 2 PCT_H
#define PCT_FW (2 * PCT_W + 16)       // full-resolution columns 2 i0 - 8 .. 2 i0 + 2 PCT_W + 7, in runs of 8
#define PCT_SP_BYTES (((PCT_FH * PCT_FW) + 15) & ~15)          // the footprint (bytes) ...
#define PCT_SMEM_BYTES (PCT_SP_BYTES + PCT_FH * (PCT_W / 2) * 4)  // ... and the row pass' low band behind it
#define PCT_T 256                     // threads per workgroup (576 = one footprint run per thread was measured: slower, fewer workgroups per CU)
// sP (bytes, rows yb .., columns xb ..) -> sL: the row pass' low band, two adjacent columns per word.  Call with the footprint
// written and a barrier passed; returns with sL complete (barrier included).
__device__ __forceinline__ void chroma_tile_lows(uint8_t (*sP)[PCT_FW], uint32_t (*sL)[PCT_W / 2], const LevelGeom& g, int i0, int xb, int yb,
                                                 int tid) {
  const int cw = g.W >> 1;
  // the picture's left / right ends: whole-sample symmetric extension (see the file header)
  if (xb < 0 && tid < PCT_FH) { sP[tid][6] = sP[tid][10]; sP[tid][7] = sP[tid][9]; }
  if (xb + PCT_FW > g.W && g.W - xb >= 0 && tid >= 64 && tid < 64 + PCT_FH) sP[tid - 64][g.W - xb] = sP[tid - 64][g.W - xb - 2];
  __syncthreads();
  // B. row pass: lows of 4 outputs per task out of 4 words of the row
  for (int k = tid; k < PCT_FH * (PCT_W / 4); k += PCT_T) {
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
}
// the column pass for 4 columns of output row jr of the tile: ll[0] = columns 4 gq, 4 gq + 1, ll[1] = the next two (16-bit pairs)
__device__ __forceinline__ void chroma_tile_ll(uint32_t (*sL)[PCT_W / 2], int jr, int gq, uint32_t ll[2]) {
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const uint32_t r0 = sL[2 * jr][2 * gq + w], r1 = sL[2 * jr + 1][2 * gq + w], r2 = sL[2 * jr + 2][2 * gq + w],
                   r3 = sL[2 * jr + 3][2 * gq + w], r4 = sL[2 * jr + 4][2 * gq + w];
    const uint32_t hA = pk_sub(r1, pk_tdiv2(pk_add(r0, r2))), hB = pk_sub(r3, pk_tdiv2(pk_add(r2, r4)));
    ll[w] = pk_add(r2, pk_tdiv4(pk_add(hA, hB)));
  }
}

}  // namespace qsvc
