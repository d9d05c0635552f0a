// qsvc_amd/csrc/j2k_dwt.hip -- forward DWT of the JPEG 2000 stage (ITU-T T.800 Annex F) as ONE kernel per decomposition
// level: a workgroup stages a haloed tile of the level's input in LDS, runs the vertical and then the horizontal lifting
// there, and the four subbands leave the chip exactly once:
//   * LL goes to the next level's input buffer (int16 for the reversible 5/3 path, float32 for 9/7), compact, its own stride;
//   * HL / LH / HH (and LL of the last level) are quantised (Annex E) and written straight into the Tier-1 layout
//     (`mags`: per code-block, per 4-row stripe column one u64 of 4 x (sign<<15 | 15-bit magnitude)), with the block's
//     largest magnitude collected on the way -- there is no coefficient plane in HBM and no separate pack pass.
// It replaces the part of the reference's external `kdu_compress` call (trunk/src/texture_compress_fb_j2k.py:183-195) that
// transforms and quantises a component.  Bit-exactness: every lifting step is the restatement's expression (oracle/j2k_oracle.c)
// with the same boundary rule -- a neighbour outside the signal is replaced by the neighbour on the other side, which is what
// the restatement's index clamps amount to for 5/3 and 9/7 alike -- products and sums rounded separately (no FMA contraction).
//
// Tile: 64 x 16 samples of each subband = 128 x 32 input samples (+ halo 2 for 5/3, 4 for 9/7), staged as 32-bit words with an
// odd row stride: the vertical steps run with lanes along a row, the horizontal steps with lanes along a column, both free of
// LDS bank conflicts.  A lifting step only writes samples of one parity and only reads samples of the other, so each step is
// a plain parallel map over the tile with one barrier after it.  No MFMA: there is no contraction here.
#include <hip/hip_runtime.h>
#include <vector>
#include <type_traits>
#include <stdlib.h>
#include "j2k_kernels.h"

namespace qsvc {

#define DT_TW 64                    // subband columns per tile (= the widest code-block)
#ifndef DT_TH
#define DT_TH 16                    // subband rows per tile (16: 19.7 KB of LDS, 8 workgroups per CU; measured faster than 32)
#endif
#ifndef DT_UNROLL
#define DT_UNROLL 4
#endif
#define DT_NC (2 * DT_TW + 8)       // staged input columns [2*tx0 - 4, 2*tx0 + 132): a multiple of 4 for wide loads
#define DT_NCP (DT_NC + 1)          // odd row stride (words)
#define DT_CL 4                     // columns left of the tile's first output column

#define F97_ALPHA (-1.586134342059924f)
#define F97_BETA (-0.052980118572961f)
#define F97_GAMMA (0.882911075530934f)
#define F97_DELTA (0.443506852043971f)
#define F97_K (1.230174104914001f)

template <bool REV> struct DwtTraits { typedef int T; typedef int16_t LL; static const int HALO = 2; };
template <> struct DwtTraits<false> { typedef float T; typedef float LL; static const int HALO = 4; };

// one lifting step on the element `x` with its two neighbours (already boundary-resolved)
template <int STEP> __device__ __forceinline__ int lift53(int x, int lo, int hi) {
  return STEP == 0 ? x - ((lo + hi) >> 1) : x + ((lo + hi + 2) >> 2);   // T.800 F.4.8.1 (floor by arithmetic shift)
}
template <int STEP> __device__ __forceinline__ float lift97(float x, float lo, float hi) {
  const float c = STEP == 0 ? F97_ALPHA : (STEP == 1 ? F97_BETA : (STEP == 2 ? F97_GAMMA : F97_DELTA));
  return __fadd_rn(x, __fmul_rn(c, __fadd_rn(lo, hi)));                 // T.800 F.4.8.2, rounded separately
}

// Vertical step STEP over the staged tile: updates rows of parity (STEP even -> odd rows), lanes along a row.
// INTERIOR (tile-uniform): every staged row has both neighbours inside the signal, so the boundary rule never applies and the
// rows above / below the staged range are padding rows of the LDS array (what they feed is halo that nobody reads).
template <bool REV, int STEP, int NR, bool INTERIOR>
__device__ __forceinline__ void sweep_v(typename DwtTraits<REV>::T* T, int tid, int g0, int n) {
  typedef typename DwtTraits<REV>::T TT;
  const int par = (STEP & 1) ^ 1;
  auto one = [&](int lr, int c) {
    const int l = 2 * lr + par, g = g0 + l;
    TT* p = T + l * DT_NCP + c;
    TT lo, hi;
    if (INTERIOR) { lo = p[-DT_NCP]; hi = p[DT_NCP]; }
    else {
      if (g < 0 || g >= n) return;
      const TT a = l > 0 ? p[-DT_NCP] : p[0], b = l + 1 < NR ? p[DT_NCP] : p[0];
      lo = g > 0 ? a : b; hi = g + 1 < n ? b : a;
    }
    if (REV) p[0] = (TT)lift53<STEP>((int)p[0], (int)lo, (int)hi); else p[0] = (TT)lift97<STEP>((float)p[0], (float)lo, (float)hi);
  };
  // columns 0..127: two rows of the parity per pass of the 256 threads (no division); the last DT_NC - 128 columns afterwards
  static_assert(DT_NC >= 128 && DT_NC - 128 == 8, "column split of sweep_v");
  const int c = tid & 127;
#pragma unroll DT_UNROLL
  for (int lr = tid >> 7; lr < NR / 2; lr += 2) one(lr, c);
  for (int e = tid; e < (NR / 2) * 8; e += 256) one(e >> 3, 128 + (e & 7));
}
// Horizontal step over the output rows [r0, r0 + nrows) of the tile: lanes along a column (row stride is odd).
template <bool REV, int STEP, bool INTERIOR>
__device__ __forceinline__ void sweep_h(typename DwtTraits<REV>::T* T, int tid, int r0, int nrows, int g0, int n) {
  typedef typename DwtTraits<REV>::T TT;
  const int par = (STEP & 1) ^ 1;
#pragma unroll DT_UNROLL
  for (int idx = tid; idx < nrows * (DT_NC / 2); idx += 256) {
    const int r = r0 + idx % nrows, l = 2 * (idx / nrows) + par, g = g0 + l;
    TT* p = T + r * DT_NCP + l;
    TT lo, hi;
    if (INTERIOR) { lo = p[-1]; hi = p[1]; }   // column -1 / DT_NC of a row is the stride's padding word
    else {
      if (g < 0 || g >= n) continue;
      const TT a = l > 0 ? p[-1] : p[0], b = l + 1 < DT_NC ? p[1] : p[0];
      lo = g > 0 ? a : b; hi = g + 1 < n ? b : a;
    }
    if (REV) p[0] = (TT)lift53<STEP>((int)p[0], (int)lo, (int)hi); else p[0] = (TT)lift97<STEP>((float)p[0], (float)lo, (float)hi);
  }
}

// OR over each aligned group of `span` lanes (span a power of two, 4 .. 64) with DPP row shifts / broadcasts: no LDS
// round trips.  The LAST lane of every group ends up with the group's OR (other lanes hold partial values).
__device__ __forceinline__ int wave_or_segments(int v, int span) {
  v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);               // row_shr:1
  v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);               // row_shr:2
  if (span >= 8) v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
  if (span >= 16) v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
  if (span >= 32) v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  if (span >= 64) v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

struct DwtArgs {
  const PlaneDesc* planes;
  const DwtSb* sbtab;        // [nplanes][1 + 3 L]: LL_L, then HL, LH, HH for d = L .. 1 (code-stream order)
  int nsb;                   // 1 + 3 L
  int lev, L;                // this launch: decomposition level lev + 1 of L
  const void* llin;          // LL_lev (lev >= 1): compact planes at PlaneDesc::ll[(lev - 1) & 1]
  void* llout;               // LL_{lev+1}
  unsigned long long* mags;  // Tier-1 magnitudes, 1024 u64 per code-block
  int* maxmag;               // per code-block OR of the magnitudes (atomicOr; zeroed by the caller): same bit length as the maximum
  void* plane_out;           // OUT_PLANE: the Mallat-layout plane of the unit-test entry point (int32 / float32)
  int cbwl, cbhl;            // log2 of the code-block size
  int* err;
  const uint32_t* tilemap;   // this level's workgroups: plane << 20 | tile row << 10 | tile column (nullptr: the 3-D grid, planes of one size)
};
// (a batch's planes differ in size -- luma / chroma -- and an empty workgroup still costs its dispatch: with a common 3-D grid half of
// config 3's level-1 workgroups were empty, about 10 of the launch's 62 us)
__device__ __forceinline__ void dwt_tile_of(const DwtArgs& a, int& plane, int& tx, int& ty) {
  if (a.tilemap) { const uint32_t t = a.tilemap[blockIdx.x]; plane = (int)(t >> 20); ty = (int)((t >> 10) & 1023u); tx = (int)(t & 1023u); }
  else { plane = blockIdx.z; ty = blockIdx.y; tx = blockIdx.x; }
}

// LL stride of a plane's level-l low band (elements): a multiple of 4 so that rows stay 8 / 16-byte aligned
__host__ __device__ __forceinline__ int dwt_ll_stride(int w) { return (w + 3) & ~3; }

template <bool REV, bool FROM_U8, bool OUT_PLANE>
__global__ void __launch_bounds__(256) k_dwt_tile(DwtArgs a) {
  typedef typename DwtTraits<REV>::T TT;
  typedef typename DwtTraits<REV>::LL LLT;
  constexpr int HALO = DwtTraits<REV>::HALO;
  constexpr int NR = 2 * DT_TH + 2 * HALO;
  __shared__ TT sTpad[(NR + 2) * DT_NCP];   // one padding row above and below the staged tile (interior sweeps read them)
  TT* const sT = sTpad + DT_NCP;
  const int tid = threadIdx.x;
  int plane_i, tile_x, tile_y;
  dwt_tile_of(a, plane_i, tile_x, tile_y);
  const PlaneDesc pd = a.planes[plane_i];
  int w = pd.W, h = pd.H;
  for (int l = 0; l < a.lev; l++) { w = (w + 1) >> 1; h = (h + 1) >> 1; }
  const int lw = (w + 1) >> 1, lh = (h + 1) >> 1, hw = w >> 1, hh = h >> 1;
  const int tx0 = tile_x * DT_TW, ty0 = tile_y * DT_TH;
  if (tx0 >= lw || ty0 >= lh) return;                      // (3-D grid: planes of one batch may differ in size)
  const int gc0 = 2 * tx0 - DT_CL, gr0 = 2 * ty0 - HALO;   // global coordinates of the staged tile's origin

  // ---- stage the tile: rows [gr0, gr0 + NR) x columns [gc0, gc0 + DT_NC), zero outside the signal -----------------------
  {
    // (pd.src comes out of a descriptor in memory: without the address space the compiler emits FLAT loads, which also count in lgkmcnt)
    typedef const uint8_t __attribute__((address_space(1))) * gu8;
    typedef const uint32_t __attribute__((address_space(1))) * gu32;
    const gu8 u = (gu8)(uintptr_t)pd.src;
    const LLT* lin = (const LLT*)a.llin + (((a.lev - 1) & 1) ? pd.ll[1] : pd.ll[0]);  // (a dynamic index would send pd to scratch)
    const int sstr = FROM_U8 ? pd.u8stride : dwt_ll_stride(w);
    const bool wide = FROM_U8 ? ((((uintptr_t)u) & 3) == 0 && (sstr & 3) == 0) : true;
    // all of a thread's wide loads are issued before the first one is unpacked (memory-level parallelism: ~10 per thread)
    constexpr int NQ = NR * (DT_NC / 4), NIT = (NQ + 255) / 256;
    typedef typename std::conditional<FROM_U8, uint32_t, typename std::conditional<REV, uint2, float4>::type>::type Raw;
    Raw raw[NIT];
    unsigned fast = 0;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + it * 256, lr = idx / (DT_NC / 4), q = idx % (DT_NC / 4), gr = gr0 + lr, gc = gc0 + 4 * q;
      if (idx < NQ && gr >= 0 && gr < h && gc >= 0 && gc + 3 < w && wide) {
        fast |= 1u << it;
        if constexpr (FROM_U8) raw[it] = *(gu32)(u + (size_t)gr * sstr + gc);
        else raw[it] = *(const Raw*)(lin + (size_t)gr * sstr + gc);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + it * 256, lr = idx / (DT_NC / 4), q = idx % (DT_NC / 4), gr = gr0 + lr, gc = gc0 + 4 * q;
      if (idx >= NQ) continue;  // (no `break`: the loop must unroll completely so that raw[] stays in registers)
      TT v[4] = {0, 0, 0, 0};
      if (fast & (1u << it)) {
        if constexpr (FROM_U8) {
          const uint32_t d = raw[it];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = (TT)((int)((d >> (8 * k)) & 255) - 128);
        } else if constexpr (REV) {
          const uint2 d = raw[it];
          v[0] = (TT)(int16_t)(d.x & 0xFFFF); v[1] = (TT)(int16_t)(d.x >> 16);
          v[2] = (TT)(int16_t)(d.y & 0xFFFF); v[3] = (TT)(int16_t)(d.y >> 16);
        } else {
          const float4 d = raw[it];
          v[0] = (TT)d.x; v[1] = (TT)d.y; v[2] = (TT)d.z; v[3] = (TT)d.w;
        }
      } else if (gr >= 0 && gr < h && gc + 3 >= 0 && gc < w) {  // picture edge or an unaligned source: sample by sample
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (gc + k >= 0 && gc + k < w)
            v[k] = FROM_U8 ? (TT)((int)u[(size_t)gr * sstr + gc + k] - 128) : (TT)lin[(size_t)gr * sstr + gc + k];
      }
      TT* p = sT + lr * DT_NCP + 4 * q;
#pragma unroll
      for (int k = 0; k < 4; k++) p[k] = v[k];
    }
  }
  __syncthreads();
  // ---- vertical lifting, then horizontal lifting on the rows that are output ----------------------------------------------
  const bool inner_v = gr0 >= 1 && gr0 + NR <= h - 1, inner_h = gc0 >= 1 && gc0 + DT_NC <= w - 1;   // workgroup-uniform
  if (h >= 2) {
    if (inner_v) {
      sweep_v<REV, 0, NR, true>(sT, tid, gr0, h); __syncthreads();
      sweep_v<REV, 1, NR, true>(sT, tid, gr0, h); __syncthreads();
      if (!REV) {
        sweep_v<REV, 2, NR, true>(sT, tid, gr0, h); __syncthreads();
        sweep_v<REV, 3, NR, true>(sT, tid, gr0, h); __syncthreads();
      }
    } else {
      sweep_v<REV, 0, NR, false>(sT, tid, gr0, h); __syncthreads();
      sweep_v<REV, 1, NR, false>(sT, tid, gr0, h); __syncthreads();
      if (!REV) {
        sweep_v<REV, 2, NR, false>(sT, tid, gr0, h); __syncthreads();
        sweep_v<REV, 3, NR, false>(sT, tid, gr0, h); __syncthreads();
      }
    }
    if (!REV) {
      const float invK = 1.0f / F97_K;
      for (int idx = tid; idx < NR * DT_NC; idx += 256) {  // low (even rows) * 1/K, high (odd rows) * K
        const int l = idx / DT_NC, c = idx % DT_NC;
        float* p = (float*)sT + l * DT_NCP + c;
        p[0] = __fmul_rn(p[0], ((gr0 + l) & 1) ? F97_K : invK);
      }
      __syncthreads();
    }
  }
  if (w >= 2) {
    if (inner_h) {
      sweep_h<REV, 0, true>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
      sweep_h<REV, 1, true>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
      if (!REV) {
        sweep_h<REV, 2, true>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
        sweep_h<REV, 3, true>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
      }
    } else {
      sweep_h<REV, 0, false>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
      sweep_h<REV, 1, false>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
      if (!REV) {
        sweep_h<REV, 2, false>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
        sweep_h<REV, 3, false>(sT, tid, HALO, 2 * DT_TH, gc0, w); __syncthreads();
      }
    }
    if (!REV) {
      const float invK = 1.0f / F97_K;
      for (int idx = tid; idx < 2 * DT_TH * DT_NC; idx += 256) {
        const int r = HALO + idx % (2 * DT_TH), l = idx / (2 * DT_TH);
        float* p = (float*)sT + r * DT_NCP + l;
        p[0] = __fmul_rn(p[0], ((gc0 + l) & 1) ? F97_K : invK);
      }
      __syncthreads();
    }
  }
  // Sample (ry, rx) parity classes of the tile are now the subbands: (0,0) LL, (0,1) HL, (1,0) LH, (1,1) HH.
  // A signal of one sample passes through as a low sample (the restatement's `n >= 2 ? lift : x`).
  auto coef = [&](int ky, int kx, int ry, int rx) { return sT[(HALO + 2 * ky + ry) * DT_NCP + DT_CL + 2 * kx + rx]; };
  const bool last = a.lev == a.L - 1;

  if (OUT_PLANE) {  // unit-test form: the Mallat layout of the whole plane, 32-bit
    TT* out = (TT*)a.plane_out + pd.off;
    for (int idx = tid; idx < 4 * DT_TH * DT_TW; idx += 256) {
      const int sub = idx / (DT_TH * DT_TW), ky = (idx / DT_TW) % DT_TH, kx = idx % DT_TW, ry = sub >> 1, rx = sub & 1;
      const int y = ty0 + ky, x = tx0 + kx;
      if (y >= (ry ? hh : lh) || x >= (rx ? hw : lw)) continue;
      if (sub == 0 && !last) continue;
      out[(size_t)(y + (ry ? lh : 0)) * pd.W + x + (rx ? lw : 0)] = coef(ky, kx, ry, rx);
    }
  }
  // ---- LL -> next level's input (compact, 4 samples per thread-store) ----------------------------------------------------
  if (!last) {
    LLT* lout = (LLT*)a.llout + ((a.lev & 1) ? pd.ll[1] : pd.ll[0]);
    const int ostr = dwt_ll_stride(lw);
    for (int idx = tid; idx < DT_TH * (DT_TW / 4); idx += 256) {
      const int ky = idx / (DT_TW / 4), kx = 4 * (idx % (DT_TW / 4)), y = ty0 + ky, x = tx0 + kx;
      if (y >= lh || x >= ostr) continue;
      TT v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = coef(ky, kx + k, 0, 0);   // columns in [lw, ostr) are padding, never read back
      if (REV) {
        bool big = false;
#pragma unroll
        for (int k = 0; k < 4; k++) big |= (x + k < lw) && ((int)v[k] > 32767 || (int)v[k] < -32768);
        if (big) atomicOr(a.err, 1);  // cannot happen for 8-bit input through <= 10 levels at the guard bits in use
        uint2 d;
        d.x = ((uint32_t)(int)v[0] & 0xFFFF) | ((uint32_t)(int)v[1] << 16);
        d.y = ((uint32_t)(int)v[2] & 0xFFFF) | ((uint32_t)(int)v[3] << 16);
        *(uint2*)(lout + (size_t)y * ostr + x) = d;
      } else {
        *(float4*)(lout + (size_t)y * ostr + x) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
      }
    }
  }
  if (OUT_PLANE) return;
  // ---- HL, LH, HH (and the last LL): quantise, pack 4 rows of a stripe column, store into the block's Tier-1 layout ---------
  const DwtSb* sbp = a.sbtab + (size_t)plane_i * a.nsb;
  const int lane = tid & 63, wv = tid >> 6;
  const int span = a.cbwl >= 6 ? 64 : (1 << a.cbwl);            // lanes of one code-block
  const bool tall = (1 << a.cbhl) >= DT_TH;                      // the tile's stripes of a subband lie in one block row
  for (int sub = last ? 0 : 1; sub < 4; sub++) {                 // wave-uniform (subband, stripe); lane = column
    const int ry = sub >> 1, rx = sub & 1;
    const int sbh = ry ? hh : lh, sbw = rx ? hw : lw;
    const int x = tx0 + lane;
    // code-stream order: LL_L first, then for d = L .. 1: HL, LH, HH; this level is d = lev + 1
    const DwtSb sb = sbp[sub == 0 ? 0 : 1 + 3 * (a.L - 1 - a.lev) + (sub == 1 ? 0 : (sub == 2 ? 1 : 2))];
    int acc = 0, accy = ty0;
    auto flush = [&]() {  // OR over the lanes of each code-block (DPP, no LDS), one atomic per block
      int v = wave_or_segments(acc, span);
      // the group's last lane speaks for the block (it may itself lie beyond the subband's last column: its acc was 0)
      if ((lane & (span - 1)) == span - 1 && v)
        atomicOr(&a.maxmag[sb.first_cb + (accy >> a.cbhl) * sb.gx + (x >> a.cbwl)], v);
      acc = 0;
    };
    for (int s = wv; s < DT_TH / 4; s += 4) {
      const int y = ty0 + 4 * s;
      if (y >= sbh) break;                                       // uniform
      uint64_t word = 0;
      if (x < sbw) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (y + k >= sbh) break;
          const TT v = coef(4 * s + k, lane, ry, rx);
          int mag, neg;
          if (REV) { const int iv = (int)v; neg = iv < 0; mag = neg ? -iv : iv; }
          else { const float fv = (float)v; mag = (int)floorf(__fdiv_rn(fabsf(fv), sb.delta)); neg = fv < 0.0f && mag; }  // T.800 E.1
          acc |= mag;
          word |= (uint64_t)((mag & 0x7FFF) | (neg << 15)) << (16 * k);
        }
        const int bx = x >> a.cbwl, by = y >> a.cbhl;
        const size_t b = (size_t)sb.first_cb + (size_t)by * sb.gx + bx;
        a.mags[b * 1024 + (size_t)(((y & ((1 << a.cbhl) - 1)) >> 2) * 64 + (x & ((1 << a.cbwl) - 1)))] = word;
      }
      accy = y;
      if (!tall) flush();
    }
    if (tall) flush();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Reversible 5/3, packed form (the Tier-1 path of lossless coding; k_dwt_tile above stays for 9/7, for the unit-test plane
// output and for levels smaller than 4 samples).  Same tile, same outputs, but the samples live in LDS as PAIRS of 16-bit
// values (two adjacent columns per word: half the LDS, and every lifting operation works on two samples -- v_pk_add_i16,
// v_pk_ashrrev_i16), the tile is staged with the signal's ends already mirrored (whole-sample symmetric extension is exactly
// the rule "a neighbour outside the signal is replaced by the one on the other side", so the lifting itself has no boundary
// case), each of the two directions is ONE pass with the intermediate highs in registers instead of two read-modify-write
// sweeps, and the four rows of a stripe column are produced by the same lane, so the Tier-1 words (4 x 16 bit per column)
// leave without any exchange between lanes.
//   vertical:   lane = (column pair, 8-row segment): 11 words in, highs and lows of 8 rows out (out of place)
//   horizontal: lane = (2 subband columns, 4 rows of one parity = one stripe): even / odd columns de-interleaved with v_perm
typedef short dp_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short dp_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dp_s2 dpS(uint32_t v) { return __builtin_bit_cast(dp_s2, v); }
__device__ __forceinline__ uint32_t dpU(dp_s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t dp_add(uint32_t a, uint32_t b) { return dpU(dpS(a) + dpS(b)); }
__device__ __forceinline__ uint32_t dp_sub(uint32_t a, uint32_t b) { return dpU(dpS(a) - dpS(b)); }
__device__ __forceinline__ uint32_t dp_sar(uint32_t a, int k) { return dpU(dpS(a) >> (short)k); }
// T.800 F.4.8.1 on both lanes: high = x - floor((l + r) / 2); low = x + floor((hl + hr + 2) / 4)
__device__ __forceinline__ uint32_t dp_high(uint32_t x, uint32_t l, uint32_t r) { return dp_sub(x, dp_sar(dp_add(l, r), 1)); }
__device__ __forceinline__ uint32_t dp_low(uint32_t x, uint32_t hl, uint32_t hr) { return dp_add(x, dp_sar(dp_add(dp_add(hl, hr), 0x00020002u), 2)); }
// sign << 15 | magnitude of both lanes
__device__ __forceinline__ uint32_t dp_signmag(uint32_t v) {
  const uint32_t neg = dp_sar(v, 15);
  return dp_sub(v ^ neg, neg) | (neg & 0x80008000u);
}
__device__ __forceinline__ int dwt_reflect(int i, int n) {  // whole-sample symmetric extension of index i into [0, n), n >= 2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);  // (farther out than the lifting reads: any valid sample)
}
#define DP_NP (DT_NC / 2)           // column pairs per staged row
#define DP_STR (DP_NP + 1)          // odd row stride (words)
// The two subband records a wave of the packed / register-blocked kernels needs -- low band (LL / LH) and high band (HL / HH) of its
// row parity -- fetched at kernel entry with a wave-uniform index: scalar loads, which do not count in vmcnt.  Loaded where they are
// used (after the LL stores, between the bands' stores) they were vector loads whose `s_waitcnt vmcnt(0)` also waited for every
// store issued before them: three store round trips in a row at the end of every wave.
struct DwtSb2 { DwtSb lo, hi; };
__device__ __forceinline__ DwtSb2 dwt_sb_pair(const DwtArgs& a, int plane_i, int tid) {
  const int ry = __builtin_amdgcn_readfirstlane(tid >> 7);
  const DwtSb* sbp = a.sbtab + (size_t)plane_i * a.nsb;
  const int sbi = 1 + 3 * (a.L - 1 - a.lev);   // code-stream order: LL_L, then HL, LH, HH for d = L .. 1; this level is d = lev + 1
  DwtSb2 r;
  r.lo = sbp[ry ? sbi + 1 : 0]; r.hi = sbp[ry ? sbi + 2 : sbi];
  return r;
}
// Output of the packed 5/3 kernels: the lane holds, for the 4 rows of stripe sidx of row parity ry, the low (Lw) and high (Hw) band's
// columns 2 cq, 2 cq + 1 of the tile as 16-bit pairs.
__device__ __forceinline__ void dwt53_pk_output(const DwtArgs& a, const PlaneDesc& pd, int plane_i, int tid, bool last, int ry, int sidx, int cq, int tx0, int ty0,
                                                int lw, int lh, int hw, int hh, const uint32_t (&Lw)[4], const uint32_t (&Hw)[4], const DwtSb2& sb2) {
  const int kx = 2 * cq;
  const int ky0 = 4 * sidx, y = ty0 + ky0, x = tx0 + kx;
  // LL -> next level's input (int16 pairs)
  if (!last && ry == 0) {
    int16_t* lout = (int16_t*)a.llout + ((a.lev & 1) ? pd.ll[1] : pd.ll[0]);
    const int ostr = dwt_ll_stride(lw);
    if (x < ostr)
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (y + r < lh) *(uint32_t*)(lout + (size_t)(y + r) * ostr + x) = Lw[r];   // columns in [lw, ostr) are padding, never read back
  }
  // HL / LH / HH (and LL of the last level): sign-magnitude, 4 rows of a stripe column per u64, into the block's Tier-1 words
  const int lane = tid & 63;
  const int span2 = a.cbwl >= 6 ? 32 : (1 << (a.cbwl - 1));   // lanes (of 2 columns) per code-block
#pragma unroll
  for (int band = 0; band < 2; band++) {       // 0: the low band of this row parity (LL / LH), 1: the high band (HL / HH)
    const int sub = 2 * ry + band;
    if (sub == 0 && !last) continue;
    const int sbh = ry ? hh : lh, sbw = band ? hw : lw;
    const DwtSb sb = band ? sb2.hi : sb2.lo;
    int acc = 0;
    if (y < sbh && x < sbw) {
      uint32_t q[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        q[r] = y + r < sbh ? dp_signmag(band ? Hw[r] : Lw[r]) : 0u;
        acc |= (int)(q[r] & 0x7FFF7FFFu);
      }
      if (x + 1 >= sbw) acc &= 0xFFFF;         // the second column lies beyond the subband
      // column x: low halves of the four rows; column x + 1: high halves
      const uint32_t c0lo = __builtin_amdgcn_perm(q[1], q[0], 0x05040100u), c0hi = __builtin_amdgcn_perm(q[3], q[2], 0x05040100u);
      const uint32_t c1lo = __builtin_amdgcn_perm(q[1], q[0], 0x07060302u), c1hi = __builtin_amdgcn_perm(q[3], q[2], 0x07060302u);
      const int bx = x >> a.cbwl, by = y >> a.cbhl;
      const size_t b = (size_t)sb.first_cb + (size_t)by * sb.gx + bx;
      unsigned long long* m = a.mags + b * 1024 + (size_t)(((y & ((1 << a.cbhl) - 1)) >> 2) * 64 + (x & ((1 << a.cbwl) - 1)));
      // (one 16-byte store in every case: x is even, so the word of column x + 1 lies in the same block's row even where the subband ends
      // at x -- a slot nobody reads, written as zero; two forms of the store cost three store instructions)
      const bool two = x + 1 < sbw;
      *(uint4*)m = make_uint4(c0lo, c0hi, two ? c1lo : 0u, two ? c1hi : 0u);
    }
    acc = (acc | (acc >> 16)) & 0x7FFF;
    // OR over the lanes of each code-block (DPP row shifts), one atomic per (block, stripe): the group's last lane speaks
    const int v = wave_or_segments(acc, span2);
    if ((lane & (span2 - 1)) == span2 - 1 && v && y < sbh) {
      const int xl = tx0 + 2 * (cq | (span2 - 1)) ;   // a column of this lane group's block (may lie beyond the subband: same block index)
      atomicOr(&a.maxmag[sb.first_cb + (y >> a.cbhl) * sb.gx + ((xl < sbw ? xl : sbw - 1) >> a.cbwl)], v);
    }
  }
}

template <bool FROM_U8>
__global__ void __launch_bounds__(256) k_dwt53_pk(DwtArgs a) {
  constexpr int HALO = 2, NR = 2 * DT_TH + 2 * HALO;   // 36 rows: [2 ty0 - 2, 2 ty0 + 34)
  static_assert(DT_TH == 16 && DT_TW == 64 && DT_CL == 4, "task maps of k_dwt53_pk");
  __shared__ uint32_t sX[NR * DP_STR];   // staged input pairs
  __shared__ uint32_t sV[NR * DP_STR];   // after the vertical lifting
  const int tid = threadIdx.x;
  int plane_i, tile_x, tile_y;
  dwt_tile_of(a, plane_i, tile_x, tile_y);
  const PlaneDesc pd = a.planes[plane_i];
  int w = pd.W, h = pd.H;
  for (int l = 0; l < a.lev; l++) { w = (w + 1) >> 1; h = (h + 1) >> 1; }
  const int lw = (w + 1) >> 1, lh = (h + 1) >> 1, hw = w >> 1, hh = h >> 1;
  const int tx0 = tile_x * DT_TW, ty0 = tile_y * DT_TH;
  if (tx0 >= lw || ty0 >= lh) return;
  const DwtSb2 sb2 = dwt_sb_pair(a, plane_i, tid);
  const int gc0 = 2 * tx0 - DT_CL, gr0 = 2 * ty0 - HALO;
  // ---- stage: 4 samples per thread-load; the signal's ends mirrored ----------------------------------------------------------
  {
    typedef const uint8_t __attribute__((address_space(1))) * gu8;
    typedef const uint32_t __attribute__((address_space(1))) * gu32;
    const gu8 u = (gu8)(uintptr_t)pd.src;
    const int16_t* lin = (const int16_t*)a.llin + (((a.lev - 1) & 1) ? pd.ll[1] : pd.ll[0]);
    const int sstr = FROM_U8 ? pd.u8stride : dwt_ll_stride(w);
    const bool wide = FROM_U8 ? ((((uintptr_t)u) & 3) == 0 && (sstr & 3) == 0) : true;
    constexpr int NQ = NR * (DT_NC / 4), NIT = (NQ + 255) / 256;
    typedef typename std::conditional<FROM_U8, uint32_t, uint2>::type Raw;
    // A tile whose whole footprint lies inside the signal (9 of 10 at 1080p) is staged without an index division, a range test or
    // 64-bit address arithmetic per element: thread = (row mod 8, quad of 4 columns), five rows 8 apart, and the footprint's last two
    // quads by 72 threads.  (The kernel is issue-bound: staging was 40 % of its instructions.)
    static_assert(NR == 36 && DT_NC == 136, "row / quad map of the interior staging");
    if (gr0 >= 0 && gr0 + NR <= h && gc0 >= 0 && gc0 + DT_NC <= w && wide) {   // (uniform)
      const int rr = tid >> 5, q = tid & 31;
      auto put = [&](int lr, int qq, Raw d) {
        uint32_t p0, p1;
        if constexpr (FROM_U8) {
          p0 = dp_sub(__builtin_amdgcn_perm(0u, d, 0x0c010c00u), 0x00800080u);
          p1 = dp_sub(__builtin_amdgcn_perm(0u, d, 0x0c030c02u), 0x00800080u);
        } else { p0 = d.x; p1 = d.y; }
        uint32_t* p = sX + lr * DP_STR + 2 * qq;
        p[0] = p0; p[1] = p1;
      };
      Raw rw[5], rx;
      if constexpr (FROM_U8) {
        const gu8 b = u + (size_t)(gr0 + rr) * sstr + (gc0 + 4 * q);
#pragma unroll
        for (int it = 0; it < 5; it++) if (it < 4 || rr < 4) rw[it] = *(gu32)(b + (size_t)it * 8 * sstr);
        if (tid < 2 * NR) rx = *(gu32)(u + (size_t)(gr0 + (tid >> 1)) * sstr + (gc0 + 128 + 4 * (tid & 1)));
      } else {
        const int16_t* b = lin + (size_t)(gr0 + rr) * sstr + (gc0 + 4 * q);
#pragma unroll
        for (int it = 0; it < 5; it++) if (it < 4 || rr < 4) rw[it] = *(const uint2*)(b + (size_t)it * 8 * sstr);
        if (tid < 2 * NR) rx = *(const uint2*)(lin + (size_t)(gr0 + (tid >> 1)) * sstr + (gc0 + 128 + 4 * (tid & 1)));
      }
#pragma unroll
      for (int it = 0; it < 5; it++) if (it < 4 || rr < 4) put(rr + 8 * it, q, rw[it]);
      if (tid < 2 * NR) put(tid >> 1, 32 + (tid & 1), rx);
    } else {
    Raw raw[NIT];
    unsigned fast = 0;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + it * 256, lr = idx / (DT_NC / 4), q = idx % (DT_NC / 4), gr = gr0 + lr, gc = gc0 + 4 * q;
      if (idx < NQ && gr >= 0 && gr < h && gc >= 0 && gc + 3 < w && wide) {
        fast |= 1u << it;
        if constexpr (FROM_U8) raw[it] = *(gu32)(u + (size_t)gr * sstr + gc);
        else raw[it] = *(const uint2*)(lin + (size_t)gr * sstr + gc);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + it * 256, lr = idx / (DT_NC / 4), q = idx % (DT_NC / 4), gr = gr0 + lr, gc = gc0 + 4 * q;
      if (idx >= NQ) continue;
      uint32_t p0, p1;
      if (fast & (1u << it)) {
        if constexpr (FROM_U8) {
          const uint32_t d = raw[it];  // level shift: two samples per word, - 128 each
          p0 = dp_sub(__builtin_amdgcn_perm(0u, d, 0x0c010c00u), 0x00800080u);
          p1 = dp_sub(__builtin_amdgcn_perm(0u, d, 0x0c030c02u), 0x00800080u);
        } else { p0 = raw[it].x; p1 = raw[it].y; }
      } else {  // the signal's ends (or an unaligned source): sample by sample at the mirrored position
        const int rr = dwt_reflect(gr, h);
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int cc = dwt_reflect(gc + k, w);
          v[k] = FROM_U8 ? (int)u[(size_t)rr * sstr + cc] - 128 : (int)lin[(size_t)rr * sstr + cc];
        }
        p0 = ((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16);
        p1 = ((uint32_t)v[2] & 0xFFFFu) | ((uint32_t)v[3] << 16);
      }
      uint32_t* p = sX + lr * DP_STR + 2 * q;
      p[0] = p0; p[1] = p1;
    }
    }  // border tile
  }
  __syncthreads();
  // ---- vertical lifting: rows 1 .. 33 of the tile (row 0 is global row 2 ty0 - 2: even rows are lows, odd rows highs) --------
  for (int k = tid; k < DP_NP * 4; k += 256) {
    const int cp = k % DP_NP, q = k / DP_NP, r0 = 8 * q;
    const uint32_t* x = sX + r0 * DP_STR + cp;
    uint32_t xr[11];
#pragma unroll
    for (int r = 0; r < 11; r++) xr[r] = x[r * DP_STR];
    uint32_t hv[5];
#pragma unroll
    for (int j = 0; j < 5; j++) hv[j] = dp_high(xr[2 * j + 1], xr[2 * j], xr[2 * j + 2]);   // rows r0 + 1, 3, 5, 7, 9
    uint32_t* o = sV + r0 * DP_STR + cp;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      o[(2 * j + 1) * DP_STR] = hv[j];                                            // rows r0 + 1, 3, 5, 7
      o[(2 * j + 2) * DP_STR] = dp_low(xr[2 * j + 2], hv[j], hv[j + 1]);          // rows r0 + 2, 4, 6, 8
    }
    if (q == 3) o[9 * DP_STR] = hv[4];                                            // row 33
  }
  __syncthreads();
  // ---- horizontal lifting + output: lane = (parity ry, stripe s of the tile, 2 subband columns 2 cq, 2 cq + 1) -----------------
  const bool last = a.lev == a.L - 1;
  const int cq = tid & 31, sidx = (tid >> 5) & 3, ry = tid >> 7;
  const int kx = 2 * cq;                       // the lane's first subband column within the tile
  // columns: subband column kx <-> tile columns 4 + 2 kx (low) and 5 + 2 kx (high); even index e = 2 + kx counts tile columns 2 e
  uint32_t Lw[4], Hw[4];                       // per row: {col kx, col kx + 1} of the low / high band
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int t = HALO + 2 * (4 * sidx + r) + ry;
    const uint32_t* row = sV + t * DP_STR + (1 + kx);   // pairs holding tile columns 2 + 2 kx .. 9 + 2 kx
    const uint32_t d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3];
    const uint32_t E0 = __builtin_amdgcn_perm(d1, d0, 0x05040100u), E1 = __builtin_amdgcn_perm(d3, d2, 0x05040100u);  // evens e-1, e | e+1, e+2
    const uint32_t O0 = __builtin_amdgcn_perm(d1, d0, 0x07060302u), O1 = __builtin_amdgcn_perm(d3, d2, 0x07060302u);  // odds between them
    const uint32_t Es0 = __builtin_amdgcn_alignbit(E1, E0, 16);            // evens e, e + 1 (the lows' centres)
    const uint32_t Es1 = __builtin_amdgcn_alignbit(0u, E1, 16);            // even e + 2, -
    const uint32_t H0 = dp_high(O0, E0, Es0);                              // highs right of evens e - 1, e
    const uint32_t H1 = dp_high(O1, E1, Es1);                              // high right of even e + 1 (second lane unused)
    const uint32_t Hs = __builtin_amdgcn_alignbit(H1, H0, 16);             // highs right of evens e, e + 1 = subband columns kx, kx + 1
    Lw[r] = dp_low(Es0, H0, Hs);
    Hw[r] = Hs;
  }
  dwt53_pk_output(a, pd, plane_i, tid, last, ry, sidx, cq, tx0, ty0, lw, lh, hw, hh, Lw, Hw, sb2);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Reversible 5/3, packed AND register-blocked (round 4, after k_dwt97_rb below showed what the staging pass costs): the vertical
// lifting no longer reads a staged tile.  A lane owns one pair of columns (one packed word per row) and 8 output rows; it loads
// the 11 rows [8 seg, 8 seg + 11) of its pair straight from memory -- one 2-byte (pictures) / 4-byte (int16 low bands) load per
// row, all in flight, a row or pair beyond the signal's end at its mirrored position (the image of a pair is a pair in reverse
// order) -- lifts them with the packed 16-bit operations and leaves 8 rows in LDS.  272 tasks on 256 threads: the first 16
// threads also take the 16 tasks of the tile's last four column pairs, both tasks' loads requested together.  The horizontal
// pass and the output are k_dwt53_pk's.  One barrier instead of two, 8.8 instead of 19.9 KB of LDS.
// NT tiles per workgroup (round 4, last step): the two tiles' loads are requested together, so a wave has twice the bytes in flight for
// the same fixed latencies (its scalar set-up, one memory round trip, one barrier, the stores' issue) -- these kernels wait on round
// trips at full occupancy, not on issue slots.  NT = 2 on levels of at least DWT_RB2_MIN_TILES tiles (below that a launch is one
// round of workgroups and fewer, longer workgroups do not shorten it: 1080p levels 3 and 4 measured equal); the tiles are consecutive entries of the level's tile list and
// need not be neighbours.
#define DWT_RB2_MIN_TILES 2048
struct Rb53Tile {
  PlaneDesc pd;
  DwtSb2 sb2;
  int plane_i, w, h, lw, lh, hw, hh, tx0, ty0;
};
template <bool FROM_U8, int NT>
__global__ void __launch_bounds__(256) k_dwt53_rb(DwtArgs a, int ntiles) {
  constexpr int HALO = 2;
  static_assert(DT_TH == 16 && DT_TW == 64 && DT_CL == 4 && DT_NC == 136, "task maps of k_dwt53_rb");
  static_assert(NT == 1 || NT == 2, "tiles per workgroup");
  __shared__ uint32_t sV[NT][2 * DT_TH * DP_STR];   // per tile: its 32 output rows after the vertical lifting (row t = tile row t + 2)
  const int tid = threadIdx.x;
  Rb53Tile T[NT];
  bool valid[NT];
#pragma unroll
  for (int u = 0; u < NT; u++) {
    int tile_x, tile_y;
    if (NT == 1) { dwt_tile_of(a, T[u].plane_i, tile_x, tile_y); valid[u] = true; }
    else {
      const int b = (int)blockIdx.x * NT + u;
      valid[u] = b < ntiles;
      const uint32_t t = a.tilemap[valid[u] ? b : ntiles - 1];
      T[u].plane_i = (int)(t >> 20); tile_y = (int)((t >> 10) & 1023u); tile_x = (int)(t & 1023u);
    }
    T[u].pd = a.planes[T[u].plane_i];
    int w = T[u].pd.W, h = T[u].pd.H;
    for (int l = 0; l < a.lev; l++) { w = (w + 1) >> 1; h = (h + 1) >> 1; }
    T[u].w = w; T[u].h = h; T[u].lw = (w + 1) >> 1; T[u].lh = (h + 1) >> 1; T[u].hw = w >> 1; T[u].hh = h >> 1;
    T[u].tx0 = tile_x * DT_TW; T[u].ty0 = tile_y * DT_TH;
    if (NT == 1 && (T[u].tx0 >= T[u].lw || T[u].ty0 >= T[u].lh)) return;   // (3-D grid: planes of one batch may differ in size)
    T[u].sb2 = dwt_sb_pair(a, T[u].plane_i, tid);
  }
  // ---- vertical lifting out of registers ---------------------------------------------------------------------------------------
  {
    typedef uint16_t __attribute__((aligned(1))) u16a1;
    typedef uint32_t __attribute__((aligned(2))) u32a2;
    typedef const uint8_t __attribute__((address_space(1))) * gu8;
    typedef const u16a1 __attribute__((address_space(1))) * gu16;
    typedef const int16_t __attribute__((address_space(1))) * gs16;
    typedef const u32a2 __attribute__((address_space(1))) * gu32;
    // slots 0 .. NT - 1: the thread's task `tid` of tile 0 .. NT - 1 (no branch around them: a workgroup's missing second tile is a copy
    // of the level's last one, computed and thrown away); slot NT: one of the 16 tasks 256 .. 271 of tile tid / 16, by 16 NT threads.
    // The extra slot's loads are issued FIRST: behind its branch the wait counts of the main slots' loads stay exact.
    constexpr int NS = NT + 1;
    typename std::conditional<FROM_U8, uint16_t, uint32_t>::type raw[NS][11];   // (as 16-bit values: the zero extension of a 32-bit array is an instruction behind every load, and its wait comes before the next slot's loads)
    bool direct[NS];
    int cp[NS], sg[NS], ut[NS];
    const bool has_extra = tid < 16 * NT;
#pragma unroll
    for (int tt = 0; tt < NS; tt++) {
      const int t = tt == 0 ? NT : tt - 1;
      const bool extra = t == NT;
      ut[t] = extra ? (NT == 2 && tid >= 16 ? 1 : 0) : t;
      const int id = extra ? 256 + (tid & 15) : tid;
      cp[t] = id % DP_NP; sg[t] = id / DP_NP;
      const bool second = NT == 2 && ut[t] == 1;   // the slot's tile: uniform for the main slots, per thread for the extra one
      const int w = second ? T[NT - 1].w : T[0].w, h = second ? T[NT - 1].h : T[0].h;
      const int gc0 = 2 * (second ? T[NT - 1].tx0 : T[0].tx0) - DT_CL, gr0 = 2 * (second ? T[NT - 1].ty0 : T[0].ty0) - HALO;
      const int sstr = FROM_U8 ? (second ? T[NT - 1].pd.u8stride : T[0].pd.u8stride) : dwt_ll_stride(w);
      const gu8 u = (gu8)(uintptr_t)(second ? T[NT - 1].pd.src : T[0].pd.src);
      const size_t llo = ((a.lev - 1) & 1) ? (second ? T[NT - 1].pd.ll[1] : T[0].pd.ll[1]) : (second ? T[NT - 1].pd.ll[0] : T[0].pd.ll[0]);
      const gs16 lin = (gs16)(uintptr_t)((const int16_t*)a.llin + llo);
      const int gc = gc0 + 2 * cp[t], gr = gr0 + 8 * sg[t];
      direct[t] = gc >= 0 && gc + 1 < w;
      // the mirrored pair (gc + 1, gc) starts at the image of gc + 1.  Without a branch (mask arithmetic, not ?:): a branch here splits
      // the block, every 16-bit value loaded for an earlier slot is then widened where it crosses the boundary, and the waits of those
      // eleven `v_and` come before this slot's loads are issued -- the slots' round trips in a row instead of side by side
      const int dm = -(int)direct[t];
      int m = (gc & dm) | (dwt_reflect(gc + 1, w) & ~dm);
      m = m > w - 2 ? w - 2 : m;                         // (farther out than the lifting reads: any two samples of the row)
      if (!extra || has_extra) {
#pragma unroll
        for (int j = 0; j < 11; j++) {
          const uint32_t o = (__umul24((unsigned)dwt_reflect(gr + j, h), (unsigned)sstr) + (unsigned)m);
          if constexpr (FROM_U8) raw[t][j] = *(gu16)(u + o); else raw[t][j] = *(gu32)(lin + o);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // every load requested before the first is unpacked
#pragma unroll
    for (int t = 0; t < NS; t++) {
      if (t == NT && !has_extra) continue;
      uint32_t x[11];
#pragma unroll
      for (int j = 0; j < 11; j++) {
        const uint32_t d = raw[t][j];
        if constexpr (FROM_U8) x[j] = dp_sub(direct[t] ? __builtin_amdgcn_perm(0u, d, 0x0c010c00u) : __builtin_amdgcn_perm(0u, d, 0x0c000c01u), 0x00800080u);   // level shift
        else x[j] = direct[t] ? d : __builtin_amdgcn_alignbit(d, d, 16);
      }
      uint32_t hv[5];
#pragma unroll
      for (int j = 0; j < 5; j++) hv[j] = dp_high(x[2 * j + 1], x[2 * j], x[2 * j + 2]);   // tile rows 8 seg + 1, 3, 5, 7, 9
      uint32_t* o = &sV[0][0] + ut[t] * (2 * DT_TH * DP_STR) + (8 * sg[t]) * DP_STR + cp[t];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        o[(2 * j) * DP_STR] = dp_low(x[2 * j + 2], hv[j], hv[j + 1]);   // tile rows 8 seg + 2, 4, 6, 8 (lows)
        o[(2 * j + 1) * DP_STR] = hv[j + 1];                            // tile rows 8 seg + 3, 5, 7, 9 (highs)
      }
    }
  }
  __syncthreads();
  // ---- horizontal lifting + output: lane = (parity ry, stripe s of the tile, 2 subband columns 2 cq, 2 cq + 1) -----------------
  const bool last = a.lev == a.L - 1;
  const int cq = tid & 31, sidx = (tid >> 5) & 3, ry = tid >> 7;
  const int kx = 2 * cq;
#pragma unroll
  for (int u = 0; u < NT; u++) {
    if (!valid[u]) continue;   // (uniform)
    uint32_t Lw[4], Hw[4];                       // per row: {col kx, col kx + 1} of the low / high band
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t* row = &sV[u][0] + (2 * (4 * sidx + r) + ry) * DP_STR + (1 + kx);   // pairs holding tile columns 2 + 2 kx .. 9 + 2 kx
      const uint32_t d0 = row[0], d1 = row[1], d2 = row[2], d3 = row[3];
      const uint32_t E0 = __builtin_amdgcn_perm(d1, d0, 0x05040100u), E1 = __builtin_amdgcn_perm(d3, d2, 0x05040100u);  // evens e-1, e | e+1, e+2
      const uint32_t O0 = __builtin_amdgcn_perm(d1, d0, 0x07060302u), O1 = __builtin_amdgcn_perm(d3, d2, 0x07060302u);  // odds between them
      const uint32_t Es0 = __builtin_amdgcn_alignbit(E1, E0, 16);            // evens e, e + 1 (the lows' centres)
      const uint32_t Es1 = __builtin_amdgcn_alignbit(0u, E1, 16);            // even e + 2, -
      const uint32_t H0 = dp_high(O0, E0, Es0);                              // highs right of evens e - 1, e
      const uint32_t H1 = dp_high(O1, E1, Es1);                              // high right of even e + 1 (second lane unused)
      const uint32_t Hs = __builtin_amdgcn_alignbit(H1, H0, 16);             // highs right of evens e, e + 1 = subband columns kx, kx + 1
      Lw[r] = dp_low(Es0, H0, Hs);
      Hw[r] = Hs;
    }
    dwt53_pk_output(a, T[u].pd, T[u].plane_i, tid, last, ry, sidx, cq, T[u].tx0, T[u].ty0, T[u].lw, T[u].lh, T[u].hw, T[u].hh, Lw, Hw, T[u].sb2);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Irreversible 9/7, register-blocked form (round 4; what lossy coding runs -- k_dwt_tile<false> stays for the unit-test plane
// output, for levels under 8 samples and behind QSVC_DWT_PK=0).  The general kernel's eight read-modify-write sweeps over the
// staged tile (index arithmetic, three LDS reads and one write per lifted sample and step, a barrier after each) were 80 of its
// 103 lane-operations per sample.  Here each direction is ONE pass with all four lifting steps in registers:
//   vertical:   a lane owns 2 adjacent columns and 8 output rows: it loads the 15 rows [8 seg - 4, 8 seg + 11) of its pair
//               straight from memory (one 2-byte / 8-byte load per row, all in flight; a row or column beyond the signal's
//               end is read at its mirrored position -- whole-sample symmetric extension IS the restatement's boundary rule
//               "the missing neighbour is replaced by the one on the other side", for all four steps, because the lifted
//               extension stays symmetric), lifts them in registers and leaves 8 scaled rows in LDS.  68 pairs x 4 segments =
//               272 tasks: the workgroup has five waves for this pass (42 VGPRs.  Quads of columns on 136 threads of 256 needed
//               108 / 151 and left half of the workgroup idle: 1.33 ms per 33-picture 4K batch against 0.96 in this form and
//               1.54 for k_dwt_tile<false>);
//   horizontal: a lane owns 2 low and 2 high columns of the 4 rows of one stripe (as in k_dwt53_pk): three float4 reads per row
//               (12 samples: 4 outputs + halo 4), the steps in registers, then quantiser, sign-magnitude and the Tier-1 words
//               without any exchange between lanes.
// One barrier, 17 KB of LDS (all accesses lane-consecutive: no odd stride needed).  Every lifted value is the same chain of
// separately rounded products and sums as in the restatement (lift97), so the magnitudes are bit-identical.  What is left
// (builds with one part removed, 4K batch): the atomics of the bit-plane counts 6 %, the correctly rounded divisions of the
// quantiser 6 %, the LL stores 10 %: no single cost; 388 M vector wave-instructions at level 1 = 0.32 of its 0.66 ms at the issue peak.
#define RB_Q (DT_NC / 4)            // quads of 4 columns per staged row
#define RB_P (DT_NC / 2)            // pairs of columns per staged row
#define RB_THREADS 320              // 4 x 68 (segment, pair) tasks of the vertical pass on five waves; the horizontal pass runs on four
template <bool FROM_U8>
__global__ void __launch_bounds__(RB_THREADS) k_dwt97_rb(DwtArgs a) {
  constexpr int HALO = 4;
  static_assert(DT_TH == 16 && DT_TW == 64 && DT_CL == 4 && DT_NC == 136, "task maps of k_dwt97_rb");
  __shared__ float4 sV[2 * DT_TH * RB_Q];   // the tile's 32 output rows after the vertical lifting, columns [gc0, gc0 + 136)
  const int tid = threadIdx.x;
  int plane_i, tile_x, tile_y;
  dwt_tile_of(a, plane_i, tile_x, tile_y);
  const PlaneDesc pd = a.planes[plane_i];
  int w = pd.W, h = pd.H;
  for (int l = 0; l < a.lev; l++) { w = (w + 1) >> 1; h = (h + 1) >> 1; }
  const int lw = (w + 1) >> 1, lh = (h + 1) >> 1, hw = w >> 1, hh = h >> 1;
  const int tx0 = tile_x * DT_TW, ty0 = tile_y * DT_TH;
  if (tx0 >= lw || ty0 >= lh) return;
  const DwtSb2 sb2 = dwt_sb_pair(a, plane_i, tid);
  const int gc0 = 2 * tx0 - DT_CL, gr0 = 2 * ty0 - HALO;   // both even: local parity = global parity
  const float invK = 1.0f / F97_K;
  // ---- vertical lifting out of registers: thread = (pair of columns, segment of 8 output rows) -------------------------------
  if (tid < 4 * RB_P) {
    // A pair that leaves the signal at its left / right end is read at its mirrored position, which is again two adjacent samples
    // in reverse order: ONE (unaligned) load per row in every case, no second code path.
    typedef uint16_t __attribute__((aligned(1))) u16a1;
    typedef unsigned long long __attribute__((aligned(4))) u64a4;
    typedef const uint8_t __attribute__((address_space(1))) * gu8;
    typedef const u16a1 __attribute__((address_space(1))) * gu16;
    typedef const float __attribute__((address_space(1))) * gf1;
    typedef const u64a4 __attribute__((address_space(1))) * gu64;
    const gu8 u = (gu8)(uintptr_t)pd.src;
    const gf1 lin = (gf1)(uintptr_t)((const float*)a.llin + (((a.lev - 1) & 1) ? pd.ll[1] : pd.ll[0]));
    const int sstr = FROM_U8 ? pd.u8stride : dwt_ll_stride(w);
    const int p = tid % RB_P, seg = tid / RB_P;
    const int gc = gc0 + 2 * p, gr = gr0 + 8 * seg;   // rows gr .. gr + 14 feed the output rows gr + 4 .. gr + 11
    const bool direct = gc >= 0 && gc + 1 < w;
    int m = direct ? gc : dwt_reflect(gc + 1, w);     // the mirrored pair (gc + 1, gc) starts at the image of gc + 1
    m = m > w - 2 ? w - 2 : m;                        // (farther out than the lifting reads: any two samples of the row)
    float v[15][2];
    if constexpr (FROM_U8) {
      uint32_t raw[15];
#pragma unroll
      for (int j = 0; j < 15; j++) raw[j] = *(gu16)(u + (__umul24((unsigned)dwt_reflect(gr + j, h), (unsigned)sstr) + (unsigned)m));   // (32-bit offsets from a uniform base: one address register per load)
      __builtin_amdgcn_sched_barrier(0);   // all fifteen requested before the first is unpacked
#pragma unroll
      for (int j = 0; j < 15; j++) {
        const float lo = (float)((int)(raw[j] & 255) - 128), hi = (float)((int)(raw[j] >> 8) - 128);
        v[j][0] = direct ? lo : hi; v[j][1] = direct ? hi : lo;
      }
    } else {
      unsigned long long raw[15];
#pragma unroll
      for (int j = 0; j < 15; j++) raw[j] = *(gu64)(lin + (__umul24((unsigned)dwt_reflect(gr + j, h), (unsigned)sstr) + (unsigned)m));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 15; j++) {
        const float lo = __uint_as_float((uint32_t)raw[j]), hi = __uint_as_float((uint32_t)(raw[j] >> 32));
        v[j][0] = direct ? lo : hi; v[j][1] = direct ? hi : lo;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
#pragma unroll
      for (int j = 1; j <= 13; j += 2) v[j][k] = lift97<0>(v[j][k], v[j - 1][k], v[j + 1][k]);
#pragma unroll
      for (int j = 2; j <= 12; j += 2) v[j][k] = lift97<1>(v[j][k], v[j - 1][k], v[j + 1][k]);
#pragma unroll
      for (int j = 3; j <= 11; j += 2) v[j][k] = lift97<2>(v[j][k], v[j - 1][k], v[j + 1][k]);
#pragma unroll
      for (int j = 4; j <= 10; j += 2) v[j][k] = lift97<3>(v[j][k], v[j - 1][k], v[j + 1][k]);
    }
    float2* const sV2 = (float2*)sV;
#pragma unroll
    for (int j = 4; j < 12; j++) {   // low (even rows) * 1/K, high (odd rows) * K
      const float s = (j & 1) ? F97_K : invK;
      sV2[(8 * seg + j - 4) * RB_P + p] = make_float2(__fmul_rn(v[j][0], s), __fmul_rn(v[j][1], s));
    }
  }
  __syncthreads();
  if (tid >= 256) return;   // (after the only barrier)
  // ---- horizontal lifting + output: lane = (row parity ry, stripe sidx of the tile, subband columns 2 cq and 2 cq + 1) ---------
  const bool last = a.lev == a.L - 1;
  const int cq = tid & 31, sidx = (tid >> 5) & 3, ry = __builtin_amdgcn_readfirstlane(tid >> 7);   // (uniform per wave)
  float Lf[4][2], Hf[4][2];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float4* row = sV + (2 * (4 * sidx + r) + ry) * RB_Q + cq;   // tile columns 4 cq .. 4 cq + 11; the outputs are 4 cq + 4 .. + 7
    const float4 A = row[0], B = row[1], C = row[2];
    float x[11] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, C.x, C.y, C.z};
#pragma unroll
    for (int i = 1; i <= 9; i += 2) x[i] = lift97<0>(x[i], x[i - 1], x[i + 1]);
#pragma unroll
    for (int i = 2; i <= 8; i += 2) x[i] = lift97<1>(x[i], x[i - 1], x[i + 1]);
#pragma unroll
    for (int i = 3; i <= 7; i += 2) x[i] = lift97<2>(x[i], x[i - 1], x[i + 1]);
#pragma unroll
    for (int i = 4; i <= 6; i += 2) x[i] = lift97<3>(x[i], x[i - 1], x[i + 1]);
    Lf[r][0] = __fmul_rn(x[4], invK); Hf[r][0] = __fmul_rn(x[5], F97_K);
    Lf[r][1] = __fmul_rn(x[6], invK); Hf[r][1] = __fmul_rn(x[7], F97_K);
  }
  const int y = ty0 + 4 * sidx, x = tx0 + 2 * cq;
  const int lane = tid & 63;
  const int span2 = a.cbwl >= 6 ? 32 : (1 << (a.cbwl - 1));   // lanes (of 2 columns) per code-block
#pragma unroll
  for (int band = 0; band < 2; band++) {       // 0: the low band of this row parity (LL / LH), 1: the high band (HL / HH)
    const int sub = 2 * ry + band;
    if (sub == 0 && !last) continue;
    const int sbh = ry ? hh : lh, sbw = band ? hw : lw;
    const DwtSb sb = band ? sb2.hi : sb2.lo;
    int acc = 0;
    if (y < sbh && x < sbw) {
      const bool two = x + 1 < sbw;            // else the second column lies beyond the subband
      uint32_t q0[4], q1[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        q0[r] = q1[r] = 0u;
        if (y + r < sbh) {
          const float f0 = band ? Hf[r][0] : Lf[r][0], f1 = band ? Hf[r][1] : Lf[r][1];
          const int m0 = (int)floorf(__fdiv_rn(fabsf(f0), sb.delta)), m1 = (int)floorf(__fdiv_rn(fabsf(f1), sb.delta));   // T.800 E.1
          acc |= m0 | (two ? m1 : 0);
          q0[r] = (uint32_t)(m0 & 0x7FFF) | ((f0 < 0.0f && m0) ? 0x8000u : 0u);
          q1[r] = (uint32_t)(m1 & 0x7FFF) | ((f1 < 0.0f && m1) ? 0x8000u : 0u);
        }
      }
      const int bx = x >> a.cbwl, by = y >> a.cbhl;
      const size_t b = (size_t)sb.first_cb + (size_t)by * sb.gx + bx;
      unsigned long long* m = a.mags + b * 1024 + (size_t)(((y & ((1 << a.cbhl) - 1)) >> 2) * 64 + (x & ((1 << a.cbwl) - 1)));
      *(uint4*)m = make_uint4(q0[0] | q0[1] << 16, q0[2] | q0[3] << 16, two ? q1[0] | q1[1] << 16 : 0u, two ? q1[2] | q1[3] << 16 : 0u);   // (one store: see dwt53_pk_output)
    }
    // OR over the lanes of each code-block (DPP row shifts), one atomic per (block, stripe): the group's last lane speaks.  The OR
    // keeps every bit of the magnitudes: a step size that needs more than 15 bit-planes is reported by the host, not truncated here.
    const int v = wave_or_segments(acc, span2);
    if ((lane & (span2 - 1)) == span2 - 1 && v && y < sbh) {
      const int xl = tx0 + 2 * (cq | (span2 - 1));   // a column of this lane group's block (may lie beyond the subband: same block index)
      atomicOr(&a.maxmag[sb.first_cb + (y >> a.cbhl) * sb.gx + ((xl < sbw ? xl : sbw - 1) >> a.cbwl)], v);
    }
  }
  if (!last && ry == 0) {   // LL -> next level's input
    float* lout = (float*)a.llout + ((a.lev & 1) ? pd.ll[1] : pd.ll[0]);
    const int ostr = dwt_ll_stride(lw);
    if (x < ostr)
#pragma unroll
      for (int r = 0; r < 4; r++)
        if (y + r < lh) *(float2*)(lout + (size_t)(y + r) * ostr + x) = make_float2(Lf[r][0], Lf[r][1]);   // columns in [lw, ostr) are padding, never read back
  }
}

// Forward DWT of a batch of planes sharing L >= 1 levels.  mode 0: Tier-1 magnitudes; mode 1: Mallat plane (unit tests).
// Workgroup list of every level of a batch (dwt_tile_of): host side.  first[l] .. first[l + 1] = level l's entries.
void dwt_build_tilemap(const PlaneDesc* planes, int nplanes, int L, std::vector<uint32_t>& map, std::vector<int>& first) {
  map.clear(); first.assign(L + 1, 0);
  for (int l = 0; l < L; l++) {
    first[l] = (int)map.size();
    for (int p = 0; p < nplanes; p++) {
      int w = planes[p].W, h = planes[p].H;
      for (int k = 0; k < l; k++) { w = (w + 1) >> 1; h = (h + 1) >> 1; }
      const int lw = (w + 1) >> 1, lh = (h + 1) >> 1, gx = (lw + DT_TW - 1) / DT_TW, gy = (lh + DT_TH - 1) / DT_TH;
      for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) map.push_back((uint32_t)p << 20 | (uint32_t)ty << 10 | (uint32_t)tx);
    }
  }
  first[L] = (int)map.size();
}
bool dwt_tilemap_fits(int nplanes, int maxW, int maxH) { return nplanes < 4096 && (maxW + 1) / 2 <= 1024 * DT_TW && (maxH + 1) / 2 <= 1024 * DT_TH; }

// d_tilemap / first (dwt_build_tilemap; nullptr: a 3-D grid over the largest plane)
void launch_dwt_tiles(hipStream_t st, const PlaneDesc* d_planes, int nplanes, int maxW, int maxH, int minW0, int minH0, const DwtSb* d_sbtab, int L,
                      bool reversible, void* llA, void* llB, unsigned long long* mags, int* maxmag, void* plane_out,
                      int cbw_log2, int cbh_log2, int* err, const uint32_t* d_tilemap, const int* first) {
  int w = maxW, h = maxH, minW = minW0, minH = minH0;   // the largest and the smallest plane of the batch at this level
  for (int l = 0; l < L; l++) {
    const int lw = (w + 1) >> 1, lh = (h + 1) >> 1;
    DwtArgs a;
    a.planes = d_planes; a.sbtab = d_sbtab; a.nsb = 1 + 3 * L; a.lev = l; a.L = L;
    a.llin = (l & 1) ? llA : llB;    // LL_l lives in A for odd l, in B for even l >= 2
    a.llout = (l & 1) ? llB : llA;   // LL_{l+1}
    a.mags = mags; a.maxmag = maxmag; a.plane_out = plane_out; a.cbwl = cbw_log2; a.cbhl = cbh_log2; a.err = err;
    a.tilemap = d_tilemap ? d_tilemap + first[l] : nullptr;
    dim3 grid((lw + DT_TW - 1) / DT_TW, (lh + DT_TH - 1) / DT_TH, nplanes), block(256);
    if (d_tilemap) grid = dim3(first[l + 1] - first[l], 1, 1);
#define DWT_LAUNCH(REV, U8, PL) hipLaunchKernelGGL((k_dwt_tile<REV, U8, PL>), grid, block, 0, st, a)
    if (plane_out) {
      if (reversible) { if (l == 0) DWT_LAUNCH(true, true, true); else DWT_LAUNCH(true, false, true); }
      else { if (l == 0) DWT_LAUNCH(false, true, true); else DWT_LAUNCH(false, false, true); }
    } else {
      // the packed 5/3 kernels: levels of at least 4 x 4 samples, code-blocks at least 8 wide (QSVC_DWT_PK=0: the general kernel, 2: the staged packed kernel)
      const bool pk = reversible && tuning().dwt_pk && minW >= 4 && minH >= 4 && cbw_log2 >= 3 && cbh_log2 >= 2;
      // the register-blocked 9/7 kernel: levels of at least 8 x 8 samples (its mirrored reads reach 4 samples beyond either end)
      const bool rb = !reversible && tuning().dwt_pk && minW >= 8 && minH >= 8 && cbw_log2 >= 3 && cbh_log2 >= 2;
      if (pk && tuning().dwt_pk == 2) { if (l == 0) hipLaunchKernelGGL(k_dwt53_pk<true>, grid, block, 0, st, a); else hipLaunchKernelGGL(k_dwt53_pk<false>, grid, block, 0, st, a); }   // (the staged form, for comparison)
      else if (pk) {
        const int nt = d_tilemap ? first[l + 1] - first[l] : 0;
        if ((nt >= DWT_RB2_MIN_TILES && tuning().dwt_pk != 3) || (nt >= 1 && tuning().dwt_pk == 4)) {   // two tiles per workgroup (QSVC_DWT_PK=3: always one, 4: always two)
          const dim3 g2((nt + 1) / 2);
          if (l == 0) hipLaunchKernelGGL((k_dwt53_rb<true, 2>), g2, block, 0, st, a, nt); else hipLaunchKernelGGL((k_dwt53_rb<false, 2>), g2, block, 0, st, a, nt);
        } else {
          if (l == 0) hipLaunchKernelGGL((k_dwt53_rb<true, 1>), grid, block, 0, st, a, nt); else hipLaunchKernelGGL((k_dwt53_rb<false, 1>), grid, block, 0, st, a, nt);
        }
      }
      else if (rb) { if (l == 0) hipLaunchKernelGGL(k_dwt97_rb<true>, grid, dim3(RB_THREADS), 0, st, a); else hipLaunchKernelGGL(k_dwt97_rb<false>, grid, dim3(RB_THREADS), 0, st, a); }
      else if (reversible) { if (l == 0) DWT_LAUNCH(true, true, false); else DWT_LAUNCH(true, false, false); }
      else { if (l == 0) DWT_LAUNCH(false, true, false); else DWT_LAUNCH(false, false, false); }
    }
#undef DWT_LAUNCH
    w = lw; h = lh; minW = (minW + 1) >> 1; minH = (minH + 1) >> 1;
  }
}

}  // namespace qsvc
