// qsvc_amd/csrc/j2k_kernels.hip -- gfx950 kernels of the JPEG 2000 texture stage that replaces the
// reference's external `kdu_compress` call (trunk/src/texture_compress_fb_j2k.py:183-195):
//   * forward DWT, ITU-T T.800 Annex F: reversible 5/3 (int32) and irreversible 9/7 (float32),
//     vertical pass then horizontal pass per level, in-place Mallat layout, batched over planes;
//   * dead-zone quantiser (Annex E) + packing into the lane-interleaved Tier-1 layout;
//   * EBCOT Tier-1 (Annex C + D): 64 code-blocks per wavefront, one per lane (t1_core.h);
//   * gather of the code-block segments into one contiguous buffer for the host's Tier-2.
// No MFMA: integer/bit work and short float lifting chains.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "j2k_kernels.h"
#include "t1_cm2.h"

namespace qsvc {

// ---------------------------------------------------------------------------------------------
// DWT.  One thread per output coefficient of one 1-D pass; the lifting chain is re-evaluated per
// output with the oracle's exact operation order (clamped neighbour indices = whole-sample
// symmetric extension), products and sums rounded separately (no FMA contraction).
#define F97_ALPHA (-1.586134342059924f)
#define F97_BETA (-0.052980118572961f)
#define F97_GAMMA (0.882911075530934f)
#define F97_DELTA (0.443506852043971f)
#define F97_K (1.230174104914001f)

template <class Ld>
__device__ __forceinline__ int dwt53_coef(const Ld& X, int k, int n) {  // T.800 F.4.8.1
  int nl = (n + 1) >> 1, nh = n >> 1;
  auto Hc = [&](int i) {
    i = i < 0 ? 0 : (i >= nh ? nh - 1 : i);
    int a = X(2 * i), b = (2 * i + 2 < n) ? X(2 * i + 2) : a;
    return X(2 * i + 1) - ((a + b) >> 1);
  };
  if (k >= nl) return Hc(k - nl);
  return X(2 * k) + ((Hc(k - 1) + Hc(k) + 2) >> 2);
}
template <class Ld>
__device__ __forceinline__ float dwt97_coef(const Ld& X, int k, int n) {  // T.800 F.4.8.2
  int nl = (n + 1) >> 1, nh = n >> 1;
  auto E0 = [&](int i) { return X(2 * (i >= nl ? nl - 1 : i)); };
  auto O1 = [&](int i) {
    i = i < 0 ? 0 : (i >= nh ? nh - 1 : i);
    return __fadd_rn(X(2 * i + 1), __fmul_rn(F97_ALPHA, __fadd_rn(E0(i), E0(i + 1))));
  };
  auto E1 = [&](int i) {
    i = i >= nl ? nl - 1 : i;
    return __fadd_rn(X(2 * i), __fmul_rn(F97_BETA, __fadd_rn(O1(i - 1), O1(i))));
  };
  auto O2 = [&](int i) {
    i = i < 0 ? 0 : (i >= nh ? nh - 1 : i);
    return __fadd_rn(O1(i), __fmul_rn(F97_GAMMA, __fadd_rn(E1(i), E1(i + 1))));
  };
  if (k >= nl) return __fmul_rn(O2(k - nl), F97_K);
  float e2 = __fadd_rn(E1(k), __fmul_rn(F97_DELTA, __fadd_rn(O2(k - 1), O2(k))));
  const float invK = 1.0f / F97_K;
  return __fmul_rn(e2, invK);
}

// vertical pass of level `lev` (region hr x wr of each plane): src -> dst (same layout)
template <bool REV, bool FROM_U8>
__global__ void __launch_bounds__(256) k_dwt_v(const PlaneDesc* planes, const void* srcv, void* dstv, int lev) {
  PlaneDesc pd = planes[blockIdx.z];
  int wr = pd.W, hr = pd.H;
  for (int l = 0; l < lev; l++) { wr = (wr + 1) >> 1; hr = (hr + 1) >> 1; }
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= wr || y >= hr) return;
  size_t o = pd.off + (size_t)y * pd.W + x;
  if (REV) {
    const int32_t* s = (const int32_t*)srcv + pd.off;
    const uint8_t* u = pd.src;
    auto X = [&](int yy) {
      if (!FROM_U8) return s[(size_t)yy * pd.W + x];
      return pd.s16 ? (int)((const int16_t*)u)[(size_t)yy * pd.u8stride + x] : (int)u[(size_t)yy * pd.u8stride + x] - 128;
    };
    ((int32_t*)dstv)[o] = hr >= 2 ? dwt53_coef(X, y, hr) : X(y);
  } else {
    const float* s = (const float*)srcv + pd.off;
    const uint8_t* u = pd.src;
    auto X = [&](int yy) { return FROM_U8 ? (float)((int)u[(size_t)yy * pd.u8stride + x] - 128) : s[(size_t)yy * pd.W + x]; };
    ((float*)dstv)[o] = hr >= 2 ? dwt97_coef(X, y, hr) : X(y);
  }
}
template <bool REV>
__global__ void __launch_bounds__(256) k_dwt_h(const PlaneDesc* planes, const void* srcv, void* dstv, int lev) {
  PlaneDesc pd = planes[blockIdx.z];
  int wr = pd.W, hr = pd.H;
  for (int l = 0; l < lev; l++) { wr = (wr + 1) >> 1; hr = (hr + 1) >> 1; }
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= wr || y >= hr) return;
  size_t row = pd.off + (size_t)y * pd.W;
  if (REV) {
    const int32_t* s = (const int32_t*)srcv + row;
    auto X = [&](int xx) { return s[xx]; };
    ((int32_t*)dstv)[row + x] = wr >= 2 ? dwt53_coef(X, x, wr) : X(x);
  } else {
    const float* s = (const float*)srcv + row;
    auto X = [&](int xx) { return s[xx]; };
    ((float*)dstv)[row + x] = wr >= 2 ? dwt97_coef(X, x, wr) : X(x);
  }
}
// level 0 with no decomposition at all: just the level shift
template <bool REV>
__global__ void __launch_bounds__(256) k_level_shift(const PlaneDesc* planes, void* dstv) {
  PlaneDesc pd = planes[blockIdx.z];
  int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= pd.W || y >= pd.H) return;
  int v = pd.s16 ? (int)((const int16_t*)pd.src)[(size_t)y * pd.u8stride + x] : (int)pd.src[(size_t)y * pd.u8stride + x] - 128;
  if (REV) ((int32_t*)dstv)[pd.off + (size_t)y * pd.W + x] = v;
  else ((float*)dstv)[pd.off + (size_t)y * pd.W + x] = (float)v;
}

void launch_dwt(hipStream_t st, const PlaneDesc* d_planes, int nplanes, int maxW, int maxH, void* A,
                void* B, int levels, bool reversible) {
  dim3 block(64, 4);
  if (levels == 0) {
    dim3 grid((maxW + 63) / 64, (maxH + 3) / 4, nplanes);
    if (reversible) hipLaunchKernelGGL(k_level_shift<true>, grid, block, 0, st, d_planes, A);
    else hipLaunchKernelGGL(k_level_shift<false>, grid, block, 0, st, d_planes, A);
    return;
  }
  int w = maxW, h = maxH;
  for (int l = 0; l < levels; l++) {
    dim3 grid((w + 63) / 64, (h + 3) / 4, nplanes);
    // vertical: A (or the u8 source at level 0) -> B ; horizontal: B -> A
    if (reversible) {
      if (l == 0) hipLaunchKernelGGL((k_dwt_v<true, true>), grid, block, 0, st, d_planes, A, B, l);
      else hipLaunchKernelGGL((k_dwt_v<true, false>), grid, block, 0, st, d_planes, A, B, l);
      hipLaunchKernelGGL(k_dwt_h<true>, grid, block, 0, st, d_planes, B, A, l);
    } else {
      if (l == 0) hipLaunchKernelGGL((k_dwt_v<false, true>), grid, block, 0, st, d_planes, A, B, l);
      else hipLaunchKernelGGL((k_dwt_v<false, false>), grid, block, 0, st, d_planes, A, B, l);
      hipLaunchKernelGGL(k_dwt_h<false>, grid, block, 0, st, d_planes, B, A, l);
    }
    w = (w + 1) >> 1; h = (h + 1) >> 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Quantise + pack one code-block per workgroup into the lane-interleaved Tier-1 layout.
template <bool REV>
__global__ void __launch_bounds__(256) k_t1_pack(const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coefv,
                                                  uint32_t* mags, int* nbps_out, int* err, int lanes) {
  int b = blockIdx.x;
  if (b >= ncb) return;
  CbDesc cb = cbs[b];
  PlaneDesc pd = planes[cb.plane];
  int g = b / lanes, lane = b % lanes;
  uint32_t* mg = mags + (size_t)g * T1_MAGS_WORDS * lanes + lane;
  __shared__ int smax[256];
  int x = threadIdx.x & 63, sq = threadIdx.x >> 6;
  int mymax = 0;
  for (int s = sq; s < 16; s += 4) {
    uint64_t word = 0;
    if (x < cb.w) {
      for (int k = 0; k < 4; k++) {
        int y = 4 * s + k;
        if (y >= cb.h) break;
        size_t i = pd.off + (size_t)(cb.y0 + y) * pd.W + cb.x0 + x;
        int mag, neg;
        if (REV) {
          int v = ((const int32_t*)coefv)[i];
          neg = v < 0; mag = neg ? -v : v;
        } else {
          float v = ((const float*)coefv)[i];
          mag = (int)floorf(__fdiv_rn(fabsf(v), cb.delta));  // dead-zone quantiser, T.800 E.1
          neg = v < 0.0f && mag;
        }
        mymax = mag > mymax ? mag : mymax;
        word |= (uint64_t)((mag & 0x7FFF) | (neg << 15)) << (16 * k);
      }
    }
    mg[(size_t)((s * 64 + x) * 2 + 0) * lanes] = (uint32_t)word;
    mg[(size_t)((s * 64 + x) * 2 + 1) * lanes] = (uint32_t)(word >> 32);
  }
  smax[threadIdx.x] = mymax;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + o] ? smax[threadIdx.x] : smax[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int m = smax[0], nb = 0;
    while (nb < 31 && (m >> nb)) nb++;
    nbps_out[b] = nb > 15 ? 15 : nb;
    if (nb > cb.Mb || nb > 15) atomicOr(err, 1);  // magnitude does not fit M_b / the 15-bit lane format
  }
}
// The same for the 64-lane layout of the split Tier-1, one workgroup per (stripe, group of 64 blocks): coefficient rows are
// read coalesced along x, the packed words go through an LDS transpose and leave as full 256-byte runs along the lane
// index (the per-block form above scatters 4-byte stores 256 bytes apart: 12x write amplification in HBM).
// maxmag[b] collects the block's largest magnitude (atomicMax; zeroed by the launcher), k_t1_nbps turns it into nbps.
#define PK_ROW 65
template <bool REV>
__global__ void __launch_bounds__(256) k_t1_pack64(const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coefv,
                                                    uint32_t* mags, int* maxmag) {
  __shared__ uint32_t sT[128 * PK_ROW];  // [(x*2 + k)][lane]
  __shared__ int smax[64];
  const int s = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  const int x = tid & 63, lq = tid >> 6;
  if (tid < 64) smax[tid] = 0;
  __syncthreads();
  for (int l = lq; l < 64; l += 4) {
    const int b = g * 64 + l;
    uint64_t word = 0;
    int mymax = 0;
    if (b < ncb) {
      const CbDesc cb = cbs[b];
      if (x < cb.w && 4 * s < cb.h) {
        const PlaneDesc pd = planes[cb.plane];
        for (int k = 0; k < 4; k++) {
          int y = 4 * s + k;
          if (y >= cb.h) break;
          size_t i = pd.off + (size_t)(cb.y0 + y) * pd.W + cb.x0 + x;
          int mag, neg;
          if (REV) {
            int v = ((const int32_t*)coefv)[i];
            neg = v < 0; mag = neg ? -v : v;
          } else {
            float v = ((const float*)coefv)[i];
            mag = (int)floorf(__fdiv_rn(fabsf(v), cb.delta));  // dead-zone quantiser, T.800 E.1
            neg = v < 0.0f && mag;
          }
          mymax = mag > mymax ? mag : mymax;
          word |= (uint64_t)((mag & 0x7FFF) | (neg << 15)) << (16 * k);
        }
      }
    }
    sT[(x * 2 + 0) * PK_ROW + l] = (uint32_t)word;
    sT[(x * 2 + 1) * PK_ROW + l] = (uint32_t)(word >> 32);
    if (mymax) atomicMax(&smax[l], mymax);
  }
  __syncthreads();
  uint32_t* mg = mags + (size_t)g * T1_MAGS_WORDS * 64 + (size_t)s * 128 * 64;
  for (int i = tid; i < 128 * 64; i += 256) mg[i] = sT[(i >> 6) * PK_ROW + (i & 63)];
  if (tid < 64 && g * 64 + tid < ncb && smax[tid]) atomicMax(&maxmag[g * 64 + tid], smax[tid]);
}
__global__ void __launch_bounds__(256) k_t1_nbps(const CbDesc* cbs, int ncb, int* nbps, int* err) {
  int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= ncb) return;
  int m = nbps[b], nb = 0;
  while (nb < 31 && (m >> nb)) nb++;
  nbps[b] = nb > 15 ? 15 : nb;
  if (nb > cbs[b].Mb || nb > 15) atomicOr(err, 1);  // magnitude does not fit M_b / the 15-bit lane format
}
void launch_t1_pack(hipStream_t st, const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coef, uint32_t* mags,
                    int* nbps, int* err, bool reversible, int lanes) {
  if (lanes == 64) {
    hipMemsetAsync(nbps, 0, (size_t)ncb * 4, st);
    dim3 grid(16, (ncb + 63) / 64);
    if (reversible) hipLaunchKernelGGL(k_t1_pack64<true>, grid, dim3(256), 0, st, cbs, ncb, planes, coef, mags, nbps);
    else hipLaunchKernelGGL(k_t1_pack64<false>, grid, dim3(256), 0, st, cbs, ncb, planes, coef, mags, nbps);
    hipLaunchKernelGGL(k_t1_nbps, dim3((ncb + 255) / 256), dim3(256), 0, st, cbs, ncb, nbps, err);
    return;
  }
  if (reversible) hipLaunchKernelGGL(k_t1_pack<true>, dim3(ncb), dim3(256), 0, st, cbs, ncb, planes, coef, mags, nbps, err, lanes);
  else hipLaunchKernelGGL(k_t1_pack<false>, dim3(ncb), dim3(256), 0, st, cbs, ncb, planes, coef, mags, nbps, err, lanes);
}

// ---------------------------------------------------------------------------------------------
// EBCOT Tier-1: one wavefront = 64 code-blocks, one per lane (t1_core.h).
//
// Block state streams through an LDS window: per pass every flag row (66 words x 64 lanes) and every
// stripe's magnitudes are brought in ONCE by LDS-DMA (global_load_lds, 256 contiguous bytes per
// wave-instruction, no VGPR round trip), one stripe ahead of the stripe being coded, and each flag row is
// written back once.  The coding loop itself touches only LDS and registers, so the serial per-lane chain
// never waits on HBM.  LDS per wave: 4 flag-row slots (67.6 KB) + 2 magnitude buffers (64 KB) + coder
// tables (3 KB): one workgroup per CU.
__constant__ uint32_t c_mqtab[47] = T1_MQ_TABLE_INIT;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int LANES>
struct T1LdsStore {
  uint32_t* F;            // LDS [4][T1_FW][LANES]
  uint32_t* M;            // LDS [2][128][LANES]
  uint32_t* gflags;       // this group's flag words in HBM (lane-interleaved); already offset by lane
  const uint32_t* gmags;  // this group's magnitude words in HBM; already offset by lane
  int lane;
  __device__ __forceinline__ uint32_t ld(int r, int c) const { return F[((r & 3) * T1_FW + c) * LANES + lane]; }
  __device__ __forceinline__ void st(int r, int c, uint32_t v) { F[((r & 3) * T1_FW + c) * LANES + lane] = v; }
  __device__ __forceinline__ void orr(int r, int c, uint32_t b) { F[((r & 3) * T1_FW + c) * LANES + lane] |= b; }
  __device__ __forceinline__ uint64_t mag(int s, int x) const {
    const uint32_t* m = M + ((s & 1) * 128 + x * 2) * LANES + lane;
    return (uint64_t)m[0] | ((uint64_t)m[LANES] << 32);
  }
  __device__ __forceinline__ void dma_row(int r) {  // flag row r -> slot r&3
    const uint32_t* g = gflags + (size_t)r * T1_FW * LANES;
    uint32_t* l = F + (r & 3) * T1_FW * LANES;
#pragma unroll 6
    for (int c = 0; c < T1_FW; c++)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + c * LANES), (lds_ptr_t)(l + c * LANES), 4, 0, 0);
  }
  __device__ __forceinline__ void dma_mags(int s) {  // magnitudes of stripe s -> buffer s&1
    const uint32_t* g = gmags + (size_t)s * 128 * LANES;
    uint32_t* l = M + (s & 1) * 128 * LANES;
#pragma unroll 8
    for (int i = 0; i < 128; i++)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + i * LANES), (lds_ptr_t)(l + i * LANES), 4, 0, 0);
  }
  __device__ __forceinline__ void wb_row(int r) {  // slot r&3 -> flag row r in HBM
    uint32_t* g = gflags + (size_t)r * T1_FW * LANES;
    const uint32_t* l = F + (r & 3) * T1_FW * LANES + lane;
#pragma unroll 6
    for (int c = 0; c < T1_FW; c++) g[c * LANES] = l[c * LANES];
  }
  __device__ __forceinline__ void pass_begin(int nstripes) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous pass's write-backs
    dma_row(0); dma_row(1); dma_row(2);
    dma_mags(0);
  }
  __device__ __forceinline__ void stripe_begin(int s, int nstripes) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // rows <= s+2 and magnitudes of stripe s have landed
    if (s + 1 < nstripes) { dma_row(s + 3); dma_mags(s + 1); }  // one stripe ahead; slot (s+3)&3 held row s-1
  }
  __device__ __forceinline__ void stripe_end(int s) { wb_row(s); }
  __device__ __forceinline__ void pass_end(int nstripes) { wb_row(nstripes); wb_row(nstripes + 1); }
};

template <int LANES>
__global__ void __launch_bounds__(LANES) k_t1_encode(const CbDesc* cbs, int ncb, const int* nbps_in, const uint32_t* mags,
                                                      uint32_t* flags, uint8_t* outbuf, int outcap, CbOut* res, int* pass_len,
                                                      long long* pass_dist, int want_dist) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* s_F = (uint32_t*)smem;                           // 4*66*LANES words
  uint32_t* s_M = s_F + 4 * T1_FW * LANES;                   // 2*128*LANES words
  uint32_t* s_mq = s_M + 2 * 128 * LANES;                    // 47 (padded to 64)
  uint8_t* s_zc = (uint8_t*)(s_mq + 64);                     // 3*512
  uint32_t* s_cx = (uint32_t*)(s_zc + 3 * 512);              // 19*LANES words
  int lane = threadIdx.x, g = blockIdx.x, b = g * LANES + lane;
  for (int i = lane; i < 47; i += LANES) s_mq[i] = c_mqtab[i];
  for (int i = lane; i < 3 * 512; i += LANES) {
    int c = i >> 9;
    s_zc[i] = (uint8_t)t1_zc_from_nb(i & 511, c == 0 ? 0 : (c == 1 ? 1 : 3));
  }
  __syncthreads();
  T1Lane t;
  t.cx = s_cx + lane; t.cs = LANES;
  t.mqtab = s_mq; t.zclut = s_zc;
  t.w = 0; t.h = 0; t.orient = 0; t.nbps = 0;
  t.out = outbuf; t.cap = 0;
  if (b < ncb) {
    CbDesc cb = cbs[b];
    t.w = cb.w; t.h = cb.h; t.orient = cb.orient; t.nbps = nbps_in[b];
    t.out = outbuf + (size_t)b * outcap; t.cap = outcap;
  }
  T1LdsStore<LANES> S;
  S.F = s_F; S.M = s_M; S.lane = lane;
  S.gflags = flags + (size_t)g * T1_FLAGS_WORDS * LANES + lane;
  S.gmags = mags + (size_t)g * T1_MAGS_WORDS * LANES + lane;
  // wave-uniform scan bounds
  int top = t.nbps - 1, wmax = t.nbps ? t.w : 0, hmax = t.nbps ? t.h : 0;
  for (int o = LANES / 2; o >= 1; o >>= 1) {
    int a = __shfl_xor(top, o, 64), c = __shfl_xor(wmax, o, 64), d = __shfl_xor(hmax, o, 64);
    top = a > top ? a : top; wmax = c > wmax ? c : wmax; hmax = d > hmax ? d : hmax;
  }
  top = __builtin_amdgcn_readfirstlane(top); wmax = __builtin_amdgcn_readfirstlane(wmax); hmax = __builtin_amdgcn_readfirstlane(hmax);
  T1Result r;
  int bb = b < ncb ? b : 0;
  r.pass_len = pass_len + (size_t)bb * T1_MAXPASS;
  r.pass_dist = want_dist ? pass_dist + (size_t)bb * T1_MAXPASS : nullptr;
  t1_encode_lane(t, S, top, wmax, hmax, want_dist != 0, r);
  if (b < ncb) {
    CbOut o;
    o.len = r.len; o.npasses = r.npasses; o.nbps = r.nbps; o.overflow = r.overflow; o.nsym = r.nsym;
    res[b] = o;
  }
}
template <int LANES>
static void launch_t1_encode_n(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const uint32_t* mags, uint32_t* flags,
                               uint8_t* outbuf, int outcap, CbOut* res, int* pass_len, long long* pass_dist, int want_dist) {
  const int smem = ((4 * T1_FW + 2 * 128) * LANES + 64) * 4 + 3 * 512 + 19 * LANES * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_t1_encode<LANES>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  int ngroups = (ncb + LANES - 1) / LANES;
  hipLaunchKernelGGL(k_t1_encode<LANES>, dim3(ngroups), dim3(LANES), smem, st, cbs, ncb, nbps, mags, flags, outbuf, outcap, res,
                     pass_len, pass_dist, want_dist);
}
void launch_t1_encode(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const uint32_t* mags, uint32_t* flags,
                      uint8_t* outbuf, int outcap, CbOut* res, int* pass_len, long long* pass_dist, int want_dist, int lanes) {
  switch (lanes) {
    case 8: launch_t1_encode_n<8>(st, cbs, ncb, nbps, mags, flags, outbuf, outcap, res, pass_len, pass_dist, want_dist); break;
    case 16: launch_t1_encode_n<16>(st, cbs, ncb, nbps, mags, flags, outbuf, outcap, res, pass_len, pass_dist, want_dist); break;
    case 32: launch_t1_encode_n<32>(st, cbs, ncb, nbps, mags, flags, outbuf, outcap, res, pass_len, pass_dist, want_dist); break;
    default: launch_t1_encode_n<64>(st, cbs, ncb, nbps, mags, flags, outbuf, outcap, res, pass_len, pass_dist, want_dist); break;
  }
}

// ---------------------------------------------------------------------------------------------
// Split Tier-1: bit-plane-parallel context modelling + MQ kernel (t1_core.h "Split form").
// A CM wave = (group of 64 code-blocks, bit-plane p): its 64 lanes model plane p of 64 different blocks.
// The start-of-plane flag words are a pure function of the magnitudes, produced by k_t1_flags_init
// (thread per flag word), so all ~10 planes of all blocks run concurrently: ~10x the waves of the fused
// kernel, each with 1/10 of its serial chain.  Decisions (cx | d<<5, one byte) go to per-(block, plane, pass)
// segments; k_t1_mq runs the MQ coder of each block over its segments, 16 blocks per wave.
struct T1GlobalStore {  // flag words and magnitudes straight from HBM/L2, lane stride 64
  uint32_t* flags;
  const uint32_t* mags;
  // Neighbour updates are fire-and-forget L2 atomics (no round trip in the lane's dependent chain); flag loads
  // and stores are agent-scope so they meet those atomics in L2 instead of a stale L1 line.
  __device__ __forceinline__ uint32_t ld(int r, int c) const {
    return __hip_atomic_load(flags + (r * T1_FW + c) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void st(int r, int c, uint32_t v) {
    __hip_atomic_store(flags + (r * T1_FW + c) * 64, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ void orr(int r, int c, uint32_t b) {
    (void)__hip_atomic_fetch_or(flags + (r * T1_FW + c) * 64, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ uint64_t mag(int s, int x) const {
    const uint32_t* m = mags + (size_t)((s * 64 + x) * 2) * 64;
    return (uint64_t)m[0] | ((uint64_t)m[64] << 32);
  }
  __device__ __forceinline__ void pass_begin(int) {}
  __device__ __forceinline__ void stripe_begin(int, int) {}
  __device__ __forceinline__ void stripe_end(int) {}
  __device__ __forceinline__ void pass_end(int) {}
};

// waves[k] = group << 8 | plane.  One thread per (flag word, lane) of a CM wave.
__global__ void __launch_bounds__(256) k_t1_flags_init(const int* waves, const CbDesc* cbs, int ncb, const int* nbps_in,
                                                        const uint32_t* mags, uint32_t* flags) {
  int wv = blockIdx.y, g = waves[wv] >> 8, p = waves[wv] & 255;
  int idx = blockIdx.x * 256 + threadIdx.x;  // word*64 + lane
  if (idx >= T1_FLAGS_WORDS * 64) return;
  int lane = idx & 63, word = idx >> 6, r = word / T1_FW, c = word % T1_FW;
  int b = g * 64 + lane;
  uint32_t f = 0;
  if (b < ncb && r >= 1 && r <= 16 && c >= 1 && c <= 64 && p < nbps_in[b]) {
    CbDesc cb = cbs[b];
    int s = r - 1, x = c - 1, nstr = (cb.h + 3) >> 2;
    if (x < cb.w && s < nstr) {
      const uint32_t* mg = mags + (size_t)g * T1_MAGS_WORDS * 64 + lane;
      auto M = [&](int ss, int xx) {
        const uint32_t* m = mg + (size_t)((ss * 64 + xx) * 2) * 64;
        return (uint64_t)m[0] | ((uint64_t)m[64] << 32);
      };
      f = t1_initial_flag(M, s, x, p, nstr, cb.w);
    }
  }
  flags[(size_t)wv * T1_FLAGS_WORDS * 64 + idx] = f;
}

template <int CML>  // lanes per CM wave: 64, 32 or 16 consecutive blocks of the group
__global__ void __launch_bounds__(CML) k_t1_cm(const int* waves, const CbDesc* cbs, int ncb, const int* nbps_in, const int* joboff,
                                               const uint32_t* mags, uint32_t* flags, uint8_t* segs, int* cnt, long long* dist,
                                               int want_dist) {
  __shared__ uint8_t s_zc[3 * 512];
  int wv = blockIdx.x, g = waves[wv] >> 8, p = waves[wv] & 255;
  int lane = blockIdx.y * CML + threadIdx.x, b = g * 64 + lane;
  for (int i = threadIdx.x; i < 3 * 512; i += CML) {
    int c = i >> 9;
    s_zc[i] = (uint8_t)t1_zc_from_nb(i & 511, c == 0 ? 0 : (c == 1 ? 1 : 3));
  }
  __syncthreads();
  int w = 0, h = 0, orient = 0, nb = 0;
  if (b < ncb) { CbDesc cb = cbs[b]; w = cb.w; h = cb.h; orient = cb.orient; nb = nbps_in[b]; }
  const bool mine = p < nb;
  int wmax = mine ? w : 0, hmax = mine ? h : 0;
  for (int o = CML / 2; o >= 1; o >>= 1) {
    int a = __shfl_xor(wmax, o, 64), d = __shfl_xor(hmax, o, 64);
    wmax = a > wmax ? a : wmax; hmax = d > hmax ? d : hmax;
  }
  wmax = __builtin_amdgcn_readfirstlane(wmax); hmax = __builtin_amdgcn_readfirstlane(hmax);
  T1GlobalStore S;
  S.flags = flags + (size_t)wv * T1_FLAGS_WORDS * 64 + lane;
  S.mags = mags + (size_t)g * T1_MAGS_WORDS * 64 + lane;
  size_t job = mine ? (size_t)joboff[b] + p : 0;
  int c3[3]; long long d3[3];
  t1_cm_plane(S, s_zc, segs + job * T1_SEG_BYTES, p, w, h, orient, mine ? nb : 0, wmax, hmax, want_dist != 0, c3, d3);
  if (mine) {
    for (int k = 0; k < 3; k++) { cnt[job * 3 + k] = c3[k]; if (want_dist) dist[job * 3 + k] = d3[k]; }
  }
}

template <int T1_MQ_LANES>
__global__ void __launch_bounds__(T1_MQ_LANES) k_t1_mq(const CbDesc* cbs, int ncb, const int* nbps_in, const int* joboff,
                                                        const int* order, int nslots, const uint8_t* segs, const int* cnt, const long long* dist, uint8_t* outbuf,
                                                        int outcap, CbOut* res, int* pass_len, long long* pass_dist) {
  __shared__ uint32_t s_cx[19 * T1_MQ_LANES];
  __shared__ uint32_t s_mq[64];
  __shared__ uint64_t s_mq2[94];
  // order[]: the host's slot table (block index, or ncb for an idle lane)
  int lane = threadIdx.x, slot = blockIdx.x * T1_MQ_LANES + lane, b = slot < nslots ? order[slot] : ncb;
  for (int i = lane; i < 47; i += T1_MQ_LANES) s_mq[i] = c_mqtab[i];
  for (int i = lane; i < 94; i += T1_MQ_LANES) s_mq2[i] = mq2_pair(c_mqtab, i);
  __syncthreads();
  T1Lane t;
  t.cx = s_cx + lane; t.cs = T1_MQ_LANES; t.mqtab = s_mq; t.mq2 = s_mq2; t.zclut = nullptr;
  t.w = t.h = t.orient = 0; t.nbps = 0; t.out = outbuf; t.cap = 0;
  size_t job0 = 0;
  if (b < ncb) { t.nbps = nbps_in[b] > 15 ? 15 : nbps_in[b]; t.out = outbuf + (size_t)b * outcap; t.cap = outcap; job0 = (size_t)joboff[b]; }
  int top = t.nbps - 1;
  for (int o = T1_MQ_LANES / 2; o >= 1; o >>= 1) { int a = __shfl_xor(top, o, 64); top = a > top ? a : top; }
  top = __builtin_amdgcn_readfirstlane(top);
  int bb = b < ncb ? b : 0;
  int* pl = pass_len + (size_t)bb * T1_MAXPASS;
  long long* pd = pass_dist ? pass_dist + (size_t)bb * T1_MAXPASS : nullptr;
  mq2_lane_init(t);
  int np = 0;
  for (int p = top; p >= 0; p--) {
    const bool active = p < t.nbps, first = p == t.nbps - 1;
    const size_t job = job0 + (active ? p : 0);
    for (int pass = 0; pass < 3; pass++) {
      if (pass < 2 && p == top) continue;
      const bool lane_pass = active && (pass == 2 || !first);
      const int n = lane_pass ? cnt[job * 3 + pass] : 0;
      int nmax = n;
      for (int o = T1_MQ_LANES / 2; o >= 1; o >>= 1) { int a = __shfl_xor(nmax, o, 64); nmax = a > nmax ? a : nmax; }
      nmax = __builtin_amdgcn_readfirstlane(nmax);
      const uint4* sg = (const uint4*)(segs + job * T1_SEG_BYTES + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2)));
      uint4 cur = make_uint4(0, 0, 0, 0);
      if (n > 0) cur = sg[0];
      for (int i = 0; i < nmax; i += 16) {
        // `cur` must have LANDED before the next chunk's load is issued: otherwise hipcc, which cannot count the
        // divergent stores in between, waits vmcnt(0) at the first use of `cur` and so also for the prefetch.
        asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w)::"memory");
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (i + 16 < n) nxt = sg[(i >> 4) + 1];  // next 16 decisions in flight while these are coded
        mq2_chunk16(t, cur.x, cur.y, cur.z, cur.w, n - i);
        cur = nxt;
      }
      t.nsym += (uint32_t)n;
      if (lane_pass && np < T1_MAXPASS) { pl[np] = t.n + 4; if (pd) pd[np] = dist[job * 3 + pass]; np++; }
    }
  }
  T1Result r;
  r.pass_len = pl; r.pass_dist = pd;
  t1_lane_finish<true>(t, np, r);
  if (b < ncb) {
    CbOut o;
    o.len = r.len; o.npasses = r.npasses; o.nbps = r.nbps; o.overflow = r.overflow; o.nsym = r.nsym;
    res[b] = o;
  }
}

void launch_t1_split(hipStream_t st, const int* waves, int nwaves, const CbDesc* cbs, int ncb, const int* nbps, const int* joboff,
                     const int* order, int nslots, const uint32_t* mags, uint32_t* flags, uint8_t* segs, int* cnt, long long* dist, uint8_t* outbuf, int outcap,
                     CbOut* res, int* pass_len, long long* pass_dist, int want_dist, int stage) {
  if (stage == 0) {
    hipLaunchKernelGGL(k_t1_flags_init, dim3((T1_FLAGS_WORDS * 64 + 255) / 256, nwaves), dim3(256), 0, st, waves, cbs, ncb, nbps, mags, flags);
  } else if (stage == 1) {
    static const int cl = [] { const char* e = getenv("QSVC_CM_LANES"); int v = e ? atoi(e) : 64; return (v == 16 || v == 32) ? v : 64; }();
    if (cl == 16) hipLaunchKernelGGL(k_t1_cm<16>, dim3(nwaves, 4), dim3(16), 0, st, waves, cbs, ncb, nbps, joboff, mags, flags, segs, cnt, dist, want_dist);
    else if (cl == 32) hipLaunchKernelGGL(k_t1_cm<32>, dim3(nwaves, 2), dim3(32), 0, st, waves, cbs, ncb, nbps, joboff, mags, flags, segs, cnt, dist, want_dist);
    else hipLaunchKernelGGL(k_t1_cm<64>, dim3(nwaves, 1), dim3(64), 0, st, waves, cbs, ncb, nbps, joboff, mags, flags, segs, cnt, dist, want_dist);
  } else {
    // lanes (code-blocks) per MQ wave: 16 gives the shortest chain for one sequence alone, 64 the best throughput when several
    // encodes share the GPU (fewer issue slots per decision).  Read at every launch so a host can switch between the two.
    const int ml = [] { const char* e = getenv("QSVC_MQ_LANES"); int v = e ? atoi(e) : 16; return (v == 4 || v == 8 || v == 32 || v == 64) ? v : 16; }();
    long long* pdp = want_dist ? pass_dist : nullptr;
#define MQ_LAUNCH(N) hipLaunchKernelGGL(k_t1_mq<N>, dim3((nslots + N - 1) / N), dim3(N), 0, st, cbs, ncb, nbps, joboff, order, nslots, segs, cnt, dist, outbuf, outcap, res, pass_len, pdp)
    switch (ml) { case 4: MQ_LAUNCH(4); break; case 8: MQ_LAUNCH(8); break; case 32: MQ_LAUNCH(32); break; case 64: MQ_LAUNCH(64); break; default: MQ_LAUNCH(16); }
#undef MQ_LAUNCH
  }
}

// ---------------------------------------------------------------------------------------------
// Sample-parallel context modelling (t1_cm2.h): ONE WAVE per (code-block, bit-plane) job.  The block's magnitudes are read
// once (block-contiguous layout, 512-byte rows), the per-column state words live in LDS, the decisions of a pass are laid
// out in LDS in scan order by a prefix sum over the lanes' counts and leave as 16-byte stores.  No flag words in HBM.
template <bool REV>
__global__ void __launch_bounds__(256) k_t1_pack_blk(const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coefv,
                                                      unsigned long long* mags, int* nbps_out, int* err) {
  int b = blockIdx.x;
  if (b >= ncb) return;
  CbDesc cb = cbs[b];
  PlaneDesc pd = planes[cb.plane];
  unsigned long long* mg = mags + (size_t)b * 1024;
  __shared__ int smax[256];
  int x = threadIdx.x & 63, sq = threadIdx.x >> 6;
  int mymax = 0;
  for (int s = sq; s < 16; s += 4) {
    uint64_t word = 0;
    if (x < cb.w) {
      for (int k = 0; k < 4; k++) {
        int y = 4 * s + k;
        if (y >= cb.h) break;
        size_t i = pd.off + (size_t)(cb.y0 + y) * pd.W + cb.x0 + x;
        int mag, neg;
        if (REV) {
          int v = ((const int32_t*)coefv)[i];
          neg = v < 0; mag = neg ? -v : v;
        } else {
          float v = ((const float*)coefv)[i];
          mag = (int)floorf(__fdiv_rn(fabsf(v), cb.delta));  // dead-zone quantiser, T.800 E.1
          neg = v < 0.0f && mag;
        }
        mymax = mag > mymax ? mag : mymax;
        word |= (uint64_t)((mag & 0x7FFF) | (neg << 15)) << (16 * k);
      }
    }
    mg[s * 64 + x] = word;
  }
  smax[threadIdx.x] = mymax;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) smax[threadIdx.x] = smax[threadIdx.x] > smax[threadIdx.x + o] ? smax[threadIdx.x] : smax[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int m = smax[0], nb = 0;
    while (nb < 31 && (m >> nb)) nb++;
    nbps_out[b] = nb > 15 ? 15 : nb;  // the error flag stops the host before any Tier-1 launch
    if (nb > cb.Mb || nb > 15) atomicOr(err, 1);
  }
}
void launch_t1_pack_blk(hipStream_t st, const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coef,
                        unsigned long long* mags, int* nbps, int* err, bool reversible) {
  if (reversible) hipLaunchKernelGGL(k_t1_pack_blk<true>, dim3(ncb), dim3(256), 0, st, cbs, ncb, planes, coef, mags, nbps, err);
  else hipLaunchKernelGGL(k_t1_pack_blk<false>, dim3(ncb), dim3(256), 0, st, cbs, ncb, planes, coef, mags, nbps, err);
}

#define CM2_OUT_BYTES 10240  // the largest pass segment (clean-up: 10 decisions per stripe column at most)
#ifndef CM2_JOBS
#define CM2_JOBS 1           // jobs (waves) per workgroup (<= 4): their membership recurrences share wave 0, 16 lanes each.
#endif                       // Measured: 1 -> 3.8 ms, 4 -> 4.8 ms per 17-picture step (the other waves idle through the chain)
__global__ void __launch_bounds__(64 * CM2_JOBS) k_t1_cm2(const int* jobs, int njobs, const CbDesc* cbs, const int* nbps_in,
                                                           const unsigned long long* mags, uint8_t* segs, int* cnt,
                                                           long long* dist, int want_dist) {
  __shared__ uint32_t s_cwq[CM2_JOBS][18][66];
  __shared__ uint32_t s_sp[64];
  __shared__ uint8_t s_zcq[CM2_JOBS][512];
  __shared__ __attribute__((aligned(16))) uint8_t s_outq[CM2_JOBS][CM2_OUT_BYTES];
  __shared__ int s_geo[CM2_JOBS][2];
  __shared__ uint64_t s_rowM[CM2_JOBS][64];
  const int jq = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int job = blockIdx.x * CM2_JOBS + jq;
  const bool live = job < njobs;
  uint32_t (*s_cw)[66] = s_cwq[jq];
  uint8_t* s_zc = s_zcq[jq];
  uint8_t* s_out = s_outq[jq];
  int b = 0, p = 0, w = 0, h = 0, nstr = 0, nb = 0, orient = 0;
  if (live) {
    const int jb = jobs[job];
    b = jb >> 4; p = jb & 15;
    const CbDesc cb = cbs[b];
    w = cb.w; h = cb.h; nstr = (h + 3) >> 2; nb = nbps_in[b]; orient = cb.orient;
  }
  const unsigned long long* mg = mags + (size_t)b * 1024;
  // tables + column words (lane = column x, all stripes: 512-byte rows of magnitudes)
  if (jq == 0) s_sp[lane] = CM2_SPREAD_INIT((uint32_t)lane);
  if (lane == 0) { s_geo[jq][0] = w; s_geo[jq][1] = nstr; }
  {
    const int oc = orient == 1 ? 1 : (orient == 3 ? 3 : 0);
    for (int i = lane; i < 512; i += 64) s_zc[i] = (uint8_t)t1_zc_from_nb(i, oc);
  }
  for (int i = lane; i < 18 * 66; i += 64) (&s_cw[0][0])[i] = 0;
  __syncthreads();
  // column words (lane = column x); at the same time the rows' sig0 / bit masks are collected by ballots: afterwards lane y
  // holds row y (bit x = column x)
  uint64_t Zr = 0, Br = 0;
  uint32_t cwreg[16];
#pragma unroll
  for (int s = 0; s < 16; s++) {
    uint32_t c = 0;
    if (s < nstr) {  // uniform
      const int rows = h - 4 * s > 4 ? 4 : h - 4 * s;
      if (lane < w) c = cm2_colword(mg[s * 64 + lane], p, rows, true);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint64_t zb = __ballot((c >> (CW_SIG0 + k)) & 1), bb = __ballot((c >> (CW_BIT + k)) & 1);
        if (lane == 4 * s + k) { Zr = zb; Br = bb; }
      }
    }
    cwreg[s] = c;
  }
  // membership: fixpoint of cm2_row_update over the 64 rows (t1_cm2.h)
  {
    const uint64_t wmask = w == 64 ? ~0ull : ((1ull << w) - 1);
    const uint64_t cand = (lane < h ? wmask : 0ull) & ~Zr;
    auto up = [&](uint64_t v) {  // the value of lane - 1 (0 for lane 0)
      uint32_t lo = __shfl_up((uint32_t)v, 1, 64), hi = __shfl_up((uint32_t)(v >> 32), 1, 64);
      return lane ? ((uint64_t)lo | ((uint64_t)hi << 32)) : 0ull;
    };
    auto dn = [&](uint64_t v) {  // the value of lane + 1 (0 for lane 63)
      uint32_t lo = __shfl_down((uint32_t)v, 1, 64), hi = __shfl_down((uint32_t)(v >> 32), 1, 64);
      return lane < 63 ? ((uint64_t)lo | ((uint64_t)hi << 32)) : 0ull;
    };
    const uint64_t seed = cm2_row_seed(Zr, up(Zr), dn(Zr));
    uint64_t M = 0;
    for (;;) {
      const uint64_t S1 = M & Br;
      const uint64_t Mn = cm2_row_update(cand, Br, seed, up(S1), dn(S1), lane & 3);
      const bool changed = Mn != M;
      M = Mn;
      if (!__any(changed)) break;
    }
    s_rowM[jq][lane] = M;
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 16; s++) {
    if (s < nstr && lane < w) {
      const uint32_t c = cwreg[s];
      uint32_t M4 = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) M4 |= (uint32_t)((s_rowM[jq][4 * s + k] >> lane) & 1) << k;
      const uint32_t S1 = M4 & (c >> CW_BIT) & 0xF;
      const uint32_t S2 = (c >> CW_VAL) & ~(c >> CW_SIG0) & ~M4 & (c >> CW_BIT) & 0xF;
      s_cw[s + 1][lane + 1] = c | (M4 << CW_M) | (S1 << CW_S1) | (S2 << CW_S2);
    }
  }
  __syncthreads();
  if (!live) return;  // no workgroup-wide barrier below this line
  Cm2Ctx cx;
  cx.zc = s_zc; cx.sp = s_sp; cx.p = p; cx.want_dist = want_dist != 0;
  const bool first = p == nb - 1;
  const int pass0 = first ? 2 : 0;
  // count the three passes' decisions of this lane's 16 stripe columns (one neighbourhood assembly per column)
  int n3[3] = {0, 0, 0};
  for (int i = lane * 16; i < lane * 16 + 16; i++) {
    const int s = i >> 6, x = i & 63;
    if (s >= nstr || x >= w) continue;
    Cm2Col c;
    cm2_gather(c, &s_cw[s][x], &s_cw[s + 1][x], &s_cw[s + 2][x]);
    int n[3];
    cm2_counts(c, n);
    if (!first) { n3[0] += n[0]; n3[1] += n[1]; }
    n3[2] += n[2];
  }
  int off3[3], tot3[3];
  for (int pass = 0; pass < 3; pass++) {
    int incl = n3[pass];
    for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    off3[pass] = incl - n3[pass];
    tot3[pass] = __shfl(incl, 63, 64);
  }
  uint8_t* seg = segs + (size_t)job * T1_SEG_BYTES;
  long long d3[3] = {0, 0, 0};
  // 16-byte aligned bases of the three streams inside s_out when all of them fit at once
  const int base1 = (tot3[0] + 15) & ~15, base2 = base1 + ((tot3[1] + 15) & ~15);
  if (base2 + tot3[2] <= CM2_OUT_BYTES) {
    Cm2ByteSink B0, B1, B2;
    B0.out = s_out + off3[0]; B0.n = 0;
    B1.out = s_out + base1 + off3[1]; B1.n = 0;
    B2.out = s_out + base2 + off3[2]; B2.n = 0;
    for (int i = lane * 16; i < lane * 16 + 16; i++) {
      const int s = i >> 6, x = i & 63;
      if (s >= nstr || x >= w) continue;
      Cm2Col c;
      cm2_gather(c, &s_cw[s][x], &s_cw[s + 1][x], &s_cw[s + 2][x]);
      const uint64_t mag = want_dist ? mg[i] : 0ull;
      if (!first) { cm2_pass(B0, cx, 0, c, mag, d3[0]); cm2_pass(B1, cx, 1, c, mag, d3[1]); }
      cm2_pass(B2, cx, 2, c, mag, d3[2]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int pass = pass0; pass < 3; pass++) {
      const uint8_t* src = s_out + (pass == 0 ? 0 : (pass == 1 ? base1 : base2));
      uint8_t* dst = seg + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      for (int o = lane * 16; o < tot3[pass]; o += 64 * 16) *(uint4*)(dst + o) = *(const uint4*)(src + o);
    }
  } else {  // a dense plane: one pass at a time through the buffer
    for (int pass = pass0; pass < 3; pass++) {
      Cm2ByteSink B;
      B.out = s_out + off3[pass]; B.n = 0;
      for (int i = lane * 16; i < lane * 16 + 16; i++) {
        const int s = i >> 6, x = i & 63;
        if (s < nstr && x < w)
          cm2_column(B, cx, pass, &s_cw[s][x], &s_cw[s + 1][x], &s_cw[s + 2][x], want_dist ? mg[i] : 0ull, d3[pass]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
      uint8_t* dst = seg + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      for (int o = lane * 16; o < tot3[pass]; o += 64 * 16) *(uint4*)(dst + o) = *(const uint4*)(s_out + o);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    }
  }
  for (int pass = 0; pass < 3; pass++) {
    if (want_dist) {
      long long dd = d3[pass];
      for (int o = 32; o >= 1; o >>= 1) dd += __shfl_xor(dd, o, 64);
      if (lane == 0) dist[(size_t)job * 3 + pass] = dd;
    }
    if (lane == 0) cnt[(size_t)job * 3 + pass] = tot3[pass];
  }
}
void launch_t1_cm2(hipStream_t st, const int* jobs, int njobs, const CbDesc* cbs, const int* nbps, const unsigned long long* mags,
                   uint8_t* segs, int* cnt, long long* dist, int want_dist) {
  if (njobs > 0) hipLaunchKernelGGL(k_t1_cm2, dim3((njobs + CM2_JOBS - 1) / CM2_JOBS), dim3(64 * CM2_JOBS), 0, st, jobs, njobs, cbs, nbps, mags, segs, cnt, dist, want_dist);
}

// gather: segment of block b (bytes 1..len of its lane buffer) -> dst + offs[b]
__global__ void __launch_bounds__(256) k_gather(const uint8_t* outbuf, int outcap, const unsigned long long* offs, const int* lens,
                                                 int ncb, uint8_t* dst) {
  int b = blockIdx.x;
  if (b >= ncb) return;
  const uint8_t* s = outbuf + (size_t)b * outcap + 1;
  uint8_t* d = dst + offs[b];
  for (int i = threadIdx.x; i < lens[b]; i += 256) d[i] = s[i];
}
void launch_gather(hipStream_t st, const uint8_t* outbuf, int outcap, const unsigned long long* offs, const int* lens, int ncb,
                   uint8_t* dst) {
  hipLaunchKernelGGL(k_gather, dim3(ncb), dim3(256), 0, st, outbuf, outcap, offs, lens, ncb, dst);
}

}  // namespace qsvc
