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
// chroma: a workgroup produces a 64 x 16 tile of one half-resolution component.
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
  // (the odd picture's samples of phase C: requested now, needed after three barriers)
  uint32_t o_c = 0;
  {
    const int jr = tid / (PCT_W / 4), gq = tid % (PCT_W / 4), cj = j0 + jr, ci = i0 + 4 * gq;
    if (tid < PCT_H * (PCT_W / 4) && cj < chh && ci < cw) o_c = *(const uint32_t*)(frame_c(a.frames, g, 2 * i + 1, c) + (size_t)cj * cw + ci);
  }
  // A. full-resolution prediction of the footprint.  A task = one run of 8 columns x FOUR rows of one MC block (rows 4 m .. 4 m + 3 of
  // the picture: a multiple of 4 rows lies inside one block for every block size on this path), so the vectors, the address
  // arithmetic and the in-picture tests are paid once per four runs (round 4: they were two thirds of the per-run form's
  // instructions).  Four displaced full-resolution rows read three stored rows R0..R2 of each reference: with Y the first
  // displaced row, the rows are [R0, (R0+R1)>>1, R1, (R1+R2)>>1] for even Y and [(R0+R1)>>1, R1, (R1+R2)>>1, R2] for odd Y
  // (decorrelate.cpp:610-648, columns first: the vertical mean is taken on stored samples, the horizontal one on the result).
  {
    static_assert((2 * PCT_H) % 4 == 0, "the footprint starts two rows above a multiple of 4");
    constexpr int NG = (PCT_FH + 2 + 3) / 4;            // groups of 4 picture rows the footprint's rows yb .. yb + PCT_FH - 1 touch (yb = 4 m - 2)
    static_assert(NG * (PCT_FW / 8) <= PCT_T, "one task per thread");
    const int gi = tid / (PCT_FW / 8), run = tid % (PCT_FW / 8);
    const int y4 = yb - 2 + 4 * gi, xx = xb + 8 * run;   // the group's first picture row (yb - 2 is a multiple of 4)
    if (gi < NG) {
      const bool in_pic = (unsigned)y4 < (unsigned)g.H && (unsigned)xx < (unsigned)g.W;   // H % 4 == 0 is not given: rows are tested again below
      uint64_t p[4] = {0, 0, 0, 0};
      if (in_pic) {
        const int16_t* m = a.mv + (size_t)i * 4 * fld + (size_t)(y4 >> lbs) * g.BX + (xx >> lbs);
        const int mvx0 = m[0], mvy0 = m[fld], mvx1 = m[2 * fld], mvy1 = m[3 * fld];
        const int Y0 = y4 + mvy0, X0 = xx + mvx0, Y1 = y4 + mvy1, X1 = xx + mvx1;
        const bool fast = Y0 >= 0 && Y0 + 3 < g.H && X0 >= 0 && X0 + 8 < g.W && Y1 >= 0 && Y1 + 3 < g.H && X1 >= 0 && X1 + 8 < g.W && y4 + 3 < g.H;
        if (fast) {
          auto rows3 = [&](const uint8_t* u, int Y, int X, uint64_t R[3]) {
            const int ja = Y >> 1, k0 = X >> 1;
            const uint8_t* b = u + (size_t)ja * cw + k0;
            R[0] = ld8(b); R[1] = ld8(b + (size_t)(imin(ja + 1, chh - 1) - ja) * cw); R[2] = ld8(b + (size_t)(imin(ja + 2, chh - 1) - ja) * cw);
          };
          uint64_t A[3], B[3];
          rows3(u0, Y0, X0, A); rows3(u1, Y1, X1, B);
          auto four = [&](const uint64_t R[3], int Y, int X, uint64_t out[4]) {
            const uint64_t M01 = avg8(R[0], R[1]), M12 = avg8(R[1], R[2]);
            const bool oy = Y & 1, ox = X & 1;
            const uint64_t c[4] = {oy ? M01 : R[0], oy ? R[1] : M01, oy ? M12 : R[1], oy ? R[2] : M12};
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const uint32_t c0 = (uint32_t)c[r], c1 = (uint32_t)(c[r] >> 32);
              const uint32_t cs = __builtin_amdgcn_alignbyte(c1, c0, 1), mm = avg4(c0, cs);
              const uint32_t Aw = ox ? mm : c0, Bw = ox ? cs : mm;
              out[r] = (uint64_t)__builtin_amdgcn_perm(Bw, Aw, 0x05010400u) | ((uint64_t)__builtin_amdgcn_perm(Bw, Aw, 0x07030602u) << 32);
            }
          };
          uint64_t pa[4], pb[4];
          four(A, Y0, X0, pa); four(B, Y1, X1, pb);
#pragma unroll
          for (int r = 0; r < 4; r++) p[r] = avg8(pa[r], pb[r]);
        } else {
          for (int r = 0; r < 4; r++)
            if (y4 + r < g.H) p[r] = avg8(chroma_run8(u0, g, y4 + r + mvy0, X0), chroma_run8(u1, g, y4 + r + mvy1, X1));
        }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ry = y4 + r - yb;
        if ((unsigned)ry < (unsigned)PCT_FH) *(uint64_t*)&sP[ry][8 * run] = (y4 + r < g.H) ? p[r] : 0ull;
      }
    }
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
      const uint32_t o = o_c;
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
