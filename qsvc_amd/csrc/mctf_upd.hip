// qsvc_amd/csrc/mctf_upd.hip -- the update step of a whole temporal level (update.cpp:85-146, 472-503, 632-643) as an ORDERED
// GATHER BY RUNS OF 8 TARGETS, the form used when the block size is a power of two >= 8, the picture width a multiple of 8 and
// the block grid tiles the picture (everything else: the tile kernels of mctf_kernels.hip).
//
// The reference scatters: every sample of every block of a B picture adds residue * update_factor into the even picture at its
// displaced position, with a clamp after every addition, in raster order of blocks (H2: order matters where footprints overlap).
// Here a lane owns 8 neighbouring targets of one row and walks the few blocks whose displaced footprint can meet them, in that
// same raster order.  A block's footprint meets the run in an interval [lo, hi) of it, and the 8 source samples that land on the
// run are ONE unaligned 8-byte load of the B picture's row (ty - dy) at column (tx - dx), masked to the interval.  As long as the
// intervals of different blocks do not overlap -- everywhere except at motion discontinuities -- each target has at most one
// source per side and the masked loads are simply OR-ed together (a zero residue changes nothing: clamp(v + 0) = v) and applied
// once with update.cpp:93-115's float operations; a lane that sees two blocks on one target replays its blocks one by one.
// Targets on the picture's border collect every source that clip() folds onto them: the border kernel (luma) and the border
// pass of the chroma tile evaluate those cells with the ordered gather of mctf_core.h (update_gather) from the input picture.
// Chroma: targets are the x2-interpolated planes (computed in the tile from the stored half-resolution plane), sources the
// half-resolution residue in the top-left quadrant (Q3), and the stored result is the LL band of the updated tile.
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"
#include "mctf_runs.h"

namespace qsvc {

struct UpdSrc {
  const uint8_t* high;   // plane of residue + 128
  int stride, Hs, Ws;    // its row stride and the area that carries residues
  const int16_t* mvx;
  const int16_t* mvy;
  int BX, lbs, nby, nbx;  // field stride, log2(block size), blocks that can carry a residue
};
__device__ __forceinline__ uint64_t run_mask(int lo, int hi) { return (~0ull << (8 * lo)) & (~0ull >> (64 - 8 * hi)); }
// The reference's default update factor 0.25 without floating point: v + 0.25 r is exact in float (two fraction bits), the clamp
// and the truncating conversion of update.cpp:93-115 make it min(max(4 v + r, 0) >> 2, 255) -- on packed 16-bit lanes, four
// instead of eleven instructions per sample.
__device__ __forceinline__ uint64_t upd_apply8_quarter(uint64_t v, uint64_t r) {
  uint32_t out[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const uint32_t vw = (uint32_t)(v >> (32 * h)), uw = (uint32_t)(r >> (32 * h)) ^ 0x80808080u;   // residue + 128, unsigned bytes
    uint32_t t[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t sel = q ? 0x0c030c02u : 0x0c010c00u;   // bytes 2 q, 2 q + 1 -> the two 16-bit lanes
      const s16x2 V = S2(__builtin_amdgcn_perm(0u, vw, sel)), R = S2(pk_sub(__builtin_amdgcn_perm(0u, uw, sel), 0x00800080u));
      const s16x2 z = {0, 0}, m = {255, 255}, four = {4, 4};
      s16x2 T = V * four + R;
      T = __builtin_elementwise_max(T, z) >> 2;
      t[q] = U(__builtin_elementwise_min(T, m));
    }
    out[h] = __builtin_amdgcn_perm(t[1], t[0], 0x06040200u);
  }
  return (uint64_t)out[0] | (uint64_t)out[1] << 32;
}
// the 8 targets' values after one source run (signed residue bytes, 0 = no source)
__device__ __forceinline__ uint64_t upd_apply8(uint64_t v, uint64_t r, float uf) {
  if (uf == 0.25f) return upd_apply8_quarter(v, r);   // (uniform)
  uint64_t o = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int vk = (int)((v >> (8 * k)) & 255), rk = (int)(int8_t)(r >> (8 * k));
    o |= (uint64_t)upd_one(vk, rk, uf) << (8 * k);
  }
  return o;
}
// One side's contribution to targets (ty, tx0 .. tx0+7).  SEQ = false: the sources OR-ed into one run (ovl reports a target with
// two sources); SEQ = true: v updated block after block.
template <bool SEQ>
__device__ __forceinline__ uint64_t upd_side(const UpdSrc& s, int ty, int tx0, int R, float uf, uint64_t v, bool& ovl) {
  const int bs = 1 << s.lbs;
  const int by_lo = imax((ty - R) >> s.lbs, 0), by_hi = imin((ty + R) >> s.lbs, s.nby - 1);
  const int bx_lo = imax((tx0 - R) >> s.lbs, 0), bx_hi = imin((tx0 + 7 + R) >> s.lbs, s.nbx - 1);
  uint64_t acc = 0, accm = 0, clash = 0;
  for (int by = by_lo; by <= by_hi; by++)
    for (int bx = bx_lo; bx <= bx_hi; bx++) {
      const int o = by * s.BX + bx;
      const int sy = ty - s.mvy[o], sx0 = tx0 - s.mvx[o], b0 = bx << s.lbs;
      const int lo = imax(0, b0 - sx0), hi = imin(8, imin(b0 + bs, s.Ws) - sx0);
      if ((sy >> s.lbs) == by && sy < s.Hs && lo < hi) {
        const uint64_t m = run_mask(lo, hi);
        const uint64_t r = (ld8(s.high + (size_t)sy * s.stride + sx0) ^ 0x8080808080808080ull) & m;
        if (SEQ) v = upd_apply8(v, r, uf);
        else { clash |= accm & m; acc |= r; accm |= m; }
      }
    }
  if (SEQ) return v;
  ovl = clash != 0;
  return acc;
}
// The same collection for NS sides at once with the memory requests batched: per block row, the NX candidates' vectors of every
// side are loaded together, then their NX source runs (unconditionally, at a harmless address when the candidate misses), then
// everything is masked and OR-ed -- two round trips to memory per block row instead of two per candidate.
template <int NX, int NS>
__device__ __forceinline__ void upd_collect(const UpdSrc* s, int ty, int tx0, int R, uint64_t* acc, bool* ovl) {
  const int lbs = s[0].lbs, bs = 1 << lbs;
  const int by_lo = imax((ty - R) >> lbs, 0), by_hi = imin((ty + R) >> lbs, s[0].nby - 1);
  const int bx_lo = imax((tx0 - R) >> lbs, 0), bx_hi = imin((tx0 + 7 + R) >> lbs, s[0].nbx - 1);
  uint64_t accm[NS], clash[NS];
#pragma unroll
  for (int q = 0; q < NS; q++) { acc[q] = 0; accm[q] = 0; clash[q] = 0; }
  for (int by = by_lo; by <= by_hi; by++) {
    int dy[NS][NX], dx[NS][NX];
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
      for (int b = 0; b < NX; b++) {
        const int o = by * s[q].BX + imin(bx_lo + b, s[q].nbx - 1);
        dy[q][b] = s[q].mvy[o]; dx[q][b] = s[q].mvx[o];
      }
    uint64_t r[NS][NX], m[NS][NX];
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
      for (int b = 0; b < NX; b++) {
        const int sy = ty - dy[q][b], sx0 = tx0 - dx[q][b], b0 = (bx_lo + b) << lbs;
        const int lo = imax(0, b0 - sx0), hi = imin(8, imin(b0 + bs, s[q].Ws) - sx0);
        const bool ok = bx_lo + b <= bx_hi && (sy >> lbs) == by && sy < s[q].Hs && lo < hi;
        m[q][b] = ok ? run_mask(lo, hi) : 0ull;
        r[q][b] = ld8(s[q].high + (ok ? (size_t)sy * s[q].stride + sx0 : (size_t)0));
      }
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
      for (int b = 0; b < NX; b++) {
        clash[q] |= accm[q] & m[q][b];
        acc[q] |= (r[q][b] ^ 0x8080808080808080ull) & m[q][b];
        accm[q] |= m[q][b];
      }
  }
#pragma unroll
  for (int q = 0; q < NS; q++) ovl[q] = clash[q] != 0;
}
// both sides of the update of targets (ty, tx0 .. tx0+7), in the reference's order: side 0, then side 1 (absent sides: high == nullptr).
// NX = the most block columns a run's sources can come from (0: unknown, candidate by candidate)
template <int NX>
__device__ __forceinline__ uint64_t upd_run(const UpdSrc* s, int ty, int tx0, int R, float uf, uint64_t v) {
  uint64_t acc[2] = {0, 0};
  bool ovl[2] = {false, false};
  const bool h0 = s[0].high != nullptr, h1 = s[1].high != nullptr;  // wave-uniform
  if (NX == 0) {
    if (h0) acc[0] = upd_side<false>(s[0], ty, tx0, R, uf, v, ovl[0]);
    if (h1) acc[1] = upd_side<false>(s[1], ty, tx0, R, uf, v, ovl[1]);
  } else if (h0 && h1) upd_collect<NX ? NX : 1, 2>(s, ty, tx0, R, acc, ovl);
  else if (h0) upd_collect<NX ? NX : 1, 1>(s, ty, tx0, R, acc, ovl);
  else if (h1) upd_collect<NX ? NX : 1, 1>(s + 1, ty, tx0, R, acc + 1, ovl + 1);
  if (h0) v = ovl[0] ? upd_side<true>(s[0], ty, tx0, R, uf, v, ovl[0]) : upd_apply8(v, acc[0], uf);
  if (h1) v = ovl[1] ? upd_side<true>(s[1], ty, tx0, R, uf, v, ovl[1]) : upd_apply8(v, acc[1], uf);
  return v;
}
// the most block columns the sources of a run of 8 targets can come from (host side)
static int upd_max_cols(int bs, int R) {
  int lbs = 0;
  while ((1 << (lbs + 1)) <= bs) lbs++;
  int best = 1;
  for (int c = 0; c < bs / 8 + 1; c++) {
    const int lo = 8 * c + 1024 - R, hi = 8 * c + 1024 + 7 + R;  // offset by a multiple of every block size: shifts of non-negative numbers
    best = imax(best, (hi >> lbs) - (lo >> lbs) + 1);
  }
  return best;
}
__device__ __forceinline__ UpdSrc upd_src_luma(const LevelGeom& g, const UpdSide& sd) {
  UpdSrc s;
  s.high = sd.high; s.stride = g.W; s.Hs = g.H; s.Ws = g.W; s.mvx = sd.mvx; s.mvy = sd.mvy;
  s.BX = g.BX; s.lbs = 31 - __builtin_clz(g.bs); s.nby = g.BY; s.nbx = g.BX;
  return s;
}

// ---------------------------------------------------------------------------------------------
// luma: a lane per run of 8 targets.  Cells on the picture's border belong to the border role (below): this role never
// stores them, so both can run in one launch.
template <int NX>
__device__ __forceinline__ void upd_luma_body(const UpdArgs& a, const int bx_, const int by_, const int j, const int tx_, const int ty_) {
  const LevelGeom& g = a.g;
  const int tx0 = (bx_ * 64 + tx_) * 8, ty = by_ * 4 + ty_;
  if (tx0 >= g.W || ty >= g.H) return;
  UpdSide sd[2];
  upd_sides(a, j, sd);
  uint64_t v = ld8(frame_y(a.frames, g, a.src_mul * j) + (size_t)ty * g.W + tx0);
  const UpdSrc s[2] = {upd_src_luma(g, sd[0]), upd_src_luma(g, sd[1])};
  const bool any = sd[0].high || sd[1].high;
  if (any) v = upd_run<NX>(s, ty, tx0, a.reach, a.uf, v);
  uint8_t* d = a.low + (size_t)(a.dst_mul * j) * g.F + (size_t)ty * g.W + tx0;
  const bool edge = any && (ty == 0 || ty == g.H - 1 || tx0 == 0 || tx0 + 8 == g.W);
  if (!edge) *(uint2*)d = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
  else if (ty != 0 && ty != g.H - 1) {  // the first / last run of an inner row: everything but the border cell
    for (int k = 0; k < 8; k++)
      if (tx0 + k != 0 && tx0 + k != g.W - 1) d[k] = (uint8_t)(v >> (8 * k));
  }
}
// the 2 W + 2 (H - 2) cells of the picture's border: everything clip() folds onto them, in order (update_gather)
__device__ __forceinline__ void upd_border_body(const UpdArgs& a, const int k, const int j) {
  const LevelGeom& g = a.g;
  int y, x;
  if (k < g.W) { y = 0; x = k; }
  else if (k < 2 * g.W) { y = g.H - 1; x = k - g.W; }
  else if (k < 2 * g.W + (g.H - 2)) { y = 1 + k - 2 * g.W; x = 0; }
  else if (k < 2 * g.W + 2 * (g.H - 2)) { y = 1 + k - 2 * g.W - (g.H - 2); x = g.W - 1; }
  else return;
  UpdSide sd[2];
  upd_sides(a, j, sd);
  if (!sd[0].high && !sd[1].high) return;  // the luma role has copied the cell
  int v = frame_y(a.frames, g, a.src_mul * j)[(size_t)y * g.W + x];
  for (int q = 0; q < 2; q++)
    if (sd[q].high) {
      const uint8_t* h = sd[q].high;
      auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * g.W + sx] - 128; };
      v = update_gather(v, R, sd[q].mvx, sd[q].mvy, g.BY, g.BX, g.bs, g.H, g.W, g.H, g.W, a.reach, y, x, a.uf);
    }
  a.low[(size_t)(a.dst_mul * j) * g.F + (size_t)y * g.W + x] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------
// chroma: one 64 x 14 tile of one stored component per workgroup.  Tiles no update can reach (Q3: sources exist in the top-left
// quadrant only) and pictures without a contributing B picture keep the stored chroma: analysis(synthesis(u)) == u exactly.
template <int NX>
__device__ __forceinline__ void upd_chroma_body(const UpdArgs& a, uint8_t* smem, const int bx_, const int by_, const int bz_, const int tid) {
  uint8_t (*sP)[PCT_FW] = (uint8_t (*)[PCT_FW])smem;
  uint32_t (*sL)[PCT_W / 2] = (uint32_t (*)[PCT_W / 2])(smem + PCT_SP_BYTES);
  const LevelGeom& g = a.g;
  const int j = bz_ >> 1, c = 1 + (bz_ & 1);
  const int i0 = bx_ * PCT_W, j0 = by_ * PCT_H;
  const int cw = g.W >> 1, chh = g.H >> 1;
  const int xb = 2 * i0 - 8, yb = 2 * j0 - 2;
  UpdSide sd[2];
  upd_sides(a, j, sd);
  const size_t coff = (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh;
  const uint8_t* u = frame_c(a.frames, g, a.src_mul * j, c);
  uint8_t* dst = a.low + (size_t)(a.dst_mul * j) * g.F + coff;
  if ((!sd[0].high && !sd[1].high) || yb >= chh + a.reach || 2 * i0 - 2 >= cw + a.reach) {  // workgroup-uniform: a plain copy
    if (tid < PCT_H * (PCT_W / 4)) {
      const int cj = j0 + tid / (PCT_W / 4), ci = i0 + 4 * (tid % (PCT_W / 4));
      if (cj < chh && ci < cw) *(uint32_t*)(dst + (size_t)cj * cw + ci) = *(const uint32_t*)(u + (size_t)cj * cw + ci);
    }
    return;
  }
  UpdSrc s[2];
  const int lbs = 31 - __builtin_clz(g.bs);
  for (int q = 0; q < 2; q++) {
    s[q].high = sd[q].high ? sd[q].high + coff : nullptr; s[q].stride = cw; s[q].Hs = chh; s[q].Ws = cw;
    s[q].mvx = sd[q].mvx; s[q].mvy = sd[q].mvy; s[q].BX = g.BX; s[q].lbs = lbs;
    s[q].nby = imin(g.BY, ((chh - 1) >> lbs) + 1); s[q].nbx = imin(g.BX, ((cw - 1) >> lbs) + 1);
  }
  // A. the footprint: interpolated stored chroma (update.cpp:472-503), then both sides' updates
  for (int k = tid; k < PCT_FH * (PCT_FW / 8); k += PCT_T) {
    const int ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8), yy = yb + ry, xx = xb + 8 * run;
    uint64_t v = 0;
    if ((unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) {
      v = upd_run<NX>(s, yy, xx, a.reach, a.uf, chroma_run8(u, g, yy, xx));
    }
    *(uint64_t*)&sP[ry][8 * run] = v;
  }
  __syncthreads();
  // cells on the picture's border: the ordered gather over everything that clips onto them
  {
    auto border_cell = [&](int y, int x) {
      int v = chroma_up(u, y, x, g.H, g.W);
      for (int q = 0; q < 2; q++)
        if (s[q].high) {
          const uint8_t* h = s[q].high;
          auto R = [&](int sy, int sx) { return (int)h[(size_t)sy * cw + sx] - 128; };
          v = update_gather(v, R, s[q].mvx, s[q].mvy, g.BY, g.BX, g.bs, g.H, g.W, chh, cw, a.reach, y, x, a.uf);
        }
      sP[y - yb][x - xb] = (uint8_t)v;
    };
    for (int e = 0; e < 2; e++) {  // picture rows 0 and H-1
      const int y = e ? g.H - 1 : 0, x = xb + tid;
      if (y >= yb && y < yb + PCT_FH && tid < PCT_FW && x >= 0 && x < g.W) border_cell(y, x);
    }
    for (int e = 0; e < 2; e++) {  // picture columns 0 and W-1 (corners are done)
      const int x = e ? g.W - 1 : 0, y = yb + tid;
      if (x >= xb && x < xb + PCT_FW && tid < PCT_FH && y > 0 && y < g.H - 1) border_cell(y, x);
    }
  }
  __syncthreads();
  chroma_tile_lows(sP, sL, g, i0, xb, yb, tid);
  if (tid < PCT_H * (PCT_W / 4)) {
    const int jr = tid / (PCT_W / 4), gq = tid % (PCT_W / 4), cj = j0 + jr, ci = i0 + 4 * gq;
    if (cj < chh && ci < cw) {
      uint32_t ll[2];
      chroma_tile_ll(sL, jr, gq, ll);
      *(uint32_t*)(dst + (size_t)cj * cw + ci) = __builtin_amdgcn_perm(ll[1], ll[0], 0x06040200u);  // Q4 wrap
    }
  }
}

// One launch per level: the chroma tiles first (the longest workgroups), then the luma strips, then the border cells.
template <int NX>
__global__ void __launch_bounds__(256) k_upd_level(UpdArgs a, int ncx, int ncy, int nlx, int nly) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[PCT_SMEM_BYTES];
  const int np = a.g.nB + 1;
  int b = blockIdx.x;
  const int nchroma = ncx * ncy * np * 2, nluma = nlx * nly * np;
  if (b < nchroma) { upd_chroma_body<NX>(a, smem, b % ncx, (b / ncx) % ncy, b / (ncx * ncy), threadIdx.x); return; }
  b -= nchroma;
  if (b < nluma) { upd_luma_body<NX>(a, b % nlx, (b / nlx) % nly, b / (nlx * nly), threadIdx.x & 63, threadIdx.x >> 6); return; }
  b -= nluma;
  const int nbb = (2 * a.g.W + 2 * (a.g.H - 2) + 255) / 256;
  upd_border_body(a, (b % nbb) * 256 + threadIdx.x, b / nbb);
}

bool upd_run_applies(const UpdArgs& a) {
  const LevelGeom& g = a.g;
  return g.bs >= 8 && (g.bs & (g.bs - 1)) == 0 && g.W % 8 == 0 && g.H % 2 == 0 && g.H % g.bs == 0 && g.W % g.bs == 0 && g.H >= 8 && g.W >= 16 &&
         a.reach <= 255;
}
void launch_update_run(hipStream_t st, const UpdArgs& a) {
  const LevelGeom& g = a.g;
  const int np = g.nB + 1, nx = upd_max_cols(g.bs, a.reach);
  const int nlx = (g.W + 511) / 512, nly = (g.H + 3) / 4, ncx = (g.W / 2 + PCT_W - 1) / PCT_W, ncy = (g.H / 2 + PCT_H - 1) / PCT_H;
  const int nbb = (2 * g.W + 2 * (g.H - 2) + 255) / 256;
  const dim3 grid((unsigned)(ncx * ncy * np * 2 + nlx * nly * np + nbb * np)), block(256);
  switch (nx) {
    case 2: hipLaunchKernelGGL(k_upd_level<2>, grid, block, 0, st, a, ncx, ncy, nlx, nly); break;
    case 3: hipLaunchKernelGGL(k_upd_level<3>, grid, block, 0, st, a, ncx, ncy, nlx, nly); break;
    case 5: hipLaunchKernelGGL(k_upd_level<5>, grid, block, 0, st, a, ncx, ncy, nlx, nly); break;
    default: hipLaunchKernelGGL(k_upd_level<0>, grid, block, 0, st, a, ncx, ncy, nlx, nly); break;
  }
}

}  // namespace qsvc
