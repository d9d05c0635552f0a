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
template <bool NEG>
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
      s16x2 T = NEG ? V * four - R : V * four + R;   // un_update: the factor is -0.25 (a - r/4 is as exact as a + r/4)
      T = __builtin_elementwise_max(T, z) >> 2;
      t[q] = U(__builtin_elementwise_min(T, m));
    }
    out[h] = __builtin_amdgcn_perm(t[1], t[0], 0x06040200u);
  }
  return (uint64_t)out[0] | (uint64_t)out[1] << 32;
}
// the 8 targets' values after one source run (signed residue bytes, 0 = no source)
__device__ __forceinline__ uint64_t upd_apply8(uint64_t v, uint64_t r, float uf) {
  if (uf == 0.25f) return upd_apply8_quarter<false>(v, r);   // (uniform)
  if (uf == -0.25f) return upd_apply8_quarter<true>(v, r);  // (the inverse step of the default factor)
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
// SCATTER form (round 4).  The ordered gather above tests every block within reach of a run of 8 targets -- (2 R / bs + 2)^2
// candidates per side, nearly all of them misses -- and that test, not the memory system, was what the update step cost (32 M
// vector wave-instructions per launch, 32 lane-operations per byte, at 10 % of the HBM peak).  Here a workgroup owns a TILE of targets and walks the sources instead: every (block,
// run of 8 columns, quad of rows) within reach of the tile is one task; its four aligned 8-byte rows of residue are OR-ed into a
// byte image of the tile in LDS at the displaced position (three aligned words per row, no return value needed) and a second
// image counts the non-zero contributions per target.  A target with at most one non-zero source per side is order-free
// (clamp(v + 0) = v), so the tile's runs are then applied exactly as before: one packed update per side from the image; a run
// whose counter image shows two sources on one target -- motion discontinuities -- replays that side block by block in the
// reference's order (upd_side<true>).  The counters are bytes: the form is used while fewer than 256 blocks lie within reach of
// a target (upd_scatter_applies); beyond that the gather above runs.
#ifndef US_W
#define US_W 128                      // luma tile: targets per row ...
#define US_H 32                       // ... and rows
#endif
#define US_RX (US_W / 8)              // runs of 8 per tile row
#define US_NK ((US_H * US_RX + 255) / 256)   // runs per thread
#define US_STR (US_W + 16)            // bytes per image row: 8 guard bytes on either side
#define UC_STR (PCT_FW + 16)          // the chroma tile's footprint likewise
#define US_SMEM (4 * US_H * US_STR)   // luma: image + counters of both sides
#define UC_SMEM (4 * PCT_FH * UC_STR)   // chroma: image + counters of both sides; the finished footprint and the row pass' low band then take their place
#define UPD_SC_SMEM (US_SMEM > UC_SMEM ? US_SMEM : UC_SMEM)
template <int STR>
__device__ __forceinline__ void upd_scatter(const UpdSrc& s, int y0, int nrows, int x0, int ncols, int R, uint8_t* acc, uint8_t* cnt, int tid) {
  const int lbs = s.lbs;
  const int by_lo = imax((y0 - R) >> lbs, 0), by_hi = imin((y0 + nrows - 1 + R) >> lbs, s.nby - 1);
  const int bx_lo = imax((x0 - R) >> lbs, 0), bx_hi = imin((x0 + ncols - 1 + R) >> lbs, s.nbx - 1);
  if (by_hi < by_lo || bx_hi < bx_lo) return;
  const int ncx = (bx_hi - bx_lo + 1) << (lbs - 3);          // runs of 8 source columns in a block row's candidates
  const int ntask = ncx * ((by_hi - by_lo + 1) << (lbs - 2));  // x quads of source rows
  const float inv = 1.0f / (float)ncx;
  for (int t = tid; t < ntask; t += 256) {
    const int qi = (int)(((float)t + 0.5f) * inv), cxi = t - qi * ncx;
    const int sy0 = (by_lo << lbs) + 4 * qi, sx0 = (bx_lo << lbs) + 8 * cxi;
    const int o = (sy0 >> lbs) * s.BX + (sx0 >> lbs);
    const int dy = s.mvy[o], dx = s.mvx[o];
    const int pos = sx0 + dx - x0 + 8, row0 = sy0 + dy - y0;   // where the run's first sample lands in the image
    if (pos <= 0 || pos >= ncols + 8 || sx0 >= s.Ws || row0 <= -4 || row0 >= nrows || sy0 >= s.Hs) continue;
    const int nv = s.Ws - sx0;
    const uint64_t smask = nv >= 8 ? ~0ull : (~0ull >> (64 - 8 * nv));
    uint64_t raw[4];
#pragma unroll
    for (int k = 0; k < 4; k++) raw[k] = ld8(s.high + (size_t)imin(sy0 + k, s.Hs - 1) * s.stride + sx0);
    const int sh = (pos & 3) * 8, wofs = pos & ~3;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int row = row0 + k;
      if ((unsigned)row >= (unsigned)nrows || sy0 + k >= s.Hs) continue;
      const uint64_t r = (raw[k] ^ 0x8080808080808080ull) & smask;
      if (!r) continue;
      const uint32_t rl = (uint32_t)r, rh = (uint32_t)(r >> 32);
      const uint32_t tl = ((((rl & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | rl) & 0x80808080u) >> 7;   // 1 per non-zero byte
      const uint32_t th = ((((rh & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | rh) & 0x80808080u) >> 7;
      const uint64_t c = (uint64_t)tl | (uint64_t)th << 32;
      const uint64_t r01 = r << sh, c01 = c << sh;
      const uint32_t r2 = (uint32_t)((r >> 32) >> (32 - sh)), c2 = (uint32_t)((c >> 32) >> (32 - sh));
      uint32_t* ap = (uint32_t*)(acc + row * STR + wofs);
      uint32_t* cp = (uint32_t*)(cnt + row * STR + wofs);
      __hip_atomic_fetch_or(ap, (uint32_t)r01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_or(ap + 1, (uint32_t)(r01 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(cp, (uint32_t)c01, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(cp + 1, (uint32_t)(c01 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (sh) {
        __hip_atomic_fetch_or(ap + 2, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(cp + 2, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
}
// one side's update of the run at (row, 8 run) of the tile from the images
template <int STR>
__device__ __forceinline__ uint64_t upd_from_image(const UpdSrc& s, const uint8_t* acc, const uint8_t* cnt, int row, int run, int ty, int tx0, int R,
                                                   float uf, uint64_t v) {
  const uint64_t a = *(const uint64_t*)(acc + row * STR + 8 + 8 * run), c = *(const uint64_t*)(cnt + row * STR + 8 + 8 * run);
  if (c & 0xFEFEFEFEFEFEFEFEull) { bool d; return upd_side<true>(s, ty, tx0, R, uf, v, d); }   // two sources on one target: in order
  return upd_apply8(v, a, uf);
}
__device__ __forceinline__ void upd_zero(uint8_t* p, int bytes, int tid) {   // bytes % 16 == 0, p 16-byte aligned
  for (int i = tid; i < bytes / 16; i += 256) ((uint4*)p)[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// chroma: one 64 x 16 tile of one stored component per workgroup.  Tiles no update can reach (Q3: sources exist in the top-left
// quadrant only) and pictures without a contributing B picture keep the stored chroma: analysis(synthesis(u)) == u exactly.
template <int NX, bool SC>
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
  if (!SC) {
    for (int k = tid; k < PCT_FH * (PCT_FW / 8); k += PCT_T) {
      const int ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8), yy = yb + ry, xx = xb + 8 * run;
      uint64_t v = 0;
      if ((unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) {
        v = upd_run<NX>(s, yy, xx, a.reach, a.uf, chroma_run8(u, g, yy, xx));
      }
      *(uint64_t*)&sP[ry][8 * run] = v;
    }
  } else {  // scatter form: both sides' images first (they fill the tile's LDS), then the finished footprint in their place
    static_assert(4 * PCT_FH * UC_STR >= PCT_SMEM_BYTES, "footprint + low band fit where the images were");
    uint8_t* acc = smem;                       // [2][PCT_FH][UC_STR]
    uint8_t* cnt = smem + 2 * PCT_FH * UC_STR;
    constexpr int NRUN = PCT_FH * (PCT_FW / 8), NIT = (NRUN + PCT_T - 1) / PCT_T;
    uint64_t v[NIT];
    ChromaLd ld[NIT];
#pragma unroll
    for (int i = 0; i < NIT; i++) {
      const int k = tid + i * PCT_T, ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8);
      ld[i] = chroma_run8_ld(u, g, yb + ry, xb + 8 * run);
    }
    upd_zero(smem, UC_SMEM, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++)
      if (s[q].high) upd_scatter<UC_STR>(s[q], yb, PCT_FH, xb, PCT_FW, a.reach, acc + q * PCT_FH * UC_STR, cnt + q * PCT_FH * UC_STR, tid);
#pragma unroll
    for (int i = 0; i < NIT; i++) {
      const int k = tid + i * PCT_T, ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8), yy = yb + ry, xx = xb + 8 * run;
      v[i] = (k < NRUN && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) ? chroma_run8_fin(ld[i], u, g, yy, xx) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NIT; i++) {
      const int k = tid + i * PCT_T, ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8), yy = yb + ry, xx = xb + 8 * run;
      if (k < NRUN && (unsigned)yy < (unsigned)g.H && (unsigned)xx < (unsigned)g.W) {
#pragma unroll
        for (int q = 0; q < 2; q++)
          if (s[q].high)
            v[i] = upd_from_image<UC_STR>(s[q], acc + q * PCT_FH * UC_STR, cnt + q * PCT_FH * UC_STR, ry, run, yy, xx, a.reach, a.uf, v[i]);
      }
    }
    __syncthreads();   // every read of the images is done: the footprint takes their place
#pragma unroll
    for (int i = 0; i < NIT; i++) {
      const int k = tid + i * PCT_T, ry = k / (PCT_FW / 8), run = k % (PCT_FW / 8);
      if (k < NRUN) *(uint64_t*)&sP[ry][8 * run] = v[i];
    }
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

__device__ __forceinline__ void upd_luma_tile(const UpdArgs& a, uint8_t* smem, const int bx_, const int by_, const int j, const int tid) {
  const LevelGeom& g = a.g;
  const int x0 = bx_ * US_W, y0 = by_ * US_H;
  const int ncols = imin(US_W, g.W - x0), nrows = imin(US_H, g.H - y0);
  UpdSide sd[2];
  upd_sides(a, j, sd);
  const uint8_t* src = frame_y(a.frames, g, a.src_mul * j);
  uint8_t* dst = a.low + (size_t)(a.dst_mul * j) * g.F;
  uint64_t v[US_NK];
  bool ok[US_NK];
#pragma unroll
  for (int k = 0; k < US_NK; k++) {
    const int i = tid + 256 * k, row = i / US_RX, run = i % US_RX;
    ok[k] = row < nrows && 8 * run < ncols;
    v[k] = ok[k] ? ld8(src + (size_t)(y0 + row) * g.W + x0 + 8 * run) : 0ull;
  }
  const bool any = sd[0].high || sd[1].high;
  if (any) {
    const UpdSrc s[2] = {upd_src_luma(g, sd[0]), upd_src_luma(g, sd[1])};
    uint8_t* acc = smem;                       // [2][US_H][US_STR]
    uint8_t* cnt = smem + 2 * US_H * US_STR;
    upd_zero(smem, US_SMEM, tid);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++)
      if (s[q].high) upd_scatter<US_STR>(s[q], y0, nrows, x0, ncols, a.reach, acc + q * US_H * US_STR, cnt + q * US_H * US_STR, tid);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < US_NK; k++) {
      if (!ok[k]) continue;
      const int i = tid + 256 * k, row = i / US_RX, run = i % US_RX;
#pragma unroll
      for (int q = 0; q < 2; q++)
        if (s[q].high)
          v[k] = upd_from_image<US_STR>(s[q], acc + q * US_H * US_STR, cnt + q * US_H * US_STR, row, run, y0 + row, x0 + 8 * run, a.reach, a.uf, v[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < US_NK; k++) {
    if (!ok[k]) continue;
    const int i = tid + 256 * k, ty = y0 + i / US_RX, tx0 = x0 + 8 * (i % US_RX);
    uint8_t* d = dst + (size_t)ty * g.W + tx0;
    const bool edge = any && (ty == 0 || ty == g.H - 1 || tx0 == 0 || tx0 + 8 == g.W);   // border cells belong to the border role
    if (!edge) *(uint2*)d = make_uint2((uint32_t)v[k], (uint32_t)(v[k] >> 32));
    else if (ty != 0 && ty != g.H - 1) {
      for (int c = 0; c < 8; c++)
        if (tx0 + c != 0 && tx0 + c != g.W - 1) d[c] = (uint8_t)(v[k] >> (8 * c));
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
  if (b < nchroma) { upd_chroma_body<NX, false>(a, smem, b % ncx, (b / ncx) % ncy, b / (ncx * ncy), threadIdx.x); return; }
  b -= nchroma;
  if (b < nluma) { upd_luma_body<NX>(a, b % nlx, (b / nlx) % nly, b / (nlx * nly), threadIdx.x & 63, threadIdx.x >> 6); return; }
  b -= nluma;
  const int nbb = (2 * a.g.W + 2 * (a.g.H - 2) + 255) / 256;
  upd_border_body(a, (b % nbb) * 256 + threadIdx.x, b / nbb);
}

// The stored chroma no update can reach (Q3: three quarters of every plane) is a plain copy; one workgroup per tile made
// that copy cost its workgroup dispatches (about a nanosecond each, 8 000 per launch at 1080p).  Here the plane outside the
// rectangle of active tiles [0, ya) x [0, xa) is two regions -- A: rows ya.., contiguous; B: rows 0..ya, columns xa.. -- copied 16 KB
// per workgroup.
#define UPD_COPY_BYTES 16384
struct UpdCopyGeom { int ya, xa, nWA, nWB, rowsB; };
__device__ __forceinline__ void upd_copy_span(uint8_t* d, const uint8_t* s, int n, int tid) {   // n % 4 == 0, both 4-byte aligned
  if ((((uintptr_t)d | (uintptr_t)s) & 15) == 0) {
    const int nv = n >> 4;
    for (int i = tid; i < nv; i += 256) ((uint4*)d)[i] = ((const uint4*)s)[i];
    for (int i = (nv << 2) + tid; i < (n >> 2); i += 256) ((uint32_t*)d)[i] = ((const uint32_t*)s)[i];
  } else {
    for (int i = tid; i < (n >> 2); i += 256) ((uint32_t*)d)[i] = ((const uint32_t*)s)[i];
  }
}
__device__ __forceinline__ void upd_chroma_copy(const UpdArgs& a, const UpdCopyGeom& cg, int k, int plane, int tid) {
  const LevelGeom& g = a.g;
  const int j = plane >> 1, c = 1 + (plane & 1), cw = g.W >> 1, chh = g.H >> 1;
  const uint8_t* u = frame_c(a.frames, g, a.src_mul * j, c);
  uint8_t* dst = a.low + (size_t)(a.dst_mul * j) * g.F + (size_t)g.W * g.H + (size_t)(c - 1) * cw * chh;
  if (k < cg.nWA) {
    const size_t o0 = (size_t)cg.ya * cw + (size_t)k * UPD_COPY_BYTES, end = (size_t)chh * cw;
    upd_copy_span(dst + o0, u + o0, (int)(end - o0 < UPD_COPY_BYTES ? end - o0 : UPD_COPY_BYTES), tid);
    return;
  }
  const int r0 = (k - cg.nWA) * cg.rowsB, r1 = imin(cg.ya, r0 + cg.rowsB), wB = cw - cg.xa;
  const bool wide = ((((uintptr_t)dst | (uintptr_t)u) & 15) == 0) && (cw & 15) == 0 && (cg.xa & 15) == 0;   // (uniform)
  const int sh = wide ? 4 : 2, per = wB >> sh, total = (r1 - r0) * per;
  const float inv = 1.0f / (float)per;
  for (int i = tid; i < total; i += 256) {
    const int r = (int)(((float)i + 0.5f) * inv), q = i - r * per;
    const size_t o = (size_t)(r0 + r) * cw + cg.xa + ((size_t)q << sh);
    if (wide) *(uint4*)(dst + o) = *(const uint4*)(u + o);
    else *(uint32_t*)(dst + o) = *(const uint32_t*)(u + o);
  }
}
// the scatter form of the same launch: the chroma tiles an update can reach, luma tiles of US_W x US_H targets, border cells, the
// copy of the rest of the chroma
__global__ void __launch_bounds__(256) k_upd_level_sc(UpdArgs a, int ncxa, int ncya, int nlx, int nly, UpdCopyGeom cg) {
  __shared__ __attribute__((aligned(16))) uint8_t smem[UPD_SC_SMEM];
  const int np = a.g.nB + 1;
  int b = blockIdx.x;
  const int nchroma = ncxa * ncya * np * 2, nluma = nlx * nly * np;
  // the border cells first: a cell's ordered gather over every block within reach is the launch's longest chain of dependent loads;
  // started first it runs beside the tiles, at the end of the grid it was the launch's tail
  const int nbb = (2 * a.g.W + 2 * (a.g.H - 2) + 255) / 256;
  if (b < nbb * np) { upd_border_body(a, (b % nbb) * 256 + threadIdx.x, b / nbb); return; }
  b -= nbb * np;
  if (b < nchroma) { upd_chroma_body<0, true>(a, smem, b % ncxa, (b / ncxa) % ncya, b / (ncxa * ncya), threadIdx.x); return; }
  b -= nchroma;
  if (b < nluma) { upd_luma_tile(a, smem, b % nlx, (b / nlx) % nly, b / (nlx * nly), threadIdx.x); return; }
  b -= nluma;
  const int ncopy = cg.nWA + cg.nWB;
  upd_chroma_copy(a, cg, b % ncopy, b / ncopy, threadIdx.x);
}

bool upd_run_applies(const UpdArgs& a) {
  const LevelGeom& g = a.g;
  return g.bs >= 8 && (g.bs & (g.bs - 1)) == 0 && g.W % 8 == 0 && g.H % 2 == 0 && g.H % g.bs == 0 && g.W % g.bs == 0 && g.H >= 8 && g.W >= 16 &&
         a.reach <= 255;
}
// fewer than 256 blocks within reach of a target: the byte counters of the scatter form cannot wrap
static bool upd_scatter_applies(const UpdArgs& a) {
  const int nb = 2 * a.reach / a.g.bs + 2;
  return nb * nb < 256;
}
void launch_update_run(hipStream_t st, const UpdArgs& a) {
  const LevelGeom& g = a.g;
  const int np = g.nB + 1, nx = upd_max_cols(g.bs, a.reach);
  if (tuning().upd_scatter && upd_scatter_applies(a)) {
    const int cw = g.W / 2, chh = g.H / 2;
    const int nlx = (g.W + US_W - 1) / US_W, nly = (g.H + US_H - 1) / US_H, ncx = (cw + PCT_W - 1) / PCT_W, ncy = (chh + PCT_H - 1) / PCT_H;
    // tile (bx, by) is active iff 2 by PCT_H - 2 < chh + reach and 2 bx PCT_W - 2 < cw + reach (upd_chroma_body)
    const int ncya = imin(ncy, (chh + a.reach + 1) / (2 * PCT_H) + 1), ncxa = imin(ncx, (cw + a.reach + 1) / (2 * PCT_W) + 1);
    UpdCopyGeom cg;
    cg.ya = imin(chh, ncya * PCT_H); cg.xa = imin(cw, ncxa * PCT_W);
    const size_t nA = (size_t)(chh - cg.ya) * cw;
    cg.nWA = (int)((nA + UPD_COPY_BYTES - 1) / UPD_COPY_BYTES);
    const int wB = cw - cg.xa;
    cg.rowsB = wB > 0 ? imax(1, UPD_COPY_BYTES / wB) : 1;
    cg.nWB = wB > 0 ? (cg.ya + cg.rowsB - 1) / cg.rowsB : 0;
    const int nbb = (2 * g.W + 2 * (g.H - 2) + 255) / 256;
    const unsigned nwg = (unsigned)(ncxa * ncya * np * 2 + nlx * nly * np + nbb * np + (cg.nWA + cg.nWB) * np * 2);
    hipLaunchKernelGGL(k_upd_level_sc, dim3(nwg), dim3(256), 0, st, a, ncxa, ncya, nlx, nly, cg);
    return;
  }
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
