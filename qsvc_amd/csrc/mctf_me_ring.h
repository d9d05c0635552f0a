// qsvc_amd/csrc/mctf_me_ring.h -- the border-rule forms of the block search (a wave per group of blocks; every sample through
// MePlane::at, i.e. replicate border with Q1/Q2 and, at pyramid levels, the pixel-domain border around the Mallat interior, H3):
// bodies shared by the stand-alone kernels of mctf_kernels.hip and the ring role of mctf_me.hip's level kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "mctf_threads.h"

namespace qsvc {

__device__ __forceinline__ uint64_t ld8u(const uint8_t* p) {  // unaligned 8-byte load
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

// ---------------------------------------------------------------------------------------------
// ME search: one wave per MC block; lanes split the block's pixels, 18 SADs reduced across the
// wave with DPP-free xor shuffles (64 lanes).
// LPB = lanes per MC block: 64 (one block per wave) or 16 (four 8x8 blocks per wave, 4 pixels per lane: the
// xor-shuffle reduction then needs 4 steps instead of 6 and serves four blocks at once).
template <class T, int LPB>
__device__ __forceinline__ void me_search_body(const MeArgs& a, const int gx_, const int gy_, const int i, const int tx_) {
  const int sub = tx_ / LPB, lane = tx_ % LPB;
  int bx = gx_ * (64 / LPB) + sub, by = gy_;
  bool live = a.ring ? me_ring_block(a, bx, by, bx) : bx < a.nbx;
  MePlane<T> r0, r1, pr;
  me_block_setup<T>(a, i, r0, r1, pr);
  int pvx = 0, pvy = 0, nvx = 0, nvy = 0;
  int sad[18];
  if (live) {
    me_block_init_mv(a, i, by, bx, pvx, pvy, nvx, nvy);
    if (a.ring && me_inner(a.g.H, a.g.W, a.g.bs, by * a.g.bs, bx * a.g.bs, pvy, pvx, nvy, nvx)) live = false;  // k_me_piece's
  }
  if (live) {
    me_block_sads<T>(r0, r1, pr, a.g.bs, by, bx, pvy, pvx, nvy, nvx, lane, LPB, sad, a.g.bord);
  } else {
#pragma unroll
    for (int k = 0; k < 18; k++) sad[k] = 0;
  }
  // 18 sums over the block's LPB lanes with DPP row shifts (a DPP row = 16 lanes = one 8x8 block's lanes when LPB = 16): the
  // LAST lane of the block ends up with the totals; no LDS round trips (the xor-shuffle form costs 18 x log2(LPB) ds_bpermute)
#pragma unroll
  for (int k = 0; k < 18; k++) {
    int v = sad[k];
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    if (LPB == 64) {
      v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
      v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    }
    sad[k] = v;
  }
  if (live && lane == LPB - 1) me_block_store(a, i, by, bx, sad, pvx, pvy, nvx, nvy);
}
// 8x8 blocks, packed form: one lane per block ROW (8 lanes per block, 8 blocks per wave).  A lane holds its row of the predicted
// picture and, per reference, the three displaced rows it can meet (mv.y - 1 .. mv.y + 1); the nine candidates are the 3 x 3
// (row, byte offset) windows of those, each compared with v_sad_u8 / v_sad_u16 on packed samples.  18 sums over the block's 8
// lanes by DPP row shifts.  Blocks whose windows leave the picture take the per-sample path with the border rule (Q1/Q2, H3).
template <class T> struct MeRow;
template <> struct MeRow<uint8_t> {
  uint32_t w[2];
  __device__ __forceinline__ void load(const uint8_t* p) { uint64_t v = ld8u(p); w[0] = (uint32_t)v; w[1] = (uint32_t)(v >> 32); }
  __device__ __forceinline__ int sad(const MeRow& o) const {
    return (int)__builtin_amdgcn_sad_u8(w[1], o.w[1], __builtin_amdgcn_sad_u8(w[0], o.w[0], 0u));
  }
};
template <> struct MeRow<int16_t> {
  uint32_t w[4];  // 8 samples, biased by 0x8000 so that the unsigned SAD equals the signed one
  __device__ __forceinline__ void load(const int16_t* p) {
    uint64_t a = ld8u((const uint8_t*)p), b = ld8u((const uint8_t*)(p + 4));
    w[0] = (uint32_t)a ^ 0x80008000u; w[1] = (uint32_t)(a >> 32) ^ 0x80008000u;
    w[2] = (uint32_t)b ^ 0x80008000u; w[3] = (uint32_t)(b >> 32) ^ 0x80008000u;
  }
  __device__ __forceinline__ int sad(const MeRow& o) const {
    uint32_t s = __builtin_amdgcn_sad_u16(w[0], o.w[0], 0u);
    s = __builtin_amdgcn_sad_u16(w[1], o.w[1], s);
    s = __builtin_amdgcn_sad_u16(w[2], o.w[2], s);
    return (int)__builtin_amdgcn_sad_u16(w[3], o.w[3], s);
  }
};
template <class T>
__device__ __forceinline__ void me_search8_body(const MeArgs& a, const int gx_, const int gy_, const int i, const int tx_) {
  const int sub = tx_ >> 3, row = tx_ & 7;
  int bx = gx_ * 8 + sub, by = gy_;
  bool live = a.ring ? me_ring_block(a, bx, by, bx) : bx < a.nbx;
  MePlane<T> r0, r1, pr;
  me_block_setup<T>(a, i, r0, r1, pr);
  int pvx = 0, pvy = 0, nvx = 0, nvy = 0;
  int sad[18];
#pragma unroll
  for (int k = 0; k < 18; k++) sad[k] = 0;
  if (live) {
    me_block_init_mv(a, i, by, bx, pvx, pvy, nvx, nvy);
    const int y0 = by * 8, x0 = bx * 8, y = y0 + row;
    const bool inner = me_inner(r0.g.H, r0.g.W, 8, y0, x0, pvy, pvx, nvy, nvx);
    if (inner && a.ring) live = false;  // k_me_piece's
    else if (inner) {
      MeRow<T> p, w;
      p.load(pr.snap + (size_t)y * pr.stride + x0);
      const int cand[9][2] = QSVC_ME_CANDS;
#pragma unroll
      for (int c = 0; c < 9; c++) {
        w.load(r0.snap + (size_t)(y + pvy + cand[c][0]) * r0.stride + x0 + pvx + cand[c][1]);
        sad[2 * c] = p.sad(w);
        w.load(r1.snap + (size_t)(y + nvy - cand[c][0]) * r1.stride + x0 + nvx - cand[c][1]);
        sad[2 * c + 1] = p.sad(w);
      }
    } else {
      const int cand[9][2] = QSVC_ME_CANDS;
      for (int k = 0; k < 8; k++) {
        const int pv = pr.in(y, x0 + k);
        for (int c = 0; c < 9; c++) {
          int d0 = pv - r0.at(y + pvy + cand[c][0], x0 + k + pvx + cand[c][1]);
          int d1 = pv - r1.at(y + nvy - cand[c][0], x0 + k + nvx - cand[c][1]);
          sad[2 * c] += d0 < 0 ? -d0 : d0;
          sad[2 * c + 1] += d1 < 0 ? -d1 : d1;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 18; k++) {  // lane 7 of each group of 8 ends up with the block's total
    int v = sad[k];
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    sad[k] = v;
  }
  if (live && row == 7) me_block_store(a, i, by, bx, sad, pvx, pvy, nvx, nvy);
}

}  // namespace qsvc
