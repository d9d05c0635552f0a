// qsvc_amd/csrc/mctf_me.hip -- the +-1 block search (local_me_for_block, motion_estimate.cpp:70-184) with ONE LANE PER 8x8
// SAMPLES of a block, for the blocks whose nine displaced windows of both directions lie inside the picture (me_inner).
//
// A lane keeps the 8 rows of its piece of the predicted picture in registers and streams the ten rows (-1 .. 8) of each reference
// window past them: a row arrives as two unaligned wide loads (samples -1..6 and 1..8), the window displaced by 0 is cut out of
// them with v_alignbyte, and every (row, vertical offset) pair adds into the nine running sums of that direction with packed
// v_sad_u8 (picture samples) / v_sad_u16 on 0x8000-biased samples (pyramid levels).  Nothing crosses lanes until the end: a block
// of 8x8 is one lane (64 blocks per wave), larger blocks add their (bs/8)^2 pieces with DPP row shifts.  The candidate order and
// the "last minimum wins" rule are me_pick's (mctf_core.h).  Blocks that are not inner are left to the border-rule kernels of
// mctf_kernels.hip, which enumerate only the ring of the block grid where such blocks can lie (MeArgs::ring).
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"
#include "mctf_me_ring.h"

namespace qsvc {

template <class T> struct MePiece;
// 8 samples of u8: two dwords
template <> struct MePiece<uint8_t> {
  static constexpr int NW = 2;
  uint32_t p[8][2];
  __device__ __forceinline__ void load_pred(const uint8_t* s, uint32_t off, int stride) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      uint64_t v;
      __builtin_memcpy(&v, s + off + (uint32_t)(r * stride), 8);
      p[r][0] = (uint32_t)v; p[r][1] = (uint32_t)(v >> 32);
    }
  }
  // row of the reference window: w[0] = samples -1..6, w[1] = 0..7, w[2] = 1..8 (two steps: the loads of several rows in flight)
  struct Raw { uint64_t a, e; };
  __device__ __forceinline__ Raw load_raw(const uint8_t* s, uint32_t off) const {
    Raw r;
    __builtin_memcpy(&r.a, s + off, 8);
    __builtin_memcpy(&r.e, s + off + 2, 8);
    return r;
  }
  __device__ __forceinline__ void windows(const Raw& r, uint32_t w[3][2]) const {
    const uint32_t a0 = (uint32_t)r.a, a1 = (uint32_t)(r.a >> 32), e0 = (uint32_t)r.e, e1 = (uint32_t)(r.e >> 32);
    w[0][0] = a0; w[0][1] = a1;
    w[1][0] = __builtin_amdgcn_alignbyte(a1, a0, 1); w[1][1] = __builtin_amdgcn_alignbyte(e1, e0, 3);
    w[2][0] = e0; w[2][1] = e1;
  }
  __device__ __forceinline__ uint32_t sad(int r, const uint32_t* w, uint32_t acc) const {
    return __builtin_amdgcn_sad_u8(p[r][1], w[1], __builtin_amdgcn_sad_u8(p[r][0], w[0], acc));
  }
};
// 8 samples of int16, biased by 0x8000 so that the unsigned SAD equals the signed one: four dwords
template <> struct MePiece<int16_t> {
  static constexpr int NW = 4;
  uint32_t p[8][4];
  struct Q { uint32_t w[4]; };
  __device__ __forceinline__ void load_pred(const int16_t* s, uint32_t off, int stride) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      Q v;
      __builtin_memcpy(&v, s + off + (uint32_t)(r * stride), 16);
#pragma unroll
      for (int k = 0; k < 4; k++) p[r][k] = v.w[k] ^ 0x80008000u;
    }
  }
  struct Raw { Q a, e; };
  __device__ __forceinline__ Raw load_raw(const int16_t* s, uint32_t off) const {
    Raw r;
    __builtin_memcpy(&r.a, s + off, 16);      // samples -1..6
    __builtin_memcpy(&r.e, s + off + 2, 16);  // samples 1..8
    return r;
  }
  __device__ __forceinline__ void windows(const Raw& r, uint32_t w[3][4]) const {
    Q a = r.a, e = r.e;
#pragma unroll
    for (int k = 0; k < 4; k++) { a.w[k] ^= 0x80008000u; e.w[k] ^= 0x80008000u; }
#pragma unroll
    for (int k = 0; k < 4; k++) { w[0][k] = a.w[k]; w[2][k] = e.w[k]; }
    w[1][0] = __builtin_amdgcn_alignbyte(a.w[1], a.w[0], 2);
    w[1][1] = __builtin_amdgcn_alignbyte(a.w[2], a.w[1], 2);
    w[1][2] = __builtin_amdgcn_alignbyte(a.w[3], a.w[2], 2);
    w[1][3] = __builtin_amdgcn_alignbyte(e.w[3], e.w[2], 2);
  }
  __device__ __forceinline__ uint32_t sad(int r, const uint32_t* w, uint32_t acc) const {
    acc = __builtin_amdgcn_sad_u16(p[r][0], w[0], acc);
    acc = __builtin_amdgcn_sad_u16(p[r][1], w[1], acc);
    acc = __builtin_amdgcn_sad_u16(p[r][2], w[2], acc);
    return __builtin_amdgcn_sad_u16(p[r][3], w[3], acc);
  }
};

// acc[wy + 1][wx + 1] += SAD(piece, window displaced by (vy + wy, vx + wx)), wy, wx in -1..1; (y, x) = the piece's origin
template <class T>
__device__ __forceinline__ void me_piece_sums(const MePiece<T>& pc, const T* ref, int stride, int y, int x, int vy, int vx,
                                              uint32_t acc[3][3]) {
  uint32_t off = (uint32_t)((y + vy - 1) * stride + x + vx - 1);
  constexpr int CH = sizeof(T) == 1 ? 10 : 5;  // window rows whose loads are in flight together
#pragma unroll
  for (int c0 = -1; c0 <= 8; c0 += CH) {
    typename MePiece<T>::Raw raw[CH];
#pragma unroll
    for (int q = 0; q < CH; q++) { raw[q] = pc.load_raw(ref, off); off += (uint32_t)stride; }
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const int rr = c0 + q;
      uint32_t w[3][MePiece<T>::NW];
      pc.windows(raw[q], w);
#pragma unroll
      for (int wy = -1; wy <= 1; wy++) {
        const int r = rr - wy;  // the piece's row that meets window row rr under vertical offset wy
        if (r < 0 || r > 7) continue;
#pragma unroll
        for (int wx = 0; wx < 3; wx++) acc[wy + 1][wx] = pc.sad(r, w[wx], acc[wy + 1][wx]);
      }
    }
  }
}

// SUB = bs / 8 pieces per block side; LPB = SUB * SUB lanes per block, 64 / LPB blocks per wave, 4 waves per workgroup
// Workgroups [0, npx * nby * nB): the piece role; the rest: the ring role, a wave per ring block, which skips the blocks the piece role takes (MeArgs::ring).
template <class T, int SUB>
__global__ void __launch_bounds__(256) k_me_piece(MeArgs a, int npx, int ring_groups) {
  constexpr int LPB = SUB * SUB, BPW = 64 / LPB;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int npiece = npx * a.nby * a.g.nB;
  if (a.zero_words)  // (the predict kernel that follows on the stream adds into these histograms)
    for (int k = (int)blockIdx.x * 256 + (int)threadIdx.x; k < a.zero_count; k += (int)gridDim.x * 256) a.zero_words[k] = 0;
  if ((int)blockIdx.x >= npiece) {
    const int r = (int)blockIdx.x - npiece, per_pic = (ring_groups + 3) / 4, gi = (r % per_pic) * 4 + wv;
    if (gi >= ring_groups) return;
    me_search_body<T, 64>(a, gi, 0, r / per_pic, lane);  // one ring block per wave, a lane per sample (or a few): the shortest chain
    return;
  }
  const int grp = lane / LPB, li = lane % LPB;
  const int bx = ((int)(blockIdx.x % npx) * 4 + wv) * BPW + grp, by = (int)(blockIdx.x / npx) % a.nby, i = (int)blockIdx.x / (npx * a.nby);
  const int bs = a.g.bs, H = a.g.H, W = a.g.W;
  int pvx = 0, pvy = 0, nvx = 0, nvy = 0;
  bool go = bx < a.nbx;
  if (go) {
    me_block_init_mv(a, i, by, bx, pvx, pvy, nvx, nvy);
    go = me_inner(H, W, bs, by * bs, bx * bs, pvy, pvx, nvy, nvx);
  }
  int sad[18];
#pragma unroll
  for (int k = 0; k < 18; k++) sad[k] = 0;
  if (go) {
    const size_t N = !a.snap ? a.g.F : a.snap_pic_stride;
    const T* base = !a.snap ? (const T*)a.frames : (const T*)a.snap;
    const int i0 = (!a.snap ? 2 : a.pic_mul) * i;
    const T* r0 = base + (size_t)i0 * N;
    const T* pr = base + (size_t)(i0 + 1) * N;
    const T* r1 = base + (size_t)(i0 + 2) * N;
    const int y = by * bs + 8 * (li / SUB), x = bx * bs + 8 * (li % SUB);
    MePiece<T> pc;
    pc.load_pred(pr, (uint32_t)(y * W + x), W);
    uint32_t acc0[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, acc1[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    me_piece_sums<T>(pc, r0, W, y, x, pvy, pvx, acc0);
    me_piece_sums<T>(pc, r1, W, y, x, nvy, nvx, acc1);
    const int cand[9][2] = QSVC_ME_CANDS;
#pragma unroll
    for (int c = 0; c < 9; c++) {  // PREV tries mv + (dy, dx), NEXT mv - (dy, dx) (motion_estimate.cpp:90-91)
      sad[2 * c] = (int)acc0[cand[c][0] + 1][cand[c][1] + 1];
      sad[2 * c + 1] = (int)acc1[1 - cand[c][0]][1 - cand[c][1]];
    }
  }
  if (LPB > 1) {  // the block's pieces: the group's last lane ends up with the totals
#pragma unroll
    for (int k = 0; k < 18; k++) {
      int v = sad[k];
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);                  // row_shr:1
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);                  // row_shr:2
      if (LPB >= 16) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);                // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);                // row_shr:8
      }
      if (LPB == 64) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);                // row_bcast:15 -> rows 1, 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);                // row_bcast:31 -> rows 2, 3
      }
      sad[k] = v;
    }
  }
  if (go && li == LPB - 1) me_block_store(a, i, by, bx, sad, pvx, pvy, nvx, nvy);
}

template <class T>
static void launch_piece(hipStream_t st, const MeArgs& a, int ring_blocks) {
  const int sub = a.g.bs / 8, bpw = 64 / (sub * sub);
  const int npx = (a.nbx + 4 * bpw - 1) / (4 * bpw);
  const int ring_groups = ring_blocks;      // waves the ring needs per picture (one block each)
  const unsigned nwg = (unsigned)(npx * a.nby * a.g.nB + ((ring_groups + 3) / 4) * a.g.nB);
  dim3 grid(nwg), block(256);
  switch (sub) {
    case 1: hipLaunchKernelGGL((k_me_piece<T, 1>), grid, block, 0, st, a, npx, ring_groups); break;
    case 2: hipLaunchKernelGGL((k_me_piece<T, 2>), grid, block, 0, st, a, npx, ring_groups); break;
    case 4: hipLaunchKernelGGL((k_me_piece<T, 4>), grid, block, 0, st, a, npx, ring_groups); break;
    default: hipLaunchKernelGGL((k_me_piece<T, 8>), grid, block, 0, st, a, npx, ring_groups); break;
  }
}
bool me_piece_applies(const MeArgs& a) {
  const int bs = a.g.bs;
  // 32-bit sample offsets inside a plane; the search windows of a piece are read without any border rule
  return a.g.bord == 0 && (bs == 8 || bs == 16 || bs == 32 || bs == 64) && (size_t)a.g.H * a.g.W < (1u << 30);
}
void launch_me_piece(hipStream_t st, const MeArgs& a, int ring_blocks) {
  if (!a.snap) launch_piece<uint8_t>(st, a, ring_blocks);
  else launch_piece<int16_t>(st, a, ring_blocks);
}

}  // namespace qsvc
