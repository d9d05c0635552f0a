// qsvc_amd/csrc/j2k_kernels.hip -- gfx950 kernels of the JPEG 2000 texture stage that replaces the
// reference's external `kdu_compress` call (trunk/src/texture_compress_fb_j2k.py:183-195):
//   * (forward DWT + quantiser: j2k_dwt.hip; planes without any decomposition level go through k_level_shift +
//     k_t1_pack_blk here);
//   * EBCOT Tier-1 (Annex C + D): sample-parallel context modelling (t1_cm2.h) + MQ coder (t1_core.h);
//   * gather of the code-block segments into one contiguous buffer for the host's Tier-2.
// No MFMA: integer/bit work and short float lifting chains.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include "j2k_kernels.h"
#include "t1_cm2.h"

namespace qsvc {

// ---------------------------------------------------------------------------------------------
// The forward DWT lives in j2k_dwt.hip (one LDS-tiled kernel per level, subbands written as Tier-1 magnitudes).
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

void launch_level_shift(hipStream_t st, const PlaneDesc* d_planes, int nplanes, int maxW, int maxH, void* A, bool reversible) {
  dim3 block(64, 4), grid((maxW + 63) / 64, (maxH + 3) / 4, nplanes);
  if (reversible) hipLaunchKernelGGL(k_level_shift<true>, grid, block, 0, st, d_planes, A);
  else hipLaunchKernelGGL(k_level_shift<false>, grid, block, 0, st, d_planes, A);
}

// ---------------------------------------------------------------------------------------------
// Motion-field coding chain on the device (motion_compress.py:146-186, motion_compress_j2k.py:103-107): residue of every level but
// the last against the level above (field - reference / 2, C division, two fields per reference field, the last reference kept:
// interlevel_motion_decorrelate.cpp:32-70,249-283), NEXT -= PREV on the last (bidirectional_motion_decorrelate.cpp:25-52), and the
// component planes as the reference's `demux 8 2c 2 | split` cuts the planar residue file: component c, sample j <- int16 4 j + c.
__global__ void __launch_bounds__(256) k_motion_planes(MvLevels ml, int16_t* dst, size_t n) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const size_t per = 4 * (size_t)ml.fld;
  const int fi = (int)(e / per);                 // field over all levels
  const int w = (int)(e % per), comp = w / ml.fld, j = w % ml.fld;
  int k = 0;
  while (k + 1 < ml.nlevels && fi >= ml.first[k + 1]) k++;
  const int i = fi - ml.first[k];
  const int m = 4 * j + comp;                    // position in the planar field
  const int16_t* f = ml.filt[k] + (size_t)i * per;
  int v = f[m];
  if (k + 1 < ml.nlevels) {
    const int nref = ml.fields[k + 1];
    const int r = i / 2 < nref ? i / 2 : nref - 1;
    if (r >= 0) v -= ml.filt[k + 1][(size_t)r * per + m] / 2;
  } else if (m >= 2 * ml.fld) {
    v -= f[m - 2 * ml.fld];
  }
  dst[e] = (int16_t)v;
}
void launch_motion_planes(hipStream_t st, const MvLevels& ml, int16_t* dst, size_t n) {
  if (n) hipLaunchKernelGGL(k_motion_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ml, dst, n);
}

// ---------------------------------------------------------------------------------------------
// largest magnitude of every block (collected by the DWT kernel) -> number of magnitude bit-planes
__global__ void __launch_bounds__(256) k_t1_nbps(const CbDesc* cbs, int ncb, int* nbps, int* err) {
  int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= ncb) return;
  int m = nbps[b], nb = 0;
  while (nb < 31 && (m >> nb)) nb++;
  nbps[b] = nb > 15 ? 15 : nb;
  if (nb > cbs[b].Mb || nb > 15) atomicOr(err, 1);  // magnitude does not fit M_b / the 15-bit lane format
}
void launch_t1_nbps(hipStream_t st, const CbDesc* cbs, int ncb, int* nbps, int* err) {
  hipLaunchKernelGGL(k_t1_nbps, dim3((ncb + 255) / 256), dim3(256), 0, st, cbs, ncb, nbps, err);
}

// ---------------------------------------------------------------------------------------------
// The Tier-1 job list on the device (one workgroup): nb[b] = the block's bit-planes, joboff[b] = the exclusive prefix sum over the
// blocks (plane p of block b is job joboff[b] + p: its decision segments' slot), order[] = the MQ stage's slot table -- blocks
// sorted by expected length nb x area, longest first (counting sort on 961 keys; ties in any order: the slot table only decides
// which blocks share a wave, never what is coded), padded with ncb (= idle lane) to nslots.  cap_jobs = the jobs the segment
// buffers hold: if the batch needs more, nb is written as 0 for every block (the Tier-1 kernels then touch nothing) and err
// gets bit 4; the host re-runs the stage with its own job list and larger buffers.
__global__ void __launch_bounds__(1024) k_t1_joblist(const CbDesc* cbs, int ncb, const int* nbps, int cap_jobs, int* nb_eff, int* joboff,
                                                      int* order, int nslots, int* njobs_out, int* err) {
  // The blocks go through LDS in chunks of STAGE (a 1080p GOP: 15 657 blocks, one chunk; a 4K GOP window: seven): coalesced loads
  // of the (bit-planes, area) pairs, then thread t owns 16 consecutive blocks of the chunk -- what the running sum of joboff needs.
  constexpr int STAGE = 16384, PER = STAGE / 1024;
  __shared__ int s_scan[1024];
  __shared__ int s_hist[1024];      // 15 * 64 + 2 = 962 in use
  __shared__ uint16_t s_blk[STAGE]; // nb | area64 << 4
  const int tid = threadIdx.x;
  auto stage = [&](int c0, int n) {
    for (int b = tid; b < n; b += 1024) {
      const int nb = nbps[c0 + b] > 15 ? 15 : nbps[c0 + b];
      s_blk[b] = (uint16_t)(nb | ((((int)cbs[c0 + b].w * cbs[c0 + b].h + 63) >> 6) << 4));
    }
  };
  auto scan1024 = [&](int* a) {  // inclusive, in place
    for (int o = 1; o < 1024; o <<= 1) {
      const int v = tid >= o ? a[tid - o] : 0;
      __syncthreads();
      a[tid] += v;
      __syncthreads();
    }
  };
  // the jobs of the whole batch (decides whether the buffers hold them)
  int sum = 0;
  for (int b = tid; b < ncb; b += 1024) sum += nbps[b] > 15 ? 15 : nbps[b];
  s_scan[tid] = sum;
  s_hist[tid] = 0;
  __syncthreads();
  scan1024(s_scan);
  const int total = s_scan[1023];
  const bool over = total > cap_jobs;
  if (tid == 0) { *njobs_out = total; if (over) atomicOr(err, 4); }
  __syncthreads();
  // first job of every block + the histogram of the sort keys
  int carry = 0;
  for (int c0 = 0; c0 < ncb; c0 += STAGE) {
    const int n = ncb - c0 < STAGE ? ncb - c0 : STAGE;
    stage(c0, n);
    __syncthreads();
    const int b0 = tid * PER < n ? tid * PER : n, b1 = b0 + PER < n ? b0 + PER : n;
    int mine = 0;
    for (int b = b0; b < b1; b++) mine += s_blk[b] & 15;
    s_scan[tid] = mine;
    __syncthreads();
    scan1024(s_scan);
    int run = carry + s_scan[tid] - mine;
    for (int b = b0; b < b1; b++) {
      const int nb = s_blk[b] & 15, ne = over ? 0 : nb;
      nb_eff[c0 + b] = ne;
      joboff[c0 + b] = over ? 0 : run;
      run += nb;
      atomicAdd(&s_hist[15 * 64 - ne * (s_blk[b] >> 4) + 1], 1);
    }
    carry += s_scan[1023];
    __syncthreads();
  }
  scan1024(s_hist);   // over the counts at [key + 1]: s_hist[k] = first slot of key k
  for (int c0 = 0; c0 < ncb; c0 += STAGE) {
    const int n = ncb - c0 < STAGE ? ncb - c0 : STAGE;
    if (ncb > STAGE) { __syncthreads(); stage(c0, n); __syncthreads(); }   // (a single chunk is still there)
    for (int b = tid; b < n; b += 1024) order[atomicAdd(&s_hist[15 * 64 - (over ? 0 : (s_blk[b] & 15)) * (s_blk[b] >> 4)], 1)] = c0 + b;
  }
  for (int k = ncb + tid; k < nslots; k += 1024) order[k] = ncb;
}
void launch_t1_joblist(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, int cap_jobs, int* nb_eff, int* joboff, int* order, int nslots,
                       int* njobs_out, int* err) {
  hipLaunchKernelGGL(k_t1_joblist, dim3(1), dim3(1024), 0, st, cbs, ncb, nbps, cap_jobs, nb_eff, joboff, order, nslots, njobs_out, err);
}

// ---------------------------------------------------------------------------------------------
// EBCOT Tier-1 (T.800 Annex C + D) in two kernels: sample-parallel context modelling (k_t1_cm2, t1_cm2.h: one wave per
// (code-block, bit-plane), decisions as one byte each into per-(block, plane, pass) segments) and the MQ coder over those
// segments (k_t1_mq, t1_core.h: one code-block per lane).
__constant__ uint32_t c_mqtab[47] = T1_MQ_TABLE_INIT;

// Every lane walks ITS block's coding passes on its own: the wave runs until the last lane has coded its last chunk, i.e. for
// max over lanes of the block's decisions -- not for the sum over the passes of the per-pass maxima, which a pass-synchronous
// loop costs (measured on config 3: 51 200 steps for the slowest wave against 38 048 decisions of its longest block).  The
// decisions-per-pass table of the block sits in LDS; the first chunk of the next non-empty pass is prefetched while the last
// chunk of the current one is coded.
#define MQ_CNT_STRIDE 50  // u16 entries per lane (>= T1_MAXPASS; 25 words: odd)
template <int T1_MQ_LANES>
__global__ void __launch_bounds__(T1_MQ_LANES) k_t1_mq(const CbDesc* cbs, int ncb, const int* nbps_in, const int* joboff,
                                                        const int* order, int nslots, const uint8_t* segs, const int* cnt, const long long* dist, uint8_t* outbuf,
                                                        int outcap, CbOut* res, int* pass_len, long long* pass_dist) {
  __shared__ uint32_t s_cx[21 * T1_MQ_LANES];  // 20 context words per lane, odd stride
  __shared__ uint32_t s_stage[(T1_STAGE / 4) * T1_MQ_LANES];
  __shared__ uint16_t s_cnt[MQ_CNT_STRIDE * T1_MQ_LANES];
  __shared__ uint32_t s_mq[64];
  __shared__ uint64_t s_mq2[94];
  static_assert(T1_MAXPASS <= MQ_CNT_STRIDE, "pass table");
  // order[]: the host's slot table (block index, or ncb for an idle lane)
  int lane = threadIdx.x, slot = blockIdx.x * T1_MQ_LANES + lane, b = slot < nslots ? order[slot] : ncb;
  for (int i = lane; i < 47; i += T1_MQ_LANES) s_mq[i] = c_mqtab[i];
  for (int i = lane; i < 94; i += T1_MQ_LANES) s_mq2[i] = mq2_pair(c_mqtab, i);
  __syncthreads();
  T1Lane t;
  t.cx = s_cx + 21 * lane; t.cs = 1; t.mqtab = s_mq; t.mq2 = s_mq2; t.zclut = nullptr;
  t.stg = (uint8_t*)(s_stage + (T1_STAGE / 4) * lane);
  t.w = t.h = t.orient = 0; t.nbps = 0; t.out = outbuf; t.cap = 0;
  size_t job0 = 0;
  if (b < ncb) { t.nbps = nbps_in[b] > 15 ? 15 : nbps_in[b]; t.out = outbuf + (size_t)b * outcap; t.cap = outcap; job0 = (size_t)joboff[b]; }
  int bb = b < ncb ? b : 0;
  int* pl = pass_len + (size_t)bb * T1_MAXPASS;
  long long* pd = pass_dist ? pass_dist + (size_t)bb * T1_MAXPASS : nullptr;
  mq2_lane_init(t);
  // coding pass q of the block: q = 0 is the clean-up pass of the top bit-plane, then three passes per plane
  const int nbps = t.nbps, npass = nbps ? 3 * nbps - 2 : 0;
  auto plane_of = [&](int q) { return q ? nbps - 2 - (((q - 1) * 43) >> 7) : nbps - 1; };   // (q - 1) / 3 for q - 1 < 128
  auto kind_of = [&](int q) { return q ? (q - 1) - 3 * (((q - 1) * 43) >> 7) : 2; };
  uint16_t* mycnt = s_cnt + MQ_CNT_STRIDE * lane;
  for (int q = 0; q < npass; q++) {
    const size_t e = (job0 + plane_of(q)) * 3 + kind_of(q);
    mycnt[q] = (uint16_t)cnt[e];
    if (pd) pd[q] = dist[e];
  }
  auto seg_of = [&](int q) {
    const int k = kind_of(q);
    return (const uint4*)(segs + (job0 + plane_of(q)) * T1_SEG_BYTES + (k == 0 ? 0 : (k == 1 ? T1_SEG1 : T1_SEG2)));
  };
  // (q, n, sg): the pass being coded; (q2, n2, sg2): the next pass that has decisions (q2 == npass: none)
  int q = -1, n = 0, i = 0, q2 = 0, n2 = 0;
  const uint4 *sg = nullptr, *sg2 = nullptr;
  auto look_ahead = [&]() {
    q2 = q + 1; n2 = 0;
    while (q2 < npass && (n2 = mycnt[q2]) == 0) q2++;
    sg2 = q2 < npass ? seg_of(q2) : nullptr;
  };
  auto enter_next = [&]() {   // passes q .. q2-1 are complete (all but q without a decision): their lengths, then pass q2 becomes current
    const int c4 = mq2_count(t) + 4;
    for (int k = q < 0 ? 0 : q; k < q2; k++) pl[k] = c4;
    q = q2; n = n2; sg = sg2; i = 0;
    look_ahead();
  };
  look_ahead();
  enter_next();
  const uint4 pad = make_uint4(T1_PAD4, T1_PAD4, T1_PAD4, T1_PAD4);  // a lane that is done codes the idle context
  uint4 cur = pad;
  if (q < npass) cur = sg[0];
  while (__any(q < npass)) {
    // `cur` must have LANDED before the next chunk's load is issued: otherwise hipcc, which cannot count the
    // divergent stores in between, waits vmcnt(0) at the first use of `cur` and so also for the prefetch.
    asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cur.z), "+v"(cur.w)::"memory");
    const bool more = i + 16 < n;
    const uint4* pa = more ? sg + (i >> 4) + 1 : sg2;
    uint4 nxt = pad;
    if (more || (q < npass && q2 < npass)) nxt = *pa;  // next 16 decisions in flight while these are coded
    mq2_chunk16(t, cur.x, cur.y, cur.z, cur.w);
    if (__any(t.sp - t.stg >= T1_STAGE_FLUSH)) mq2_stage_flush(t);  // all lanes at once: the wave pays for a flush whoever needs it
    cur = nxt;
    i += 16;
    if (q < npass && i >= n) { t.nsym += (uint32_t)n; enter_next(); }
  }
  T1Result r;
  r.pass_len = pl; r.pass_dist = pd;
  t1_lane_finish<true>(t, npass, r);
  if (b < ncb) {
    CbOut o;
    o.len = r.len; o.npasses = r.npasses; o.nbps = r.nbps; o.overflow = r.overflow; o.nsym = r.nsym;
    res[b] = o;
  }
}

// The coder as a two-wave pipeline (mq2_front16 / mq2_back16, t1_core.h): wave 0 of the workgroup walks the decisions, keeps the
// contexts and the interval register and leaves a record per decision in an LDS ring; wave 1 follows one or more chunks behind
// with the code register and the bytes.  Each half is about half a decision long, and a single wave issues one instruction
// per ~4.3 cycles whatever it does, so the pair codes a block's chain in a little over half the time of one wave.
// Both waves derive the same chunk schedule from the block's pass table (chunk c of lane l = the c-th 16-decision piece of its
// passes, the idle context after its last), so the ring needs two counters only: chunks produced, chunks consumed.
template <int L, int MQ_RING, int STAGE, int STAGE_FLUSH>
__global__ void __launch_bounds__(2 * L) k_t1_mq_pipe(const CbDesc* cbs, int ncb, const int* nbps_in, const int* joboff, const int* order, int nslots,
                                                       const uint8_t* segs, const int* cnt, const long long* dist, uint8_t* outbuf, int outcap, CbOut* res,
                                                       int* pass_len, long long* pass_dist, int* err) {
  static_assert(L == 64, "one front wave and one back wave");
  __shared__ uint32_t s_cx[21 * L];
  __shared__ uint32_t s_stage[(STAGE / 4) * L];
  __shared__ uint16_t s_cnt[MQ_CNT_STRIDE * L];
  __shared__ uint32_t s_mq[64];
  __shared__ uint64_t s_mq2[94];
  __shared__ uint32_t s_ring[MQ_RING][16][L];
  __shared__ int s_sync[4];  // [0] chunks produced, [1] chunks consumed, [2] a wait gave up, [3] front done (LDS atomics: ds_read / ds_write)
  const int wv = threadIdx.x / L, lane = threadIdx.x % L;
  const int slot = blockIdx.x * L + lane, b = slot < nslots ? order[slot] : ncb;
  for (int i = threadIdx.x; i < 47; i += 2 * L) s_mq[i] = c_mqtab[i];
  for (int i = threadIdx.x; i < 94; i += 2 * L) s_mq2[i] = mq2_pair(c_mqtab, i);
  if (threadIdx.x < 4) s_sync[threadIdx.x] = 0;
  T1Lane t;
  t.cx = s_cx + 21 * lane; t.cs = 1; t.mqtab = s_mq; t.mq2 = s_mq2; t.zclut = nullptr;
  t.stg = (uint8_t*)(s_stage + (STAGE / 4) * lane);
  t.w = t.h = t.orient = 0; t.nbps = 0; t.out = outbuf; t.cap = 0;
  size_t job0 = 0;
  if (b < ncb) { t.nbps = nbps_in[b] > 15 ? 15 : nbps_in[b]; t.out = outbuf + (size_t)b * outcap; t.cap = outcap; job0 = (size_t)joboff[b]; }
  const int bb = b < ncb ? b : 0;
  int* pl = pass_len + (size_t)bb * T1_MAXPASS;
  long long* pd = pass_dist ? pass_dist + (size_t)bb * T1_MAXPASS : nullptr;
  const int nbps = t.nbps, npass = nbps ? 3 * nbps - 2 : 0;
  auto plane_of = [&](int q) { return q ? nbps - 2 - (((q - 1) * 43) >> 7) : nbps - 1; };
  auto kind_of = [&](int q) { return q ? (q - 1) - 3 * (((q - 1) * 43) >> 7) : 2; };
  uint16_t* mycnt = s_cnt + MQ_CNT_STRIDE * lane;
  if (wv == 0) {  // the block's pass table; all of a lane's loads are in flight together (plane pl = passes 3 (nbps-1-pl) - 2 .. + 2)
    int c3[15][3];
#pragma unroll
    for (int pl = 0; pl < 15; pl++) {
      // a block without bit-planes owns no job: its joboff is the NEXT block's first job, or njobs (one past the table) when it
      // is the last one -- never form an address from it
      const int* src = cnt + (nbps ? job0 + (pl < nbps ? pl : 0) : (size_t)0) * 3;
      c3[pl][0] = src[0]; c3[pl][1] = src[1]; c3[pl][2] = src[2];
    }
#pragma unroll
    for (int pl = 0; pl < 15; pl++)
      if (pl < nbps) {
        const int q0 = 3 * (nbps - 1 - pl) - 2;   // the plane's significance pass (the top plane has only its clean-up pass, q = 0)
        if (q0 >= 0) { mycnt[q0] = (uint16_t)c3[pl][0]; mycnt[q0 + 1] = (uint16_t)c3[pl][1]; }
        mycnt[q0 + 2] = (uint16_t)c3[pl][2];
      }
    if (pd)
      for (int q = 0; q < npass; q++) pd[q] = dist[(job0 + plane_of(q)) * 3 + kind_of(q)];
  }
  __syncthreads();
  // this lane's chunks and the wave's: the same in both waves
  int my_chunks = 0;
  for (int q = 0; q < npass; q++) my_chunks += (mycnt[q] + 15) >> 4;
  int nchunks = my_chunks;
  for (int o = 32; o >= 1; o >>= 1) { const int v = __shfl_xor(nchunks, o, 64); nchunks = v > nchunks ? v : nchunks; }
  nchunks = __builtin_amdgcn_readfirstlane(nchunks);
  // relaxed workgroup-scope atomics on LDS words: plain ds_read_b32 / ds_write_b32 (a volatile access through a generic pointer
  // became a FLAT load with vmcnt(0) -- it waited for the wave's decision prefetches every time it looked at a counter)
  auto sync_get = [&](int which) { return __hip_atomic_load(&s_sync[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  auto sync_set = [&](int which, int v) { __hip_atomic_store(&s_sync[which], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
  int seen = 0;  // the partner's counter as last read (a wave waits on one counter only): it is re-read only when that is not enough
  auto wait_for = [&](int which, int at_least) {  // bounded: a lost update must not hang the GPU
    if (seen >= at_least) return;
    seen = sync_get(which);
    for (int spin = 0; seen < at_least && !sync_get(2); spin++) {
      __builtin_amdgcn_s_sleep(1);
      seen = sync_get(which);
      if (spin > (1 << 24)) { sync_set(2, 1); atomicOr(err, 2); }
    }
  };
  // (q, n): the pass in work; i: decisions of it already taken; q2 / n2: the next pass with decisions
  int q = -1, n = 0, i = 0, q2 = 0, n2 = 0;
  auto look_ahead = [&]() {
    q2 = q + 1; n2 = 0;
    while (q2 < npass && (n2 = mycnt[q2]) == 0) q2++;
  };
  if (wv == 0) {  // ---- front
    t.A = 0x8000;
    for (int c = 0; c < 19; c++) t.cx[c] = mq2_word(t.mqtab, 0, 0);
    t.cx[0] = mq2_word(t.mqtab, 4, 0); t.cx[T1_CTX_RUN] = mq2_word(t.mqtab, 3, 0); t.cx[T1_CTX_UNI] = mq2_word(t.mqtab, 46, 0);
    t.cx[T1_CTX_PAD] = 0;
    auto seg_of = [&](int qq) {
      const int k = kind_of(qq);
      return (const uint4*)(segs + (job0 + plane_of(qq)) * T1_SEG_BYTES + (k == 0 ? 0 : (k == 1 ? T1_SEG1 : T1_SEG2)));
    };
    // the decisions are fetched three chunks ahead of the coder: with half a decision per wave a chunk lasts less than a load from HBM
    const uint4* sg = nullptr;
    look_ahead();
    q = q2; n = n2; sg = q < npass ? seg_of(q) : nullptr;
    const uint4 pad = make_uint4(T1_PAD4, T1_PAD4, T1_PAD4, T1_PAD4);
    auto fetch = [&]() {
      uint4 v = pad;
      if (q < npass) {
        v = sg[i >> 4];
        i += 16;
        if (i >= n) { look_ahead(); q = q2; n = n2; i = 0; sg = q < npass ? seg_of(q) : nullptr; }
      }
      return v;
    };
    uint4 d0 = fetch(), d1 = fetch(), d2 = fetch();
    for (int c = 0; c < nchunks; c++) {
      const uint4 cur = d0;
      d0 = d1; d1 = d2; d2 = fetch();
      wait_for(1, c - MQ_RING + 1);  // the ring slot is free again
      asm volatile("" ::: "memory");
      mq2_front16(t, cur.x, cur.y, cur.z, cur.w, &s_ring[c % MQ_RING][0][lane], L);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the records are in LDS (only LDS: the fetches stay in flight)
      if (lane == 0) sync_set(0, c + 1);
    }
    t.cx[20] = t.A;  // FLUSH (the back's) needs the final interval
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) sync_set(3, 1);
    return;
  }
  // ---- back
  t.C = 0; t.CT = 12; t.B = 0; t.n = 0; t.ob = 0; t.nsym = 0; t.sp = t.stg; t.nfl = 0; t.A = 0x8000;
  auto enter_next = [&]() {
    const int c4 = mq2_count(t) + 4;
    for (int k = q < 0 ? 0 : q; k < q2; k++) pl[k] = c4;
    q = q2; n = n2; i = 0;
    look_ahead();
  };
  look_ahead();
  enter_next();
  for (int c = 0; c < nchunks; c++) {
    wait_for(0, c + 1);
    asm volatile("" ::: "memory");
    mq2_back_chunk(t, &s_ring[c % MQ_RING][0][lane], L);
    asm volatile("" ::: "memory");  // (LDS runs a wave's accesses in order: the records were read before this store lands)
    if (lane == 0) sync_set(1, c + 1);
    if (__any(t.sp - t.stg >= STAGE_FLUSH)) mq2_stage_flush(t);  // all lanes at once: the wave pays for a flush whoever needs it
    i += 16;
    if (q < npass && i >= n) { t.nsym += (uint32_t)n; enter_next(); }
  }
  seen = 0;
  wait_for(3, 1);
  asm volatile("" ::: "memory");
  t.A = t.cx[20];
  T1Result r;
  r.pass_len = pl; r.pass_dist = pd;
  t1_lane_finish<true>(t, npass, r);
  if (b < ncb) {
    CbOut o;
    o.len = r.len; o.npasses = r.npasses; o.nbps = r.nbps; o.overflow = r.overflow; o.nsym = r.nsym;
    res[b] = o;
  }
}

void launch_t1_mq(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const int* joboff, const int* order, int nslots,
                  const uint8_t* segs, const int* cnt, const long long* dist, uint8_t* outbuf, int outcap, CbOut* res, int* pass_len,
                  long long* pass_dist, int want_dist, int* err) {
  // QSVC_MQ_PIPE (read at every launch): 1 = the two-wave pipeline (default), 0 = one wave per 64 blocks.
  // QSVC_MQ_LDS=small: the pipeline with a two-slot ring and a third of the staging area (25 instead of 40 KB of LDS per wave
  // pair): + 5 % throughput when many sequences share the GPU (its workgroups leave room for the other kernels' on a CU),
  // + 4 % latency for one sequence alone.
  const bool pipe = tuning().mq_pipe != 0, small = tuning().mq_lds_small != 0;
  if (pipe) {
    if (small)
      hipLaunchKernelGGL((k_t1_mq_pipe<64, 2, T1_STAGE_SMALL, T1_STAGE_SMALL_FLUSH>), dim3((nslots + 63) / 64), dim3(128), 0, st, cbs, ncb, nbps, joboff, order, nslots,
                         segs, cnt, dist, outbuf, outcap, res, pass_len, want_dist ? pass_dist : nullptr, err);
    else
      hipLaunchKernelGGL((k_t1_mq_pipe<64, 4, T1_STAGE, T1_STAGE_FLUSH>), dim3((nslots + 63) / 64), dim3(128), 0, st, cbs, ncb, nbps, joboff, order, nslots,
                         segs, cnt, dist, outbuf, outcap, res, pass_len, want_dist ? pass_dist : nullptr, err);
    return;
  }
  // lanes (code-blocks) per MQ wave: the coder is straight-line (no divergent path in a decision), so a wave costs the same
  // whatever the number of lanes in work; 64 is best for one sequence (245 waves, one per CU) and for many.  QSVC_MQ_LANES
  // (4, 8, 16, 32; read at every launch) is kept for experiments.
  const int ml = tuning().mq_lanes;
  long long* pdp = want_dist ? pass_dist : nullptr;
#define MQ_LAUNCH(N) hipLaunchKernelGGL(k_t1_mq<N>, dim3((nslots + N - 1) / N), dim3(N), 0, st, cbs, ncb, nbps, joboff, order, nslots, segs, cnt, dist, outbuf, outcap, res, pass_len, pdp)
  switch (ml) { case 4: MQ_LAUNCH(4); break; case 8: MQ_LAUNCH(8); break; case 16: MQ_LAUNCH(16); break; case 32: MQ_LAUNCH(32); break; default: MQ_LAUNCH(64); }
#undef MQ_LAUNCH
}

// ---------------------------------------------------------------------------------------------
// Sample-parallel context modelling (t1_cm2.h): ONE WAVE per code-block, bit-plane after bit-plane (each plane is a "job" with its own
// decision segments); the magnitudes stay in registers, the column words and row masks are carried from plane to plane.  The block's magnitudes are read
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

// lane i <- lane i - 1 (lane 0 <- 0) / lane i <- lane i + 1 (lane 63 <- 0): DPP wave shifts, no trip through the LDS crossbar
__device__ __forceinline__ uint64_t cm2_wave_up(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x138 /*wave_shr:1*/, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, true);
  return (uint64_t)lo | (uint64_t)hi << 32;
}
__device__ __forceinline__ uint64_t cm2_wave_dn(uint64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x130 /*wave_shl:1*/, 0xF, 0xF, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x130, 0xF, 0xF, true);
  return (uint64_t)lo | (uint64_t)hi << 32;
}
// 64 x 64 bit-matrix transpose across the wave: lane r holds row r (bit c = column c) and receives column r (bit c = row c).  Six
// butterfly stages, one per index bit: the lanes r and r ^ d exchange the off-diagonal d x d blocks; for d < 32 what a lane sends
// -- the half of each 32-bit word its partner wants -- fits one word.
__device__ __forceinline__ uint64_t cm2_transpose64(uint64_t v, int lane) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  {
    const bool rj = (lane & 32) != 0;
    const uint32_t got = (uint32_t)__shfl_xor((int)(rj ? lo : hi), 32, 64);
    lo = rj ? got : lo; hi = rj ? hi : got;
  }
#pragma unroll
  for (int j = 4; j >= 0; j--) {
    const int d = 1 << j;
    const uint32_t m0 = d == 16 ? 0x0000FFFFu : (d == 8 ? 0x00FF00FFu : (d == 4 ? 0x0F0F0F0Fu : (d == 2 ? 0x33333333u : 0x55555555u))), m1 = ~m0;
    const bool rj = (lane & d) != 0;
    const uint32_t P = rj ? ((lo & m0) | ((hi & m0) << d)) : (((lo & m1) >> d) | (hi & m1));
    const uint32_t G = (uint32_t)__shfl_xor((int)P, d, 64);
    const uint32_t lo1 = (lo & m1) | (G & m0), hi1 = (hi & m1) | ((G & m1) >> d);
    const uint32_t lo0 = (lo & m0) | ((G & m0) << d), hi0 = (hi & m0) | (G & m1);
    lo = rj ? lo1 : lo0; hi = rj ? hi1 : hi0;
  }
  return (uint64_t)lo | (uint64_t)hi << 32;
}

// inclusive prefix sum over the wave's 64 lanes by DPP row shifts / broadcasts (no LDS crossbar: __shfl_up is a ds_bpermute, six
// dependent round trips per scan, and the modelling kernel runs four scans per bit-plane)
__device__ __forceinline__ int wave_scan_add(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
#define CM2_CW_STRIDE 67
// The staging buffer of a plane's three decision streams.  16 360 bytes of LDS per wave in all = ten waves per CU (round 4: with
// room for the worst case, 9 984 bytes = 20.4 KB per wave, seven waves fitted; the modelling kernel is VALU-bound and every wave
// more shows in the throughput figure).  A plane whose decisions do not fit (more than about 1.4 per sample: rare) writes its
// passes straight to memory, byte by byte.
#ifndef CM2_OUT_BYTES
#define CM2_OUT_BYTES 5888
#endif
static_assert(CM2_LIST_AT + CM2_LIST_MAX * 8 <= CM2_OUT_BYTES, "the sparse form's list lies inside the staging buffer");
static_assert(T1_TABLES_BYTES == CM2_TAB_BYTES, "table block size");
__global__ void __launch_bounds__(64) k_t1_tables(uint8_t* tabs) {
  const int t = threadIdx.x;
  for (int i = t; i < 1536; i += 64) tabs[CM2_TAB_ZC + i] = (uint8_t)t1_zc_from_nb(i & 511, i < 512 ? 0 : (i < 1024 ? 1 : 3));
  for (int i = t; i < 512; i += 64) tabs[CM2_TAB_REF + i] = (uint8_t)(((i >> 4) & 1) ? 16 : ((i & 0x1EF) ? 15 : 14));
  for (int i = t; i < 256; i += 64) tabs[CM2_TAB_SIGN + i] = cm2_sign_entry(i);
  ((uint32_t*)(tabs + CM2_TAB_SP))[t] = CM2_SPREAD_INIT((uint32_t)t);
  for (int i = t; i < 256; i += 64) ((uint64_t*)(tabs + CM2_TAB_CMP8))[i] = cm2_cmp8_entry(i);
}
void launch_t1_tables(hipStream_t st, uint8_t* tabs) { hipLaunchKernelGGL(k_t1_tables, dim3(1), dim3(64), 0, st, tabs); }

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) k_t1_cm2(const CbDesc* cbs, int ncb, const int* nbps_in, const int* joboff, const int* order,
                                                const unsigned long long* mags, uint8_t* segs, int* cnt, long long* dist, int want_dist, const uint8_t* tabs) {
  __shared__ uint32_t s_cw[18][CM2_CW_STRIDE];   // one-cell apron of zeros; odd stride: the emission lanes (stripe, 16-column run) read 2-way, the best 64 lanes can get
  __shared__ __attribute__((aligned(16))) uint32_t s_sp[64];
  __shared__ __attribute__((aligned(16))) uint8_t s_zc[1024];  // [0..511] zero coding, [512..1023] refinement (t1_cm2.h, cm2_emit3w)
  __shared__ __attribute__((aligned(16))) uint8_t s_out[CM2_OUT_BYTES + 16];  // zeroed per plane; the streams OR whole words into it
  __shared__ __attribute__((aligned(16))) uint8_t s_sign[256];
  __shared__ __attribute__((aligned(16))) uint32_t s_pk[64][8];   // per column: its 16 stripes' decisions per pass + "listed" flag (cm2_pack_pk)
  __shared__ __attribute__((aligned(16))) uint64_t s_cmp8[256];
  // order != nullptr: workgroup k models block order[k] (the MQ stage's slot table: longest blocks first; ncb = an idle slot)
  const int lane = threadIdx.x, b = order ? order[blockIdx.x] : (int)blockIdx.x;
  if (b >= ncb) return;
  const CbDesc cb = cbs[b];
  const int w = cb.w, h = cb.h, nstr = (h + 3) >> 2, orient = cb.orient;
  const int nb = nbps_in[b] > 15 ? 15 : nbps_in[b];
  if (nb <= 0) return;  // (the whole workgroup)
  const size_t job0 = (size_t)joboff[b];
  const unsigned long long* mg = mags + (size_t)b * 1024;
  // tables, once per block: 16-byte copies out of the context's table block
  {
    const int oc = orient == 1 ? 1 : (orient == 3 ? 2 : 0);
    const uint4* zc = (const uint4*)(tabs + CM2_TAB_ZC + 512 * oc);
    const uint4* rf = (const uint4*)(tabs + CM2_TAB_REF);
    if (lane < 32) ((uint4*)s_zc)[lane] = zc[lane]; else ((uint4*)s_zc)[lane] = rf[lane - 32];
    if (lane < 16) ((uint4*)s_sign)[lane] = ((const uint4*)(tabs + CM2_TAB_SIGN))[lane];
    else if (lane < 32) ((uint4*)s_sp)[lane - 16] = ((const uint4*)(tabs + CM2_TAB_SP))[lane - 16];
    ((uint4*)s_cmp8)[lane] = ((const uint4*)(tabs + CM2_TAB_CMP8))[lane];
    ((uint4*)s_cmp8)[lane + 64] = ((const uint4*)(tabs + CM2_TAB_CMP8))[lane + 64];
  }
  for (int i = lane; i < 18 * CM2_CW_STRIDE; i += 64) (&s_cw[0][0])[i] = 0;
  // The block's magnitudes stay in registers for all of its bit-planes (lane = column x: 16 stripes x 4 rows).  The state of a
  // plane is a handful of 64-bit masks in two layouts: by COLUMN (lane = x, bit y) where the column words are put together, by ROW
  // (lane = y, bit x) where the membership recurrence runs; a bit-matrix transpose across the wave (cm2_transpose64) takes the
  // plane's bits to the row layout and the membership back.  Both are carried from plane to plane:
  // sig0(p) = sig0(p+1) | bit(p+1), mu(p) = sig0(p+1).
  // ... as bit-planes: P[p] = bit p of the column's 64 samples (cm2_bitplanes: four 16 x 16 bit transposes per block instead of a
  // sample-by-sample extraction per plane); P[15] is the column's sign mask
  uint64_t P[16];
  {
    uint32_t m32[32];
#pragma unroll
    for (int s = 0; s < 16; s++) {
      unsigned long long m = 0;
      if (s < nstr && lane < w) m = mg[s * 64 + lane];
      m32[2 * s] = (uint32_t)m; m32[2 * s + 1] = (uint32_t)(m >> 32);
    }
    cm2_bitplanes(m32, P);
  }
  const uint64_t NEGc = P[15];
  const uint64_t VALc = lane < w ? (h == 64 ? ~0ull : ((1ull << h) - 1)) : 0ull;   // samples of the block in this column
  uint64_t Zc = 0, Bc = 0;   // by column: significant before the plane; the plane's bits (of the plane above on entry)
  uint64_t Zr = 0, Br = 0;   // the same by row
  __syncthreads();
  for (int p = nb - 1; p >= 0; p--) {
  const size_t job = job0 + p;
  for (int i = lane; i < (CM2_OUT_BYTES + 16) / 16; i += 64) ((uint4*)s_out)[i] = make_uint4(0, 0, 0, 0);
  const uint64_t MUc = Zc;   // significant before the plane above: "refined before" of the refinement pass
  Zc |= Bc;
  Zr |= Br;
  Bc = P[p];   // (p is uniform: an indexed register read)
  Br = cm2_transpose64(Bc, lane);
  // membership: fixpoint of cm2_row_update over the 64 rows (t1_cm2.h)
  uint64_t Mc;
  {
    const uint64_t wmask = w == 64 ? ~0ull : ((1ull << w) - 1);
    const uint64_t cand = (lane < h ? wmask : 0ull) & ~Zr;
    const uint64_t seed = cm2_row_seed(Zr, cm2_wave_up(Zr), cm2_wave_dn(Zr));
    uint64_t M = 0;
    for (;;) {
      const uint64_t S1 = M & Br;
      const uint64_t Mn = cm2_row_update(cand, Br, seed, cm2_wave_up(S1), cm2_wave_dn(S1), lane & 3);
      const bool changed = Mn != M;
      M = Mn;
      if (!__any(changed)) break;
    }
    Mc = cm2_transpose64(M, lane);
  }
  {
    const uint64_t S1c = Mc & Bc, S2c = VALc & ~Zc & ~Mc & Bc;
    {  // decisions per pass of the column's 16 stripes, bit-parallel on the masks (cm2_counts_swar); the neighbours' by DPP
      const uint64_t T = Zc | S1c, T2 = T | S2c;
      Cm2Counts cc;
      cm2_counts_swar(cc, VALc, Zc, Bc, Mc, cm2_wave_up(T), cm2_wave_up(T2), cm2_wave_dn(T), cm2_wave_dn(T2));
      uint32_t pk[8];
      cm2_pack_pk(cc, pk);
      *(uint4*)&s_pk[lane][0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      *(uint4*)&s_pk[lane][4] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    }
    // the column words of the 16 stripes: a nibble transpose of the eight masks (cm2_colwords8); lanes beyond the width hold 0 in every mask
    const uint64_t F[8] = {Zc, Bc, NEGc, MUc, VALc, Mc, S1c, S2c};
#pragma unroll
    for (int hh = 0; hh < 2; hh++) {
      uint32_t f[8], o8[8];
#pragma unroll
      for (int q = 0; q < 8; q++) f[q] = (uint32_t)(F[q] >> (32 * hh));
      cm2_colwords8(f, o8);
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (8 * hh + k < nstr) s_cw[8 * hh + k + 1][lane + 1] = o8[k];   // (uniform)
    }
  }
  __syncthreads();
  Cm2Ctx cx;
  cx.zc = s_zc; cx.sp = s_sp; cx.p = p; cx.want_dist = want_dist != 0;
  const bool first = p == nb - 1;
  const int pass0 = first ? 2 : 0;
  // a lane walks 16 consecutive columns of ONE stripe (s = lane / 4, x from 16 (lane % 4)): first it counts the three passes'
  // decisions from the raw column words, then it emits them through a sliding window of spread neighbourhood columns
  const int ls = lane >> 2, lx0 = (lane & 3) * 16;
  const bool lane_live = ls < nstr && lx0 < w;
  int n3[3] = {0, 0, 0};
  uint32_t act = 0;    // columns that are not "plain" (more to do than one RUN = 0)
  const uint16_t* mypk = (const uint16_t*)&s_pk[lx0][0] + CM2_PK_AT(ls);   // column lx0 + j: + 16 j halfwords
  if (lane_live) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t pkj = mypk[16 * j];   // zero for a column beyond the block's width
      n3[0] += pkj & 15; n3[1] += (pkj >> 4) & 15; n3[2] += (pkj >> 8) & 15;
      act |= (pkj >> 12) << j;
    }
  }
  int off3[3], tot3[3];
  for (int pass = 0; pass < 3; pass++) {
    const int incl = wave_scan_add(n3[pass]);
    off3[pass] = incl - n3[pass];
    tot3[pass] = __builtin_amdgcn_readlane(incl, 63);
  }
  uint8_t* seg = segs + (size_t)job * T1_SEG_BYTES;
  long long d3[3] = {0, 0, 0};
  // 16-byte aligned bases of the three streams inside s_out when all of them fit at once
  const int base1 = (tot3[0] + 15) & ~15, base2 = base1 + ((tot3[1] + 15) & ~15);
  // the columns with more to do than one RUN = 0, over the wave
  int apos, nact;
  {
    const int mine = __popc(act);
    const int incl = wave_scan_add(mine);
    apos = incl - mine;
    nact = __builtin_amdgcn_readlane(incl, 63);
  }
  if (nact <= CM2_LIST_MAX && base2 + tot3[2] + 16 <= CM2_LIST_AT) {
    // ---- sparse plane: plain columns put their RUN byte, the others are listed and worked off 64 at a time ---------------
    Cm2Entry* list = (Cm2Entry*)(s_out + CM2_LIST_AT);
    // (the RUN bytes of the lane's plain columns through a sequential stream: whole words by plain stores, zeros where the listed
    // columns' clean-up decisions are OR-ed in afterwards)
    Cm2Seq R2;
    cm2_seq_begin(R2, s_out + base2, off3[2], nullptr);
    if (lane_live) {
      int o0 = off3[0], o1 = off3[1], o2 = off3[2], lp = apos;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t pkj = mypk[16 * j];
        const int n0 = pkj & 15, n1 = (pkj >> 4) & 15, n2 = (pkj >> 8) & 15;
        if ((act >> j) & 1) {
          Cm2Entry e;
          e.col = (uint16_t)(ls * 64 + lx0 + j); e.o0 = (uint16_t)o0; e.o1 = (uint16_t)o1; e.o2 = (uint16_t)o2;
          list[lp++] = e;
          cm2_seq_skip_b(R2, n2);
        } else if (n2) {
          cm2_seq_put1_b(R2, (uint32_t)T1_CTX_RUN);
        }
        o0 += n0; o1 += n1; o2 += n2;
      }
    }
    cm2_seq_end(R2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int base = 0; base < nact; base += 64) {
      const int k = base + lane;
      if (k < nact) {
        const Cm2Entry e = list[k];
        const int s = e.col >> 6, x = e.col & 63;
        Cm2Win c;
        cm2_win_at(c, s_sp, &s_cw[0][0], CM2_CW_STRIDE, s, x);
        Cm2Out3 O;
        cm2_emit3s(O, s_zc, s_sign, s_cmp8, c);
        Cm2Stream T0, T1, T2;
        cm2_stream_begin(T0, s_out, e.o0); cm2_stream_begin(T1, s_out + base1, e.o1); cm2_stream_begin(T2, s_out + base2, e.o2);
        cm2_stream_put8(T0, O.p0); cm2_stream_put4(T1, O.p1); cm2_stream_put4(T2, O.hdr); cm2_stream_put8(T2, O.p2);
        cm2_stream_end(T0); cm2_stream_end(T1); cm2_stream_end(T2);
        if (want_dist) cm2_dist3(cx, c.w, mg[e.col], d3);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (lane < 16)  // the tails of the last 16-byte chunks: the idle decision (t1_core.h, T1_PAD)
      for (int pass = pass0; pass < 3; pass++) {
        const int i = tot3[pass] + lane;
        if (i < ((tot3[pass] + 15) & ~15)) s_out[(pass == 0 ? 0 : (pass == 1 ? base1 : base2)) + i] = (uint8_t)T1_PAD;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int pass = pass0; pass < 3; pass++) {
      const uint8_t* src = s_out + (pass == 0 ? 0 : (pass == 1 ? base1 : base2));
      uint8_t* dst = seg + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      for (int o = lane * 16; o < tot3[pass]; o += 64 * 16) *(uint4*)(dst + o) = *(const uint4*)(src + o);
    }
  } else if (base2 + tot3[2] <= CM2_OUT_BYTES) {
    // (the lane's dummy word for stores that are not due yet: the packed counts are dead from here to the next plane)
    const bool chk_plain = nact <= 768;
    uint32_t* const trash = &s_pk[0][0] + lane;
    Cm2Seq T0, T1, T2;
    cm2_seq_begin(T0, s_out, off3[0], trash);
    cm2_seq_begin(T1, s_out + base1, off3[1], trash);
    cm2_seq_begin(T2, s_out + base2, off3[2], trash);
    if (lane_live) {
      Cm2Win c;
      c.w = 0; c.B1 = c.D1 = c.Bn = c.Dn = 0;
      for (int k = 0; k < 3; k++) c.AB[k] = c.P[k] = c.Q[k] = c.DL[k] = c.DB[k] = c.X[k] = 0;
      cm2_win_push(c, s_sp, 0, s_cw[ls][lx0], s_cw[ls + 1][lx0], s_cw[ls + 2][lx0]);
      uint32_t own = s_cw[ls + 1][lx0 + 1];
      cm2_win_push(c, s_sp, 0, s_cw[ls][lx0 + 1], own, s_cw[ls + 2][lx0 + 1]);
#ifndef CM2_DENSE_UNROLL
#define CM2_DENSE_UNROLL 16
#endif
#pragma unroll CM2_DENSE_UNROLL
      for (int j = 0; j < 16; j++) {
        const uint32_t nm = s_cw[ls + 1][lx0 + j + 2];
        cm2_win_push(c, s_sp, own, s_cw[ls][lx0 + j + 2], nm, s_cw[ls + 2][lx0 + j + 2]);
        // In the upper bit-planes most columns are a run of four zeros without a significant neighbour: one RUN decision.  When
        // that holds for the column in work of EVERY lane, the wave skips the general form.
        const bool col_on = lx0 + j < w;
        // (with three quarters of the columns listed as not plain the test would never hold for all 64 lanes: not made)
        const bool plain = chk_plain && (!col_on || cm2_plain_run(c));
        if (chk_plain && !__any(!plain)) {
          if (col_on) { Cm2Bytes run1; run1.a = T1_CTX_RUN; run1.n = 1; cm2_seq_put4(T2, run1); }
        } else if (col_on) {
          Cm2Out3 O;
          cm2_emit3s(O, s_zc, s_sign, s_cmp8, c);
          cm2_seq_put8(T0, O.p0); cm2_seq_put4(T1, O.p1);
          if (__any(O.hdr.n != 0)) cm2_seq_put4(T2, O.hdr);
          cm2_seq_put8(T2, O.p2);
          if (want_dist) cm2_dist3(cx, own, mg[lane * 16 + j], d3);
        }
        own = nm;
      }
    }
    cm2_seq_end(T0); cm2_seq_end(T1); cm2_seq_end(T2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (lane < 16)  // the tails of the last 16-byte chunks: the idle decision (t1_core.h, T1_PAD)
      for (int pass = pass0; pass < 3; pass++) {
        const int i = tot3[pass] + lane;
        if (i < ((tot3[pass] + 15) & ~15)) s_out[(pass == 0 ? 0 : (pass == 1 ? base1 : base2)) + i] = (uint8_t)T1_PAD;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int pass = pass0; pass < 3; pass++) {
      const uint8_t* src = s_out + (pass == 0 ? 0 : (pass == 1 ? base1 : base2));
      uint8_t* dst = seg + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      for (int o = lane * 16; o < tot3[pass]; o += 64 * 16) *(uint4*)(dst + o) = *(const uint4*)(src + o);
    }
  } else {  // more decisions than the buffer holds: one pass at a time, straight to memory (the per-pass column form)
    for (int pass = pass0; pass < 3; pass++) {
      uint8_t* dst = seg + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      Cm2ByteSink B;
      B.out = dst + off3[pass]; B.n = 0;
      for (int i = lane * 16; i < lane * 16 + 16; i++) {
        const int s = i >> 6, x = i & 63;
        if (s < nstr && x < w)
          cm2_column(B, cx, pass, &s_cw[s][x], &s_cw[s + 1][x], &s_cw[s + 2][x], want_dist ? mg[i] : 0ull, d3[pass]);
      }
      if (lane < 16 && tot3[pass] + lane < ((tot3[pass] + 15) & ~15)) dst[tot3[pass] + lane] = (uint8_t)T1_PAD;
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
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads();  // the next plane rewrites the column words and the streams
  }  // planes
}
// order == nullptr: every block, in index order; else the `count` slots order[0 .. count)
void launch_t1_cm2(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const int* joboff, const int* order, int count,
                   const unsigned long long* mags, uint8_t* segs, int* cnt, long long* dist, int want_dist, const uint8_t* tabs) {
  const int n = order ? count : ncb;
  if (n > 0) hipLaunchKernelGGL(k_t1_cm2, dim3(n), dim3(64), 0, st, cbs, ncb, nbps, joboff, order, mags, segs, cnt, dist, want_dist, tabs);
}

// Which passes of a block go into which layer: one thread per block, the restatement's loop (oracle/j2k_oracle.c) in the same double
// arithmetic (no contraction: the library is built with -ffp-contract=off; IEEE division), so the choice is bit-identical with the
// restatement's.  Round 4: on the host this was 4 ms of 16 threads per 4K sequence, behind 37 MB of distortion sums over the bus.
__global__ void __launch_bounds__(256) k_rate_alloc(const CbOut* res, const int* plen_all, const long long* pdist_all, const double* wgt, const int* cb_plane,
                                                    const RaPlane* planes, int ncb, uint8_t* layer_passes) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= ncb) return;
  const int npasses = res[k].npasses;
  const int* plen = plen_all + (size_t)k * T1_MAXPASS;
  const long long* pdist = pdist_all + (size_t)k * T1_MAXPASS;
  const RaPlane& pp = planes[cb_plane[k]];
  const double weight = wgt[k];
  uint8_t* lp = layer_passes + (size_t)k * 16;
  int prev = 0;
  for (int l = 0; l < 16; l++) {
    int n = 0;
    if (l < pp.layers) {
      const double lambda = pp.lam[l];
      if (l == pp.layers - 1 && pp.last_all) n = npasses;
      else if (lambda <= 0) n = 0;
      else {
        int best = 0, best_l = 0, cur = 0;
        while (cur < npasses) {
          double bs = -1; int bi = -1;
          double cd = 0;
          for (int i = cur; i < npasses; i++) {
            cd += (double)pdist[i] * weight;
            const int dl = plen[i] - best_l;
            if (dl <= 0) { if (cd > 0) { bs = 1e300; bi = i; } continue; }
            const double s = cd / dl;
            if (s >= bs) { bs = s; bi = i; }
          }
          if (bi < 0 || bs < lambda) break;
          best_l = plen[bi]; best = bi + 1; cur = bi + 1;
        }
        n = best;
      }
      if (l > 0 && n < prev) n = prev;
      prev = n;
    }
    lp[l] = (uint8_t)n;
  }
}
void launch_rate_alloc(hipStream_t st, const CbOut* res, const int* plen, const long long* pdist, const double* wgt, const int* cb_plane,
                       const RaPlane* planes, int ncb, uint8_t* layer_passes) {
  if (ncb > 0) hipLaunchKernelGGL(k_rate_alloc, dim3((ncb + 255) / 256), dim3(256), 0, st, res, plen, pdist, wgt, cb_plane, planes, ncb, layer_passes);
}

// gather: segment of block b (bytes 1..len of its lane buffer) -> dst + offs[b]
__global__ void __launch_bounds__(256) k_gather(const uint8_t* outbuf, int outcap, const unsigned long long* offs, const int* lens,
                                                 int ncb, uint8_t* dst) {
  int b = blockIdx.x;
  if (b >= ncb) return;
  const uint8_t* s = outbuf + (size_t)b * outcap + 1;
  uint8_t* d = dst + offs[b];
  const int n = lens[b];
  // bytes up to the destination's first 16-byte boundary, then 16 bytes per thread and step (aligned stores, unaligned loads: the
  // source starts one byte into its slot), then the tail
  const int head = (int)((16 - ((uintptr_t)d & 15)) & 15), h = head < n ? head : n, nv = (n - h) >> 4;
  if ((int)threadIdx.x < h) d[threadIdx.x] = s[threadIdx.x];
  typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u4u;
  for (int i = threadIdx.x; i < nv; i += 256) *(uint4*)(d + h + 16 * i) = __builtin_bit_cast(uint4, *(const u4u*)(s + h + 16 * i));
  const int t0 = h + 16 * nv;
  if (t0 + (int)threadIdx.x < n) d[t0 + threadIdx.x] = s[t0 + threadIdx.x];
}
void launch_gather(hipStream_t st, const uint8_t* outbuf, int outcap, const unsigned long long* offs, const int* lens, int ncb,
                   uint8_t* dst) {
  hipLaunchKernelGGL(k_gather, dim3(ncb), dim3(256), 0, st, outbuf, outcap, offs, lens, ncb, dst);
}

}  // namespace qsvc
