// qsvc_amd/csrc/j2k_kernels.h -- device-side structures and launch wrappers of j2k_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "t1_core.h"

namespace qsvc {

struct PlaneDesc {   // one image component plane of the batch
  size_t off;        // element offset of the plane in the coefficient buffers (stride = W)
  const uint8_t* src; // source samples (device): 8-bit unsigned (level shift -128) or 16-bit signed (no shift)
  int u8stride;      // source row stride in samples
  int W, H;
  int s16;           // 1: src holds int16 signed samples
};
struct CbDesc {      // one code-block
  int plane;
  int x0, y0;        // position in the plane's Mallat layout
  short w, h;
  short orient;      // 0 LL, 1 HL, 2 LH, 3 HH
  short Mb;          // magnitude bit-planes of its subband
  float delta;       // quantiser step (irreversible path)
};
struct CbOut {
  int len, npasses, nbps, overflow;
  unsigned long long nsym;
};

// Tier-1 working set per code-block, lane-interleaved within a group of `lanes` blocks
#define T1_MAGS_WORDS (16 * 64 * 2)               // u32 words per block: [stripe][x][half] (x lanes)
#define T1_FLAGS_WORDS (18 * T1_FW)               // u32 words per block (x lanes)

void launch_dwt(hipStream_t st, const PlaneDesc* d_planes, int nplanes, int maxW, int maxH, void* A,
                void* B, int levels, bool reversible);
void launch_t1_pack(hipStream_t st, const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coef, uint32_t* mags,
                    int* nbps, int* err, bool reversible, int lanes);
void launch_t1_encode(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const uint32_t* mags, uint32_t* flags,
                      uint8_t* outbuf, int outcap, CbOut* res, int* pass_len, long long* pass_dist, int want_dist, int lanes);
void launch_t1_split(hipStream_t st, const int* waves, int nwaves, const CbDesc* cbs, int ncb, const int* nbps, const int* joboff,
                     const int* order, int nslots,
                     const uint32_t* mags, uint32_t* flags, uint8_t* segs, int* cnt, long long* dist, uint8_t* outbuf, int outcap,
                     CbOut* res, int* pass_len, long long* pass_dist, int want_dist, int stage);
void launch_t1_pack_blk(hipStream_t st, const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coef,
                        unsigned long long* mags, int* nbps, int* err, bool reversible);
void launch_t1_cm2(hipStream_t st, const int* jobs, int njobs, const CbDesc* cbs, const int* nbps, const unsigned long long* mags,
                   uint8_t* segs, int* cnt, long long* dist, int want_dist);
void launch_gather(hipStream_t st, const uint8_t* outbuf, int outcap, const unsigned long long* offs, const int* lens, int ncb,
                   uint8_t* dst);
}  // namespace qsvc
