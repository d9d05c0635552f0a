// qsvc_amd/csrc/j2k_kernels.h -- device-side structures and launch wrappers of j2k_kernels.hip
#pragma once
#include "tuning.h"
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "t1_core.h"

namespace qsvc {

struct PlaneDesc {   // one image component plane of the batch
  size_t off;        // element offset of the plane in a full-size 32-bit plane buffer (stride = W): planes without any
                     // decomposition level, and the Mallat-layout output of the unit-test entry point
  size_t ll[2];      // element offsets of this plane's compact low bands: [0] LL_1, LL_3, ... ; [1] LL_2, LL_4, ...
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

// Rate allocation on the device (PCRD-opt, T.800 J.14; oracle/j2k_oracle.c holds the same loop): per plane the
// layers' thresholds, per block its subband's distortion weight and its plane
struct RaPlane { double lam[16]; int layers; int last_all; };   // lam[l] <= 0: layer l has no slope; last_all: the last layer takes every pass
void launch_rate_alloc(hipStream_t st, const CbOut* res, const int* plen, const long long* pdist, const double* wgt, const int* cb_plane,
                       const RaPlane* planes, int ncb, uint8_t* layer_passes /*[ncb][16]*/);

struct DwtSb {       // one subband of one plane as the DWT kernel addresses its code-blocks
  int first_cb;      // index of the subband's first code-block in the batch
  int gx;            // code-blocks per row
  int Mb;
  float delta;       // quantiser step (irreversible path)
};

// Tier-1 magnitudes: 1024 u64 per code-block, [stripe][x] = 4 x (sign << 15 | 15-bit magnitude) of rows 4s .. 4s+3
#define T1_MAGS_BYTES (1024 * 8)

// forward DWT, L >= 1 levels (j2k_dwt.hip); plane_out != nullptr selects the Mallat-plane form of the unit-test entry point
void launch_dwt_tiles(hipStream_t st, const PlaneDesc* d_planes, int nplanes, int maxW, int maxH, int minW, int minH, const DwtSb* d_sbtab, int L,
                      bool reversible, void* llA, void* llB, unsigned long long* mags, int* maxmag, void* plane_out,
                      int cbw_log2, int cbh_log2, int* err, const uint32_t* d_tilemap, const int* first);
// the levels' workgroup lists of a batch whose planes differ in size (no empty workgroups): map[first[l] .. first[l + 1]) = level l
void dwt_build_tilemap(const PlaneDesc* planes, int nplanes, int L, std::vector<uint32_t>& map, std::vector<int>& first);
bool dwt_tilemap_fits(int nplanes, int maxW, int maxH);
// planes without a decomposition level: level shift into a 32-bit plane, then launch_t1_pack_blk
void launch_level_shift(hipStream_t st, const PlaneDesc* d_planes, int nplanes, int maxW, int maxH, void* A, bool reversible);
void launch_t1_nbps(hipStream_t st, const CbDesc* cbs, int ncb, int* nbps, int* err);
void launch_t1_mq(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const int* joboff, const int* order, int nslots,
                  const uint8_t* segs, const int* cnt, const long long* dist, uint8_t* outbuf, int outcap, CbOut* res, int* pass_len,
                  long long* pass_dist, int want_dist, int* err);
void launch_t1_pack_blk(hipStream_t st, const CbDesc* cbs, int ncb, const PlaneDesc* planes, const void* coef,
                        unsigned long long* mags, int* nbps, int* err, bool reversible);
void launch_t1_joblist(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, int cap_jobs, int* nb_eff, int* joboff, int* order, int nslots,
                       int* njobs_out, int* err);
void launch_t1_cm2(hipStream_t st, const CbDesc* cbs, int ncb, const int* nbps, const int* joboff, const int* order, int count,
                   const unsigned long long* mags, uint8_t* segs, int* cnt, long long* dist, int want_dist, const uint8_t* tabs);
#define T1_TABLES_BYTES 4608   // = CM2_TAB_BYTES (t1_cm2.h)
void launch_t1_tables(hipStream_t st, uint8_t* tabs);   // the table block of k_t1_cm2, once per context
void launch_gather(hipStream_t st, const uint8_t* outbuf, int outcap, const unsigned long long* offs, const int* lens, int ncb,
                   uint8_t* dst);
// the motion fields' component planes on the device (qsvc_compress_code): inter-level / bidirectional residues of the resident
// motion_filtered_t + the reference's byte cut (component c of a field = every 4th int16), one thread per output sample
#define MV_MAXLEV 16
struct MvLevels {
  const int16_t* filt[MV_MAXLEV];   // [fields][4][BY][BX] of level k (ascending temporal subbands)
  int fields[MV_MAXLEV];
  int first[MV_MAXLEV + 1];         // fields before level k
  int nlevels, fld;                 // fld = BY * BX
};
void launch_motion_planes(hipStream_t st, const MvLevels& ml, int16_t* dst, size_t n);
}  // namespace qsvc
