// qsvc_amd/csrc/mctf_kernels.h -- launch wrappers of mctf_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "mctf_threads.h"
#include "tuning.h"

namespace qsvc {
void launch_pyr_level_u8(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int P, int R, int C);
bool pyr_tile_applies(int W, int H);             // mctf_pyr.hip
void launch_pyr_ll_tile(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int P, int R, int C);
bool pyr_tile16_applies(int W, int ny, int nx);
void launch_pyr_ll_tile16(hipStream_t st, const int16_t* src, int16_t* dst, int W, int H, int P, int ny, int nx, int R, int C);  // LL (rows < R, columns < C) of the level below's ny x nx quadrant
void launch_pyr_level_s16(hipStream_t st, const int16_t* src, int16_t* dst, int W, int H, int P, int l, int R, int C);
// mctf_replay.hip: the literal analyse / synthesise replay of geometries with an odd size along the pyramid (Q5)
void launch_rp_load(hipStream_t st, const uint8_t* frames, size_t F, int W, int H, int16_t* dst, int slot0, int slot_step, int pic0, int pic_step, int count);
void launch_rp_dwt(hipStream_t st, int16_t* planes, int16_t* tmp, size_t plane_stride, int count, int W, int ny, int nx, int inverse);
void launch_me_search(hipStream_t st, const MeArgs& a, int mv_bound);
bool me_piece_applies(const MeArgs& a);          // mctf_me.hip
void launch_me_piece(hipStream_t st, const MeArgs& a, int ring_blocks);  // a.ring set: the ring role rides in the same launch
// chroma x2 interpolation (decorrelate.cpp:610-648, update.cpp:472-503) of pictures frames[mul*k], k < count, both components,
// into cup[k][c][H][W]: computed once per even picture instead of inside every prediction / update footprint
void launch_chroma_up(hipStream_t st, const uint8_t* frames, const LevelGeom& g, int mul, int count, uint8_t* cup);
void launch_predict(hipStream_t st, const PredArgs& a);
bool pred_run_applies(const PredArgs& a);        // mctf_pred.hip
void launch_predict_run(hipStream_t st, const PredArgs& a);
bool pred_fast_path(const PredArgs& a);          // launch_predict takes the mctf_pred.hip kernels (no interpolated chroma planes needed)
// --block_overlaping >= 2: predict through per-block analysis + picture synthesis (two [nB][3][H][W] int16 work buffers)
bool launch_predict_overlap(hipStream_t st, const PredArgs& a, int16_t* ppl0, int16_t* ppl1);
bool launch_ovl_blocks(hipStream_t st, const PredArgs& a, int16_t* ppl0, int16_t* ppl1);   // the prediction planes only (into ppl0)
// residue / histograms (or correlate) from full-resolution prediction planes ppl [nB][3][H][W]; clip: to 0..255 first
void launch_pred_finish(hipStream_t st, const PredArgs& a, const int16_t* ppl, int clip);
// mctf_subpel.hip (--subpixel_accuracy)
void launch_sp_load(hipStream_t st, const uint8_t* frames, size_t F, const SpGeom& g, int16_t* dst, const int16_t* src16, int slot0, int slot_step,
                    int pic0, int pic_step, int count, int is_ref);
void launch_sp_search(hipStream_t st, const SpGeom& g, const uint8_t* frames, size_t F, const int16_t* planes, int nB, int BY, int BX, int bsl, int bordl,
                      int lim, const int16_t* mv_in, int16_t* mv_out);
uint8_t* launch_sp_refs(hipStream_t st, const uint8_t* frames, const LevelGeom& g, int mul, int count, int a, uint8_t* up, uint8_t* tmp);
void launch_sp_predict(hipStream_t st, const LevelGeom& ge, const uint8_t* up, const int16_t* mv, int16_t* ppl);
void launch_sp_predict_one(hipStream_t st, const LevelGeom& ge, const uint8_t* up, const int16_t* mv, int i, int16_t* pp, int16_t* tmp, int H, int W, int a,
                           int16_t* out);
int16_t* launch_sp_down(hipStream_t st, int16_t* ppl, int16_t* tmp, int count, int H, int W, int a);
void launch_pred_margin(hipStream_t st, const PredArgs& a);  // mctf_margin.hip (Q7)
void launch_update(hipStream_t st, const UpdArgs& a);
bool upd_run_applies(const UpdArgs& a);          // mctf_upd.hip
void launch_update_run(hipStream_t st, const UpdArgs& a);
bool upd_fast_path(const UpdArgs& a);            // launch_update takes the mctf_upd.hip kernels (no interpolated chroma planes needed)
void launch_mv_maxabs(hipStream_t st, const int16_t* mv, size_t n, int* out);
void launch_dwt53t_level(hipStream_t st, const int16_t* src, int16_t* dst, int stride, int ny, int nx, int inverse);
}  // namespace qsvc
