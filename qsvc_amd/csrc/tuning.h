// qsvc_amd/csrc/tuning.h -- the library's A/B switches (kernel forms kept for comparison; results never depend on them).
// Every context takes its own copy of the defaults when it is created (the environment variables QSVC_* are read THEN, once) and
// qsvc_set_tuning changes that copy; the launchers read the copy of the context whose call they serve (a thread-local pointer set
// by the entry point: calls on one context are serialised by the caller, different contexts may run on different threads).
#pragma once
#include <stdlib.h>
#include <string.h>

namespace qsvc {

struct Tuning {
  int mq_pipe = 1;        // QSVC_MQ_PIPE      1: two-wave MQ pipeline (k_t1_mq_pipe); 0: one wave per mq_lanes blocks (k_t1_mq)
  int mq_lds_small = 0;   // QSVC_MQ_LDS=small the pipeline with 25 instead of 40 KB of LDS per wave pair
  int mq_lanes = 64;      // QSVC_MQ_LANES     4, 8, 16, 32, 64 blocks per wave of the one-wave form
  int mq_sort = 1;        // QSVC_MQ_SORT      0: natural slot order
  int t1_parts = 0;       // QSVC_T1_PARTS     0 (by batch size: 3 from 65 536 slots up, else 1), 1 .. 4: Tier-1 in parts, MQ of part k beside modelling of part k + 1
  int t1_devlist = 1;     // QSVC_T1_DEVLIST   0: always the host's job list
  int pyr_tile = 1;       // QSVC_PYR_TILE     set (any value): thread-per-coefficient pyramid instead of the LDS tiles
  int me_piece = 1;       // QSVC_ME_PIECE     set: the border-rule search kernels for every block
  int pred_run = 1;       // QSVC_PRED_RUN     set: the general predict kernels
  int upd_run = 1;        // QSVC_UPD_RUN      set: the general update kernels
  int upd_scatter = 1;    // QSVC_UPD_SCATTER=0 the ordered gather by runs instead of the scatter tiles
  int dwt_pk = 1;         // QSVC_DWT_PK       1: register-blocked kernels (k_dwt53_rb, k_dwt97_rb); 2: 5/3 by the staged packed kernel (k_dwt53_pk); 3 / 4: k_dwt53_rb with one / two tiles per workgroup on every level; 0: the general tile kernel
  int defer_ib = 1;       // QSVC_DEFER_IB=0   the I/B decision of every level taken at once (one host round trip per level)
  int dec_lanes = 1;      // QSVC_DEC_LANES     1, 2, 4, 6, 8 code-blocks per wave of the Tier-1 decoder
  int sync_block = 1;     // QSVC_SYNC_BLOCK=0 the runtime's own (spinning) waits instead of query + sleep (qsvc_ctx.h ctx_sync)
  static Tuning from_env() {
    Tuning t;
    const char* e;
    if ((e = getenv("QSVC_MQ_PIPE"))) t.mq_pipe = atoi(e) != 0;
    if ((e = getenv("QSVC_MQ_LDS"))) t.mq_lds_small = !strcmp(e, "small");
    if ((e = getenv("QSVC_MQ_LANES"))) { const int v = atoi(e); t.mq_lanes = (v == 4 || v == 8 || v == 16 || v == 32) ? v : 64; }
    if ((e = getenv("QSVC_MQ_SORT"))) t.mq_sort = atoi(e) != 0;
    if ((e = getenv("QSVC_T1_PARTS"))) { const int v = atoi(e); t.t1_parts = v < 0 ? 0 : (v > 4 ? 4 : v); }
    if ((e = getenv("QSVC_T1_DEVLIST"))) t.t1_devlist = e[0] != '0';
    if (getenv("QSVC_PYR_TILE")) t.pyr_tile = 0;
    if (getenv("QSVC_ME_PIECE")) t.me_piece = 0;
    if (getenv("QSVC_PRED_RUN")) t.pred_run = 0;
    if (getenv("QSVC_UPD_RUN")) t.upd_run = 0;
    if ((e = getenv("QSVC_UPD_SCATTER"))) t.upd_scatter = e[0] != '0';
    if ((e = getenv("QSVC_DWT_PK"))) { const int v = atoi(e); t.dwt_pk = v < 0 || v > 4 ? 1 : v; }
    if ((e = getenv("QSVC_DEFER_IB"))) t.defer_ib = e[0] != '0';
    if ((e = getenv("QSVC_DEC_LANES"))) { const int v = atoi(e); t.dec_lanes = (v == 2 || v == 4 || v == 6 || v == 8) ? v : 1; }
    if ((e = getenv("QSVC_SYNC_BLOCK"))) t.sync_block = e[0] != '0';
    return t;
  }
  bool set(const char* name, int v) {
    struct { const char* n; int* p; } tab[] = {{"mq_pipe", &mq_pipe}, {"mq_lds_small", &mq_lds_small}, {"mq_lanes", &mq_lanes}, {"mq_sort", &mq_sort},
                                               {"t1_parts", &t1_parts}, {"t1_devlist", &t1_devlist}, {"pyr_tile", &pyr_tile}, {"me_piece", &me_piece},
                                               {"pred_run", &pred_run}, {"upd_run", &upd_run}, {"upd_scatter", &upd_scatter}, {"dwt_pk", &dwt_pk}, {"defer_ib", &defer_ib}, {"sync_block", &sync_block}, {"dec_lanes", &dec_lanes}};
    for (auto& t : tab)
      if (!strcmp(t.n, name)) {
        if (t.p == &mq_lanes && !(v == 4 || v == 8 || v == 16 || v == 32 || v == 64)) return false;
        if (t.p == &t1_parts && (v < 0 || v > 4)) return false;
        if (t.p == &dwt_pk && (v < 0 || v > 4)) return false;
        if (t.p == &dec_lanes && !(v == 1 || v == 2 || v == 4 || v == 6 || v == 8)) return false;
        *t.p = v;
        return true;
      }
    return false;
  }
};
// the switches of the context whose call is running on this thread (process defaults outside any call)
const Tuning& tuning();
struct TuningScope {
  const Tuning* prev;
  explicit TuningScope(const Tuning* t);
  ~TuningScope();
};

}  // namespace qsvc
