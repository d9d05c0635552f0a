// qsvc_amd/csrc/qsvc_ctx.h -- the context behind the C ABI (internal).
#pragma once
#include <time.h>
#define QSVC_ALLB_MAX 4096
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "../../include/qsvc_hip.h"
#include "mctf_kernels.h"
#include "tuning.h"

#define QSVC_MAX_T 15

struct qsvc_ctx;

// p points DEVBUF_PAD bytes into the allocation and DEVBUF_PAD more follow the requested size: the MCTF kernels read whole
// 8-byte runs whose unused tail (or head) bytes may lie a few bytes outside the plane they belong to.
#define DEVBUF_PAD 256
struct DevBuf {  // grow-only device allocation: steady-state stage calls never hipMalloc
  void* p = nullptr;
  size_t cap = 0;
  int ensure(qsvc_ctx* c, size_t n);
  void release();
};

struct PinBuf {  // grow-only page-locked host allocation (small result / table transfers without a staging copy)
  void* p = nullptr;
  size_t cap = 0;
  bool ensure(size_t n) {
    if (n <= cap) return true;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = n + n / 4 + 256;
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return false;
    cap = want;
    return true;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Level {  // the device-resident "files" of temporal subband t
  DevBuf low, high, pred;        // low_t, high_t, prediction_even_t
  int low_n = 0, high_n = 0, pred_n = 0;
  DevBuf mv_raw, mv_filt;        // motion_t, motion_filtered_t
  int mv_raw_n = 0, mv_filt_n = 0, mv_bs = 0;
  DevBuf hist;                   // [nB][3][256] u32
  std::string types;             // frame_types_t
  DevBuf types_d; PinBuf types_pin; bool types_d_ok = false; bool types_allB = false;   // types_allB: every picture 'B', update reads the context's constant run of 'B';  // the same on the device (written by decorrelate; put_frame_types invalidates)
  int mv_bound = -1;             // >= 0: no component of motion_t / motion_filtered_t exceeds it (set by motion_estimate, cleared by put_motion)
  const void* halo[2] = {nullptr, nullptr};
  bool ragged = false;           // Q7: decorrelate ran on a picture the block grid does not tile
  // Deferred I/B decision (qsvc_api.hip, "deferred decisions"): decorrelate has queued the histograms' way to `hhist` and gone
  // on as if every picture were 'B'; the float entropy rule runs on the host at the next point that needs its outcome.
  bool hist_zeroed = false;      // motion_estimate's last launch has zeroed `hist` for this level's B pictures (consumed by decorrelate)
  bool pending = false;
  PinBuf hhist;                  // [nB][3][256] u32 on the host
  int dec_nB = 0, dec_BY = 0, dec_BX = 0;
};
struct StageCall {  // a stage call made while decisions were pending: replayed if one of them turns out 'I'
  int kind;         // 0 motion_estimate, 1 decorrelate, 2 update
  int t, P, bs, S, always_B;
  float uf;
};

struct Seg {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;  // event pairs, created once and reused
  size_t used = 0;                                    // pairs recorded since profiling was (re)enabled
};
struct Prof {
  Seg* cur = nullptr;
  void begin(qsvc_ctx* c, const char* name);
  void end(qsvc_ctx* c);
};

struct HostSeg {  // host-side wall time of a named stage (profiling on): reported next to the kernel segments as "host_<name>"
  std::string name;
  double ms = 0;
  int n = 0;
};

struct J2kState;
struct J2kDecState;

struct qsvc_ctx {
  int dev = 0;
  hipStream_t st = nullptr;
  int W = 0, H = 0;
  size_t F = 0;
  int keep_pred = 0;
  qsvc::Tuning tune;         // this context's A/B switches (tuning.h): the QSVC_* environment as it was at qsvc_init, then qsvc_set_tuning
  int opt_ov = 0, opt_bord = 0, opt_sub = 0;  // --block_overlaping, --border_size, --subpixel_accuracy (qsvc_set_mctf_options)
  std::vector<Level> lv;
  std::vector<DevBuf> snap;  // ME pyramid snapshots per DWT level
  DevBuf sp[2];              // --subpixel_accuracy (mctf_subpel.hip): the search's role planes at (H << a) x (W << a) + scratch; the
                             // up-sampled references of the prediction + scratch
  DevBuf rp[2];              // Q5 geometries (mctf_replay.hip): [3 nB][H][W] int16 role planes of the literal replay + scratch
  DevBuf margin;             // Q7 chroma margin state of the level being decorrelated
  DevBuf ovl[2];             // --block_overlaping: prediction planes in the Mallat layout, [nB][3][H][W] int16 each
  const void* cup_src = nullptr; int cup_mul = 0, cup_n = 0;  // what `cup` currently holds (reset by anything that writes pictures)
  DevBuf cup;                // chroma x2 interpolation of the even pictures of the level in work: [nB+1][2][H][W] u8
  DevBuf mvtmp[2], scal, tmp16[2];
  DevBuf allB;   // QSVC_ALLB_MAX bytes of 'B': frame_types_t of a level without 'I' pictures, already on the device
  DevBuf halo_tx[2], halo_rx[2];  // boundary records: [0] left neighbour, [1] right neighbour
  DevBuf halo_btx[2], halo_brx[2];  // batched exchange (qsvc_halo_batch_*): n records back to back, owned by the batch's first context
  hipEvent_t halo_ev = nullptr;   // this context's record is packed / the batch's records have arrived
  hipEvent_t sync_ev = nullptr;   // ctx_sync's marker
  uint8_t halo_tb[2][16];
  void* comm = nullptr;            // ncclComm_t (RCCL), line topology
  int rank = 0, nranks = 1;
  PinBuf hhist;                  // the B pictures' histograms on the host: [nB][3][256] u32 (immediate decisions)
  bool defer_ok = true;          // decorrelate may defer its I/B decisions (off while a replay runs; QSVC_DEFER_IB=0 turns it off)
  bool resolving = false;
  bool spec_call = false;        // the stage call in progress queued work on a speculative ('B') state: logged for replay once it has succeeded
  std::vector<int> pending;      // levels with pending decisions, in the order decorrelate was called
  std::vector<StageCall> calls;  // stage calls since the first pending decision
  std::string err;
  int profiling = 0;
  int t2_threads = 0;            // qsvc_set_tier2_threads: 0 = default (QSVC_T2_THREADS at the context's first encode, else 8)
  int t2_default = 0;            // that default, once read
  std::vector<Seg> segs;
  std::vector<HostSeg> hsegs;
  void host_time(const char* name, double ms);
  J2kState* j2k = nullptr;
  J2kState* j2k_mv = nullptr;    // a second encoder state for the motion-field planes (qsvc_j2k_encode_planes_s16): `mctf compress` codes them
                                 // between the analysis and the textures of every sequence, and the texture state's resident tables stay valid
  J2kDecState* j2kdec = nullptr;
  int fail(int code, const char* fmt, ...);
  int hipfail(hipError_t e, const char* what);
};

// CPU time of the calling thread spent in a scope, as a host segment (qsvc_get_timings; profiling on): tells waiting from working
struct HostCpu {
  qsvc_ctx* c; const char* name; timespec t0;
  HostCpu(qsvc_ctx* c_, const char* n) : c(c_), name(n) { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0); }
  ~HostCpu() {
    timespec t1; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1);
    c->host_time(name, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
  }
};
// Waiting for the device without burning a CPU.  The runtime's waits (hipStreamSynchronize, hipEventSynchronize) spin -- with
// hipDeviceScheduleBlockingSync still for most of a wait of a few milliseconds (measured: 4.7 ms of CPU for 8 ms of waiting) -- and a
// host that keeps a dozen sequences in flight per GPU, each with a session thread and Tier-2 workers that wait several times per step,
// then spends its CPUs on waiting: bench.py host_cpu showed 15.7 of the box's 16-CPU quota busy and the container throttled.
// tuning().sync_block (default): 100 us of spinning for the quick completions, then query + sleep in steps of 20 us (one to two hundred
// microseconds more latency per step of 23 ms, a fraction of the CPU); 0: the runtime's own wait.
inline hipError_t wait_event_sleeping(hipEvent_t ev) {
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) * 1000000000LL + (t1.tv_nsec - t0.tv_nsec) < 100000) continue;   // the first 100 us: spin
    const timespec nap = {0, 20000};
    nanosleep(&nap, nullptr);
  }
}
inline hipError_t ctx_sync(qsvc_ctx* c) {
  if (!c->tune.sync_block) return hipStreamSynchronize(c->st);
  hipError_t e;
  if (!c->sync_ev && (e = hipEventCreateWithFlags(&c->sync_ev, hipEventDisableTiming)) != hipSuccess) return e;
  if ((e = hipEventRecord(c->sync_ev, c->st)) != hipSuccess) return e;
  if ((e = wait_event_sleeping(c->sync_ev)) != hipSuccess) return e;
  return hipStreamSynchronize(c->st);   // (nothing left to wait for: lets the runtime retire the stream's finished commands)
}

// Take every pending I/B decision (one stream synchronisation); if a picture turns out 'I', redo what was computed from the
// speculative 'B'.  Every entry point that reads or replaces level state calls this first; returns QSVC_OK or an error.
extern "C" int qsvc_resolve(qsvc_ctx* c);  // internal (not in include/qsvc_hip.h)
// internal: the host half of the motion-field coding chain (qsvc_api.hip) for the entry points that code the planes (j2k_api.hip)
extern "C" int qsvc_motion_planes(qsvc_ctx* c, int nlevels, const int* ts, const int* fields, int block_size, const int16_t* filtered_host,
                                  int16_t* residues, std::vector<int16_t>& planes, int* BY, int* BX);
extern "C" void qsvc_motion_prm(qsvc_j2k_params* prm);
extern "C" int qsvc_j2k_decode_to_device(qsvc_ctx* c, int nstreams, const uint8_t* const* streams, const size_t* lens, int max_layers,
                                         uint8_t* dev_dst, const size_t* dev_offs, int fw, int fh);   // j2k_decode.hip
void j2k_release(qsvc_ctx* c);
void j2k_decode_release(qsvc_ctx* c);
