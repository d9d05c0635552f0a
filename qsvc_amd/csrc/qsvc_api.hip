// qsvc_amd/csrc/qsvc_api.hip -- the C ABI of include/qsvc_hip.h: context, device-resident
// "files", and the host side of the MCTF stages (stage orchestration, the float entropy I/B rule).
//
// Host-side logic restated from the reference (paths relative to /root/reference/trunk/src):
//   entropy()            entropy.cpp:20-34        (float accumulation, log in double)
//   I/B decision         decorrelate.cpp:798-819,945-1000
//   level chain          analyze_step.py:117-232
#include "qsvc_ctx.h"

#include <mutex>
#include <math.h>
#include <stdarg.h>
#include <algorithm>
#include <stdio.h>

using namespace qsvc;

namespace qsvc {
static thread_local const Tuning* t_tuning = nullptr;
const Tuning& tuning() {
  static const Tuning defaults = Tuning::from_env();
  return t_tuning ? *t_tuning : defaults;
}
TuningScope::TuningScope(const Tuning* t) : prev(t_tuning) { t_tuning = t; }
TuningScope::~TuningScope() { t_tuning = prev; }
}  // namespace qsvc

// ------------------------------------------------------------------------------------------------
int qsvc_ctx::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  err = buf;
  return code;
}
int qsvc_ctx::hipfail(hipError_t e, const char* what) {
  return fail(QSVC_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
#define HIPCHK(ctx, call)                                    \
  do {                                                       \
    hipError_t e_ = (call);                                  \
    if (e_ != hipSuccess) return (ctx)->hipfail(e_, #call);  \
  } while (0)

int DevBuf::ensure(qsvc_ctx* c, size_t n) {
  if (n <= cap) return QSVC_OK;
  release();
  void* base = nullptr;
  hipError_t e = hipMalloc(&base, n + 2 * DEVBUF_PAD);
  if (e != hipSuccess) return c->fail(QSVC_ERR_NOMEM, "hipMalloc(%zu): %s", n, hipGetErrorString(e));
  p = (uint8_t*)base + DEVBUF_PAD;
  cap = n;
  return QSVC_OK;
}
void DevBuf::release() {
  if (p) (void)hipFree((uint8_t*)p - DEVBUF_PAD);
  p = nullptr; cap = 0;
}

void qsvc_ctx::host_time(const char* name, double ms) {
  if (!profiling) return;
  for (auto& h : hsegs) if (h.name == name) { h.ms += ms; h.n++; return; }
  hsegs.push_back(HostSeg());
  hsegs.back().name = name; hsegs.back().ms = ms; hsegs.back().n = 1;
}
void Prof::begin(qsvc_ctx* c, const char* name) {
  if (!c->profiling) return;
  Seg* s = nullptr;
  for (auto& x : c->segs) if (x.name == name) { s = &x; break; }
  if (!s) { c->segs.push_back(Seg()); s = &c->segs.back(); s->name = name; }
  if (s->used == s->ev.size()) {  // the pool grows to the largest number of launches between two qsvc_set_profiling calls
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    s->ev.push_back({a, b});
  }
  hipEventRecord(s->ev[s->used].first, c->st);
  s->used++;
  cur = s;
}
void Prof::end(qsvc_ctx* c) {
  if (!c->profiling || !cur) return;
  hipEventRecord(cur->ev[cur->used - 1].second, c->st);
  cur = nullptr;
}

static void free_seq(qsvc_ctx* c) {
  c->cup_src = nullptr;
  for (auto& l : c->lv) {
    l.low.release(); l.high.release(); l.pred.release(); l.mv_raw.release(); l.mv_filt.release(); l.hist.release();
    l.low_n = l.high_n = l.pred_n = 0; l.mv_raw_n = l.mv_filt_n = 0; l.mv_bs = 0; l.types.clear();
    l.types_d.release(); l.types_pin.release(); l.types_d_ok = false; l.types_allB = false; l.mv_bound = -1;
    l.halo[0] = l.halo[1] = nullptr;
    l.pending = false; l.hhist.release();
  }
  c->pending.clear(); c->calls.clear();
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int qsvc_init(int device, qsvc_ctx** out) {
  if (!out) return QSVC_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0 || device < 0 || device >= n) return QSVC_ERR_HIP;  // no CPU fallback
  if (hipSetDevice(device) != hipSuccess) return QSVC_ERR_HIP;
  qsvc_ctx* c = new qsvc_ctx();
  c->dev = device;
  c->tune = Tuning::from_env();
  if (hipStreamCreate(&c->st) != hipSuccess) { delete c; return QSVC_ERR_HIP; }
  c->lv.resize(QSVC_MAX_T + 1);
  *out = c;
  return QSVC_OK;
}

void qsvc_destroy(qsvc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  (void)hipStreamSynchronize(c->st);
  free_seq(c);
  for (auto& b : c->snap) b.release();
  c->mvtmp[0].release(); c->mvtmp[1].release(); c->rp[0].release(); c->rp[1].release(); c->sp[0].release(); c->sp[1].release();
  for (int s_ = 0; s_ < 2; s_++) { c->halo_tx[s_].release(); c->halo_rx[s_].release(); c->halo_btx[s_].release(); c->halo_brx[s_].release(); }
  if (c->halo_ev) (void)hipEventDestroy(c->halo_ev);
  if (c->sync_ev) (void)hipEventDestroy(c->sync_ev);
  c->margin.release(); c->cup.release(); c->ovl[0].release(); c->ovl[1].release();
  if (c->comm) qsvc_comm_destroy(c);
  c->scal.release(); c->allB.release(); c->tmp16[0].release(); c->tmp16[1].release(); c->hhist.release();
  j2k_release(c);
  j2k_decode_release(c);
  for (auto& s : c->segs) for (auto& p : s.ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
  (void)hipStreamDestroy(c->st);
  delete c;
}

const char* qsvc_last_error(qsvc_ctx* c) { return c ? c->err.c_str() : "null context"; }
void* qsvc_stream(qsvc_ctx* c) { return c ? (void*)c->st : nullptr; }
int qsvc_sync(qsvc_ctx* c) {
  if (!c) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}

int qsvc_seq_config(qsvc_ctx* c, int W, int H, int keep_prediction) {
  if (!c) return QSVC_ERR_INVALID;
  c->pending.clear(); c->calls.clear();  // the state the pending decisions belong to is about to be freed
  for (auto& l_ : c->lv) l_.pending = false;
  if (W < 16 || H < 16 || W > 16384 || H > 16384) return c->fail(QSVC_ERR_INVALID, "bad geometry %dx%d", W, H);
  if ((W & 1) || (H & 1)) return c->fail(QSVC_ERR_UNSUPPORTED, "odd picture size %dx%d (4:2:0 needs even sizes)", W, H);
  (void)hipSetDevice(c->dev);
  HIPCHK(c, ctx_sync(c));
  free_seq(c);
  c->W = W; c->H = H; c->F = (size_t)W * H + 2 * (size_t)(W / 2) * (H / 2);
  c->keep_pred = keep_prediction;
  return QSVC_OK;
}

static int seq_sel(qsvc_ctx* c, int kind, int t, DevBuf** buf, int** cnt) {
  if (t < 0 || t > QSVC_MAX_T) return c->fail(QSVC_ERR_INVALID, "temporal subband %d out of range", t);
  Level& l = c->lv[t];
  switch (kind) {
    case QSVC_LOW: *buf = &l.low; *cnt = &l.low_n; return QSVC_OK;
    case QSVC_HIGH: *buf = &l.high; *cnt = &l.high_n; return QSVC_OK;
    case QSVC_PRED: *buf = &l.pred; *cnt = &l.pred_n; return QSVC_OK;
  }
  return c->fail(QSVC_ERR_INVALID, "unknown sequence kind %d", kind);
}

int qsvc_put_frames(qsvc_ctx* c, int kind, int t, int first, int count, int total, const uint8_t* src) {
  if (!c || !src) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  DevBuf* b; int* n;
  int r = seq_sel(c, kind, t, &b, &n);
  if (r) return r;
  if (first < 0 || count < 0 || first + count > total) return c->fail(QSVC_ERR_INVALID, "bad frame range");
  if ((r = b->ensure(c, (size_t)total * c->F))) return r;
  *n = total;
  c->cup_src = nullptr;  // pictures change: the interpolated chroma planes are stale
  HIPCHK(c, hipMemcpyAsync((uint8_t*)b->p + (size_t)first * c->F, src, (size_t)count * c->F, hipMemcpyHostToDevice, c->st));
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}
int qsvc_get_frames(qsvc_ctx* c, int kind, int t, int first, int count, uint8_t* dst) {
  if (!c || !dst) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  DevBuf* b; int* n;
  int r = seq_sel(c, kind, t, &b, &n);
  if (r) return r;
  if (first < 0 || count < 0 || first + count > *n) return c->fail(QSVC_ERR_STATE, "sequence kind %d t=%d holds %d pictures", kind, t, *n);
  HIPCHK(c, hipMemcpyAsync(dst, (uint8_t*)b->p + (size_t)first * c->F, (size_t)count * c->F, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}

// The decoder's counterpart of qsvc_put_frames: pictures [firsts[k], firsts[k] + counts[k]) of sequence (kinds[k], ts[k]) of
// totals[k] pictures straight from their code-streams (3 per picture: Y, U, V, as texture_expand_fb_j2k.py:153-181 reads
// <file>_{Y,U,V}_NNNN.j2c), all sequences of the call in ONE decoding batch, into the resident sequences without a trip through
// host memory.  Destination pointers are taken after every buffer has its size (a later ensure() may move an earlier buffer only if
// it is the same buffer: two entries of one call must not name the same sequence with different totals).
int qsvc_j2k_decode_multi(qsvc_ctx* c, int nseq, const int* kinds, const int* ts, const int* firsts, const int* counts, const int* totals,
                          const uint8_t* const* streams, const size_t* lens, int max_layers) {
  if (!c || !kinds || !ts || !firsts || !counts || !totals || !streams || !lens || nseq < 1) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  (void)hipSetDevice(c->dev);
  std::vector<DevBuf*> bufs(nseq);
  int npic = 0;
  for (int k = 0; k < nseq; k++) {
    int* n;
    int r = seq_sel(c, kinds[k], ts[k], &bufs[k], &n);
    if (r) return r;
    if (firsts[k] < 0 || counts[k] < 0 || firsts[k] + counts[k] > totals[k]) return c->fail(QSVC_ERR_INVALID, "bad frame range");
    if ((r = bufs[k]->ensure(c, (size_t)totals[k] * c->F))) return r;
    *n = totals[k];
    npic += counts[k];
  }
  c->cup_src = nullptr;
  if (!npic) return QSVC_OK;
  // one base pointer for the batch: offsets relative to the lowest buffer address
  uintptr_t base = ~(uintptr_t)0;
  for (int k = 0; k < nseq; k++) if (counts[k]) base = std::min(base, (uintptr_t)bufs[k]->p);
  std::vector<size_t> offs((size_t)3 * npic);
  const size_t N = (size_t)c->W * c->H, Nc = (size_t)(c->W / 2) * (c->H / 2);
  size_t q = 0;
  for (int k = 0; k < nseq; k++)
    for (int i = 0; i < counts[k]; i++, q += 3) {
      const size_t f = (size_t)((uintptr_t)bufs[k]->p - base) + (size_t)(firsts[k] + i) * c->F;
      offs[q] = f; offs[q + 1] = f + N; offs[q + 2] = f + N + Nc;
    }
  return qsvc_j2k_decode_to_device(c, 3 * npic, streams, lens, max_layers, (uint8_t*)base, offs.data(), c->W, c->H);
}
int qsvc_j2k_decode_frames(qsvc_ctx* c, int kind, int t, int first, int count, int total, const uint8_t* const* streams, const size_t* lens,
                           int max_layers) {
  return qsvc_j2k_decode_multi(c, 1, &kind, &t, &first, &count, &total, streams, lens, max_layers);
}

int qsvc_put_motion(qsvc_ctx* c, int kind, int t, int nB, int bs, const int16_t* mv) {
  if (!c || !mv || t < 1 || t > QSVC_MAX_T || bs <= 0 || nB < 0) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  Level& l = c->lv[t];
  size_t bytes = (size_t)nB * 4 * (c->H / bs) * (c->W / bs) * sizeof(int16_t);
  DevBuf& b = kind == QSVC_MOTION_RAW ? l.mv_raw : l.mv_filt;
  int r = b.ensure(c, bytes ? bytes : 16);
  if (r) return r;
  (kind == QSVC_MOTION_RAW ? l.mv_raw_n : l.mv_filt_n) = nB;
  l.mv_bs = bs;
  l.mv_bound = -1;
  HIPCHK(c, hipMemcpyAsync(b.p, mv, bytes, hipMemcpyHostToDevice, c->st));
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}
int qsvc_get_motion(qsvc_ctx* c, int kind, int t, int b_pictures, int block_size, int16_t* mv) {
  if (!c || !mv || t < 1 || t > QSVC_MAX_T || b_pictures < 0) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  Level& l = c->lv[t];
  int nB = kind == QSVC_MOTION_RAW ? l.mv_raw_n : l.mv_filt_n;
  if (!l.mv_bs) return c->fail(QSVC_ERR_STATE, "no motion for t=%d", t);
  // the caller sized its buffer from (b_pictures, block_size): both must describe what is resident
  if (block_size != l.mv_bs || b_pictures > nB)
    return c->fail(QSVC_ERR_STATE, "motion of t=%d holds %d fields at block size %d (asked for %d at %d)", t, nB, l.mv_bs, b_pictures, block_size);
  DevBuf& b = kind == QSVC_MOTION_RAW ? l.mv_raw : l.mv_filt;
  size_t bytes = (size_t)b_pictures * 4 * (c->H / l.mv_bs) * (c->W / l.mv_bs) * sizeof(int16_t);
  if (!bytes) return QSVC_OK;
  HIPCHK(c, hipMemcpyAsync(mv, b.p, bytes, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}
int qsvc_put_frame_types(qsvc_ctx* c, int t, int nB, const char* types) {
  if (!c || !types || t < 1 || t > QSVC_MAX_T || nB < 0) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  c->lv[t].types.assign(types, types + nB);
  c->lv[t].types_d_ok = false; c->lv[t].types_allB = false;
  return QSVC_OK;
}
int qsvc_get_frame_types(qsvc_ctx* c, int t, int b_pictures, char* types) {
  if (!c || !types || t < 1 || t > QSVC_MAX_T || b_pictures < 0) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  if ((size_t)b_pictures > c->lv[t].types.size())
    return c->fail(QSVC_ERR_STATE, "frame_types_%d holds %zu entries (asked for %d)", t, c->lv[t].types.size(), b_pictures);
  memcpy(types, c->lv[t].types.data(), (size_t)b_pictures);
  return QSVC_OK;
}
int qsvc_get_histograms(qsvc_ctx* c, int t, int b_pictures, uint32_t* hist) {
  if (!c || !hist || t < 1 || t > QSVC_MAX_T || b_pictures < 0) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  Level& l = c->lv[t];
  if (!l.hist.p) return c->fail(QSVC_ERR_STATE, "no histograms for t=%d", t);
  if (b_pictures > l.high_n) return c->fail(QSVC_ERR_STATE, "histograms of t=%d cover %d pictures (asked for %d)", t, l.high_n, b_pictures);
  if (!b_pictures) return QSVC_OK;
  HIPCHK(c, hipMemcpyAsync(hist, l.hist.p, (size_t)b_pictures * 3 * 256 * 4, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}

// ------------------------------------------------------------------------------------------------
static int ilog2_exact(int s) {
  int l = 0;
  while ((1 << (l + 1)) <= s) l++;
  return (1 << l) == s ? l : -1;
}
static int level_geom(qsvc_ctx* c, int t, int P, int bs, int S, LevelGeom* g) {
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  if (t < 1 || t > QSVC_MAX_T) return c->fail(QSVC_ERR_INVALID, "temporal subband %d out of range", t);
  if (bs < 2 || bs > 256) return c->fail(QSVC_ERR_INVALID, "block_size %d", bs);
  if (P < 1) return c->fail(QSVC_ERR_INVALID, "pictures %d", P);
  // every B picture 2i+1 needs the even pictures 2i and 2i+2 (motion_estimate.cpp:792-820): 2*(P/2)+1 pictures in all
  if (c->lv[t - 1].low_n < 2 * (P / 2) + 1 || c->lv[t - 1].low_n < P)
    return c->fail(QSVC_ERR_STATE, "low_%d holds %d pictures, %d needed", t - 1, c->lv[t - 1].low_n, std::max(P, 2 * (P / 2) + 1));
  // rows/cols beyond BY*bs are never predicted by the reference (decorrelate.cpp:508-509,562-567; trunk/readme.txt:119-125):
  // Q7, reproduced by the margin state of PredArgs (mctf_threads.h)
  if (c->H < bs || c->W < bs) return c->fail(QSVC_ERR_INVALID, "picture %dx%d smaller than block_size %d", c->W, c->H, bs);
  g->W = c->W; g->H = c->H; g->bs = bs; g->BY = c->H / bs; g->BX = c->W / bs; g->S = S; g->P = P; g->nB = P / 2; g->F = c->F;
  g->ov = c->opt_ov; g->bord = c->opt_bord; g->sub = c->opt_sub;
  return QSVC_OK;
}

int qsvc_set_tuning(qsvc_ctx* c, const char* name, int value) {
  if (!c || !name) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }   // (a pending replay runs with the switches its calls were made with)
  if (!c->tune.set(name, value)) return c->fail(QSVC_ERR_INVALID, "unknown tuning switch '%s' or value %d out of range", name, value);
  return QSVC_OK;
}

int qsvc_set_mctf_options(qsvc_ctx* c, int block_overlaping, int border_size, int subpixel_accuracy) {
  if (!c) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  if (block_overlaping < 0 || border_size < 0 || subpixel_accuracy < 0) return c->fail(QSVC_ERR_INVALID, "negative MCTF option");
  if (subpixel_accuracy > 3)
    return c->fail(QSVC_ERR_UNSUPPORTED, "subpixel_accuracy %d: at most 3 (planes of 8 x 8 times the picture) on this path", subpixel_accuracy);
  if (block_overlaping > 1 && (block_overlaping & (block_overlaping - 1)))
    return c->fail(QSVC_ERR_UNSUPPORTED, "block_overlaping %d is not a power of two", block_overlaping);
  if (border_size > 64) return c->fail(QSVC_ERR_UNSUPPORTED, "border_size %d", border_size);
  c->opt_ov = block_overlaping; c->opt_bord = border_size; c->opt_sub = subpixel_accuracy;
  return QSVC_OK;
}

// --subpixel_accuracy (mctf_subpel.hip): the a sub-pixel levels of me_for_image (motion_estimate.cpp:361-406) behind the whole-pixel
// pyramid, whose level-0 vectors are in mvtmp[0].  Role planes at the up-sampled size: slot 3i PREV, 3i+1 predicted, 3i+2 NEXT of
// B picture i; state0 = the level-0 state of the literal replay (same slots) or nullptr: the pictures themselves.
static int me_subpel(qsvc_ctx* c, const LevelGeom& g, Level& lv, const uint8_t* frames, int L, const int16_t* state0) {
  const int a = g.sub, planes = 3 * g.nB;
  SpGeom sg;
  sg.H = g.H; sg.W = g.W; sg.a = a; sg.Ha = g.H << a; sg.Wa = g.W << a; sg.b = g.S + g.bord; sg.ba = sg.b << a;
  const long chunk = ((2L * (sg.Wa + 2 * sg.ba) + 8 + 15) & ~15L) | 1;  // glibc request2size | PREV_INUSE (texture.cpp:36-38)
  sg.c0 = (int)(int16_t)(chunk & 0xFFFF); sg.c1 = (int)(int16_t)((chunk >> 16) & 0xFFFF);
  const size_t N = (size_t)sg.Ha * sg.Wa, bytes = (size_t)planes * N * sizeof(int16_t);
  int r;
  if ((r = c->sp[0].ensure(c, bytes))) return r;
  if ((r = c->sp[1].ensure(c, bytes))) return r;
  int16_t* A = (int16_t*)c->sp[0].p; int16_t* T = (int16_t*)c->sp[1].p;
  Prof pf;
  pf.begin(c, "me_subpel");
  HIPCHK(c, hipMemsetAsync(A, 0, bytes, c->st));  // the planes' interior is zeroed before every read (motion_estimate.cpp:723-741,803-807)
  launch_sp_load(c->st, frames, g.F, sg, A, state0, 0, 3, 0, 2, g.nB, 1);
  launch_sp_load(c->st, frames, g.F, sg, A, state0, 1, 3, 1, 2, g.nB, 0);
  launch_sp_load(c->st, frames, g.F, sg, A, state0, 2, 3, 2, 2, g.nB, 1);
  int bound = (2 << L) - 1;
  const int lim = g.S << a;
  for (int s = 1; s <= a; s++) {
    launch_rp_dwt(c->st, A, T, N, planes, sg.Wa, g.H << s, g.W << s, 1);  // motion_estimate.cpp:368-370
    const int16_t* in = (const int16_t*)c->mvtmp[(s - 1) & 1].p;
    int16_t* out = s == a ? (int16_t*)lv.mv_raw.p : (int16_t*)c->mvtmp[s & 1].p;
    launch_sp_search(c->st, sg, frames, g.F, A, g.nB, g.BY, g.BX, g.bs << s, g.bord >> s, lim, in, out);
    bound = imin(2 * bound, lim) + 1;
  }
  pf.end(c);
  lv.mv_bound = bound;
  lv.hist_zeroed = false;
  HIPCHK(c, hipGetLastError());
  return QSVC_OK;
}

// Q5 (mctf_replay.hip): the search pyramid of a picture with an odd size along the way, as me_for_image walks it
// (motion_estimate.cpp:283-358): L analysis levels at floor-halved sizes, the search, then per level one synthesis at the
// ceil-halved size and the next search, on one int16 plane per (B picture, role).
static int me_replay(qsvc_ctx* c, const LevelGeom& g, Level& lv, const uint8_t* frames, int L) {
  const size_t N = (size_t)g.H * g.W, fld = (size_t)g.BY * g.BX, mvbytes = (size_t)g.nB * 4 * fld * sizeof(int16_t);
  const int planes = 3 * g.nB;
  int r;
  if ((r = c->rp[0].ensure(c, (size_t)planes * N * sizeof(int16_t)))) return r;
  if ((r = c->rp[1].ensure(c, (size_t)planes * N * sizeof(int16_t)))) return r;
  if ((r = c->mvtmp[0].ensure(c, mvbytes))) return r;
  if ((r = c->mvtmp[1].ensure(c, mvbytes))) return r;
  int16_t* A = (int16_t*)c->rp[0].p; int16_t* T = (int16_t*)c->rp[1].p;
  Prof pf;
  pf.begin(c, "me_pyramid");
  // pre-pass: PREV of B picture i >= 1 is even picture 2i after the whole walk it went through as NEXT of picture i-1
  // (the reference swaps its two reference planes, motion_estimate.cpp:902-906, and reloads only the new NEXT)
  if (g.nB > 1) {
    launch_rp_load(c->st, frames, g.F, g.W, g.H, A, 3, 3, 2, 2, g.nB - 1);
    for (int k = 0; k < L; k++) launch_rp_dwt(c->st, A + 3 * N, T + 3 * N, 3 * N, g.nB - 1, g.W, imax(g.H >> k, 1), imax(g.W >> k, 1), 0);
    for (int l = L - 1; l >= 0; l--) launch_rp_dwt(c->st, A + 3 * N, T + 3 * N, 3 * N, g.nB - 1, g.W, desp(g.H, l), desp(g.W, l), 1);
  }
  launch_rp_load(c->st, frames, g.F, g.W, g.H, A, 0, 3, 0, 2, 1);        // PREV of picture 0: even picture 0 as read
  launch_rp_load(c->st, frames, g.F, g.W, g.H, A, 1, 3, 1, 2, g.nB);     // predicted pictures
  launch_rp_load(c->st, frames, g.F, g.W, g.H, A, 2, 3, 2, 2, g.nB);     // NEXT references
  for (int k = 0; k < L; k++) launch_rp_dwt(c->st, A, T, N, planes, g.W, imax(g.H >> k, 1), imax(g.W >> k, 1), 0);  // dwt2d.cpp:78-79
  pf.end(c);
  const int16_t* prev = nullptr;
  int bound = 0;
  for (int l = L; l >= 0; l--) {
    if (l < L) {
      pf.begin(c, "me_pyramid");
      launch_rp_dwt(c->st, A, T, N, planes, g.W, desp(g.H, l), desp(g.W, l), 1);   // motion_estimate.cpp:307-309
      pf.end(c);
    }
    MeArgs a;
    a.g = g; a.frames = frames; a.snap = A; a.snap_pic_stride = N; a.pic_mul = 3;
    a.mv_prev = prev; a.mv_out = l == 0 && !g.sub ? (int16_t*)lv.mv_raw.p : (int16_t*)c->mvtmp[l & 1].p;
    a.l = l; a.nby = desp(g.BY, l); a.nbx = desp(g.BX, l);
    a.zero_words = nullptr; a.zero_count = 0;
    pf.begin(c, l == 0 ? "me_search_l0" : "me_search_coarse");
    launch_me_search(c->st, a, bound);
    pf.end(c);
    prev = a.mv_out;
    bound = imin(2 * (bound + 1), g.S);
  }
  lv.hist_zeroed = false;
  HIPCHK(c, hipGetLastError());
  return QSVC_OK;
}

static int motion_estimate_impl(qsvc_ctx* c, int t, int P, int bs, int S) {
  (void)hipSetDevice(c->dev);
  TuningScope tscope(&c->tune);
  LevelGeom g;
  int r = level_geom(c, t, P, bs, S, &g);
  if (r) return r;
  int lg = ilog2_exact(S);
  if (lg < 0 || S > 128) return c->fail(QSVC_ERR_UNSUPPORTED, "search_range %d is not a power of two <= 128", S);
  if (!c->pending.empty() && !c->resolving) c->spec_call = true;
  int L = lg - 1 < 0 ? 0 : lg - 1;  // motion_estimate.cpp:277 (search ranges 1 and 2: dwt_levels <= 0, one search on the pictures)
  // the level-l search state equals the l-level analysis only while every intermediate size is even; otherwise (Q5) the
  // reference's own analyse / synthesise walk is replayed literally (me_replay below)
  const bool replay = (g.H % (1 << L)) || (g.W % (1 << L));
  if (replay && ((g.H >> (L - 1)) < 2 || (g.W >> (L - 1)) < 2))
    return c->fail(QSVC_ERR_UNSUPPORTED, "picture %dx%d too small for %d pyramid levels (search_range %d)", g.W, g.H, L, S);
  if (g.sub) {  // mctf_subpel.hip
    const int b = S + g.bord;
    if (S < 2) return c->fail(QSVC_ERR_UNSUPPORTED, "subpixel_accuracy with search_range 1: the reference's motion_estimate reads outside its planes and crashes");
    if ((S << g.sub) > 126) return c->fail(QSVC_ERR_UNSUPPORTED, "search_range %d << subpixel_accuracy %d: vectors leave the 256-bin motion histogram of decorrelate.cpp:805-812", S, g.sub);
    if ((g.W << g.sub) > 16384 || (g.H << g.sub) > 16384) return c->fail(QSVC_ERR_UNSUPPORTED, "picture %dx%d << %d exceeds the reference's line buffer (16384)", g.W, g.H, g.sub);
    if (((g.H - b) << g.sub) < g.H + b)  // the band of the small picture's border would reach the planes' unshifted rows
      return c->fail(QSVC_ERR_UNSUPPORTED, "search_range %d + border_size %d too large for a %d-row picture with subpixel_accuracy %d", S, g.bord, g.H, g.sub);
  }
  if (desp(g.BY, L) * bs + S + g.bord > 32767) return c->fail(QSVC_ERR_UNSUPPORTED, "picture too large");
  Level& lv = c->lv[t];
  size_t fld = (size_t)g.BY * g.BX, mvbytes = (size_t)g.nB * 4 * fld * sizeof(int16_t);
  if ((r = lv.mv_raw.ensure(c, mvbytes ? mvbytes : 16))) return r;
  lv.mv_raw_n = g.nB; lv.mv_bs = bs;
  lv.mv_bound = (2 << L) - 1;  // one step of +-1 per pyramid level, doubled from level to level
  if (g.nB == 0) return QSVC_OK;
  const uint8_t* frames = (const uint8_t*)c->lv[t - 1].low.p;
  size_t N = (size_t)g.H * g.W;
  if (replay) {
    if ((r = me_replay(c, g, lv, frames, L))) return r;
    return g.sub ? me_subpel(c, g, lv, frames, L, (const int16_t*)c->rp[0].p) : QSVC_OK;
  }
  if ((int)c->snap.size() < L + 1) c->snap.resize(L + 1);
  Prof pf;
  pf.begin(c, "me_pyramid");
  for (int l = 1; l <= L; l++) {
    if ((r = c->snap[l].ensure(c, (size_t)P * N * sizeof(int16_t)))) return r;
    int R = imin(g.H, imax(desp(g.BY, l) * bs + S + g.bord, g.H >> l)), C = imin(g.W, imax(desp(g.BX, l) * bs + S + g.bord, g.W >> l));
    if (l == 1) launch_pyr_level_u8(c->st, frames, g.F, g.W, g.H, (int16_t*)c->snap[1].p, P, R, C);
    else launch_pyr_level_s16(c->st, (const int16_t*)c->snap[l - 1].p, (int16_t*)c->snap[l].p, g.W, g.H, P, l, R, C);
  }
  pf.end(c);
  if (L >= 1 || g.sub) {
    if ((r = c->mvtmp[0].ensure(c, mvbytes))) return r;
    if ((r = c->mvtmp[1].ensure(c, mvbytes))) return r;
  }
  const int16_t* prev = nullptr;
  int bound = 0;  // max |component| of the vectors a level starts from: one +-1 step per level, doubled (and clamped) in between
  for (int l = L; l >= 0; l--) {
    MeArgs a;
    a.g = g; a.frames = frames; a.snap = l ? c->snap[l].p : nullptr; a.snap_pic_stride = N;
    a.mv_prev = prev; a.mv_out = l == 0 && !g.sub ? (int16_t*)lv.mv_raw.p : (int16_t*)c->mvtmp[l & 1].p;  // (sub-pixel levels follow)
    a.l = l; a.nby = desp(g.BY, l); a.nbx = desp(g.BX, l);
    a.zero_words = nullptr; a.zero_count = 0;
    if (l == 0 && me_piece_applies(a)) {  // the level-0 launch also zeroes the level's histograms for the predict kernel behind it
      if ((r = lv.hist.ensure(c, (size_t)g.nB * 3 * 256 * 4))) return r;
      a.zero_words = (unsigned*)lv.hist.p; a.zero_count = g.nB * 3 * 256;
    }
    pf.begin(c, l == 0 ? "me_search_l0" : "me_search_coarse");
    launch_me_search(c->st, a, bound);
    pf.end(c);
    prev = a.mv_out;
    bound = imin(2 * (bound + 1), S);
    if (l == 0) lv.hist_zeroed = a.zero_words != nullptr && c->tune.me_piece;
  }
  HIPCHK(c, hipGetLastError());
  return g.sub ? me_subpel(c, g, lv, frames, L, nullptr) : QSVC_OK;
}

// chroma x2 planes of the even pictures frames[mul*k], k <= nB (mctf_kernels.hip: k_chroma_up)
static int make_cup(qsvc_ctx* c, const LevelGeom& g, const uint8_t* frames, int mul, const uint8_t** out) {
  int r = c->cup.ensure(c, (size_t)(g.nB + 1) * 2 * g.H * g.W);
  if (r) return r;
  // decorrelate and update of one level interpolate the same even pictures: the second call finds the planes in place
  if (!(c->cup_src == frames && c->cup_mul == mul && c->cup_n == g.nB + 1)) {
    launch_chroma_up(c->st, frames, g, mul, g.nB + 1, (uint8_t*)c->cup.p);
    c->cup_src = frames; c->cup_mul = mul; c->cup_n = g.nB + 1;
  }
  *out = (const uint8_t*)c->cup.p;
  return QSVC_OK;
}

static int predict_overlapped(qsvc_ctx* c, const LevelGeom& g, const PredArgs& a) {
  const size_t bytes = (size_t)g.nB * 3 * g.H * g.W * sizeof(int16_t);
  int r;
  if ((r = c->ovl[0].ensure(c, bytes))) return r;
  if ((r = c->ovl[1].ensure(c, bytes))) return r;
  if (!launch_predict_overlap(c->st, a, (int16_t*)c->ovl[0].p, (int16_t*)c->ovl[1].p))
    return c->fail(QSVC_ERR_UNSUPPORTED, "block_overlaping %d with block size %d on a %dx%d picture is outside this path (power of two dividing "
                   "the block size, block + 2 x overlap <= %d, picture tiled by the block grid)", g.ov, g.bs, g.W, g.H, 96);
  return QSVC_OK;
}

// --subpixel_accuracy in decorrelate / correlate (mctf_subpel.hip): references interpolated a times, predict() at the up-sampled
// resolution, the prediction brought down through the low-low bands of a analysis levels, then residue / histograms as usual.
static int predict_subpel(qsvc_ctx* c, const LevelGeom& g, const PredArgs& a) {
  const int s = g.sub, Ha = g.H << s, Wa = g.W << s;
  const bool ragged = g.H % g.bs || g.W % g.bs;
  if (ragged && g.ov >= 1)
    return c->fail(QSVC_ERR_UNSUPPORTED, "subpixel_accuracy with block_overlaping on a %dx%d picture the block grid (%d) does not tile is not "
                   "implemented on this path", g.W, g.H, g.bs);
  if ((Wa > 16384 || Ha > 16384)) return c->fail(QSVC_ERR_UNSUPPORTED, "picture %dx%d << %d exceeds the reference's line buffer (16384)", g.W, g.H, s);
  LevelGeom ge = g;
  ge.W = Wa; ge.H = Ha; ge.bs = g.bs << s; ge.S = g.S << s; ge.ov = g.ov << s; ge.sub = 0;
  const size_t upb = (size_t)(g.nB + 1) * 3 * Ha * Wa, pb = (size_t)g.nB * 3 * Ha * Wa * sizeof(int16_t);
  int r;
  if ((r = c->sp[0].ensure(c, upb))) return r;
  if ((r = c->sp[1].ensure(c, upb))) return r;
  if ((r = c->ovl[0].ensure(c, pb))) return r;
  if ((r = c->ovl[1].ensure(c, pb))) return r;
  const uint8_t* up = launch_sp_refs(c->st, a.frames, g, 2, g.nB + 1, s, (uint8_t*)c->sp[0].p, (uint8_t*)c->sp[1].p);
  int16_t* p0 = (int16_t*)c->ovl[0].p; int16_t* p1 = (int16_t*)c->ovl[1].p;
  if (ragged) {
    // Q7 at the up-sampled size: the prediction planes carry state from picture to picture, so the level is walked literally, one B
    // picture after the other, on three persistent planes (mctf_subpel.hip: launch_sp_predict_one)
    const size_t N = (size_t)Ha * Wa, n1 = (size_t)g.H * g.W;
    if ((r = c->margin.ensure(c, (3 * N + (size_t)g.nB * 3 * n1) * sizeof(int16_t)))) return r;
    int16_t* pp = (int16_t*)c->margin.p; int16_t* outp = pp + 3 * N;
    HIPCHK(c, hipMemsetAsync(pp, 0, 3 * N * sizeof(int16_t), c->st));
    for (int i = 0; i < g.nB; i++) launch_sp_predict_one(c->st, ge, up, a.mv, i, pp, p0, g.H, g.W, s, outp + (size_t)i * 3 * n1);
    PredArgs af = a;
    af.margin = nullptr;
    launch_pred_finish(c->st, af, outp, 0);
    return QSVC_OK;
  }
  if (ge.ov >= 2) {  // predict(block_overlaping << a, block_size << a, ...) (decorrelate.cpp:828-839)
    PredArgs ae = a;
    ae.g = ge; ae.up = up; ae.cup = nullptr; ae.margin = nullptr;
    if (!launch_ovl_blocks(c->st, ae, p0, p1))
      return c->fail(QSVC_ERR_UNSUPPORTED, "block_overlaping %d with block size %d at subpixel_accuracy %d is outside this path (block + 2 x overlap, "
                     "both up-sampled, <= %d)", g.ov, g.bs, s, 96);
  } else launch_sp_predict(c->st, ge, up, a.mv, p0);
  const int16_t* res = launch_sp_down(c->st, p0, p1, g.nB * 3, g.H, g.W, s);
  launch_pred_finish(c->st, a, res, 0);
  return QSVC_OK;
}

// entropy() of entropy.cpp:20-34, op for op (float accumulate, double log).
static float entropy256(const uint32_t* count) {
  float e = 0.0f;
  int total = 0;
  for (int i = 0; i < 256; i++) total += (int)count[i];
  for (int i = 0; i < 256; i++)
    if (count[i]) {
      float prob = (float)(int)count[i] / total;
      e += prob * (float)(log(prob) / log(2.0));
    }
  return -e;
}

// decorrelate.cpp:964-979 on one picture's three histograms: 'I' if the picture itself looks cheaper than residue + motion
static bool picture_is_I(const uint32_t* h, int H, int W, int BY, int BX) {
  float predicted_entropy = entropy256(h), residue_entropy = entropy256(h + 256), motion_entropy = entropy256(h + 512);
  int predicted_size = (int)(predicted_entropy * (float)H * (float)W);
  int residue_size = (int)(residue_entropy * (float)H * (float)W);
  int motion_size = (int)(motion_entropy * (float)BY * (float)BX);
  return predicted_size <= residue_size + motion_size;
}
// 'I' pictures of level t (decorrelate.cpp:982-1000): the odd picture itself is the "high" picture, its motion is zero
static int apply_I_pictures(qsvc_ctx* c, int t) {
  Level& lv = c->lv[t];
  const uint8_t* frames = (const uint8_t*)c->lv[t - 1].low.p;
  const size_t fld = (size_t)lv.dec_BY * lv.dec_BX;
  for (int i = 0; i < lv.dec_nB; i++)
    if (lv.types[i] == 'I') {
      HIPCHK(c, hipMemcpyAsync((uint8_t*)lv.high.p + (size_t)i * c->F, frames + (size_t)(2 * i + 1) * c->F, c->F, hipMemcpyDeviceToDevice, c->st));
      HIPCHK(c, hipMemsetAsync((int16_t*)lv.mv_filt.p + (size_t)i * 4 * fld, 0, 4 * fld * sizeof(int16_t), c->st));
    }
  return QSVC_OK;
}
static int types_to_device_pinned(qsvc_ctx* c, Level& lv, int nB) {
  lv.types_d_ok = false; lv.types_allB = false;
  if (!nB) return QSVC_OK;
  if (!lv.types_pin.ensure((size_t)nB)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc(%d)", nB);
  int r = lv.types_d.ensure(c, (size_t)nB + 16);
  if (r) return r;
  memcpy(lv.types_pin.p, lv.types.data(), (size_t)nB);
  HIPCHK(c, hipMemcpyAsync(lv.types_d.p, lv.types_pin.p, (size_t)nB, hipMemcpyHostToDevice, c->st));
  lv.types_d_ok = true;
  return QSVC_OK;
}
// ---- deferred decisions -------------------------------------------------------------------------------------------------
// The reference decides I or B per picture with a float entropy estimate (decorrelate.cpp:945-979) whose log() must be the
// host's to stay bit-exact, so the decision cannot move to the device.  It does not have to stall the device either: nearly
// every picture is 'B', so decorrelate queues the histograms' copy to the host and the level chain carries on with 'B'
// (update, the next level's search, ...) without a single host round trip.  The first call that needs a decision's outcome --
// reading frame types, pictures or motion, a halo export, the JPEG 2000 stage, qsvc_sync -- lands here: one synchronisation,
// the entropy rule for every pending level in order, and, if a picture turns out 'I', the reference's 'I' handling followed by a
// replay of every stage call that consumed the speculative state (with immediate decisions).  Results are those of the
// immediate path in every case; only the number of host round trips differs.
int qsvc_resolve(qsvc_ctx* c) {
  if (!c) return QSVC_ERR_INVALID;
  if (c->resolving) return QSVC_OK;
  if (c->pending.empty()) { c->calls.clear(); return QSVC_OK; }
  (void)hipSetDevice(c->dev);
  HIPCHK(c, ctx_sync(c));
  c->resolving = true;
  int redo_level = 0, r = QSVC_OK;
  for (size_t k = 0; k < c->pending.size() && !redo_level; k++) {
    const int t = c->pending[k];
    Level& lv = c->lv[t];
    lv.pending = false;
    bool anyI = false;
    for (int i = 0; i < lv.dec_nB; i++)
      if (picture_is_I((const uint32_t*)lv.hhist.p + (size_t)i * 3 * 256, c->H, c->W, lv.dec_BY, lv.dec_BX)) { lv.types[i] = 'I'; anyI = true; }
    if (anyI) {
      redo_level = t;
      if (!r) r = apply_I_pictures(c, t);
      if (!r) r = types_to_device_pinned(c, lv, lv.dec_nB);
    }
  }
  for (int t : c->pending) c->lv[t].pending = false;
  c->pending.clear();
  std::vector<StageCall> calls;
  calls.swap(c->calls);
  if (redo_level && !r) {
    // everything queued after level redo_level's decorrelate used the speculative 'B' state: run it again, deciding at once
    size_t from = calls.size();
    for (size_t k = 0; k < calls.size(); k++)
      if (calls[k].kind == 1 && calls[k].t == redo_level) { from = k + 1; break; }
    for (size_t k = from; k < calls.size() && !r; k++) {
      const StageCall& q = calls[k];
      if (q.kind == 0) r = qsvc_motion_estimate(c, q.t, q.P, q.bs, q.S);
      else if (q.kind == 1) r = qsvc_decorrelate(c, q.t, q.P, q.bs, q.S, q.always_B);
      else r = qsvc_update(c, q.t, q.P, q.bs, q.uf);
    }
  }
  c->resolving = false;
  if (r && redo_level) {
    // a replay failed: what the levels from redo_level up hold was computed from the speculative state -- nothing may read it
    for (int t = redo_level; t <= QSVC_MAX_T; t++) {
      Level& lv = c->lv[t];
      lv.high_n = 0; lv.low_n = 0; lv.mv_filt_n = 0; lv.types.clear(); lv.types_d_ok = false; lv.types_allB = false;
    }
  }
  return r;
}

static int decorrelate_impl(qsvc_ctx* c, int t, int P, int bs, int S, int always_B) {
  (void)hipSetDevice(c->dev);
  TuningScope tscope(&c->tune);
  LevelGeom g;
  int r = level_geom(c, t, P, bs, S, &g);
  if (r) return r;
  if (S < 1 || S > 128) return c->fail(QSVC_ERR_INVALID, "search_range %d", S);
  Level& lv = c->lv[t];
  if (lv.mv_raw_n < g.nB || lv.mv_bs != bs) return c->fail(QSVC_ERR_STATE, "motion_%d missing (run motion_estimate first)", t);
  if (!c->pending.empty() && !c->resolving) c->spec_call = true;
  size_t fld = (size_t)g.BY * g.BX, mvbytes = (size_t)g.nB * 4 * fld * sizeof(int16_t);
  if ((r = lv.high.ensure(c, (size_t)(g.nB ? g.nB : 1) * g.F))) return r;
  if ((r = lv.mv_filt.ensure(c, mvbytes ? mvbytes : 16))) return r;
  if ((r = lv.hist.ensure(c, (size_t)(g.nB ? g.nB : 1) * 3 * 256 * 4))) return r;
  if (c->keep_pred) { if ((r = lv.pred.ensure(c, (size_t)(g.nB ? g.nB : 1) * g.F))) return r; lv.pred_n = g.nB; }
  lv.high_n = g.nB; lv.mv_filt_n = g.nB;
  lv.types.assign((size_t)g.nB, 'B');
  if (g.nB == 0) return QSVC_OK;
  PredArgs a;
  a.g = g; a.frames = (const uint8_t*)c->lv[t - 1].low.p; a.mv = (const int16_t*)lv.mv_raw.p;
  a.high = (uint8_t*)lv.high.p; a.pred_dump = c->keep_pred ? (uint8_t*)lv.pred.p : nullptr; a.hist = (unsigned*)lv.hist.p;
  a.margin = nullptr; a.odd_out = nullptr; a.types = nullptr; a.cup = nullptr;
  a.mv_copy = nullptr; a.hist_is_zero = lv.hist_zeroed ? 1 : 0; a.up = nullptr;
  lv.hist_zeroed = false;
  lv.ragged = (g.H % bs || g.W % bs);
  if (lv.ragged && !g.sub) {  // Q7: the chroma margins carry state from picture to picture
    if ((r = c->margin.ensure(c, (size_t)g.nB * 2 * margin_geom(g).count() * sizeof(int16_t)))) return r;
    a.margin = (int16_t*)c->margin.p;
  }
  Prof pf;
  pf.begin(c, "predict");
  const bool fast = !lv.ragged && g.ov < 2 && !g.sub && pred_fast_path(a);
  if (fast) a.mv_copy = (int16_t*)lv.mv_filt.p;   // the fused launch also writes motion_filtered_t = motion_t
  if (!lv.ragged && !fast && !g.sub && (r = make_cup(c, g, a.frames, 2, &a.cup))) return r;
  if (g.sub) {
    if ((r = predict_subpel(c, g, a))) return r;
  } else if (g.ov >= 2) {
    if ((r = predict_overlapped(c, g, a))) return r;
  } else launch_predict(c->st, a);
  pf.end(c);
  if (!fast) HIPCHK(c, hipMemcpyAsync(lv.mv_filt.p, lv.mv_raw.p, mvbytes, hipMemcpyDeviceToDevice, c->st));
  const bool defer = !always_B && c->defer_ok && c->tune.defer_ib && !c->resolving;
  if (defer) {
    // deferred decision: the histograms travel to the host behind the kernels, the level goes on as if every picture were 'B'
    // (what it almost always is); qsvc_resolve applies decorrelate.cpp:964-1000 at the next point that needs the outcome
    const size_t hbytes = (size_t)g.nB * 3 * 256 * 4;
    if (!lv.hhist.ensure(hbytes)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc(%zu)", hbytes);
    HIPCHK(c, hipMemcpyAsync(lv.hhist.p, lv.hist.p, hbytes, hipMemcpyDeviceToHost, c->st));
    lv.pending = true; lv.dec_nB = g.nB; lv.dec_BY = g.BY; lv.dec_BX = g.BX;
    c->spec_call = true;  // the first pending decision: the replay starts behind this call
    c->pending.push_back(t);
  } else if (!always_B) {
    // host-side I/B rule (decorrelate.cpp:964-979) on the three device histograms
    const size_t hbytes = (size_t)g.nB * 3 * 256 * 4;
    if (!c->hhist.ensure(hbytes)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc(%zu)", hbytes);
    HIPCHK(c, hipMemcpyAsync(c->hhist.p, lv.hist.p, hbytes, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, ctx_sync(c));
    for (int i = 0; i < g.nB; i++) {
      const uint32_t* h = (const uint32_t*)c->hhist.p + (size_t)i * 3 * 256;
      if (picture_is_I(h, g.H, g.W, g.BY, g.BX)) {  // 'I': the odd picture itself, zero motion
        lv.types[i] = 'I';
        HIPCHK(c, hipMemcpyAsync((uint8_t*)lv.high.p + (size_t)i * g.F, a.frames + (size_t)(2 * i + 1) * g.F, g.F,
                                 hipMemcpyDeviceToDevice, c->st));
        HIPCHK(c, hipMemsetAsync((int16_t*)lv.mv_filt.p + (size_t)i * 4 * fld, 0, 4 * fld * sizeof(int16_t), c->st));
      }
    }
  }
  // frame_types_t on the device for `update` (page-locked source: nothing the caller can touch is read after this call returns)
  lv.types_d_ok = false; lv.types_allB = false;
  bool allB = g.nB > 0 && g.nB <= QSVC_ALLB_MAX;
  for (int i = 0; i < g.nB && allB; i++) allB = lv.types[i] == 'B';
  if (allB) {  // (the usual case, and what a deferred decision assumes: no copy between predict and update)
    if (!c->allB.p) {
      if ((r = c->allB.ensure(c, QSVC_ALLB_MAX + 16))) return r;
      HIPCHK(c, hipMemsetAsync(c->allB.p, 'B', QSVC_ALLB_MAX + 16, c->st));
    }
    lv.types_d_ok = true; lv.types_allB = true;
  } else if (g.nB) {
    if (!lv.types_pin.ensure((size_t)g.nB)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc(%d)", g.nB);
    if ((r = lv.types_d.ensure(c, (size_t)g.nB + 16))) return r;
    memcpy(lv.types_pin.p, lv.types.data(), (size_t)g.nB);
    HIPCHK(c, hipMemcpyAsync(lv.types_d.p, lv.types_pin.p, (size_t)g.nB, hipMemcpyHostToDevice, c->st));
    lv.types_d_ok = true;
  }
  HIPCHK(c, hipGetLastError());
  return QSVC_OK;
}

static int update_impl(qsvc_ctx* c, int t, int P, int bs, float uf) {
  (void)hipSetDevice(c->dev);
  TuningScope tscope(&c->tune);
  LevelGeom g;
  int r = level_geom(c, t, P, bs, 0, &g);
  if (r) return r;
  Level& lv = c->lv[t];
  if (lv.high_n < g.nB || lv.mv_filt_n < g.nB || (int)lv.types.size() < g.nB || (g.nB && lv.mv_bs != bs))
    return c->fail(QSVC_ERR_STATE, "high_%d / motion_filtered_%d / frame_types_%d missing", t, t, t);
  size_t fld = (size_t)g.BY * g.BX;
  if ((r = lv.low.ensure(c, (size_t)(g.nB + 1) * g.F))) return r;
  lv.low_n = g.nB + 1;
  if ((r = c->scal.ensure(c, 256))) return r;
  // reach >= max |mv| over everything an update may touch.  Motion that motion_estimate produced in this context has a known
  // bound and frame_types_t is on the device since decorrelate: then no host round trip stands between predict and update.
  const bool have_halo = lv.halo[0] || lv.halo[1];
  if ((have_halo || lv.mv_bound < 0 || !lv.types_d_ok) && !c->pending.empty()) {  // this path reads frame_types_t on the host
    if ((r = qsvc_resolve(c))) return r;
  }
  if (!c->pending.empty() && !c->resolving) {
    c->spec_call = true;
    if (c->calls.size() > 4096 && (r = qsvc_resolve(c))) return r;
  }
  const bool resident = lv.mv_bound >= 0 && lv.types_d_ok && !have_halo;
  int reach = resident ? lv.mv_bound : 0;
  uint8_t halo_types[2][16] = {{0}, {0}};
  const uint8_t* types_dev = lv.types_allB ? (const uint8_t*)c->allB.p : (const uint8_t*)lv.types_d.p;
  if (!resident) {
    HIPCHK(c, hipMemsetAsync(c->scal.p, 0, 16, c->st));
    if (g.nB) launch_mv_maxabs(c->st, (const int16_t*)lv.mv_filt.p, (size_t)g.nB * 4 * fld, (int*)c->scal.p);
    for (int s = 0; s < 2; s++)
      if (lv.halo[s]) launch_mv_maxabs(c->st, (const int16_t*)((const uint8_t*)lv.halo[s] + g.F), 4 * fld, (int*)c->scal.p);
    HIPCHK(c, hipMemcpyAsync(&reach, c->scal.p, sizeof(int), hipMemcpyDeviceToHost, c->st));
    for (int s = 0; s < 2; s++)
      if (lv.halo[s])
        HIPCHK(c, hipMemcpyAsync(halo_types[s], (const uint8_t*)lv.halo[s] + g.F + 4 * fld * sizeof(int16_t), 16, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, ctx_sync(c));
    // frame types to the device (tiny)
    if ((r = c->tmp16[0].ensure(c, (size_t)g.nB + 16))) return r;
    if (g.nB) HIPCHK(c, hipMemcpyAsync(c->tmp16[0].p, lv.types.data(), g.nB, hipMemcpyHostToDevice, c->st));
    types_dev = (const uint8_t*)c->tmp16[0].p;
  }
  UpdArgs a;
  a.g = g; a.frames = (const uint8_t*)c->lv[t - 1].low.p; a.high = (const uint8_t*)lv.high.p;
  a.mv = (const int16_t*)lv.mv_filt.p; a.types = types_dev; a.low = (uint8_t*)lv.low.p;
  a.uf = uf; a.reach = reach; a.src_mul = 2; a.dst_mul = 1;
  for (int s = 0; s < 2; s++) {
    a.halo_high[s] = (const uint8_t*)lv.halo[s];
    a.halo_mv[s] = lv.halo[s] ? (const int16_t*)((const uint8_t*)lv.halo[s] + g.F) : nullptr;
    a.halo_type[s] = lv.halo[s] ? halo_types[s][0] : 0;
  }
  a.cup = nullptr;
  Prof pf;
  pf.begin(c, "update");
  if (!upd_fast_path(a) && (r = make_cup(c, g, a.frames, a.src_mul, &a.cup))) return r;
  launch_update(c->st, a);
  pf.end(c);
  c->cup_src = nullptr;  // low_t was written: nothing interpolated from pictures may be reused across this point
  HIPCHK(c, hipGetLastError());
  if (!resident) HIPCHK(c, ctx_sync(c));  // lv.types / halo buffers may be changed by the caller next
  return QSVC_OK;
}

// A stage call made while I/B decisions are pending is logged for replay (qsvc_resolve) -- AFTER it has queued its work
// successfully: a call that was refused (geometry, state, memory) must not come back as an error of an unrelated entry point.
static int logged(qsvc_ctx* c, int r, const StageCall& call) {
  const bool spec = c->spec_call;
  c->spec_call = false;
  if (!r && spec && !c->resolving && !c->pending.empty()) c->calls.push_back(call);
  return r;
}
int qsvc_motion_estimate(qsvc_ctx* c, int t, int P, int bs, int S) {
  if (!c) return QSVC_ERR_INVALID;
  c->spec_call = false;
  return logged(c, motion_estimate_impl(c, t, P, bs, S), {0, t, P, bs, S, 0, 0.0f});
}
int qsvc_decorrelate(qsvc_ctx* c, int t, int P, int bs, int S, int always_B) {
  if (!c) return QSVC_ERR_INVALID;
  c->spec_call = false;
  return logged(c, decorrelate_impl(c, t, P, bs, S, always_B), {1, t, P, bs, S, always_B, 0.0f});
}
int qsvc_update(qsvc_ctx* c, int t, int P, int bs, float uf) {
  if (!c) return QSVC_ERR_INVALID;
  c->spec_call = false;
  return logged(c, update_impl(c, t, P, bs, uf), {2, t, P, bs, 0, 0, uf});
}

int qsvc_analyze_step(qsvc_ctx* c, int t, int P, int bs, int S, int always_B, float uf) {
  if (!c) return QSVC_ERR_INVALID;
  HostCpu hc(c, "host_cpu_analyze_call");
  int r = qsvc_motion_estimate(c, t, P, bs, S);
  if (r) return r;
  if ((r = qsvc_decorrelate(c, t, P, bs, S, always_B))) return r;
  return qsvc_update(c, t, P, bs, uf);
}

// ---- inverse temporal level (synthesize_step.py:96-151): un_update, correlate; "merge" is the layout itself --------
// low_t (P/2+1 pictures), high_t, motion_filtered_t (the decoded motion_t) and frame_types_t must be resident.
static int synth_geom(qsvc_ctx* c, int t, int P, int bs, int S, LevelGeom* g, bool need_low) {
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  if (t < 1 || t > QSVC_MAX_T) return c->fail(QSVC_ERR_INVALID, "temporal subband %d out of range", t);
  if (bs < 2 || bs > 256) return c->fail(QSVC_ERR_INVALID, "block_size %d", bs);
  if (P < 1) return c->fail(QSVC_ERR_INVALID, "pictures %d", P);
  if (c->H < bs || c->W < bs) return c->fail(QSVC_ERR_INVALID, "picture %dx%d smaller than block_size %d", c->W, c->H, bs);
  g->W = c->W; g->H = c->H; g->bs = bs; g->BY = c->H / bs; g->BX = c->W / bs; g->S = S; g->P = P; g->nB = P / 2; g->F = c->F;
  g->ov = c->opt_ov; g->bord = 0; g->sub = c->opt_sub;
  Level& lv = c->lv[t];
  if (need_low && lv.low_n < g->nB + 1) return c->fail(QSVC_ERR_STATE, "low_%d holds %d pictures, %d needed", t, lv.low_n, g->nB + 1);
  if (lv.high_n < g->nB || lv.mv_filt_n < g->nB || (int)lv.types.size() < g->nB || (g->nB && lv.mv_bs != bs))
    return c->fail(QSVC_ERR_STATE, "high_%d / motion_%d / frame_types_%d missing", t, t, t);
  return QSVC_OK;
}
static int types_to_device(qsvc_ctx* c, const Level& lv, int nB) {
  int r = c->tmp16[0].ensure(c, (size_t)nB + 16);
  if (r) return r;
  if (nB) HIPCHK(c, hipMemcpyAsync(c->tmp16[0].p, lv.types.data(), nB, hipMemcpyHostToDevice, c->st));
  return QSVC_OK;
}

int qsvc_un_update(qsvc_ctx* c, int t, int P, int bs, float uf) {
  if (!c) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  (void)hipSetDevice(c->dev);
  TuningScope tscope(&c->tune);
  LevelGeom g;
  int r = synth_geom(c, t, P, bs, 0, &g, true);
  if (r) return r;
  Level& lv = c->lv[t];
  Level& out = c->lv[t - 1];
  const int Pout = 2 * g.nB + 1;
  if ((r = out.low.ensure(c, (size_t)Pout * g.F))) return r;
  out.low_n = Pout;
  if ((r = c->scal.ensure(c, 256))) return r;
  size_t fld = (size_t)g.BY * g.BX;
  HIPCHK(c, hipMemsetAsync(c->scal.p, 0, 16, c->st));
  if (g.nB) launch_mv_maxabs(c->st, (const int16_t*)lv.mv_filt.p, (size_t)g.nB * 4 * fld, (int*)c->scal.p);
  int reach = 0;
  HIPCHK(c, hipMemcpyAsync(&reach, c->scal.p, sizeof(int), hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  if ((r = types_to_device(c, lv, g.nB))) return r;
  UpdArgs a;
  a.g = g; a.frames = (const uint8_t*)lv.low.p; a.high = (const uint8_t*)lv.high.p;
  a.mv = (const int16_t*)lv.mv_filt.p; a.types = (const uint8_t*)c->tmp16[0].p; a.low = (uint8_t*)out.low.p;
  a.uf = -uf; a.reach = reach; a.src_mul = 1; a.dst_mul = 2;
  for (int s = 0; s < 2; s++) { a.halo_high[s] = nullptr; a.halo_mv[s] = nullptr; a.halo_type[s] = 0; }
  a.cup = nullptr;
  Prof pf;
  pf.begin(c, "un_update");
  if (!upd_fast_path(a) && (r = make_cup(c, g, a.frames, a.src_mul, &a.cup))) return r;
  launch_update(c->st, a);
  pf.end(c);
  c->cup_src = nullptr;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}

int qsvc_correlate(qsvc_ctx* c, int t, int P, int bs, int S) {
  if (!c) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  (void)hipSetDevice(c->dev);
  TuningScope tscope(&c->tune);
  LevelGeom g;
  int r = synth_geom(c, t, P, bs, S, &g, false);
  if (r) return r;
  if (S < 1 || S > 128) return c->fail(QSVC_ERR_INVALID, "search_range %d", S);
  Level& lv = c->lv[t];
  Level& out = c->lv[t - 1];
  if (out.low_n < 2 * g.nB + 1) return c->fail(QSVC_ERR_STATE, "even_%d missing (run un_update first)", t);
  if (g.nB == 0) return QSVC_OK;
  if (c->keep_pred) { if ((r = lv.pred.ensure(c, (size_t)g.nB * g.F))) return r; lv.pred_n = g.nB; }
  if ((r = types_to_device(c, lv, g.nB))) return r;
  PredArgs a;
  a.g = g; a.frames = (const uint8_t*)out.low.p; a.mv = (const int16_t*)lv.mv_filt.p;
  a.high = (uint8_t*)lv.high.p; a.pred_dump = c->keep_pred ? (uint8_t*)lv.pred.p : nullptr; a.hist = nullptr;
  a.margin = nullptr; a.odd_out = (uint8_t*)out.low.p; a.types = (const uint8_t*)c->tmp16[0].p; a.cup = nullptr;
  a.mv_copy = nullptr; a.hist_is_zero = 0; a.up = nullptr;
  if ((g.H % bs || g.W % bs) && !g.sub) {  // Q7 applies to the decoder's prediction planes as well
    if ((r = c->margin.ensure(c, (size_t)g.nB * 2 * margin_geom(g).count() * sizeof(int16_t)))) return r;
    a.margin = (int16_t*)c->margin.p;
  }
  Prof pf;
  pf.begin(c, "correlate");
  c->cup_src = nullptr;  // un_update has just rewritten the even pictures
  if (!a.margin && !g.sub && (r = make_cup(c, g, a.frames, 2, &a.cup))) return r;
  if (g.sub) {
    if ((r = predict_subpel(c, g, a))) return r;
  } else if (g.ov >= 2) {
    if ((r = predict_overlapped(c, g, a))) return r;
  } else launch_predict(c->st, a);
  pf.end(c);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}

int qsvc_synthesize_step(qsvc_ctx* c, int t, int P, int bs, int S, float uf) {
  int r = qsvc_un_update(c, t, P, bs, uf);
  if (r) return r;
  return qsvc_correlate(c, t, P, bs, S);
}

// ---- halo ---------------------------------------------------------------------------------------
size_t qsvc_halo_bytes(qsvc_ctx* c, int bs) {
  if (!c || !c->F || bs <= 0) return 0;
  return c->F + (size_t)4 * (c->H / bs) * (c->W / bs) * sizeof(int16_t) + 16;
}
// pack our boundary B picture of level t into the context's send buffer `side`
// The stale-margin recurrence of a ragged picture (Q7) runs through every picture of the level in order: a shard cannot
// start it in the middle, so the open-GOP split is refused there.
static int halo_ragged(qsvc_ctx* c, int t) {
  return c->lv[t].ragged ? c->fail(QSVC_ERR_UNSUPPORTED, "picture %dx%d is not a multiple of the block size: its prediction margins depend "
                                   "on every earlier picture of the level, the sequence cannot be split across GPUs", c->W, c->H) : QSVC_OK;
}
// dst: where the record goes (device), or nullptr: the context's own send buffer of that side
static int halo_pack(qsvc_ctx* c, int t, int side, uint8_t* dst = nullptr) {
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  if (int rr = halo_ragged(c, t)) return rr;
  Level& lv = c->lv[t];
  int nB = lv.high_n;
  if (nB < 1 || lv.mv_filt_n < nB || (int)lv.types.size() < nB) return c->fail(QSVC_ERR_STATE, "nothing to export for t=%d", t);
  int i = side == 0 ? 0 : nB - 1;
  size_t fld = (size_t)(c->H / lv.mv_bs) * (c->W / lv.mv_bs), mvb = 4 * fld * sizeof(int16_t);
  if (!dst) {
    int r = c->halo_tx[side].ensure(c, c->F + mvb + 16);
    if (r) return r;
  }
  uint8_t* d = dst ? dst : (uint8_t*)c->halo_tx[side].p;
  HIPCHK(c, hipMemcpyAsync(d, (uint8_t*)lv.high.p + (size_t)i * c->F, c->F, hipMemcpyDeviceToDevice, c->st));
  HIPCHK(c, hipMemcpyAsync(d + c->F, (uint8_t*)lv.mv_filt.p + (size_t)i * mvb, mvb, hipMemcpyDeviceToDevice, c->st));
  memset(c->halo_tb[side], 0, 16);
  c->halo_tb[side][0] = (uint8_t)lv.types[i];
  HIPCHK(c, hipMemcpyAsync(d + c->F + mvb, c->halo_tb[side], 16, hipMemcpyHostToDevice, c->st));
  return QSVC_OK;
}
int qsvc_halo_export_host(qsvc_ctx* c, int t, int side, void* dst) {
  if (!c || !dst || t < 1 || t > QSVC_MAX_T || (side != 0 && side != 1)) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  int r = halo_pack(c, t, side);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(dst, c->halo_tx[side].p, qsvc_halo_bytes(c, c->lv[t].mv_bs), hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  return QSVC_OK;
}
int qsvc_halo_import_host(qsvc_ctx* c, int t, int side, const void* src) {
  if (!c || t < 1 || t > QSVC_MAX_T || (side != 0 && side != 1)) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  Level& lv = c->lv[t];
  if (!src) { lv.halo[side] = nullptr; return QSVC_OK; }
  if (!lv.mv_bs) return c->fail(QSVC_ERR_STATE, "halo import before decorrelate of t=%d", t);
  size_t n = qsvc_halo_bytes(c, lv.mv_bs);
  int r = c->halo_rx[side].ensure(c, n);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(c->halo_rx[side].p, src, n, hipMemcpyHostToDevice, c->st));
  HIPCHK(c, ctx_sync(c));
  lv.halo[side] = c->halo_rx[side].p;
  return QSVC_OK;
}

// RCCL is dlopen'ed on first use so that single-GPU runs never load it.
#include <dlfcn.h>
struct Id128 { char b[128]; };  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed by value
namespace {
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl g_rccl;
static const char* rccl_load() {
  if (g_rccl.h) return nullptr;
  // absolute path first: a process that imported torch already holds torch's own (ROCm 7.0) librccl, which is bound
  // to a different HIP runtime than this library's
  void* h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) return dlerror();
#define SYM(field, name) *(void**)(&g_rccl.field) = dlsym(h, name); if (!g_rccl.field) return "missing symbol " name;
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(Send, "ncclSend") SYM(Recv, "ncclRecv") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_rccl.h = h;
  return nullptr;
}
int qsvc_comm_unique_id(void* out, size_t cap) {
  if (!out || cap < 128) return QSVC_ERR_INVALID;
  if (rccl_load()) return QSVC_ERR_UNSUPPORTED;
  Id128 id;
  if (g_rccl.GetUniqueId(&id) != 0) return QSVC_ERR_HIP;
  memcpy(out, &id, 128);
  return QSVC_OK;
}
int qsvc_comm_init(qsvc_ctx* c, int rank, int nranks, const void* id, size_t id_len) {
  if (!c || !id || id_len < 128 || rank < 0 || rank >= nranks) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  const char* e = rccl_load();
  if (e) return c->fail(QSVC_ERR_UNSUPPORTED, "cannot load RCCL: %s", e);
  Id128 uid;
  memcpy(&uid, id, 128);
  int r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != 0) { c->comm = nullptr; return c->fail(QSVC_ERR_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
  c->rank = rank; c->nranks = nranks;
  return QSVC_OK;
}
int qsvc_comm_destroy(qsvc_ctx* c) {
  if (!c) return QSVC_ERR_INVALID;
  if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
  return QSVC_OK;
}
int qsvc_halo_exchange(qsvc_ctx* c, int t) {
  if (!c || t < 1 || t > QSVC_MAX_T) return QSVC_ERR_INVALID;
  if (!c->comm) return c->fail(QSVC_ERR_STATE, "qsvc_comm_init not called");
  (void)hipSetDevice(c->dev);
  Level& lv = c->lv[t];
  if (!lv.mv_bs) return c->fail(QSVC_ERR_STATE, "halo exchange before decorrelate of t=%d", t);
  size_t n = qsvc_halo_bytes(c, lv.mv_bs);
  bool left = c->rank > 0, right = c->rank + 1 < c->nranks;
  int r;
  if (left) { if ((r = halo_pack(c, t, 0))) return r; if ((r = c->halo_rx[0].ensure(c, n))) return r; }
  if (right) { if ((r = halo_pack(c, t, 1))) return r; if ((r = c->halo_rx[1].ensure(c, n))) return r; }
  const int ncclUint8 = 1;  // ncclUint8 / ncclChar enum value in nccl.h
  int e = g_rccl.GroupStart();
  if (!e && left) e = g_rccl.Send(c->halo_tx[0].p, n, ncclUint8, c->rank - 1, c->comm, c->st);
  if (!e && left) e = g_rccl.Recv(c->halo_rx[0].p, n, ncclUint8, c->rank - 1, c->comm, c->st);
  if (!e && right) e = g_rccl.Send(c->halo_tx[1].p, n, ncclUint8, c->rank + 1, c->comm, c->st);
  if (!e && right) e = g_rccl.Recv(c->halo_rx[1].p, n, ncclUint8, c->rank + 1, c->comm, c->st);
  int e2 = g_rccl.GroupEnd();
  if (e || e2) return c->fail(QSVC_ERR_HIP, "RCCL halo exchange: %s", g_rccl.GetErrorString(e ? e : e2));
  lv.halo[0] = left ? c->halo_rx[0].p : nullptr;
  lv.halo[1] = right ? c->halo_rx[1].p : nullptr;
  return QSVC_OK;
}

// ---- batched exchange: the records of n contexts of ONE rank (its sequences in flight) as one message per neighbour ----------
// ctxs[0] owns the batch buffers (and, for the RCCL transport, the rank's only communicator).  Every context packs its record on its
// own stream into its slot of the batch; the owner's stream waits for all of them before the message leaves, and every context's
// stream waits for the arrival before its update reads the slot it was given.  All contexts: same device, geometry and block size.
static int batch_check(qsvc_ctx* const* cs, int n, int t, size_t* rec) {
  if (!cs || n < 1 || !cs[0]) return QSVC_ERR_INVALID;
  qsvc_ctx* o = cs[0];
  if (t < 1 || t > QSVC_MAX_T) return o->fail(QSVC_ERR_INVALID, "temporal subband %d out of range", t);
  if (!o->lv[t].mv_bs) return o->fail(QSVC_ERR_STATE, "halo exchange before decorrelate of t=%d", t);
  *rec = qsvc_halo_bytes(o, o->lv[t].mv_bs);
  for (int k = 0; k < n; k++) {
    qsvc_ctx* c = cs[k];
    if (!c) return o->fail(QSVC_ERR_INVALID, "null context in the batch");
    if (c->dev != o->dev || c->W != o->W || c->H != o->H || c->lv[t].mv_bs != o->lv[t].mv_bs)
      return o->fail(QSVC_ERR_INVALID, "contexts of one halo batch must share the device, the picture size and the block size (context %d differs)", k);
    if (!c->halo_ev && hipEventCreateWithFlags(&c->halo_ev, hipEventDisableTiming) != hipSuccess) return o->fail(QSVC_ERR_HIP, "hipEventCreate");
  }
  return QSVC_OK;
}
// every context's record of `side` into the owner's send batch; on return the owner's stream is ordered behind all of them
static int batch_pack(qsvc_ctx* const* cs, int n, int t, int side, size_t rec) {
  qsvc_ctx* o = cs[0];
  int r = o->halo_btx[side].ensure(o, (size_t)n * rec);
  if (r) return r;
  for (int k = 0; k < n; k++) {
    if ((r = halo_pack(cs[k], t, side, (uint8_t*)o->halo_btx[side].p + (size_t)k * rec))) { if (k) o->err = cs[k]->err; return r; }
    HIPCHK(o, hipEventRecord(cs[k]->halo_ev, cs[k]->st));
    if (k) HIPCHK(o, hipStreamWaitEvent(o->st, cs[k]->halo_ev, 0));
  }
  return QSVC_OK;
}
// the owner's receive batch of `side` has been written on the owner's stream: every context takes its slot (or none: no neighbour)
static int batch_take(qsvc_ctx* const* cs, int n, int t, int side, size_t rec, bool present) {
  qsvc_ctx* o = cs[0];
  if (present) HIPCHK(o, hipEventRecord(o->halo_ev, o->st));
  for (int k = 0; k < n; k++) {
    cs[k]->lv[t].halo[side] = present ? (uint8_t*)o->halo_brx[side].p + (size_t)k * rec : nullptr;
    if (present && k) HIPCHK(o, hipStreamWaitEvent(cs[k]->st, o->halo_ev, 0));
  }
  return QSVC_OK;
}
size_t qsvc_halo_batch_bytes(qsvc_ctx* const* ctxs, int n, int t) {
  size_t rec = 0;
  return batch_check(ctxs, n, t, &rec) ? 0 : (size_t)n * rec;
}
int qsvc_halo_batch_export_host(qsvc_ctx* const* ctxs, int n, int t, int side, void* dst) {
  size_t rec = 0;
  int r = batch_check(ctxs, n, t, &rec);
  if (r) return r;
  qsvc_ctx* o = ctxs[0];
  if (!dst || (side != 0 && side != 1)) return o->fail(QSVC_ERR_INVALID, "bad halo batch export arguments");
  (void)hipSetDevice(o->dev);
  if ((r = batch_pack(ctxs, n, t, side, rec))) return r;
  HIPCHK(o, hipMemcpyAsync(dst, o->halo_btx[side].p, (size_t)n * rec, hipMemcpyDeviceToHost, o->st));
  HIPCHK(o, ctx_sync(o));
  return QSVC_OK;
}
int qsvc_halo_batch_import_host(qsvc_ctx* const* ctxs, int n, int t, int side, const void* src) {
  size_t rec = 0;
  int r = batch_check(ctxs, n, t, &rec);
  if (r) return r;
  qsvc_ctx* o = ctxs[0];
  if (side != 0 && side != 1) return o->fail(QSVC_ERR_INVALID, "bad halo side %d", side);
  (void)hipSetDevice(o->dev);
  for (int k = 0; k < n; k++) { int rr_ = qsvc_resolve(ctxs[k]); if (rr_) return rr_; }
  if (src) {
    if ((r = o->halo_brx[side].ensure(o, (size_t)n * rec))) return r;
    // (the slots may still be read by the previous level's update of any context: their streams are joined first)
    for (int k = 1; k < n; k++) { HIPCHK(o, hipEventRecord(ctxs[k]->halo_ev, ctxs[k]->st)); HIPCHK(o, hipStreamWaitEvent(o->st, ctxs[k]->halo_ev, 0)); }
    HIPCHK(o, hipMemcpyAsync(o->halo_brx[side].p, src, (size_t)n * rec, hipMemcpyHostToDevice, o->st));
    HIPCHK(o, ctx_sync(o));  // the caller's buffer is free on return
  }
  return batch_take(ctxs, n, t, side, rec, src != nullptr);
}
int qsvc_halo_exchange_batch(qsvc_ctx* const* ctxs, int n, int t) {
  size_t rec = 0;
  int r = batch_check(ctxs, n, t, &rec);
  if (r) return r;
  qsvc_ctx* o = ctxs[0];
  if (!o->comm) return o->fail(QSVC_ERR_STATE, "qsvc_comm_init not called on the batch's first context");
  (void)hipSetDevice(o->dev);
  const bool nb[2] = {o->rank > 0, o->rank + 1 < o->nranks};
  for (int s = 0; s < 2; s++)
    if (nb[s]) {
      if ((r = o->halo_brx[s].ensure(o, (size_t)n * rec))) return r;
      if ((r = batch_pack(ctxs, n, t, s, rec))) return r;
    }
  const int ncclUint8 = 1;
  int e = g_rccl.GroupStart();
  for (int s = 0; s < 2 && !e; s++)
    if (nb[s]) {
      const int peer = s == 0 ? o->rank - 1 : o->rank + 1;
      e = g_rccl.Send(o->halo_btx[s].p, (size_t)n * rec, ncclUint8, peer, o->comm, o->st);
      if (!e) e = g_rccl.Recv(o->halo_brx[s].p, (size_t)n * rec, ncclUint8, peer, o->comm, o->st);
    }
  int e2 = g_rccl.GroupEnd();
  if (e || e2) return o->fail(QSVC_ERR_HIP, "RCCL halo exchange (batch of %d): %s", n, g_rccl.GetErrorString(e ? e : e2));
  for (int s = 0; s < 2; s++)
    if ((r = batch_take(ctxs, n, t, s, rec, nb[s]))) return r;
  return QSVC_OK;
}

// ---- motion-field coding chain (compress.py:197-212 -> motion_compress.py:146-186 -> motion_compress_j2k.py:98-142) ---------
// One call for what the reference spreads over TRLs - 1 decorrelator processes, a demux / split pipeline and one kdu_compress per
// component plane: the resident motion_filtered_t of the levels ts[0] < ts[1] < ... come to the host (kilobytes), every level but the
// last is predicted from the level above (residue = field - reference / 2 with C division, two fields per reference field, the
// last reference kept when the file ends: interlevel_motion_decorrelate.cpp:32-70,249-283), the last level is decorrelated
// bidirectionally (NEXT -= PREV, bidirectional_motion_decorrelate.cpp:25-52), the planar residue file is cut as the reference's
// `demux 8 2c 2 | split` cuts it (component c of a field = every 4th int16, phase c: motion_compress_j2k.py:103-107), and all
// component planes of all levels are coded lossless in ONE batch.  Streams in (level, field, component) order.
// the host half of the chain: the component planes of all levels (4 * sum(fields) planes of BY x BX int16, back to back)
int qsvc_motion_planes(qsvc_ctx* c, int nlevels, const int* ts, const int* fields, int block_size, const int16_t* filtered_host,
                       int16_t* residues, std::vector<int16_t>& planes, int* BYo, int* BXo) {
  if (!c || !ts || !fields || nlevels < 1 || block_size < 1) return QSVC_ERR_INVALID;
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }
  (void)hipSetDevice(c->dev);
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  const int BY = c->H / block_size, BX = c->W / block_size;
  *BYo = BY; *BXo = BX;
  const size_t fld = (size_t)BY * BX, per = 4 * fld;
  size_t total = 0;
  for (int k = 0; k < nlevels; k++) {
    const int t = ts[k];
    if (t < 1 || t > QSVC_MAX_T || fields[k] < 0 || (k && ts[k] <= ts[k - 1])) return c->fail(QSVC_ERR_INVALID, "bad level list");
    const Level& lv = c->lv[t];
    if (!filtered_host && (lv.mv_filt_n < fields[k] || (fields[k] && lv.mv_bs != block_size)))
      return c->fail(QSVC_ERR_STATE, "motion_filtered_%d holds %d fields at block size %d (asked for %d at %d)", t, lv.mv_filt_n, lv.mv_bs, fields[k], block_size);
    total += (size_t)fields[k];
  }
  planes.assign(total * per, 0);
  if (!total) return QSVC_OK;
  std::vector<int16_t> filt(total * per);
  size_t o = 0;
  if (filtered_host) memcpy(filt.data(), filtered_host, total * per * sizeof(int16_t));
  else {
    for (int k = 0; k < nlevels; k++) {
      if (fields[k]) HIPCHK(c, hipMemcpyAsync(filt.data() + o, c->lv[ts[k]].mv_filt.p, (size_t)fields[k] * per * sizeof(int16_t), hipMemcpyDeviceToHost, c->st));
      o += (size_t)fields[k] * per;
    }
    HIPCHK(c, ctx_sync(c));
  }
  // residues, in place of the fields (a level's reference is the NEXT level's field, still unmodified: levels in ascending order)
  o = 0;
  for (int k = 0; k < nlevels; k++) {
    int16_t* f = filt.data() + o;
    const size_t on = o + (size_t)fields[k] * per;
    if (k + 1 < nlevels) {
      const int16_t* ref = filt.data() + on;
      const int nref = fields[k + 1];
      for (int i = 0; i < fields[k]; i++) {
        const int r = i / 2 < nref ? i / 2 : nref - 1;  // past the end of the reference file the tool keeps the last field it read
        int16_t* d = f + (size_t)i * per;
        if (r >= 0) { const int16_t* q = ref + (size_t)r * per; for (size_t j = 0; j < per; j++) d[j] = (int16_t)(d[j] - q[j] / 2); }
      }
    } else {
      for (int i = 0; i < fields[k]; i++) {
        int16_t* d = f + (size_t)i * per;
        for (size_t j = 0; j < 2 * fld; j++) d[2 * fld + j] = (int16_t)(d[2 * fld + j] - d[j]);  // NEXT-X -= PREV-X, NEXT-Y -= PREV-Y
      }
    }
    o = on;
  }
  if (residues) memcpy(residues, filt.data(), total * per * sizeof(int16_t));
  for (size_t i = 0; i < total; i++) {  // the byte cut of the planar file: component c = every 4th int16 of the field
    const int16_t* s = filt.data() + i * per;
    int16_t* d = planes.data() + i * per;
    for (int comp = 0; comp < 4; comp++)
      for (size_t j = 0; j < fld; j++) d[(size_t)comp * fld + j] = s[4 * j + comp];
  }
  return QSVC_OK;
}
void qsvc_motion_prm(qsvc_j2k_params* prm) {  // motion_compress_j2k.py:131-142: Creversible=yes Clevels=0 Cuse_sop=yes, one layer
  memset(prm, 0, sizeof *prm);
  prm->reversible = 1; prm->levels = 0; prm->layers = 1; prm->cblk_w = 64; prm->cblk_h = 64; prm->use_sop = 1; prm->base_step = 1.0f / 256;
}
int qsvc_motion_encode(qsvc_ctx* c, int nlevels, const int* ts, const int* fields, int block_size, const int16_t* filtered_host,
                       int16_t* residues, uint8_t* out, size_t cap, size_t* offs) {
  if (!c || !out || !offs) return QSVC_ERR_INVALID;
  std::vector<int16_t> planes;
  int BY = 0, BX = 0;
  int r = qsvc_motion_planes(c, nlevels, ts, fields, block_size, filtered_host, residues, planes, &BY, &BX);
  if (r) return r;
  if (planes.empty()) { offs[0] = 0; return QSVC_OK; }
  qsvc_j2k_params prm;
  qsvc_motion_prm(&prm);
  return qsvc_j2k_encode_planes_s16(c, planes.data(), (int)(planes.size() / ((size_t)BY * BX)), BY, BX, &prm, out, cap, offs);
}

// ---- truncating 5/3 on a host plane ---------------------------------------------------------------
int qsvc_dwt53_trunc(qsvc_ctx* c, int16_t* plane, int H, int W, int stride, int levels, int inverse) {
  if (!c || !plane || H < 2 || W < 2 || stride < W || levels < 0) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  if ((H >> (levels ? levels - 1 : 0)) < 2 || (W >> (levels ? levels - 1 : 0)) < 2)
    return c->fail(QSVC_ERR_INVALID, "too many levels for %dx%d", W, H);
  size_t bytes = (size_t)H * stride * sizeof(int16_t);
  int r;
  if ((r = c->tmp16[0].ensure(c, bytes))) return r;
  if ((r = c->tmp16[1].ensure(c, bytes))) return r;
  int16_t* A = (int16_t*)c->tmp16[0].p; int16_t* B = (int16_t*)c->tmp16[1].p;
  HIPCHK(c, hipMemcpyAsync(A, plane, bytes, hipMemcpyHostToDevice, c->st));
  for (int k = 0; k < levels; k++) {
    int lv = inverse ? levels - 1 - k : k;
    int ny = H >> lv, nx = W >> lv;
    launch_dwt53t_level(c->st, A, B, stride, ny, nx, inverse);
    HIPCHK(c, hipMemcpy2DAsync(A, (size_t)stride * 2, B, (size_t)stride * 2, (size_t)nx * 2, ny, hipMemcpyDeviceToDevice, c->st));
  }
  HIPCHK(c, hipMemcpyAsync(plane, A, bytes, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  HIPCHK(c, hipGetLastError());
  return QSVC_OK;
}

// ---- profiling ------------------------------------------------------------------------------------
int qsvc_set_tier2_threads(qsvc_ctx* c, int n) {
  if (!c) return QSVC_ERR_INVALID;
  if (n < 0 || n > 64) return c->fail(QSVC_ERR_INVALID, "tier-2 threads %d (0 = default, 1..64)", n);
  c->t2_threads = n;
  return QSVC_OK;
}
int qsvc_set_profiling(qsvc_ctx* c, int on) {
  if (!c) return QSVC_ERR_INVALID;
  (void)ctx_sync(c);
  for (auto& s : c->segs) s.used = 0;  // the event pairs stay (creating events costs more than the kernels they bracket)
  c->hsegs.clear();
  c->profiling = on;
  return QSVC_OK;
}
int qsvc_get_timings(qsvc_ctx* c, const char** names, float* ms, int* launches, int cap) {
  if (!c) return QSVC_ERR_INVALID;
  (void)ctx_sync(c);
  int n = 0;
  for (auto& s : c->segs) {
    if (n >= cap) break;
    if (!s.used) continue;
    float tot = 0;
    for (size_t k = 0; k < s.used; k++) { float m = 0; if (hipEventElapsedTime(&m, s.ev[k].first, s.ev[k].second) == hipSuccess) tot += m; }
    names[n] = s.name.c_str(); ms[n] = tot; if (launches) launches[n] = (int)s.used;
    n++;
  }
  for (auto& h : c->hsegs) {  // host wall time of the stages that run on the CPU (names start with "host_")
    if (n >= cap) break;
    names[n] = h.name.c_str(); ms[n] = (float)h.ms; if (launches) launches[n] = h.n;
    n++;
  }
  return n;
}

}  // extern "C"
