// qsvc_amd/csrc/j2k_api.hip -- host side of the JPEG 2000 texture stage: plan (subbands, code-blocks),
// kernel orchestration, rate allocation, Tier-2 (packet headers, tag trees) and code-stream markers.
//
// Replaces the reference's per-picture, per-component `kdu_compress` process
// (trunk/src/texture_compress_fb_j2k.py:155-195).  Parameters the reference pins there: 8-bit unsigned
// single component, Clevels = SRLs-1, Clayers = nLayers, Cuse_sop = yes, Creversible = no; Kakadu
// defaults kept: one tile, 64x64 code-blocks, LRCP, no precinct partition.  Algorithm: ITU-T T.800.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <pthread.h>
#include <functional>
#include <mutex>
#include <thread>

#include "j2k_kernels.h"
#include "qsvc_ctx.h"
#include "j2k_norms.inc"

using namespace qsvc;

#define J2K_GUARD 2
#define J2K_MAXLEV 10
#define J2K_OUTCAP (64 * 64 * 3 + 64)  // bytes per code-block lane buffer (incl. scratch byte)

#define HIPCHK(ctx, call)                                    \
  do {                                                       \
    hipError_t e_ = (call);                                  \
    if (e_ != hipSuccess) return (ctx)->hipfail(e_, #call);  \
  } while (0)

namespace {

struct Subband {
  int x0, y0, w, h, orient, dlev, eps, mant, Mb;
  float delta;
  int gx, gy, first_cb;  // code-block grid and index of its first block within the plane
};

struct PlanePlan {  // everything about one plane geometry
  int W, H, L, prec;
  std::vector<Subband> sb;
  int ncb;
};

double norm_of(bool rev, int orient, int dlev) {
  if (dlev < 1) dlev = 1;
  if (dlev > J2K_MAXLEV) dlev = J2K_MAXLEV;
  return (rev ? kNorm53 : kNorm97)[orient][dlev - 1];
}

// subbands in code-stream order: LL_L, then HL, LH, HH for d = L..1 (T.800 B.5, tile at the origin)
void build_plan(PlanePlan& pp, int W, int H, int levels, bool rev, float base_step, int cbw, int cbh, int prec) {
  int L = levels;
  while (L > 0 && ((W >> (L - 1)) < 1 || (H >> (L - 1)) < 1)) L--;
  pp.W = W; pp.H = H; pp.L = L; pp.prec = prec;
  int ws[J2K_MAXLEV + 1], hs[J2K_MAXLEV + 1];
  ws[0] = W; hs[0] = H;
  for (int d = 1; d <= L; d++) { ws[d] = (ws[d - 1] + 1) >> 1; hs[d] = (hs[d - 1] + 1) >> 1; }
  pp.sb.clear();
  auto add = [&](int x0, int y0, int w, int h, int orient, int dlev) {
    Subband s;
    s.x0 = x0; s.y0 = y0; s.w = w; s.h = h; s.orient = orient; s.dlev = dlev;
    int gain = orient == 0 ? 0 : (orient == 3 ? 2 : 1);
    int Rb = prec + gain;
    if (rev) { s.eps = Rb; s.mant = 0; s.delta = 1.0f; }
    else {  // T.800 E.1.1: delta = 2^(Rb-eps) (1 + mant/2^11), target base_step*2^8/norm
      double target = (double)base_step * 256.0 / norm_of(false, orient, dlev ? dlev : 1);
      int e = (int)floor(log2(target));
      double m = target / ldexp(1.0, e) - 1.0;
      int mant = (int)floor(m * 2048.0);
      mant = mant < 0 ? 0 : (mant > 2047 ? 2047 : mant);
      int eps = Rb - e;
      if (eps < 0) { eps = 0; mant = 0; }
      if (eps > 31) { eps = 31; mant = 0; }
      s.eps = eps; s.mant = mant;
      s.delta = (float)(ldexp(1.0, Rb - eps) * (1.0 + mant / 2048.0));
    }
    s.Mb = J2K_GUARD + s.eps - 1;
    s.gx = (w + cbw - 1) / cbw; s.gy = (h + cbh - 1) / cbh;
    pp.sb.push_back(s);
  };
  add(0, 0, ws[L], hs[L], 0, L);
  for (int d = L; d >= 1; d--) {
    int lw = ws[d], lh = hs[d], hw = ws[d - 1] - lw, hh = hs[d - 1] - lh;
    add(lw, 0, hw, lh, 1, d);
    add(0, lh, lw, hh, 2, d);
    add(lw, lh, hw, hh, 3, d);
  }
  int k = 0;
  for (auto& s : pp.sb) { s.first_cb = k; k += s.gx * s.gy; }
  pp.ncb = k;
}

// ---- Tier-2 helpers ---------------------------------------------------------------------------
struct OBuf {
  uint8_t* p; size_t n, cap; bool ovf;
  void put(int v) { if (n < cap) p[n] = (uint8_t)v; else ovf = true; n++; }
  void u16(int v) { put(v >> 8); put(v & 255); }
  void u32(uint32_t v) { u16(v >> 16); u16(v & 0xFFFF); }
  void bytes(const uint8_t* s, size_t k) { if (n + k <= cap) memcpy(p + n, s, k); else ovf = true; n += k; }
  void skip(size_t k) { n += k; }
};
struct BitW {  // packet-header bit writer with the 0xFF stuffing rule (T.800 B.10.1)
  OBuf* o; int acc = 0, nbits = 0; bool last_ff = false;
  void bit(int v) {
    int cap = last_ff ? 7 : 8;
    acc = (acc << 1) | (v & 1); nbits++;
    if (nbits == cap) { o->put(acc); last_ff = acc == 0xFF; acc = 0; nbits = 0; }
  }
  void bits(unsigned v, int n) { for (int i = n - 1; i >= 0; i--) bit((v >> i) & 1); }
  void flush() {
    if (nbits) { int cap = last_ff ? 7 : 8; int v = acc << (cap - nbits); o->put(v); last_ff = v == 0xFF; acc = 0; nbits = 0; }
    if (last_ff) { o->put(0); last_ff = false; }
  }
};
struct TagTree {  // T.800 B.10.2
  int w = 0, h = 0, nlev = 0;
  int lw[16], lh[16], off[16];
  std::vector<int> val, low;
  std::vector<uint8_t> known;
  void init(int w_, int h_, const int* leaf) {
    w = w_; h = h_; nlev = 0;
    int n = 0, cw = w, ch = h;
    for (;;) { lw[nlev] = cw; lh[nlev] = ch; off[nlev] = n; n += cw * ch; nlev++; if (cw == 1 && ch == 1) break; cw = (cw + 1) >> 1; ch = (ch + 1) >> 1; }
    val.assign(n, 0); low.assign(n, 0); known.assign(n, 0);
    for (int i = 0; i < w * h; i++) val[i] = leaf[i];
    for (int l = 1; l < nlev; l++)
      for (int y = 0; y < lh[l]; y++)
        for (int x = 0; x < lw[l]; x++) {
          int mn = 0x7FFFFFFF;
          for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) {
              int cy = 2 * y + dy, cx = 2 * x + dx;
              if (cy < lh[l - 1] && cx < lw[l - 1]) mn = std::min(mn, val[off[l - 1] + cy * lw[l - 1] + cx]);
            }
          val[off[l] + y * lw[l] + x] = mn;
        }
  }
  void encode(BitW& b, int lx, int ly, int threshold) {
    int path[16], n = 0;
    for (int l = 0; l < nlev; l++) path[n++] = off[l] + (ly >> l) * lw[l] + (lx >> l);
    int lo = 0;
    for (int k = n - 1; k >= 0; k--) {
      int nd = path[k];
      if (lo > low[nd]) low[nd] = lo; else lo = low[nd];
      while (lo < threshold) {
        if (lo >= val[nd]) { if (!known[nd]) { b.bit(1); known[nd] = 1; } break; }
        b.bit(0); lo++;
      }
      low[nd] = lo;
    }
  }
};
int ilog2i(int v) { int l = 0; while (v > 1) { v >>= 1; l++; } return l; }
void put_npasses(BitW& b, int n) {  // T.800 Table B.4
  if (n == 1) b.bit(0);
  else if (n == 2) b.bits(2, 2);
  else if (n <= 5) b.bits(0xC | (n - 3), 4);
  else if (n <= 36) b.bits(0x1E0 | (n - 6), 9);
  else b.bits(0xFF80 | (n - 37), 16);
}
// Kakadu-style logarithmic slope -> Lagrangian threshold in 8-bit-sample^2 per byte.  APPROXIMATE: the
// `-slope` semantics are Kakadu-private; s = 256*(192 + log2(dD/dL)), D in full-scale^2.
double slope_to_lambda(int slope) { return pow(2.0, slope / 256.0 - 176.0); }
// (the passes on the convex hull of (length, distortion) with slope >= lambda -- PCRD-opt, T.800 J.14 -- are chosen on the device:
// k_rate_alloc, j2k_kernels.hip)

struct HostCb {  // host view of one coded block
  int len, npasses, nbps, zbp;
  int take;                     // bytes of its segment that any layer uses (what is gathered and copied to the host)
  size_t data_off;              // into the gathered byte buffer
  int layer_passes[16];
  int first_layer, lblock;
  bool included;
};

}  // namespace

// Persistent Tier-2 workers of one encoder context.  Their number is bounded by QSVC_T2_THREADS (default min(8, planes)): a
// host that runs several contexts per process and several ranks per node sets it to cores / (ranks x sessions).
struct T2Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv, done_cv;
  std::function<void(int)> job;
  int gen = 0, pending = 0, nworkers = 0;
  bool stop = false;
  void start(int n) {
    nworkers = n;
    for (int t = 0; t < n; t++)
      th.emplace_back([this, t] {
        (void)pthread_setname_np(pthread_self(), "qsvc-t2");   // (bench.py's host_cpu tells the threads apart by name)
        int seen = 0;
        for (;;) {
          std::function<void(int)> f;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen; f = job;
          }
          f(t + 1);
          { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) done_cv.notify_all(); }
        }
      });
  }
  // run f(0) here and f(1..nworkers) on the workers; returns when all are done
  void run(const std::function<void(int)>& f) {
    if (nworkers) { std::lock_guard<std::mutex> lk(mu); job = f; pending = nworkers; gen++; }
    cv.notify_all();
    f(0);
    if (nworkers) { std::unique_lock<std::mutex> lk(mu); done_cv.wait(lk, [&] { return pending == 0; }); }
  }
  ~T2Pool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};

// Page-locked host array (grow-only): the per-block results come back, and the job tables go up, by DMA without a staging copy.
template <class T>
struct Pinned {
  T* p = nullptr; size_t cap = 0, n = 0;
  bool resize(size_t k) {
    if (k > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr; cap = 0;
      const size_t want = k + k / 4 + 64;
      if (hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) return false;
      cap = want;
    }
    n = k;
    return true;
  }
  T* data() { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
};

struct J2kState {
  T2Pool* pool = nullptr;
  std::function<int()> after_resolve;   // one-shot: queued work that needs the final I/B decisions (the motion planes of qsvc_compress_code)
  DevBuf coefA, llA, llB, sbtab_d, planes_d, cbs_d, nbps_d, mags, outbuf, res_d, plen_d, pdist_d, offs_d, lens_d, gath_d, err_d, up_d, mvup_d;
  DevBuf joboff_d, order_d, segs_d, cnt_d, dist_d;  // Tier-1: first job of each block, MQ slot table, decision segments
  DevBuf tilemap_d;                                 // the DWT levels' workgroup lists (dwt_build_tilemap)
  std::vector<uint32_t> tilemap; std::vector<int> tile_first; std::vector<int> tile_key;   // ... and the geometry they were built for
  std::vector<uint8_t> up_tilemap; const void* up_tilemap_at = nullptr;
  DevBuf t1tabs_d;                                  // lookup tables of the context-modelling kernel (built at first use)
  DevBuf nbeff_d, njobs_d;                          // device job list: bit-planes as the Tier-1 kernels see them, total jobs
  size_t jobs_cap = 0;                              // jobs the segment / count (/ distortion) buffers were last sized for
  bool jobs_cap_dist = false;
  Pinned<int> h_nbps, h_joboff;
  std::vector<int> h_order;
  std::vector<DwtSb> sbtab;
  uint8_t* gath_h = nullptr; size_t gath_h_cap = 0;
  hipEvent_t copied[4] = {nullptr, nullptr, nullptr, nullptr};  // the code bytes reach the host in up to four pieces
  // Tier-1 in parts (longest blocks first): the MQ coder of part k runs on its own stream beside the context modelling of part k+1
  hipStream_t mq_st[3] = {nullptr, nullptr, nullptr};
  hipEvent_t cm_done[3] = {nullptr, nullptr, nullptr}, mq_done[3] = {nullptr, nullptr, nullptr};
  std::vector<PlaneDesc> planes;
  std::vector<CbDesc> cbs;
  // what the device tables hold (a video encoder codes the same geometry step after step: identical tables are not uploaded again)
  std::vector<uint8_t> up_planes, up_cbs, up_sbtab;
  const void *up_planes_at = nullptr, *up_cbs_at = nullptr, *up_sbtab_at = nullptr;
  Pinned<CbOut> res;
  Pinned<int> plen;
  Pinned<long long> pdist;
  Pinned<uint8_t> lp;                 // [ncb][16] passes per layer from k_rate_alloc
  DevBuf ra_wgt_d, ra_cbp_d, ra_pl_d, lp_d;
  Pinned<int> h_err;
  std::vector<HostCb> hcb;
  uint64_t st_cb = 0, st_pass = 0, st_sym = 0;
};

// H2D copy of a host-built table unless the device buffer already holds exactly these bytes (same buffer, same content).  The
// shadow copy is what was last uploaded from; the source of the async copy is the shadow, which outlives the call.
static int upload_if_changed(qsvc_ctx* c, void* dev, const void* src, size_t bytes, std::vector<uint8_t>& shadow, const void*& at) {
  if (at == dev && shadow.size() == bytes && (bytes == 0 || memcmp(shadow.data(), src, bytes) == 0)) return QSVC_OK;
  hipError_t e = ctx_sync(c);  // (a copy still reading the old shadow must have finished before it is overwritten)
  if (e != hipSuccess) return c->hipfail(e, "hipStreamSynchronize");
  shadow.assign((const uint8_t*)src, (const uint8_t*)src + bytes);
  at = dev;
  if (bytes) {
    e = hipMemcpyAsync(dev, shadow.data(), bytes, hipMemcpyHostToDevice, c->st);
    if (e != hipSuccess) { at = nullptr; return c->hipfail(e, "hipMemcpyAsync(table)"); }
  }
  return QSVC_OK;
}

void j2k_release(qsvc_ctx* c) {
  if (c->j2k_mv) {  // the motion planes' state first (same teardown)
    J2kState* keep = c->j2k;
    c->j2k = c->j2k_mv; c->j2k_mv = nullptr;
    j2k_release(c);
    c->j2k = keep;
  }
  J2kState* s = c->j2k;
  if (!s) return;
  DevBuf* all[] = {&s->coefA, &s->llA, &s->llB, &s->sbtab_d, &s->planes_d, &s->cbs_d, &s->nbps_d, &s->mags, &s->outbuf, &s->res_d, &s->plen_d,
                   &s->pdist_d, &s->offs_d, &s->lens_d, &s->gath_d, &s->err_d, &s->up_d, &s->mvup_d, &s->joboff_d, &s->order_d, &s->segs_d, &s->cnt_d, &s->dist_d, &s->nbeff_d, &s->njobs_d, &s->t1tabs_d, &s->tilemap_d, &s->ra_wgt_d, &s->ra_cbp_d, &s->ra_pl_d, &s->lp_d};
  for (auto* b : all) b->release();
  if (s->gath_h) (void)hipHostFree(s->gath_h);
  for (auto& e : s->copied) if (e) (void)hipEventDestroy(e);
  for (int k = 0; k < 3; k++) {
    if (s->cm_done[k]) (void)hipEventDestroy(s->cm_done[k]);
    if (s->mq_done[k]) (void)hipEventDestroy(s->mq_done[k]);
    if (s->mq_st[k]) (void)hipStreamDestroy(s->mq_st[k]);
  }
  s->h_nbps.release(); s->h_joboff.release(); s->res.release(); s->plen.release(); s->pdist.release(); s->lp.release(); s->h_err.release();
  delete s->pool;
  delete s;
  c->j2k = nullptr;
}

// Write one plane's code-stream (T.800 Annex A markers + LRCP packets) from its coded blocks.
// out == nullptr: only the length is computed (no byte is read or written; the blocks' packet state is left as it was found).
static long write_codestream(const PlanePlan& pp, const qsvc_j2k_params& prm, int layers, int cbw, int cbh, HostCb* cb,
                             const int* plen_all /*[ncb][T1_MAXPASS]*/, const uint8_t* data, uint8_t* out, size_t cap) {
  const bool size_only = out == nullptr;
  if (size_only) cap = 0;
  const int L = pp.L, nsb = (int)pp.sb.size();
  OBuf o{out, 0, cap, false};
  o.u16(0xFF4F);                                      // SOC
  o.u16(0xFF51); o.u16(41); o.u16(0);                 // SIZ
  o.u32(pp.W); o.u32(pp.H); o.u32(0); o.u32(0); o.u32(pp.W); o.u32(pp.H); o.u32(0); o.u32(0);
  o.u16(1); o.put((pp.prec - 1) | (pp.prec == 16 ? 0x80 : 0)); o.put(1); o.put(1);  // Ssiz: 8-bit unsigned / 16-bit signed
  o.u16(0xFF52); o.u16(12);                           // COD
  o.put(prm.use_sop ? 0x02 : 0x00);
  o.put(0); o.u16(layers); o.put(0);
  o.put(L); o.put(ilog2i(cbw) - 2); o.put(ilog2i(cbh) - 2); o.put(0); o.put(prm.reversible ? 1 : 0);
  o.u16(0xFF5C);                                      // QCD
  if (prm.reversible) { o.u16(3 + nsb); o.put((J2K_GUARD << 5) | 0); for (auto& s : pp.sb) o.put(s.eps << 3); }
  else { o.u16(3 + 2 * nsb); o.put((J2K_GUARD << 5) | 2); for (auto& s : pp.sb) o.u16((s.eps << 11) | s.mant); }
  size_t sot = o.n;
  o.u16(0xFF90); o.u16(10); o.u16(0); o.u32(0); o.put(0); o.put(1);  // SOT (Psot patched below)
  o.u16(0xFF93);                                      // SOD
  std::vector<TagTree> incl(nsb), zbpt(nsb);
  std::vector<int> v1, v2;
  for (int s = 0; s < nsb; s++) {
    int n = pp.sb[s].gx * pp.sb[s].gy;
    if (!n) continue;
    v1.resize(n); v2.resize(n);
    for (int i = 0; i < n; i++) { v1[i] = cb[pp.sb[s].first_cb + i].first_layer; v2[i] = cb[pp.sb[s].first_cb + i].zbp; }
    incl[s].init(pp.sb[s].gx, pp.sb[s].gy, v1.data()); zbpt[s].init(pp.sb[s].gx, pp.sb[s].gy, v2.data());
  }
  int nsop = 0;
  for (int l = 0; l < layers; l++)
    for (int r = 0; r <= L; r++) {
      int s0 = r == 0 ? 0 : 1 + 3 * (r - 1), s1 = r == 0 ? 1 : s0 + 3;
      if (prm.use_sop) { o.u16(0xFF91); o.u16(4); o.u16(nsop & 0xFFFF); }
      nsop++;
      BitW bw; bw.o = &o;
      bool any = false;
      for (int s = s0; s < s1 && !any; s++)
        for (int i = 0; i < pp.sb[s].gx * pp.sb[s].gy; i++) {
          HostCb& c = cb[pp.sb[s].first_cb + i];
          if (c.layer_passes[l] > (l ? c.layer_passes[l - 1] : 0)) { any = true; break; }
        }
      bw.bit(any);
      if (any)
        for (int s = s0; s < s1; s++)
          for (int i = 0; i < pp.sb[s].gx * pp.sb[s].gy; i++) {
            int ci = pp.sb[s].first_cb + i;
            HostCb& c = cb[ci];
            const int* pl = plen_all ? plen_all + (size_t)ci * T1_MAXPASS : nullptr;
            int bx = i % pp.sb[s].gx, by = i / pp.sb[s].gx;
            int prev = l ? c.layer_passes[l - 1] : 0, now = c.layer_passes[l], add = now - prev;
            if (!c.included) {
              incl[s].encode(bw, bx, by, l + 1);
              if (add <= 0) continue;
              zbpt[s].encode(bw, bx, by, 0x7FFFFFF);
              c.included = true;
            } else {
              bw.bit(add > 0);
              if (add <= 0) continue;
            }
            put_npasses(bw, add);
            int len = pl ? pl[now - 1] - (prev ? pl[prev - 1] : 0) : c.len;   // no table: all passes in this one layer
            int need = 1; while ((len >> need) && need < 32) need++;
            int have = c.lblock + ilog2i(add);
            int inc = need > have ? need - have : 0;
            for (int q = 0; q < inc; q++) bw.bit(1);
            bw.bit(0);
            c.lblock += inc;
            bw.bits((unsigned)len, c.lblock + ilog2i(add));
          }
      bw.flush();
      if (any)
        for (int s = s0; s < s1; s++)
          for (int i = 0; i < pp.sb[s].gx * pp.sb[s].gy; i++) {
            int ci = pp.sb[s].first_cb + i;
            HostCb& c = cb[ci];
            const int* pl = plen_all ? plen_all + (size_t)ci * T1_MAXPASS : nullptr;
            int prev = l ? c.layer_passes[l - 1] : 0, now = c.layer_passes[l];
            if (now <= prev) continue;
            int a = pl && prev ? pl[prev - 1] : 0, e = pl ? pl[now - 1] : c.len;
            if (size_only) o.skip((size_t)(e - a)); else o.bytes(data + c.data_off + a, (size_t)(e - a));
          }
    }
  size_t psot = o.n - sot;
  if (sot + 10 <= cap) { out[sot + 6] = (uint8_t)(psot >> 24); out[sot + 7] = (uint8_t)(psot >> 16); out[sot + 8] = (uint8_t)(psot >> 8); out[sot + 9] = (uint8_t)psot; }
  o.u16(0xFFD9);                                      // EOC
  if (size_only) {
    for (int k = 0; k < pp.ncb; k++) { cb[k].included = false; cb[k].lblock = 3; }
    return (long)o.n;
  }
  return o.ovf ? -1 : (long)o.n;
}

struct PlaneSrc { const uint8_t* src; int stride, W, H; const qsvc_j2k_params* prm; int s16; };

// Encode a batch of device-resident planes.  All planes share the transform/quantiser/code-block parameters
// of src[0].prm; the number of layers and the slopes may differ per plane (they only steer the host's PCRD).
static int j2k_encode_chunk(qsvc_ctx* c, const std::vector<PlaneSrc>& src, uint8_t* out, size_t cap, size_t* offs) {
  TuningScope tscope(&c->tune);
  if (!c->j2k) c->j2k = new J2kState();
  J2kState& S = *c->j2k;
  // A batch is a run of picture planes (8-bit, one transform, step, level count and code-block size) optionally FOLLOWED by a run of
  // 16-bit signed planes coded reversibly without any decomposition (the motion-field components of `mctf compress`): the second
  // group goes through level shift + pack instead of the DWT kernels and then shares the Tier-1 launches, the gather and the copy.
  const int np_all = (int)src.size();
  int npA = 0;
  while (npA < np_all && !src[npA].s16) npA++;
  for (int p = npA; p < np_all; p++)
    if (!src[p].s16) return c->fail(QSVC_ERR_INVALID, "16-bit planes must follow the 8-bit planes of a batch");
  const qsvc_j2k_params* prm = src[0].prm;
  bool any_dist = false;
  for (int q = 0; q < np_all; q++) {
    const PlaneSrc& p = src[q];
    const qsvc_j2k_params* g = q < npA ? prm : src[npA].prm;   // the group's first plane
    if (p.prm->reversible != g->reversible || p.prm->levels != g->levels || p.prm->cblk_w != prm->cblk_w ||
        p.prm->cblk_h != prm->cblk_h || p.prm->base_step != g->base_step)
      return c->fail(QSVC_ERR_INVALID, "planes of one batch must share transform, step and code-block parameters");
    if (p.prm->layers > 16) return c->fail(QSVC_ERR_INVALID, "at most 16 quality layers");
    if (p.prm->layers > 1 || p.prm->slopes[0] > 0) any_dist = true;
  }
  const bool rev = prm->reversible != 0;
  int cbw = prm->cblk_w ? prm->cblk_w : 64, cbh = prm->cblk_h ? prm->cblk_h : 64;
  if (prm->levels < 0 || prm->levels > J2K_MAXLEV) return c->fail(QSVC_ERR_INVALID, "Clevels %d", prm->levels);
  if (cbw > 64 || cbh > 64 || cbw < 4 || cbh < 4 || (cbw & (cbw - 1)) || (cbh & (cbh - 1)))
    return c->fail(QSVC_ERR_UNSUPPORTED, "code-block %dx%d (powers of two in 4..64)", cbw, cbh);
  const int np = (int)src.size();
  // distinct plane geometries of the batch
  std::vector<PlanePlan> plans;
  std::vector<int> plan_of(np);
  for (int p = 0; p < np; p++) {
    int k = -1;
    const int prec = src[p].s16 ? 16 : 8;
    for (size_t q = 0; q < plans.size(); q++) if (plans[q].W == src[p].W && plans[q].H == src[p].H && plans[q].prec == prec) { k = (int)q; break; }
    if (k < 0) {
      plans.emplace_back();
      build_plan(plans.back(), src[p].W, src[p].H, src[p].prm->levels, src[p].prm->reversible != 0, src[p].prm->base_step > 0 ? src[p].prm->base_step : 1.0f / 256, cbw, cbh, prec);
      k = (int)plans.size() - 1;
    }
    plan_of[p] = k;
  }
  for (int p = 0; p < np; p++)
    if (src[p].s16 && (!src[p].prm->reversible || plans[plan_of[p]].L != 0))
      return c->fail(QSVC_ERR_UNSUPPORTED, "16-bit signed planes are coded reversibly with Clevels=0 only (motion fields)");
  int maxL = 0, maxW = 0, maxH = 0;
  for (int p = 0; p < npA; p++) maxL = std::max(maxL, plans[plan_of[p]].L);
  // the planes that go through level shift + pack: the 16-bit group, or everything when no plane has a decomposition level
  const int npD = maxL > 0 ? npA : 0, npP = np - npD;   // DWT group [0, npD), pack group [npD, np)
  const bool revP = npP && src[npD].s16 ? true : rev;
  int maxWP = 0, maxHP = 0;
  S.planes.resize(np);
  size_t total = 0;   // samples of the pack group (the only planes that exist as a coefficient plane)
  std::vector<int> first_cb(np + 1);
  int ncb = 0;
  for (int p = 0; p < np; p++) {
    S.planes[p].off = p < npD ? 0 : total; S.planes[p].src = src[p].src; S.planes[p].u8stride = src[p].stride; S.planes[p].W = src[p].W; S.planes[p].H = src[p].H; S.planes[p].s16 = src[p].s16;
    if (p < npD) { maxW = std::max(maxW, src[p].W); maxH = std::max(maxH, src[p].H); }
    else { total += (size_t)src[p].W * src[p].H; maxWP = std::max(maxWP, src[p].W); maxHP = std::max(maxHP, src[p].H); }
    first_cb[p] = ncb; ncb += plans[plan_of[p]].ncb;
  }
  first_cb[np] = ncb;
  const int ncbD = first_cb[npD], ncbP = ncb - ncbD;
  // the planes of the DWT group must share L for the batched launches (same Clevels, sizes large enough)
  for (int p = 0; p < npD; p++)
    if (plans[plan_of[p]].L != maxL) return c->fail(QSVC_ERR_UNSUPPORTED, "planes of one batch must allow the same number of DWT levels");
  for (int p = npD; p < np; p++)
    if (plans[plan_of[p]].L != 0) return c->fail(QSVC_ERR_UNSUPPORTED, "planes of one batch must allow the same number of DWT levels");
  S.cbs.resize(ncb);
  for (int p = 0; p < np; p++) {
    const PlanePlan& pl = plans[plan_of[p]];
    int k = first_cb[p];
    for (auto& s : pl.sb)
      for (int by = 0; by < s.gy; by++)
        for (int bx = 0; bx < s.gx; bx++, k++) {
          CbDesc& d = S.cbs[k];
          d.plane = p; d.x0 = s.x0 + bx * cbw; d.y0 = s.y0 + by * cbh;
          d.w = (short)((bx + 1) * cbw <= s.w ? cbw : s.w - bx * cbw);
          d.h = (short)((by + 1) * cbh <= s.h ? cbh : s.h - by * cbh);
          d.orient = (short)s.orient; d.Mb = (short)s.Mb; d.delta = s.delta;
        }
  }
  // ---- device plan: compact low-band buffers (ping-pong), subband table of the DWT kernel ----------------------------------
  const bool want_dist = any_dist;
  int r;
  size_t llA_n = 0, llB_n = 0;
  const int nsb = 1 + 3 * maxL;
  S.sbtab.resize((size_t)np * nsb);
  for (int p = 0; p < np; p++) {
    const int w1 = (src[p].W + 1) >> 1, h1 = (src[p].H + 1) >> 1, w2 = (w1 + 1) >> 1, h2 = (h1 + 1) >> 1;
    S.planes[p].ll[0] = llA_n; S.planes[p].ll[1] = llB_n;
    llA_n += ((size_t)((w1 + 3) & ~3) * h1 + 7) & ~(size_t)7;   // 16-byte aligned plane starts (int16 / float32 elements)
    llB_n += ((size_t)((w2 + 3) & ~3) * h2 + 7) & ~(size_t)7;
    const PlanePlan& pl = plans[plan_of[p]];
    for (int k = 0; k < nsb; k++) {
      if (k >= (int)pl.sb.size()) { S.sbtab[(size_t)p * nsb + k] = DwtSb{0, 0, 0, 0.0f}; continue; }   // (pack-group planes: LL only)
      const Subband& sb = pl.sb[k];
      S.sbtab[(size_t)p * nsb + k] = DwtSb{first_cb[p] + sb.first_cb, sb.gx, sb.Mb, sb.delta};
    }
  }
  const size_t ll_elem = rev ? 2 : 4;
  if (npP) { if ((r = S.coefA.ensure(c, total * 4))) return r; }
  if (maxL > 0) {
    if ((r = S.llA.ensure(c, llA_n * ll_elem + 64))) return r;
    if ((r = S.llB.ensure(c, llB_n * ll_elem + 64))) return r;
    if ((r = S.sbtab_d.ensure(c, S.sbtab.size() * sizeof(DwtSb)))) return r;
    if ((r = upload_if_changed(c, S.sbtab_d.p, S.sbtab.data(), S.sbtab.size() * sizeof(DwtSb), S.up_sbtab, S.up_sbtab_at))) return r;
  }
  if ((r = S.planes_d.ensure(c, np * sizeof(PlaneDesc)))) return r;
  if ((r = S.cbs_d.ensure(c, (size_t)ncb * sizeof(CbDesc)))) return r;
  if ((r = S.nbps_d.ensure(c, (size_t)ncb * 4))) return r;
  if ((r = S.mags.ensure(c, (size_t)ncb * T1_MAGS_BYTES))) return r;
  if ((r = S.outbuf.ensure(c, (size_t)ncb * J2K_OUTCAP))) return r;
  if ((r = S.res_d.ensure(c, (size_t)ncb * sizeof(CbOut)))) return r;
  if ((r = S.plen_d.ensure(c, (size_t)ncb * T1_MAXPASS * 4))) return r;
  if (want_dist) if ((r = S.pdist_d.ensure(c, (size_t)ncb * T1_MAXPASS * 8))) return r;
  if ((r = S.offs_d.ensure(c, (size_t)ncb * 8))) return r;
  if ((r = S.lens_d.ensure(c, (size_t)ncb * 4))) return r;
  if ((r = S.err_d.ensure(c, 16))) return r;
  if ((r = upload_if_changed(c, S.planes_d.p, S.planes.data(), np * sizeof(PlaneDesc), S.up_planes, S.up_planes_at))) return r;
  if ((r = upload_if_changed(c, S.cbs_d.p, S.cbs.data(), (size_t)ncb * sizeof(CbDesc), S.up_cbs, S.up_cbs_at))) return r;
  HIPCHK(c, hipMemsetAsync(S.err_d.p, 0, 16, c->st));
  { int rr_ = qsvc_resolve(c); if (rr_) return rr_; }  // pending I/B decisions (qsvc_api.hip): the pictures must be final from here on
  if (S.after_resolve) {
    std::function<int()> f;
    f.swap(S.after_resolve);
    if ((r = f())) return r;
  }
  Prof pf;
  if (maxL > 0) {
    // one kernel per level: tile in LDS, vertical + horizontal lifting, LL to the next level, the other subbands quantised and
    // packed straight into the Tier-1 magnitudes; the blocks' largest magnitudes are collected by atomicMax into nbps_d
    int minW = maxW, minH = maxH;
    for (int p = 0; p < npD; p++) { const PlaneDesc& q = S.planes[p]; if (q.W < minW) minW = q.W; if (q.H < minH) minH = q.H; }
    const uint32_t* tm = nullptr;
    if (dwt_tilemap_fits(npD, maxW, maxH)) {  // one workgroup per tile that exists (the planes of a batch differ in size)
      std::vector<int> key;
      key.reserve(2 * npD + 1);
      key.push_back(maxL);
      for (int p = 0; p < npD; p++) { key.push_back(S.planes[p].W); key.push_back(S.planes[p].H); }
      if (key != S.tile_key) { dwt_build_tilemap(S.planes.data(), npD, maxL, S.tilemap, S.tile_first); S.tile_key.swap(key); }
      if ((r = S.tilemap_d.ensure(c, S.tilemap.size() * 4 + 16))) return r;
      if ((r = upload_if_changed(c, S.tilemap_d.p, S.tilemap.data(), S.tilemap.size() * 4, S.up_tilemap, S.up_tilemap_at))) return r;
      tm = (const uint32_t*)S.tilemap_d.p;
    }
    // (the segment begins where the device's work does: the host-side table checks above are not the transform's time)
    pf.begin(c, rev ? "j2k_dwt53" : "j2k_dwt97");
    HIPCHK(c, hipMemsetAsync(S.nbps_d.p, 0, (size_t)ncb * 4, c->st));
    launch_dwt_tiles(c->st, (const PlaneDesc*)S.planes_d.p, npD, maxW, maxH, minW, minH, (const DwtSb*)S.sbtab_d.p, maxL, rev, S.llA.p, S.llB.p,
                     (unsigned long long*)S.mags.p, (int*)S.nbps_d.p, nullptr, ilog2i(cbw), ilog2i(cbh), (int*)S.err_d.p, tm, S.tile_first.data());
    launch_t1_nbps(c->st, (const CbDesc*)S.cbs_d.p, ncbD, (int*)S.nbps_d.p, (int*)S.err_d.p);
    pf.end(c);
  }
  if (npP) {  // Clevels = 0 (motion-field planes): level shift, then quantise + pack per code-block
    pf.begin(c, src[npD].s16 ? "mv_t1_pack" : "j2k_t1_pack");
    launch_level_shift(c->st, (const PlaneDesc*)S.planes_d.p + npD, npP, maxWP, maxHP, S.coefA.p, revP);
    launch_t1_pack_blk(c->st, (const CbDesc*)S.cbs_d.p + ncbD, ncbP, (const PlaneDesc*)S.planes_d.p, S.coefA.p,
                       (unsigned long long*)S.mags.p + (size_t)ncbD * (T1_MAGS_BYTES / 8), (int*)S.nbps_d.p + ncbD, (int*)S.err_d.p, revP);
    pf.end(c);
  }
  // Tier-1.  The job list (bit-planes per block -> first job of each block, MQ slot table) is built ON THE DEVICE when the segment
  // buffers of an earlier call can be reused: no host round trip between the DWT and Tier-1 (k_t1_joblist; QSVC_T1_DEVLIST=0 or
  // QSVC_MQ_SORT=0: always the host's list).  If the batch needs more jobs than the buffers hold, the device list disarms the
  // Tier-1 kernels and flags it, and the stage is run again with the host's list and larger buffers.
  const int MQW = 64;
  const int nslots = ((ncb + MQW - 1) / MQW) * MQW;
  const bool mq_sort = c->tune.mq_sort != 0;
  bool devlist = c->tune.t1_devlist && mq_sort && S.jobs_cap > 0 && (!want_dist || S.jobs_cap_dist);
  if (!S.res.resize(ncb) || !S.h_err.resize(2) || (want_dist && (!S.plen.resize((size_t)ncb * T1_MAXPASS) || !S.lp.resize((size_t)ncb * 16))))
    return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc (page-locked tables)");
  int& err = S.h_err[1];
  std::vector<double> ra_wgt;   // (tables of the device-side rate allocation: they live until the stream has been waited for)
  std::vector<int> ra_cbp;
  std::vector<RaPlane> ra_pl;
  for (;;) {
    const int* nbp = (const int*)S.nbps_d.p;   // the bit-plane counts the Tier-1 kernels read
    size_t njobs = 1;
    if ((r = S.order_d.ensure(c, (size_t)nslots * 4))) return r;
    if ((r = S.joboff_d.ensure(c, (size_t)ncb * 4))) return r;
    if (devlist) {
      if ((r = S.nbeff_d.ensure(c, (size_t)ncb * 4))) return r;
      if ((r = S.njobs_d.ensure(c, 16))) return r;
      launch_t1_joblist(c->st, (const CbDesc*)S.cbs_d.p, ncb, (const int*)S.nbps_d.p, (int)std::min<size_t>(S.jobs_cap, 0x7fffffff), (int*)S.nbeff_d.p,
                        (int*)S.joboff_d.p, (int*)S.order_d.p, nslots, (int*)S.njobs_d.p, (int*)S.err_d.p);
      nbp = (const int*)S.nbeff_d.p;
    } else {
    // the per-block bit-plane counts decide the job list: one context-modelling wave per (block, bit-plane)
    if (!S.h_nbps.resize(ncb)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc (page-locked tables)");
    int& perr = S.h_err[0];
    perr = 0;
    HIPCHK(c, hipMemcpyAsync(S.h_nbps.data(), S.nbps_d.p, (size_t)ncb * 4, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(&perr, S.err_d.p, 4, hipMemcpyDeviceToHost, c->st));
    {
      const auto w0 = std::chrono::steady_clock::now();
      HIPCHK(c, ctx_sync(c));
      c->host_time("host_wait_dwt", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    }
    // a block with more bit-planes than its subband allows (or than the 15-bit magnitude format holds) must never reach
    // the context-modelling / MQ kernels: their job and pass tables are sized for nbps <= 15
    if (perr & 1) return c->fail(QSVC_ERR_UNSUPPORTED, "a coefficient exceeds its subband's magnitude bit-planes (M_b) or 15 bits");
    if (!S.h_joboff.resize(ncb)) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc (page-locked tables)");
    njobs = 0;
    for (int b = 0; b < ncb; b++) { S.h_joboff[b] = (int)njobs; njobs += S.h_nbps[b] > 15 ? 15 : S.h_nbps[b]; }
    // MQ slot table: slot k of the MQ launch codes block order[k] (ncb = idle lane).  A wave runs until its longest block is
    // coded, so blocks of similar expected length (bit-planes x area) share a wave, the longest first (counting sort);
    // QSVC_MQ_SORT=0 keeps the natural order.
    {
      std::vector<int>& ord = S.h_order;
      ord.resize(ncb);
      if (mq_sort) {
        int hist[15 * 64 + 2] = {0};
        auto key = [&](int b) { int nb = S.h_nbps[b] > 15 ? 15 : S.h_nbps[b]; return 15 * 64 - nb * (((int)S.cbs[b].w * S.cbs[b].h + 63) >> 6); };
        for (int b = 0; b < ncb; b++) hist[key(b) + 1]++;
        for (int k = 1; k <= 15 * 64; k++) hist[k] += hist[k - 1];
        for (int b = 0; b < ncb; b++) ord[hist[key(b)]++] = b;
      } else {
        for (int b = 0; b < ncb; b++) ord[b] = b;
      }
      while ((int)ord.size() < nslots) ord.push_back(ncb);
    }
    HIPCHK(c, hipMemcpyAsync(S.order_d.p, S.h_order.data(), (size_t)nslots * 4, hipMemcpyHostToDevice, c->st));
    // the MQ stage reads joboff[b] of every block, also when no block of the batch has a bit-plane (all-zero planes)
    HIPCHK(c, hipMemcpyAsync(S.joboff_d.p, S.h_joboff.data(), (size_t)ncb * 4, hipMemcpyHostToDevice, c->st));
    // buffers with some headroom, so that the next calls (a video: much the same counts) can take the device list
    const size_t want_jobs = njobs + njobs / 8 + 64;
    if (want_jobs > S.jobs_cap || (want_dist && !S.jobs_cap_dist)) {
      if ((r = S.segs_d.ensure(c, want_jobs * (size_t)T1_SEG_BYTES))) return r;
      if ((r = S.cnt_d.ensure(c, want_jobs * 3 * 4))) return r;
      if (want_dist) if ((r = S.dist_d.ensure(c, want_jobs * 3 * 8))) return r;
      S.jobs_cap = std::min(S.segs_d.cap / (size_t)T1_SEG_BYTES, S.cnt_d.cap / 12);
      if (want_dist) { S.jobs_cap = std::min(S.jobs_cap, S.dist_d.cap / 24); S.jobs_cap_dist = true; }
      else S.jobs_cap_dist = S.dist_d.cap / 24 >= S.jobs_cap;
    }
    }
    // Tier-1 in parts (an experiment kept behind QSVC_T1_PARTS = 2 .. 4; default 1 = one launch each, back to back).  The MQ
    // coder is a serial chain per code-block: its kernel lasts as long as its longest block, whatever else the chip does, while
    // context modelling fills the chip.  The idea: cut the slot table (longest blocks first) into parts of growing size, model
    // part k on the main stream and start its MQ coder on a stream of its own beside the modelling of part k + 1.  MEASURED
    // (profiles/r03_*): no gain -- the modelling workgroups (11 KB of LDS each, 14 per CU) leave no room for an MQ
    // workgroup (25-40 KB) on any CU until they drain, so the side streams' kernels start when the modelling ends anyway
    // (single sequence 6.35 -> 6.77 ms with 4 parts, 12 sequences in flight 5200 -> 4280 fps).
    // Round 4: that is config 3 (15 657 blocks).  One 33-picture 4K sequence has 107 613 blocks and 7 + 5 ms of modelling + coding back
    // to back: three parts take 31.8 -> 26.9 ms off one sequence alone (two: 30.2, four: 26.9).  t1_parts = 0 (the default) decides by
    // the size of the batch.
    int nparts = c->tune.t1_parts > 0 ? c->tune.t1_parts : (nslots >= 65536 ? 3 : 1);
    if (!mq_sort || nslots < 64 * 16) nparts = 1;
    int cut[5] = {0, 0, 0, 0, 0};
    {
      static const int frac[4][4] = {{100, 0, 0, 0}, {25, 100, 0, 0}, {8, 30, 100, 0}, {5, 15, 40, 100}};
      const int nwg = nslots / MQW;
      for (int k = 0; k < nparts; k++) cut[k + 1] = k == nparts - 1 ? nslots : MQW * ((nwg * frac[nparts - 1][k] + 99) / 100);
      for (int k = 1; k <= nparts; k++) if (cut[k] < cut[k - 1]) cut[k] = cut[k - 1];
    }
    for (int k = 0; k + 1 < nparts; k++) {
      if (!S.mq_st[k]) HIPCHK(c, hipStreamCreateWithFlags(&S.mq_st[k], hipStreamNonBlocking));
      if (!S.cm_done[k]) HIPCHK(c, hipEventCreateWithFlags(&S.cm_done[k], hipEventDisableTiming));
      if (!S.mq_done[k]) HIPCHK(c, hipEventCreateWithFlags(&S.mq_done[k], hipEventDisableTiming));
    }
    auto mq_part = [&](hipStream_t st, int k) {
      if (cut[k + 1] > cut[k])
        launch_t1_mq(st, (const CbDesc*)S.cbs_d.p, ncb, nbp, (const int*)S.joboff_d.p, (const int*)S.order_d.p + cut[k], cut[k + 1] - cut[k],
                     (const uint8_t*)S.segs_d.p, (const int*)S.cnt_d.p, (const long long*)S.dist_d.p, (uint8_t*)S.outbuf.p, J2K_OUTCAP,
                     (CbOut*)S.res_d.p, (int*)S.plen_d.p, (long long*)S.pdist_d.p, want_dist ? 1 : 0, (int*)S.err_d.p);
    };
    if (!S.t1tabs_d.p) {
      if ((r = S.t1tabs_d.ensure(c, T1_TABLES_BYTES))) return r;
      launch_t1_tables(c->st, (uint8_t*)S.t1tabs_d.p);
    }
    pf.begin(c, src[0].s16 ? "mv_t1_cm" : "j2k_t1_cm");
    for (int k = 0; k < nparts; k++) {
      if (njobs && cut[k + 1] > cut[k])  // one context-modelling wave per block, its bit-planes one after the other (plane p = job joboff[block] + p)
        launch_t1_cm2(c->st, (const CbDesc*)S.cbs_d.p, ncb, nbp, (const int*)S.joboff_d.p, (const int*)S.order_d.p + cut[k],
                      cut[k + 1] - cut[k], (const unsigned long long*)S.mags.p, (uint8_t*)S.segs_d.p, (int*)S.cnt_d.p, (long long*)S.dist_d.p, want_dist ? 1 : 0,
                      (const uint8_t*)S.t1tabs_d.p);
      if (k + 1 < nparts) {
        HIPCHK(c, hipEventRecord(S.cm_done[k], c->st));
        HIPCHK(c, hipStreamWaitEvent(S.mq_st[k], S.cm_done[k], 0));
        mq_part(S.mq_st[k], k);
        HIPCHK(c, hipEventRecord(S.mq_done[k], S.mq_st[k]));
      }
    }
    pf.end(c);
    pf.begin(c, src[0].s16 ? "mv_t1_mq" : "j2k_t1_mq");  // (motion planes: their own segment names)  // (with parts: what of the MQ stage is left after the last part's context modelling)
    mq_part(c->st, nparts - 1);
    for (int k = 0; k + 1 < nparts; k++) HIPCHK(c, hipStreamWaitEvent(c->st, S.mq_done[k], 0));
    pf.end(c);
    // (the per-pass lengths and distortions only matter when some layer leaves passes out: lossless single-layer streams take every pass)
    err = 0;
    HIPCHK(c, hipMemcpyAsync(S.res.data(), S.res_d.p, (size_t)ncb * sizeof(CbOut), hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipMemcpyAsync(&err, S.err_d.p, 4, hipMemcpyDeviceToHost, c->st));
    if (want_dist) {
      // rate allocation on the device (k_rate_alloc): per block its subband's distortion weight and its plane, per plane the layers'
      // thresholds; what comes back is the per-pass lengths (Tier-2 needs them) and 16 bytes per block, not the distortion sums
      ra_wgt.resize(ncb); ra_cbp.resize(ncb); ra_pl.resize(np);
      std::vector<double>& wgt = ra_wgt;
      std::vector<int>& cbp = ra_cbp;
      std::vector<RaPlane>& rp = ra_pl;
      for (int p = 0; p < np; p++) {
        const PlanePlan& pl = plans[plan_of[p]];
        const qsvc_j2k_params* pprm = src[p].prm;
        RaPlane& q = rp[p];
        q.layers = pprm->layers < 1 ? 1 : (pprm->layers > 16 ? 16 : pprm->layers);
        q.last_all = pprm->slopes[q.layers - 1] <= 0;
        for (int l = 0; l < 16; l++) q.lam[l] = (l < q.layers && pprm->slopes[l] > 0) ? slope_to_lambda(pprm->slopes[l]) : 0.0;
        for (auto& sb : pl.sb) {
          double w = (double)sb.delta * norm_of(rev, sb.orient, sb.dlev);
          w *= w;
          for (int i = 0; i < sb.gx * sb.gy; i++) { const int k = first_cb[p] + sb.first_cb + i; wgt[k] = w; cbp[k] = p; }
        }
      }
      if ((r = S.ra_wgt_d.ensure(c, (size_t)ncb * 8)) || (r = S.ra_cbp_d.ensure(c, (size_t)ncb * 4)) || (r = S.ra_pl_d.ensure(c, (size_t)np * sizeof(RaPlane))) ||
          (r = S.lp_d.ensure(c, (size_t)ncb * 16))) return r;
      HIPCHK(c, hipMemcpyAsync(S.ra_wgt_d.p, wgt.data(), (size_t)ncb * 8, hipMemcpyHostToDevice, c->st));
      HIPCHK(c, hipMemcpyAsync(S.ra_cbp_d.p, cbp.data(), (size_t)ncb * 4, hipMemcpyHostToDevice, c->st));
      HIPCHK(c, hipMemcpyAsync(S.ra_pl_d.p, rp.data(), (size_t)np * sizeof(RaPlane), hipMemcpyHostToDevice, c->st));
      launch_rate_alloc(c->st, (const CbOut*)S.res_d.p, (const int*)S.plen_d.p, (const long long*)S.pdist_d.p, (const double*)S.ra_wgt_d.p,
                        (const int*)S.ra_cbp_d.p, (const RaPlane*)S.ra_pl_d.p, ncb, (uint8_t*)S.lp_d.p);
      HIPCHK(c, hipMemcpyAsync(S.plen.data(), S.plen_d.p, (size_t)ncb * T1_MAXPASS * 4, hipMemcpyDeviceToHost, c->st));
      HIPCHK(c, hipMemcpyAsync(S.lp.data(), S.lp_d.p, (size_t)ncb * 16, hipMemcpyDeviceToHost, c->st));
    }
    {
      HostCpu hc(c, "host_cpu_wait_tier1");
      const auto w0 = std::chrono::steady_clock::now();
      HIPCHK(c, ctx_sync(c));
      c->host_time(src[0].s16 ? "host_mv_wait_tier1" : "host_wait_tier1", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
    }
    HIPCHK(c, hipGetLastError());
    if (err & 2) return c->fail(QSVC_ERR_HIP, "internal: the MQ coder's wave pair lost step (a wait gave up)");
    if (err & 1) return c->fail(QSVC_ERR_UNSUPPORTED, "a coefficient exceeds its subband's magnitude bit-planes (M_b) or 15 bits");
    if (devlist && (err & 4)) {  // the batch outgrew the segment buffers: once more with the host's list (which resizes them)
      devlist = false;
      HIPCHK(c, hipMemsetAsync(S.err_d.p, 0, 16, c->st));
      continue;
    }
    break;
  }
  S.hcb.resize(ncb);
  S.st_cb += ncb;  // the statistics cover every plane group of the call (reset by j2k_encode_planes)
  // worker count: qsvc_set_tier2_threads, else QSVC_T2_THREADS as it was at this context's first encode (read once: the
  // environment is not consulted while other contexts' threads may be changing it), else 8; never more than planes
  if (!c->t2_default) { const char* e = getenv("QSVC_T2_THREADS"); int v = e ? atoi(e) : 8; c->t2_default = v < 1 ? 1 : (v > 64 ? 64 : v); }
  int t2_want = c->t2_threads > 0 ? c->t2_threads : c->t2_default;
  if (src[0].s16 && t2_want > 2) t2_want = 2;  // motion planes: kilobytes per stream, their state keeps at most one worker thread
  if (t2_want > np) t2_want = np;
  if (t2_want < 1) t2_want = 1;
  if (S.pool && S.pool->nworkers != t2_want - 1) { delete S.pool; S.pool = nullptr; }  // rebuilt only when the count changes
  if (!S.pool) { S.pool = new T2Pool(); S.pool->start(t2_want - 1); }
  const int nthreads = S.pool->nworkers + 1;
  std::vector<unsigned long long> goffs(ncb);
  std::vector<int> glens(ncb);
  // host, per plane and in parallel: rate allocation (which passes of a block go into which layer), then Tier-2
  auto rate_alloc = [&](int p, uint64_t& n_pass_out, uint64_t& n_sym_out) {
    const PlanePlan& pl = plans[plan_of[p]];
    const qsvc_j2k_params* pprm = src[p].prm;
    const int layers = pprm->layers < 1 ? 1 : pprm->layers;
    uint64_t n_pass = 0, n_sym = 0;   // (local: the callers' counters of different threads share cache lines)
    for (auto& sb : pl.sb) {
      for (int i = 0; i < sb.gx * sb.gy; i++) {
        int k = first_cb[p] + sb.first_cb + i;
        const CbOut& ro = S.res[k];
        HostCb& h = S.hcb[k];
        h.len = ro.len; h.npasses = ro.npasses; h.nbps = ro.nbps; h.zbp = sb.Mb - ro.nbps; h.lblock = 3; h.included = false;
        n_pass += ro.npasses; n_sym += ro.nsym;
        for (int l = 0; l < layers; l++) {
          int n;
          if (want_dist) n = l < 16 ? S.lp[(size_t)k * 16 + l] : ro.npasses;   // k_rate_alloc's choice (the PCRD loop, on the device)
          else n = l == layers - 1 ? ro.npasses : 0;                            // no slope anywhere: every pass in the last layer
          if (l > 0 && n < h.layer_passes[l - 1]) n = h.layer_passes[l - 1];
          h.layer_passes[l] = n;
        }
        for (int l = layers; l < 16; l++) h.layer_passes[l] = 0;   // (the records outlive the call: nothing stale beyond this plane's layers)
        h.first_layer = layers;
        for (int l = layers - 1; l >= 0; l--) if (h.layer_passes[l] > 0) h.first_layer = l;
        const int last = h.layer_passes[layers - 1];               // (non-decreasing over the layers)
        h.take = !want_dist ? ro.len : (last > 0 ? std::min(S.plen[(size_t)k * T1_MAXPASS + last - 1], ro.len) : 0);
      }
    }
    n_pass_out += n_pass; n_sym_out += n_sym;
  };
  // Which bytes cross the bus.  Every pass of every block (lossless, or a last layer without a slope): the copy starts at once and the
  // rate allocation runs under it.  Layers that leave passes out (`want_dist`): the allocation first -- its inputs, the per-pass
  // lengths and distortions, are already here -- and only the bytes up to each block's last included pass are gathered and copied
  // (round 4: at 4K with the reference's slope a tenth of the 185 MB; the copy of all of them was 4 of one sequence's 27 ms).
  std::vector<uint64_t> n_pass(nthreads, 0), n_sym(nthreads, 0);
  const auto tg0 = std::chrono::steady_clock::now();
  if (want_dist)
    S.pool->run([&](int tid) { for (int p = tid; p < np; p += nthreads) rate_alloc(p, n_pass[tid], n_sym[tid]); });
  const auto tg1 = std::chrono::steady_clock::now();
  size_t gtotal = 0;
  for (int k = 0; k < ncb; k++) {
    const CbOut& ro = S.res[k];
    if (ro.overflow) return c->fail(QSVC_ERR_NOMEM, "code-block segment exceeds %d bytes", J2K_OUTCAP);
    const int take = want_dist ? S.hcb[k].take : ro.len;
    goffs[k] = gtotal; glens[k] = take; gtotal += (size_t)take;
    S.hcb[k].data_off = goffs[k];
  }
  if ((r = S.gath_d.ensure(c, gtotal ? gtotal : 16))) return r;
  if (S.gath_h_cap < gtotal) {
    if (S.gath_h) (void)hipHostFree(S.gath_h);
    S.gath_h = nullptr; S.gath_h_cap = 0;
    size_t want = gtotal + gtotal / 4 + (1 << 20);
    if (hipHostMalloc((void**)&S.gath_h, want, hipHostMallocDefault) != hipSuccess) return c->fail(QSVC_ERR_NOMEM, "hipHostMalloc(%zu)", want);
    S.gath_h_cap = want;
  }
  if (!src[0].s16) {
    c->host_time("host_alloc_fill", std::chrono::duration<double, std::milli>(tg1 - tg0).count());
    c->host_time("host_gather_plan", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tg1).count());
  }
  pf.begin(c, src[0].s16 ? "mv_gather" : "j2k_gather");
  HIPCHK(c, hipMemcpyAsync(S.offs_d.p, goffs.data(), (size_t)ncb * 8, hipMemcpyHostToDevice, c->st));
  HIPCHK(c, hipMemcpyAsync(S.lens_d.p, glens.data(), (size_t)ncb * 4, hipMemcpyHostToDevice, c->st));
  launch_gather(c->st, (const uint8_t*)S.outbuf.p, J2K_OUTCAP, (const unsigned long long*)S.offs_d.p, (const int*)S.lens_d.p, ncb,
                (uint8_t*)S.gath_d.p);
  pf.end(c);
  // ... in up to four pieces of whole planes, an event after each: a plane's stream is written as soon as ITS bytes are there
  std::vector<int> piece_of(np, 0);
  {
    const int want = np < 4 ? np : 4;
    int p0 = 0;
    for (int g = 0; g < want && p0 < np; g++) {
      int p1 = p0 + 1;
      const size_t target = gtotal / want * (g + 1);
      while (p1 < np && (g == want - 1 || (size_t)goffs[first_cb[p1]] < target)) p1++;
      if (g == want - 1) p1 = np;
      const size_t b0 = (size_t)goffs[first_cb[p0]], b1 = p1 < np ? (size_t)goffs[first_cb[p1]] : gtotal;
      if (!S.copied[g]) HIPCHK(c, hipEventCreateWithFlags(&S.copied[g], hipEventDisableTiming));
      if (b1 > b0) HIPCHK(c, hipMemcpyAsync(S.gath_h + b0, (const uint8_t*)S.gath_d.p + b0, b1 - b0, hipMemcpyDeviceToHost, c->st));
      HIPCHK(c, hipEventRecord(S.copied[g], c->st));
      for (int p = p0; p < p1; p++) piece_of[p] = g;
      p0 = p1;
    }
  }
  // host: Tier-2, one code-stream per plane; planes are independent.  First every stream's length (packet headers are built, no
  // byte is copied), which fixes the streams' positions in `out`; then, once the code bytes have arrived, the streams themselves.
  {
    std::vector<long> rc(np, 0);
    auto stream_of = [&](int p, uint8_t* dst, size_t dcap) {
      return write_codestream(plans[plan_of[p]], *src[p].prm, src[p].prm->layers < 1 ? 1 : src[p].prm->layers, cbw, cbh,
                              S.hcb.data() + first_cb[p], want_dist ? S.plen.data() + (size_t)first_cb[p] * T1_MAXPASS : nullptr, S.gath_h, dst, dcap);
    };
    const auto t2a_begin = std::chrono::steady_clock::now();
    S.pool->run([&](int tid) {
      for (int p = tid; p < np; p += nthreads) { if (!want_dist) rate_alloc(p, n_pass[tid], n_sym[tid]); rc[p] = stream_of(p, nullptr, 0); }
    });
    for (int k = 0; k < nthreads; k++) { S.st_pass += n_pass[k]; S.st_sym += n_sym[k]; }
    size_t pos = 0;
    for (int p = 0; p < np; p++) { offs[p] = pos; pos += (size_t)rc[p]; }
    offs[np] = pos;
    double t2_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2a_begin).count();
    if (pos > cap) {
      (void)ctx_sync(c);
      return c->fail(QSVC_ERR_NOMEM, "output buffer too small (%zu bytes needed, %zu given)", pos, cap);
    }
    const auto t2b_begin = std::chrono::steady_clock::now();
    std::vector<long> rc2(np, 0);
    std::vector<int> waited(nthreads, -1);
    S.pool->run([&](int tid) {
      (void)hipSetDevice(c->dev);
      for (int p = tid; p < np; p += nthreads) {
        if (piece_of[p] > waited[tid]) { (void)(c->tune.sync_block ? wait_event_sleeping(S.copied[piece_of[p]]) : hipEventSynchronize(S.copied[piece_of[p]])); waited[tid] = piece_of[p]; }  // pieces complete in order
        rc2[p] = stream_of(p, out + offs[p], (size_t)rc[p]);
      }
    });
    { HostCpu hc2(c, "host_cpu_wait_copy"); HIPCHK(c, ctx_sync(c)); }
    HIPCHK(c, hipGetLastError());
    for (int p = 0; p < np; p++) if (rc2[p] != rc[p]) return c->fail(QSVC_ERR_HIP, "internal: Tier-2 length of plane %d changed between the passes", p);
    t2_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2b_begin).count();
    c->host_time(src[0].s16 ? "host_mv_tier2" : "host_tier2", t2_ms);
  }
  return QSVC_OK;
}


// Bounds the Tier-1 working set.  The decision slots are sized for the worst case (22.5 KB per code-block and bit-plane, up to
// 15 planes): a long or large sequence is coded in consecutive groups of planes whose slots fit QSVC_T1_BUDGET_GB (default 48: the 33 pictures of a 4K GOP window stay one group -- every further
// group costs a whole MQ chain of its own and the host's Tier-2 between the groups -- and four such contexts fit 288 GB in the worst case).
// The code-streams do not depend on the grouping (planes are independent).
static int j2k_encode_planes(qsvc_ctx* c, const std::vector<PlaneSrc>& src, uint8_t* out, size_t cap, size_t* offs) {
  static const double budget_gb = [] { const char* e = getenv("QSVC_T1_BUDGET_GB"); double v = e ? atof(e) : 48.0; return v > 0.001 ? v : 0.001; }();
  const size_t max_blocks = (size_t)(budget_gb * 1e9 / (15.0 * T1_SEG_BYTES));
  if (!c->j2k) c->j2k = new J2kState();
  c->j2k->st_cb = 0; c->j2k->st_pass = 0; c->j2k->st_sym = 0;
  const qsvc_j2k_params* prm = src[0].prm;
  const int cbw = prm->cblk_w ? prm->cblk_w : 64, cbh = prm->cblk_h ? prm->cblk_h : 64;
  auto est = [&](const PlaneSrc& p) {  // blocks of all subbands, a little over W*H / (cbw*cbh)
    return ((size_t)(p.W + cbw - 1) / cbw + 2) * ((size_t)(p.H + cbh - 1) / cbh + 2) + 4 * (size_t)(prm->levels + 1);
  };
  size_t total = 0;
  for (auto& p : src) total += est(p);
  if (total <= max_blocks) return j2k_encode_chunk(c, src, out, cap, offs);
  size_t pos = 0, first = 0;
  std::vector<size_t> sub;
  while (first < src.size()) {
    size_t n = 0, blocks = 0;
    while (first + n < src.size() && (n == 0 || blocks + est(src[first + n]) <= max_blocks)) { blocks += est(src[first + n]); n++; }
    std::vector<PlaneSrc> part(src.begin() + first, src.begin() + first + n);
    sub.assign(n + 1, 0);
    int r = j2k_encode_chunk(c, part, out + pos, cap - pos, sub.data());
    if (r) return r;
    for (size_t k = 0; k < n; k++) offs[first + k] = pos + sub[k];
    pos += sub[n];
    first += n;
  }
  offs[src.size()] = pos;
  return QSVC_OK;
}


extern "C" {

static int add_sequence(qsvc_ctx* c, int kind, int t, int first, int count, const qsvc_j2k_params* prm, std::vector<PlaneSrc>& src) {
  if (t < 0 || t > QSVC_MAX_T) return c->fail(QSVC_ERR_INVALID, "temporal subband %d out of range", t);
  // (pending I/B decisions of the temporal levels are taken by j2k_encode_planes, right before its first kernel: counts and
  //  buffer addresses do not depend on them, and the host builds its tables while the MCTF kernels are still running)
  Level& lv = c->lv[t];
  DevBuf* b; int n;
  switch (kind) {
    case QSVC_LOW: b = &lv.low; n = lv.low_n; break;
    case QSVC_HIGH: b = &lv.high; n = lv.high_n; break;
    case QSVC_PRED: b = &lv.pred; n = lv.pred_n; break;
    default: return c->fail(QSVC_ERR_INVALID, "unknown sequence kind %d", kind);
  }
  if (first < 0 || count < 0 || first + count > n) return c->fail(QSVC_ERR_STATE, "sequence kind %d t=%d holds %d pictures", kind, t, n);
  int cw = c->W / 2, ch = c->H / 2;
  for (int i = 0; i < count; i++) {  // order (picture, component), texture_compress_fb_j2k.py:255-257
    const uint8_t* f = (const uint8_t*)b->p + (size_t)(first + i) * c->F;
    src.push_back({f, c->W, c->W, c->H, prm, 0});
    src.push_back({f + (size_t)c->W * c->H, cw, cw, ch, prm, 0});
    src.push_back({f + (size_t)c->W * c->H + (size_t)cw * ch, cw, cw, ch, prm, 0});
  }
  return QSVC_OK;
}

int qsvc_j2k_encode_frames(qsvc_ctx* c, int kind, int t, int first, int count, const qsvc_j2k_params* prm, uint8_t* out,
                           size_t cap, size_t* offs) {
  return qsvc_j2k_encode_multi(c, 1, &kind, &t, &first, &count, prm, out, cap, offs);
}

int qsvc_j2k_encode_multi(qsvc_ctx* c, int nseq, const int* kinds, const int* ts, const int* firsts, const int* counts,
                          const qsvc_j2k_params* prms, uint8_t* out, size_t cap, size_t* offs) {
  if (!c || !kinds || !ts || !firsts || !counts || !prms || !out || !offs || nseq < 1) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  std::vector<PlaneSrc> src;
  for (int i = 0; i < nseq; i++) {
    int r = add_sequence(c, kinds[i], ts[i], firsts[i], counts[i], &prms[i], src);
    if (r) return r;
  }
  if (src.empty()) { offs[0] = 0; return QSVC_OK; }
  return j2k_encode_planes(c, src, out, cap, offs);
}

// `mctf compress` behind the analysis in ONE batch: the temporal subbands' pictures (as qsvc_j2k_encode_multi) and the motion fields'
// component planes (as qsvc_motion_encode) share the Tier-1 launches, the gather and the copy, so the (short) serial chain of
// the motion planes' code-blocks runs beside the textures' instead of in front of them.
int qsvc_compress_code(qsvc_ctx* c, int nseq, const int* kinds, const int* ts, const int* firsts, const int* counts, const qsvc_j2k_params* prms,
                       int nlevels, const int* mts, const int* mfields, int block_size, int16_t* residues, uint8_t* out, size_t cap, size_t* offs) {
  if (!c || !kinds || !ts || !firsts || !counts || !prms || !out || !offs || nseq < 1) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  if (!c->F) return c->fail(QSVC_ERR_STATE, "qsvc_seq_config not called");
  HostCpu hc(c, "host_cpu_code_call");
  std::vector<PlaneSrc> src;
  for (int i = 0; i < nseq; i++) {
    int r = add_sequence(c, kinds[i], ts[i], firsts[i], counts[i], &prms[i], src);
    if (r) return r;
  }
  std::vector<int16_t> planes;
  qsvc_j2k_params mprm;
  qsvc_motion_prm(&mprm);
  if (!src.empty()) mprm.cblk_w = prms[0].cblk_w, mprm.cblk_h = prms[0].cblk_h;  // one code-block size per batch
  if (nlevels > 0 && !residues) {
    // the motion planes are made ON THE DEVICE from the resident fields (k_motion_planes), queued right after the pending I/B
    // decisions are taken ('I' pictures zero their fields): no host round trip stands between the analysis and the batch
    if (!mts || !mfields || nlevels > MV_MAXLEV || block_size < 1) return c->fail(QSVC_ERR_INVALID, "bad motion level list");
    const int BY = c->H / block_size, BX = c->W / block_size;
    const size_t fld = (size_t)BY * BX, per = 4 * fld;
    MvLevels ml;
    memset(&ml, 0, sizeof ml);
    ml.nlevels = nlevels; ml.fld = (int)fld;
    size_t total = 0;
    for (int k = 0; k < nlevels; k++) {
      const int t = mts[k];
      if (t < 1 || t > QSVC_MAX_T || mfields[k] < 0 || (k && mts[k] <= mts[k - 1])) return c->fail(QSVC_ERR_INVALID, "bad level list");
      const Level& lv = c->lv[t];
      if (lv.mv_filt_n < mfields[k] || (mfields[k] && lv.mv_bs != block_size))
        return c->fail(QSVC_ERR_STATE, "motion_filtered_%d holds %d fields at block size %d (asked for %d at %d)", t, lv.mv_filt_n, lv.mv_bs, mfields[k], block_size);
      ml.filt[k] = (const int16_t*)lv.mv_filt.p; ml.fields[k] = mfields[k]; ml.first[k] = (int)total;
      total += (size_t)mfields[k];
    }
    ml.first[nlevels] = (int)total;
    if (total) {
      if (!c->j2k) c->j2k = new J2kState();
      int r = c->j2k->mvup_d.ensure(c, total * per * sizeof(int16_t));
      if (r) return r;
      int16_t* dst = (int16_t*)c->j2k->mvup_d.p;
      hipStream_t st = c->st;
      c->j2k->after_resolve = [ml, dst, total, per, st]() { launch_motion_planes(st, ml, dst, total * per); return (int)QSVC_OK; };
      for (size_t p = 0; p < total * 4; p++) src.push_back({(const uint8_t*)dst + p * fld * 2, BX, BX, BY, &mprm, 1});
    }
  } else if (nlevels > 0) {  // the caller wants the residue files' contents: the host half of the chain (qsvc_motion_planes)
    int BY = 0, BX = 0;
    int r = qsvc_motion_planes(c, nlevels, mts, mfields, block_size, nullptr, residues, planes, &BY, &BX);
    if (r) return r;
    if (!planes.empty()) {
      if (!c->j2k) c->j2k = new J2kState();
      const size_t bytes = planes.size() * sizeof(int16_t), N = (size_t)BY * BX;
      if ((r = c->j2k->mvup_d.ensure(c, bytes))) return r;
      HIPCHK(c, hipMemcpyAsync(c->j2k->mvup_d.p, planes.data(), bytes, hipMemcpyHostToDevice, c->st));
      HIPCHK(c, ctx_sync(c));  // (`planes` is pageable and local)
      for (size_t p = 0; p < planes.size() / N; p++) src.push_back({(const uint8_t*)c->j2k->mvup_d.p + p * N * 2, BX, BX, BY, &mprm, 1});
    }
  }
  if (src.empty()) { offs[0] = 0; return QSVC_OK; }
  const int rr = j2k_encode_planes(c, src, out, cap, offs);
  if (c->j2k) c->j2k->after_resolve = nullptr;   // (a refused batch must not leave its hook behind)
  return rr;
}

int qsvc_j2k_encode_plane(qsvc_ctx* c, const uint8_t* plane, int H, int W, const qsvc_j2k_params* prm, uint8_t* out, size_t cap,
                          size_t* len) {
  if (!c || !plane || !prm || !out || !len || W < 1 || H < 1 || W > 16384 || H > 16384) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  if (!c->j2k) c->j2k = new J2kState();
  int r = c->j2k->up_d.ensure(c, (size_t)W * H);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(c->j2k->up_d.p, plane, (size_t)W * H, hipMemcpyHostToDevice, c->st));
  std::vector<PlaneSrc> src{{(const uint8_t*)c->j2k->up_d.p, W, W, H, prm, 0}};
  size_t offs[2];
  r = j2k_encode_planes(c, src, out, cap, offs);
  if (r) return r;
  *len = offs[1];
  return QSVC_OK;
}

int qsvc_j2k_encode_planes_s16(qsvc_ctx* c, const int16_t* planes, int nplanes, int H, int W, const qsvc_j2k_params* prm,
                               uint8_t* out, size_t cap, size_t* offs) {
  if (!c || !planes || !prm || !out || !offs || nplanes < 1 || W < 1 || H < 1 || W > 16384 || H > 16384) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  // the motion planes have their own encoder state: a texture batch before and after finds its tables, job lists and buffers as it left them
  struct Swap {
    qsvc_ctx* c;
    explicit Swap(qsvc_ctx* c_) : c(c_) { std::swap(c->j2k, c->j2k_mv); }
    ~Swap() { std::swap(c->j2k, c->j2k_mv); }
  } swap_state(c);
  if (!c->j2k) c->j2k = new J2kState();
  size_t bytes = (size_t)nplanes * W * H * 2;
  int r = c->j2k->up_d.ensure(c, bytes);
  if (r) return r;
  HIPCHK(c, hipMemcpyAsync(c->j2k->up_d.p, planes, bytes, hipMemcpyHostToDevice, c->st));
  std::vector<PlaneSrc> src;
  for (int p = 0; p < nplanes; p++) src.push_back({(const uint8_t*)c->j2k->up_d.p + (size_t)p * W * H * 2, W, W, H, prm, 1});
  return j2k_encode_planes(c, src, out, cap, offs);
}

int qsvc_j2k_dwt(qsvc_ctx* c, const uint8_t* plane, int H, int W, int reversible, int levels, void* out) {
  if (!c || !plane || !out || W < 1 || H < 1 || W > 16384 || H > 16384 || levels < 0 || levels > J2K_MAXLEV) return QSVC_ERR_INVALID;
  (void)hipSetDevice(c->dev);
  if (!c->j2k) c->j2k = new J2kState();
  J2kState& S = *c->j2k;
  size_t N = (size_t)W * H;
  int r;
  if ((r = S.up_d.ensure(c, N))) return r;
  if ((r = S.coefA.ensure(c, N * 4))) return r;
  if ((r = S.planes_d.ensure(c, sizeof(PlaneDesc)))) return r;
  if ((r = S.err_d.ensure(c, 16))) return r;
  int L = levels;
  while (L > 0 && ((W >> (L - 1)) < 1 || (H >> (L - 1)) < 1)) L--;
  const int w1 = (W + 1) >> 1, h1 = (H + 1) >> 1, w2 = (w1 + 1) >> 1, h2 = (h1 + 1) >> 1;
  if ((r = S.llA.ensure(c, ((size_t)((w1 + 3) & ~3) * h1 + 8) * 4 + 64))) return r;
  if ((r = S.llB.ensure(c, ((size_t)((w2 + 3) & ~3) * h2 + 8) * 4 + 64))) return r;
  PlaneDesc pd{0, {0, 0}, (const uint8_t*)S.up_d.p, W, W, H, 0};
  HIPCHK(c, hipMemcpyAsync(S.up_d.p, plane, N, hipMemcpyHostToDevice, c->st));
  HIPCHK(c, hipMemcpyAsync(S.planes_d.p, &pd, sizeof pd, hipMemcpyHostToDevice, c->st));
  S.up_planes_at = nullptr;  // the encoder's plane table no longer is what planes_d holds: its next call uploads again
  HIPCHK(c, hipMemsetAsync(S.err_d.p, 0, 16, c->st));
  // the same tile kernel as the encoder's, with the subbands written to a Mallat-layout plane instead of the Tier-1 layout
  if (L == 0) launch_level_shift(c->st, (const PlaneDesc*)S.planes_d.p, 1, W, H, S.coefA.p, reversible != 0);
  else launch_dwt_tiles(c->st, (const PlaneDesc*)S.planes_d.p, 1, W, H, W, H, nullptr, L, reversible != 0, S.llA.p, S.llB.p, nullptr, nullptr,
                        S.coefA.p, 6, 6, (int*)S.err_d.p, nullptr, nullptr);
  HIPCHK(c, hipMemcpyAsync(out, S.coefA.p, N * 4, hipMemcpyDeviceToHost, c->st));
  HIPCHK(c, ctx_sync(c));
  HIPCHK(c, hipGetLastError());
  return QSVC_OK;
}

int qsvc_j2k_stats(qsvc_ctx* c, uint64_t* codeblocks, uint64_t* passes, uint64_t* symbols) {
  if (!c) return QSVC_ERR_INVALID;
  if (!c->j2k) return c->fail(QSVC_ERR_STATE, "no JPEG 2000 call yet");
  if (codeblocks) *codeblocks = c->j2k->st_cb;
  if (passes) *passes = c->j2k->st_pass;
  if (symbols) *symbols = c->j2k->st_sym;
  return QSVC_OK;
}

}  // extern "C"
