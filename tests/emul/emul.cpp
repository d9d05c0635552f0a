// tests/emul/emul.cpp -- TEST INFRASTRUCTURE: host loop harness over the kernel thread bodies.
//
// Compiles qsvc_amd/csrc/mctf_threads.h (the exact per-thread code the HIP kernels run) with g++
// and drives it with plain nested loops in place of the HIP grid, so kernel indexing and arithmetic
// can be checked against the oracle in a container that has no GPU.  It is NOT a product fallback:
// nothing in qsvc_amd/ loads this library, and the GPU parity tests do not use it.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../qsvc_amd/csrc/mctf_threads.h"

using namespace qsvc;

static LevelGeom make_geom(int P, int W, int H, int bs, int S) {
  LevelGeom g;
  g.W = W; g.H = H; g.bs = bs; g.BY = H / bs; g.BX = W / bs; g.S = S; g.P = P; g.nB = P / 2;
  g.F = (size_t)W * H + 2 * (size_t)(W / 2) * (H / 2);
  g.ov = 0; g.bord = 0; g.sub = 0;
  return g;
}
static int ilog2i(int s) { int l = 0; while ((1 << (l + 1)) <= s) l++; return l; }

extern "C" {

int emul_dwt53t(int16_t* plane, int stride, int H, int W, int levels, int inverse) {
  std::vector<int16_t> tmp((size_t)H * W);
  if (!inverse) {
    for (int lv = 0; lv < levels; lv++) {
      int ny = H >> lv, nx = W >> lv;
      auto P = [&](int y, int x) { return (int)plane[(size_t)y * stride + x]; };
      for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) tmp[(size_t)y * W + x] = (int16_t)analyze2d_coef(P, y, x, ny, nx);
      for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) plane[(size_t)y * stride + x] = tmp[(size_t)y * W + x];
    }
  } else {
    for (int lv = levels - 1; lv >= 0; lv--) {
      int ny = H >> lv, nx = W >> lv;
      auto P = [&](int y, int x) { return (int)plane[(size_t)y * stride + x]; };
      for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) tmp[(size_t)y * W + x] = (int16_t)synthesize2d_sample(P, y, x, ny, nx);
      for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) plane[(size_t)y * stride + x] = tmp[(size_t)y * W + x];
    }
  }
  return 0;
}

// one level of the in-place transform over the top-left ny x nx of a plane (what mctf_replay.hip's k_rp_dwt + k_rp_copy do)
static void rp_dwt(int16_t* plane, int W, int ny, int nx, int inverse, std::vector<int16_t>& tmp) {
  if (ny < 2 && nx < 2) return;
  auto P = [&](int y, int x) { return (int)plane[(size_t)y * W + x]; };
  for (int y = 0; y < ny; y++)
    for (int x = 0; x < nx; x++) tmp[(size_t)y * W + x] = (int16_t)(inverse ? synthesize2d_sample(P, y, x, ny, nx) : analyze2d_coef(P, y, x, ny, nx));
  for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) plane[(size_t)y * W + x] = tmp[(size_t)y * W + x];
}
static void me_level_search(const MeArgs& a, int bs, int16_t* mv_out, int l, size_t fld) {
  const LevelGeom& g = a.g;
  if (l == 0) memset(mv_out, 0, sizeof(int16_t) * g.nB * 4 * fld);
  for (int i = 0; i < g.nB; i++)
    for (int by = 0; by < a.nby; by++)
      for (int bx = 0; bx < a.nbx; bx++) {
        int pvx, pvy, nvx, nvy, sad[18], tot[18] = {0};
        me_block_init_mv(a, i, by, bx, pvx, pvy, nvx, nvy);
        for (int lane = 0; lane < 64; lane++) {  // 64 lanes, summed like the wave reduction
          if (!a.snap) { MePlane<uint8_t> r0, r1, pr; me_block_setup(a, i, r0, r1, pr); me_block_sads(r0, r1, pr, bs, by, bx, pvy, pvx, nvy, nvx, lane, 64, sad); }
          else { MePlane<int16_t> r0, r1, pr; me_block_setup(a, i, r0, r1, pr); me_block_sads(r0, r1, pr, bs, by, bx, pvy, pvx, nvy, nvx, lane, 64, sad); }
          for (int k = 0; k < 18; k++) tot[k] += sad[k];
        }
        me_block_store(a, i, by, bx, tot, pvx, pvy, nvx, nvy);
      }
}
// the host logic of qsvc_api.hip's me_replay (Q5 geometries) over the same thread code
static int emul_me_replay(const uint8_t* low, const LevelGeom& g, int L, int16_t* mv_out) {
  const int W = g.W, H = g.H, nB = g.nB;
  const size_t N = (size_t)H * W, fld = (size_t)g.BY * g.BX;
  std::vector<int16_t> A((size_t)3 * nB * N, 0), tmp(N);
  auto load = [&](int slot, int pic) { for (size_t k = 0; k < N; k++) A[(size_t)slot * N + k] = low[(size_t)pic * g.F + k]; };
  for (int j = 1; j < nB; j++) {
    load(3 * j, 2 * j);
    for (int k = 0; k < L; k++) rp_dwt(&A[(size_t)3 * j * N], W, imax(H >> k, 1), imax(W >> k, 1), 0, tmp);
    for (int l = L - 1; l >= 0; l--) rp_dwt(&A[(size_t)3 * j * N], W, desp(H, l), desp(W, l), 1, tmp);
  }
  load(0, 0);
  for (int i = 0; i < nB; i++) { load(3 * i + 1, 2 * i + 1); load(3 * i + 2, 2 * i + 2); }
  for (int p = 0; p < 3 * nB; p++)
    for (int k = 0; k < L; k++) rp_dwt(&A[(size_t)p * N], W, imax(H >> k, 1), imax(W >> k, 1), 0, tmp);
  std::vector<int16_t> mvb[2];
  mvb[0].assign((size_t)nB * 4 * fld, 0); mvb[1].assign((size_t)nB * 4 * fld, 0);
  const int16_t* prev = nullptr;
  for (int l = L; l >= 0; l--) {
    if (l < L) for (int p = 0; p < 3 * nB; p++) rp_dwt(&A[(size_t)p * N], W, desp(H, l), desp(W, l), 1, tmp);
    MeArgs a;
    a.g = g; a.frames = low; a.snap = A.data(); a.snap_pic_stride = N; a.pic_mul = 3;
    a.mv_prev = prev; a.mv_out = l == 0 ? mv_out : mvb[l & 1].data(); a.l = l;
    a.nby = desp(g.BY, l); a.nbx = desp(g.BX, l); a.ring = 0;
    me_level_search(a, g.bs, mv_out, l, fld);
    prev = a.mv_out;
  }
  return 0;
}

int emul_motion_estimate(const uint8_t* low, int P, int W, int H, int bs, int S, int16_t* mv_out) {
  LevelGeom g = make_geom(P, W, H, bs, S);
  int L = ilog2i(S) - 1;
  if (L < 0) L = 0;
  if ((1 << ilog2i(S)) != S) return -2;
  if ((H % (1 << L)) || (W % (1 << L))) return emul_me_replay(low, g, L, mv_out);
  size_t N = (size_t)H * W, fld = (size_t)g.BY * g.BX;
  std::vector<std::vector<int16_t>> snap(L + 1);
  for (int l = 1; l <= L; l++) {
    snap[l].assign((size_t)P * N, 0);
    int R = imin(H, imax(desp(g.BY, l) * bs + S, H >> l)), C = imin(W, imax(desp(g.BX, l) * bs + S, W >> l));
    for (int pic = 0; pic < P; pic++)
      for (int y = 0; y < R; y++)
        for (int x = 0; x < C; x++) {
          if (l == 1) pyr_thread<uint8_t>(low, g.F, W, snap[l].data(), N, H, W, l, R, C, x, y, pic);
          else pyr_thread<int16_t>(snap[l - 1].data(), N, W, snap[l].data(), N, H, W, l, R, C, x, y, pic);
        }
  }
  std::vector<int16_t> mvb[2];
  mvb[0].assign((size_t)g.nB * 4 * fld, 0); mvb[1].assign((size_t)g.nB * 4 * fld, 0);
  const int16_t* prev = nullptr;
  for (int l = L; l >= 0; l--) {
    MeArgs a;
    a.g = g; a.frames = low; a.snap = l ? snap[l].data() : nullptr; a.snap_pic_stride = N;
    a.mv_prev = prev; a.mv_out = l == 0 ? mv_out : mvb[l & 1].data(); a.l = l;
    a.nby = desp(g.BY, l); a.nbx = desp(g.BX, l); a.ring = 0;
    me_level_search(a, bs, mv_out, l, fld);
    prev = a.mv_out;
  }
  return 0;
}

int emul_predict(const uint8_t* low, int P, int W, int H, int bs, int S, const int16_t* mv, uint8_t* high, uint8_t* pred,
                 unsigned* hist /*[nB][3][256]*/) {
  LevelGeom g = make_geom(P, W, H, bs, S);
  if ((W & 1) || (H & 1)) return -2;
  std::vector<int16_t> margin;
  MarginGeom mg = margin_geom(g);
  if (H % bs || W % bs) margin.assign((size_t)(g.nB ? g.nB : 1) * 2 * mg.count(), 0);  // Q7
  PredArgs a = {g, low, mv, high, pred, hist, margin.empty() ? nullptr : margin.data(), nullptr, nullptr};
  memset(hist, 0, sizeof(unsigned) * g.nB * 3 * 256);
  size_t fld = (size_t)g.BY * g.BX;
  for (int i = 0; i < g.nB; i++) {
    if (a.margin)  // one launch per picture, in order: picture i reads the state picture i-1 left
      for (int c = 1; c < 3; c++)
        for (size_t k = 0; k < mg.count(); k++) pred_margin_thread(a, k, i, c);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        int ob, rb;
        pred_luma_thread(a, x, y, i, ob, rb);
        hist[((size_t)i * 3 + 0) * 256 + ob]++; hist[((size_t)i * 3 + 1) * 256 + rb]++;
      }
    for (int c = 1; c < 3; c++)
      for (int y = 0; y < H / 2; y++)
        for (int x = 0; x < W / 2; x++) pred_chroma_thread(a, x, y, i, c);
    for (size_t k = 0; k < 4 * fld; k++) hist[((size_t)i * 3 + 2) * 256 + ((mv[(size_t)i * 4 * fld + k] + 128) & 255)]++;
  }
  return 0;
}

int emul_update(const uint8_t* low, int P, int W, int H, int bs, const uint8_t* high, const int16_t* mv, const char* types,
                float uf, int reach, const uint8_t* halo_l, int halo_l_type, const uint8_t* halo_r, int halo_r_type,
                uint8_t* low_out) {
  LevelGeom g = make_geom(P, W, H, bs, 0);
  if ((W & 1) || (H & 1)) return -2;
  UpdArgs a;
  a.g = g; a.frames = low; a.high = high; a.mv = mv; a.types = (const uint8_t*)types; a.low = low_out; a.uf = uf; a.reach = reach;
  a.src_mul = 2; a.dst_mul = 1; a.cup = nullptr;
  a.halo_high[0] = halo_l; a.halo_mv[0] = halo_l ? (const int16_t*)(halo_l + g.F) : nullptr; a.halo_type[0] = halo_l ? halo_l_type : 0;
  a.halo_high[1] = halo_r; a.halo_mv[1] = halo_r ? (const int16_t*)(halo_r + g.F) : nullptr; a.halo_type[1] = halo_r ? halo_r_type : 0;
  for (int j = 0; j <= g.nB; j++) {
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) upd_luma_thread(a, x, y, j);
    for (int c = 1; c < 3; c++)
      for (int y = 0; y < H / 2; y++) for (int x = 0; x < W / 2; x++) upd_chroma_thread(a, x, y, j, c);
  }
  return 0;
}

// inverse level: low_t [nB+1] -> frames [2nB+1] (even pictures by un_update, odd pictures by correlate, in place)
int emul_synthesize_step(const uint8_t* low_t, int P, int W, int H, int bs, int S, const uint8_t* high, const int16_t* mv,
                         const char* types, float uf, int reach, uint8_t* frames) {
  LevelGeom g = make_geom(P, W, H, bs, S);
  if ((W & 1) || (H & 1)) return -2;
  UpdArgs u;
  u.g = g; u.frames = low_t; u.high = high; u.mv = mv; u.types = (const uint8_t*)types; u.low = frames; u.uf = -uf; u.reach = reach;
  u.src_mul = 1; u.dst_mul = 2; u.cup = nullptr;
  for (int s = 0; s < 2; s++) { u.halo_high[s] = nullptr; u.halo_mv[s] = nullptr; u.halo_type[s] = 0; }
  for (int j = 0; j <= g.nB; j++) {
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) upd_luma_thread(u, x, y, j);
    for (int c = 1; c < 3; c++)
      for (int y = 0; y < H / 2; y++) for (int x = 0; x < W / 2; x++) upd_chroma_thread(u, x, y, j, c);
  }
  std::vector<int16_t> margin;
  MarginGeom mg = margin_geom(g);
  if (H % bs || W % bs) margin.assign((size_t)(g.nB ? g.nB : 1) * 2 * mg.count(), 0);
  PredArgs a = {g, frames, mv, (uint8_t*)high, nullptr, nullptr, margin.empty() ? nullptr : margin.data(), frames, (const uint8_t*)types};
  for (int i = 0; i < g.nB; i++) {
    if (a.margin)
      for (int c = 1; c < 3; c++)
        for (size_t k = 0; k < mg.count(); k++) pred_margin_thread(a, k, i, c);
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) { int ob, rb; pred_luma_thread(a, x, y, i, ob, rb); }
    for (int c = 1; c < 3; c++)
      for (int y = 0; y < H / 2; y++)
        for (int x = 0; x < W / 2; x++) pred_chroma_thread(a, x, y, i, c);
  }
  return 0;
}

int emul_border_luma(const uint8_t* luma, int H, int W, int b, int16_t* out /*(H+2b)x(W+2b)*/) {
  BorderGeom g = make_border(H, W, b);
  for (int y = -b; y < H + b; y++)
    for (int x = -b; x < W + b; x++) out[(size_t)(y + b) * (W + 2 * b) + (x + b)] = (int16_t)luma_b(luma, g, y, x);
  return 0;
}
}

// ---- EBCOT Tier-1: one block through the lane code with lane stride 1 -------------------------
#include "../../qsvc_amd/csrc/t1_core.h"
#include "t1_legacy.h"
#include "../../qsvc_amd/csrc/t1_cm2.h"
extern "C" int emul_t1_block(const int32_t* vals, int w, int h, int orient, uint8_t* out, int* npasses, int* nbps,
                             int* pass_len, long long* pass_dist) {
  static const uint32_t mqtab[47] = T1_MQ_TABLE_INIT;
  static uint8_t zclut[3 * 512];
  static bool init = false;
  if (!init) {
    for (int c = 0; c < 3; c++)
      for (int nb = 0; nb < 512; nb++) zclut[c * 512 + nb] = (uint8_t)t1_zc_from_nb(nb, c == 0 ? 0 : (c == 1 ? 1 : 3));
    init = true;
  }
  std::vector<uint32_t> flags(18 * T1_FW, 0);
  std::vector<uint64_t> mags(16 * 64, 0);
  int maxm = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = vals[y * w + x];
      int m = v < 0 ? -v : v;
      if (m > 0x7FFF) return -1;
      if (m > maxm) maxm = m;
      uint64_t sm = (uint64_t)(m | (v < 0 ? 0x8000 : 0));
      mags[(y >> 2) * 64 + x] |= sm << (16 * (y & 3));
    }
  int nb = 0;
  while (maxm >> nb) nb++;
  uint32_t cx[19];
  std::vector<uint32_t> buf32((size_t)w * h + 16);
  uint8_t* bufp = (uint8_t*)buf32.data();
  T1Lane t;
  t.cx = cx; t.mqtab = mqtab; t.zclut = zclut; t.out = bufp;
  t.cs = 1; t.cap = (int)buf32.size() * 4; t.w = w; t.h = h; t.orient = orient; t.nbps = nb;
  T1ArrayStore S{flags.data(), mags.data(), 1};
  T1Result r;
  int pl[T1_MAXPASS]; long long pd[T1_MAXPASS];
  r.pass_len = pl; r.pass_dist = pd;
  t1_encode_lane(t, S, nb - 1, w + 7 > 64 ? 64 : w + 7, h + 5 > 64 ? 64 : h + 5, true, r);  // uniform bounds beyond the block, as in a wave
  if (r.overflow) return -2;
  memcpy(out, bufp + 1, r.len);
  *npasses = r.npasses; *nbps = r.nbps;
  for (int i = 0; i < r.npasses; i++) { pass_len[i] = pl[i]; if (pass_dist) pass_dist[i] = pd[i]; }
  return r.len;
}

// ---- split form: per-plane context modelling from initial flags + MQ over the decision segments ----
extern "C" int emul_t1_block_split(const int32_t* vals, int w, int h, int orient, uint8_t* out, int* npasses, int* nbps,
                                   int* pass_len, long long* pass_dist) {
  static const uint32_t mqtab[47] = T1_MQ_TABLE_INIT;
  static uint8_t zclut[3 * 512];
  static bool init = false;
  if (!init) {
    for (int c = 0; c < 3; c++)
      for (int nb = 0; nb < 512; nb++) zclut[c * 512 + nb] = (uint8_t)t1_zc_from_nb(nb, c == 0 ? 0 : (c == 1 ? 1 : 3));
    init = true;
  }
  std::vector<uint64_t> mags(16 * 64, 0);
  int maxm = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = vals[y * w + x];
      int m = v < 0 ? -v : v;
      if (m > 0x7FFF) return -1;
      if (m > maxm) maxm = m;
      mags[(y >> 2) * 64 + x] |= (uint64_t)(m | (v < 0 ? 0x8000 : 0)) << (16 * (y & 3));
    }
  int nb = 0;
  while (maxm >> nb) nb++;
  int nstr = (h + 3) >> 2;
  std::vector<uint8_t> segs((size_t)(nb ? nb : 1) * T1_SEG_BYTES);
  std::vector<int> cnt((size_t)(nb ? nb : 1) * 3, 0);
  std::vector<long long> dist((size_t)(nb ? nb : 1) * 3, 0);
  auto M = [&](int s, int x) { return mags[s * 64 + x]; };
  for (int p = 0; p < nb; p++) {  // every plane independently, from its initial flags
    std::vector<uint32_t> flags(18 * T1_FW, 0);
    for (int s = 0; s < nstr; s++)
      for (int x = 0; x < w; x++) flags[(s + 1) * T1_FW + x + 1] = t1_initial_flag(M, s, x, p, nstr, w);
    T1ArrayStore S{flags.data(), mags.data(), 1};
    t1_cm_plane(S, zclut, segs.data() + (size_t)p * T1_SEG_BYTES, p, w, h, orient, nb, w + 7 > 64 ? 64 : w + 7, h + 5 > 64 ? 64 : h + 5, true, &cnt[p * 3], &dist[p * 3]);
  }
  uint32_t cx[20];
  std::vector<uint32_t> buf32(((size_t)w * h + 16 + 3) & ~(size_t)3);  // capacity: a multiple of 16 bytes
  uint8_t* bufp = (uint8_t*)buf32.data();
  static uint64_t mq2[94];
  for (int i = 0; i < 94; i++) mq2[i] = mq2_pair(mqtab, i);
  T1Lane t;
  t.cx = cx; t.mqtab = mqtab; t.mq2 = mq2; t.zclut = zclut; t.out = bufp;
  t.cs = 1; t.cap = (int)buf32.size() * 4; t.w = w; t.h = h; t.orient = orient; t.nbps = nb;
  T1Result r;
  int pl[T1_MAXPASS]; long long pd[T1_MAXPASS];
  r.pass_len = pl; r.pass_dist = pd;
  auto Seg = [&](int p) { return (const uint8_t*)segs.data() + (size_t)p * T1_SEG_BYTES; };
  auto Cnt = [&](int p, int pass) { return cnt[p * 3 + pass]; };
  auto Dist = [&](int p, int pass) { return dist[p * 3 + pass]; };
  t1_mq_block(t, Seg, Cnt, Dist, r);
  if (r.overflow) return -2;
  memcpy(out, bufp + 1, r.len);
  *npasses = r.npasses; *nbps = r.nbps;
  for (int i = 0; i < r.npasses; i++) { pass_len[i] = pl[i]; if (pass_dist) pass_dist[i] = pd[i]; }
  return r.len;
}

// ---- sample-parallel context modelling (t1_cm2.h) driven like the kernel: one "wave" per (block, plane); membership by 16
// stripe lanes, each two columns behind the stripe above; then 64 "threads" x 16 stripe columns count, scan and emit ----
extern "C" int emul_t1_block_cm2(const int32_t* vals, int w, int h, int orient, uint8_t* out, int* npasses, int* nbps,
                                 int* pass_len, long long* pass_dist) {
  static const uint32_t mqtab[47] = T1_MQ_TABLE_INIT;
  static uint8_t zclut[3 * 512];
  static uint32_t sp[64];
  static bool init = false;
  if (!init) {
    for (int c = 0; c < 3; c++)
      for (int nb = 0; nb < 512; nb++) zclut[c * 512 + nb] = (uint8_t)t1_zc_from_nb(nb, c == 0 ? 0 : (c == 1 ? 1 : 3));
    for (uint32_t v = 0; v < 64; v++) sp[v] = CM2_SPREAD_INIT(v);
    init = true;
  }
  std::vector<uint64_t> mags(16 * 64, 0);
  int maxm = 0;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int v = vals[y * w + x];
      int m = v < 0 ? -v : v;
      if (m > 0x7FFF) return -1;
      if (m > maxm) maxm = m;
      mags[(y >> 2) * 64 + x] |= (uint64_t)(m | (v < 0 ? 0x8000 : 0)) << (16 * (y & 3));
    }
  int nb = 0;
  while (maxm >> nb) nb++;
  const int nstr = (h + 3) >> 2;
  std::vector<uint8_t> segs((size_t)(nb ? nb : 1) * T1_SEG_BYTES);
  std::vector<int> cnt((size_t)(nb ? nb : 1) * 3, 0);
  std::vector<long long> dist((size_t)(nb ? nb : 1) * 3, 0);
  {  // the bit-planes of every column at once (cm2_bitplanes) against the samples
    for (int x = 0; x < 64; x++) {
      uint32_t m32[32]; uint64_t P[16];
      for (int s = 0; s < 16; s++) {
        const uint64_t m = (s < nstr && x < w) ? mags[s * 64 + x] : 0;
        m32[2 * s] = (uint32_t)m; m32[2 * s + 1] = (uint32_t)(m >> 32);
      }
      cm2_bitplanes(m32, P);
      for (int y = 0; y < 64; y++) {
        const uint32_t v = (uint32_t)(((y >> 2) < nstr && x < w ? mags[(y >> 2) * 64 + x] : 0) >> (16 * (y & 3))) & 0xFFFF;
        for (int q = 0; q < 16; q++) if (((P[q] >> y) & 1) != ((v >> q) & 1)) return -17;
      }
    }
  }
  for (int p = 0; p < nb; p++) {
    uint32_t cw[18][66];  // one-cell apron of zeros
    memset(cw, 0, sizeof cw);
    for (int s = 0; s < nstr; s++)
      for (int x = 0; x < w; x++) cw[s + 1][x + 1] = cm2_colword(mags[s * 64 + x], p, h - 4 * s > 4 ? 4 : h - 4 * s, true);
    {  // membership as the kernel computes it: row masks, fixpoint of cm2_row_update over the 64 "lanes"
      uint64_t Z[66] = {0}, Bt[66] = {0}, cand[66] = {0}, M[66] = {0}, seed[66] = {0};
      const uint64_t wmask = w == 64 ? ~0ull : ((1ull << w) - 1);
      for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
          uint32_t c = cw[(y >> 2) + 1][x + 1];
          Z[y + 1] |= (uint64_t)((c >> (CW_SIG0 + (y & 3))) & 1) << x;
          Bt[y + 1] |= (uint64_t)((c >> (CW_BIT + (y & 3))) & 1) << x;
        }
      for (int y = 0; y < 64; y++) { cand[y + 1] = (y < h ? wmask : 0) & ~Z[y + 1]; seed[y + 1] = cm2_row_seed(Z[y + 1], Z[y], Z[y + 2]); }
      for (bool changed = true; changed;) {
        uint64_t Mn[66] = {0};
        changed = false;
        for (int y = 0; y < 64; y++) {
          Mn[y + 1] = cm2_row_update(cand[y + 1], Bt[y + 1], seed[y + 1], y ? (M[y] & Bt[y]) : 0, y < 63 ? (M[y + 2] & Bt[y + 2]) : 0, y & 3);
          changed |= Mn[y + 1] != M[y + 1];
        }
        memcpy(M, Mn, sizeof M);
      }
      for (int s = 0; s < nstr; s++)
        for (int x = 0; x < w; x++) {
          uint32_t c = cw[s + 1][x + 1], M4 = 0;
          for (int k = 0; k < 4; k++) M4 |= (uint32_t)((M[4 * s + k + 1] >> x) & 1) << k;
          const uint32_t S1 = M4 & (c >> CW_BIT) & 0xF;
          const uint32_t S2 = (c >> CW_VAL) & ~(c >> CW_SIG0) & ~M4 & (c >> CW_BIT) & 0xF;
          // cross-check with the scan-order recurrence of cm2_member, evaluated in place in scan order
          uint32_t ref = cm2_member(&cw[s][x], &cw[s + 1][x], &cw[s + 2][x]);
          if (ref != ((M4 << CW_M) | (S1 << CW_S1) | (S2 << CW_S2))) return -5;
          cw[s + 1][x + 1] = c | ref;
        }
    }
    Cm2Ctx cx{zclut + 512 * (orient == 1 ? 1 : (orient == 3 ? 2 : 0)), sp, p, true};
    const bool active = true, first = p == nb - 1;
    (void)active;
    for (int pass = 0; pass < 3; pass++) {
      if (first && pass < 2) { cnt[p * 3 + pass] = 0; continue; }
      int counts[64], offs[65];
      for (int th = 0; th < 64; th++) {
        Cm2CountSink C{0};
        long long dd = 0;
        for (int i = th * 16; i < th * 16 + 16; i++) {
          int s = i >> 6, x = i & 63;
          if (s >= nstr || x >= w) continue;
          cm2_column(C, cx, pass, &cw[s][x], &cw[s + 1][x], &cw[s + 2][x], mags[s * 64 + x], dd);
        }
        counts[th] = C.n;
        int nf = 0;  // the mask-only count must agree with the emission
        for (int i = th * 16; i < th * 16 + 16; i++) {
          int s = i >> 6, x = i & 63;
          if (s >= nstr || x >= w) continue;
          Cm2Col col; int n3[3];
          cm2_gather(col, &cw[s][x], &cw[s + 1][x], &cw[s + 2][x]);
          cm2_counts(col, n3);
          nf += n3[pass];
        }
        if (nf != C.n) return -4;
      }
      offs[0] = 0;
      for (int th = 0; th < 64; th++) offs[th + 1] = offs[th] + counts[th];
      uint8_t* seg = segs.data() + (size_t)p * T1_SEG_BYTES + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
      long long dsum = 0;
      for (int th = 0; th < 64; th++) {
        Cm2ByteSink B{seg + offs[th], 0};
        for (int i = th * 16; i < th * 16 + 16; i++) {
          int s = i >> 6, x = i & 63;
          if (s >= nstr || x >= w) continue;
          cm2_column(B, cx, pass, &cw[s][x], &cw[s + 1][x], &cw[s + 2][x], mags[s * 64 + x], dsum);
        }
        if (B.n != counts[th]) return -3;
      }
      cnt[p * 3 + pass] = offs[64]; dist[p * 3 + pass] = dsum;
    }
    {  // the all-passes-at-once column form (cm2_emit3) must produce the same three byte strings
      static uint8_t signlut[256];
      for (int i = 0; i < 256; i++) signlut[i] = cm2_sign_entry(i);
      std::vector<uint8_t> e3[3];
      // ... appended through the word streams of the kernel (starting at byte offsets 1, 2, 3 of zeroed buffers)
      std::vector<uint32_t> wbuf[3];
      Cm2Stream st[3];
      long long d3[3] = {0, 0, 0};
      for (int q = 0; q < 3; q++) { wbuf[q].assign(T1_SEG_BYTES / 4 + 8, 0); cm2_stream_begin(st[q], (uint8_t*)wbuf[q].data(), q + 1); }
      for (int s = 0; s < nstr; s++)
        for (int x = 0; x < w; x++) {
          Cm2Col col; Cm2Out3 O;
          cm2_gather(col, &cw[s][x], &cw[s + 1][x], &cw[s + 2][x]);
          cm2_emit3(O, cx, signlut, col, mags[s * 64 + x], d3);
          auto app = [&](std::vector<uint8_t>& v, const Cm2Bytes& b) { for (int i = 0; i < b.n; i++) v.push_back((uint8_t)(b.a >> (8 * i))); };
          app(e3[0], O.p0); app(e3[1], O.p1); app(e3[2], O.hdr); app(e3[2], O.p2);
          cm2_stream_put8(st[0], O.p0); cm2_stream_put4(st[1], O.p1); cm2_stream_put4(st[2], O.hdr); cm2_stream_put8(st[2], O.p2);
        }
      for (int q = 0; q < 3; q++) cm2_stream_end(st[q]);
      {  // the counts of all 16 stripes of a column at once (cm2_counts_swar on row masks) and their packed form, against cm2_counts_w
        uint64_t V[66] = {0}, Z[66] = {0}, Bm[66] = {0}, Mm[66] = {0}, T[66] = {0}, T2[66] = {0};   // [x + 1]
        for (int s = 0; s < nstr; s++)
          for (int x = 0; x < w; x++) {
            const uint32_t c = cw[s + 1][x + 1];
            V[x + 1] |= (uint64_t)((c >> CW_VAL) & 0xF) << (4 * s); Z[x + 1] |= (uint64_t)((c >> CW_SIG0) & 0xF) << (4 * s);
            Bm[x + 1] |= (uint64_t)((c >> CW_BIT) & 0xF) << (4 * s); Mm[x + 1] |= (uint64_t)((c >> CW_M) & 0xF) << (4 * s);
            T[x + 1] |= (uint64_t)(((c >> CW_SIG0) | (c >> CW_S1)) & 0xF) << (4 * s);
            T2[x + 1] |= (uint64_t)(((c >> CW_SIG0) | (c >> CW_S1) | (c >> CW_S2)) & 0xF) << (4 * s);
          }
        for (int x = 0; x < 64; x++) {
          Cm2Counts cc; uint32_t pk[8];
          cm2_counts_swar(cc, V[x + 1], Z[x + 1], Bm[x + 1], Mm[x + 1], T[x], T2[x], T[x + 2], T2[x + 2]);
          cm2_pack_pk(cc, pk);
          {  // the column words rebuilt from the masks by the nibble transpose
            uint64_t F[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            static const int at[8] = {CW_SIG0, CW_BIT, CW_NEG, CW_MU, CW_VAL, CW_M, CW_S1, CW_S2};
            for (int s = 0; s < 16; s++)
              for (int q = 0; q < 8; q++) F[q] |= (uint64_t)((cw[s + 1][x + 1] >> at[q]) & 0xF) << (4 * s);
            for (int hh = 0; hh < 2; hh++) {
              uint32_t f[8], o8[8];
              for (int q = 0; q < 8; q++) f[q] = (uint32_t)(F[q] >> (32 * hh));
              cm2_colwords8(f, o8);
              for (int k = 0; k < 8; k++) if (o8[k] != cw[8 * hh + k + 1][x + 1]) return -15;
            }
          }
          for (int s = 0; s < 16; s++) {
            int n3[3]; bool plain = false;
            cm2_counts_w(&cw[s][x], &cw[s + 1][x], &cw[s + 2][x], n3, &plain);
            const int act = !plain && (n3[0] | n3[1] | n3[2]);
            const int want = n3[0] | n3[1] << 4 | n3[2] << 8 | act << 12;
            const int got = ((const uint16_t*)pk)[CM2_PK_AT(s)];
            if (got != want) return -13;
          }
        }
      }
      {  // the kernel's form: 64 "lanes" x 16 consecutive columns through the sliding window of spread columns, counts from raw words
        static uint8_t zcx[1024];
        for (int i = 0; i < 512; i++) zcx[i] = cx.zc[i];
        for (int i = 512; i < 1024; i++) zcx[i] = (uint8_t)(((i >> 4) & 1) ? 16 : ((i & 0x1EF) ? 15 : 14));
        std::vector<uint8_t> w3[3];
        int nsum[3] = {0, 0, 0};
        for (int th = 0; th < 64; th++) {
          const int ls = th >> 2, lx0 = (th & 3) * 16;
          if (ls >= nstr || lx0 >= w) continue;
          Cm2Win win; memset(&win, 0, sizeof win);
          cm2_win_push(win, sp, 0, cw[ls][lx0], cw[ls + 1][lx0], cw[ls + 2][lx0]);
          cm2_win_push(win, sp, 0, cw[ls][lx0 + 1], cw[ls + 1][lx0 + 1], cw[ls + 2][lx0 + 1]);
          for (int j = 0; j < 16; j++) {
            cm2_win_push(win, sp, cw[ls + 1][lx0 + j + 1], cw[ls][lx0 + j + 2], cw[ls + 1][lx0 + j + 2], cw[ls + 2][lx0 + j + 2]);
            if (lx0 + j >= w) continue;
            Cm2Out3 O;
            cm2_emit3w(O, zcx, signlut, win);
            {  // the four-rows-at-once form must give the same four strings, zeros behind them
              static uint64_t cmp8[256]; static bool have = false;
              if (!have) { for (int i = 0; i < 256; i++) cmp8[i] = cm2_cmp8_entry(i); have = true; }
              Cm2Out3 S;
              cm2_emit3s(S, zcx, signlut, cmp8, win);
              if (S.p0.a != O.p0.a || S.p0.n != O.p0.n || S.p1.a != O.p1.a || S.p1.n != O.p1.n || S.hdr.a != O.hdr.a || S.hdr.n != O.hdr.n ||
                  S.p2.a != O.p2.a || S.p2.n != O.p2.n) return -14;
            }
            auto app = [&](std::vector<uint8_t>& v, const Cm2Bytes& b) { for (int i = 0; i < b.n; i++) v.push_back((uint8_t)(b.a >> (8 * i))); if (b.n < 8 && (b.a >> (8 * b.n))) v.push_back(0xEE); };
            app(w3[0], O.p0); app(w3[1], O.p1); app(w3[2], O.hdr); app(w3[2], O.p2);
            int n3[3];
            cm2_counts_w(&cw[ls][lx0 + j], &cw[ls + 1][lx0 + j], &cw[ls + 2][lx0 + j], n3);
            if (n3[0] != O.p0.n || n3[1] != O.p1.n || n3[2] != O.hdr.n + O.p2.n) return -9;
            for (int q = 0; q < 3; q++) nsum[q] += n3[q];
          }
        }
        for (int pass = 0; pass < 3; pass++) if (w3[pass] != e3[pass]) return -10;
        {  // the lanes' ranges through the atomic-free sequential streams: all the loops first, then all the ends (the wave's lockstep)
          std::vector<uint32_t> qb[3];
          for (int q = 0; q < 3; q++) qb[q].assign(T1_SEG_BYTES / 4 + 8, 0);
          Cm2Seq sq[64][3];
          uint32_t trash[64];
          int off[3] = {0, 0, 0};
          for (int th = 0; th < 64; th++) {
            const int ls = th >> 2, lx0 = (th & 3) * 16;
            for (int q = 0; q < 3; q++) cm2_seq_begin(sq[th][q], (uint8_t*)qb[q].data(), off[q], &trash[th]);
            if (ls >= nstr || lx0 >= w) continue;
            for (int j = 0; j < 16 && lx0 + j < w; j++) {
              int n3[3];
              cm2_counts_w(&cw[ls][lx0 + j], &cw[ls + 1][lx0 + j], &cw[ls + 2][lx0 + j], n3);
              for (int q = 0; q < 3; q++) off[q] += n3[q];
            }
          }
          for (int th = 63; th >= 0; th--) {  // (any order of the lanes within the loop phase)
            const int ls = th >> 2, lx0 = (th & 3) * 16;
            if (ls >= nstr || lx0 >= w) continue;
            for (int j = 0; j < 16 && lx0 + j < w; j++) {
              Cm2Win win; Cm2Out3 O;
              cm2_win_at(win, sp, &cw[0][0], 66, ls, lx0 + j);
              cm2_emit3w(O, zcx, signlut, win);
              cm2_seq_put8(sq[th][0], O.p0); cm2_seq_put4(sq[th][1], O.p1); cm2_seq_put4(sq[th][2], O.hdr); cm2_seq_put8(sq[th][2], O.p2);
            }
          }
          for (int th = 0; th < 64; th++) for (int q = 0; q < 3; q++) cm2_seq_end(sq[th][q]);
          for (int pass = 0; pass < 3; pass++)
            if (off[pass] != (int)e3[pass].size() || memcmp(qb[pass].data(), e3[pass].data(), e3[pass].size())) return -16;
        }
        // the sparse form: plain columns put their RUN byte, the others are listed with their stream offsets and worked off in any order
        {
          std::vector<uint32_t> sb[3];
          for (int q = 0; q < 3; q++) sb[q].assign(T1_SEG_BYTES / 4 + 8, 0);
          std::vector<Cm2Entry> list;
          int o[3] = {0, 0, 0};
          for (int s = 0; s < nstr; s++)
            for (int x = 0; x < w; x++) {
              int n3[3]; bool plain = false;
              cm2_counts_w(&cw[s][x], &cw[s + 1][x], &cw[s + 2][x], n3, &plain);
              if (plain) { if (n3[0] || n3[1] || n3[2] != 1) return -11; cm2_or_word(sb[2].data() + (o[2] >> 2), (uint32_t)T1_CTX_RUN << (8 * (o[2] & 3))); }
              else if (n3[0] + n3[1] + n3[2]) list.push_back(Cm2Entry{(uint16_t)(s * 64 + x), (uint16_t)o[0], (uint16_t)o[1], (uint16_t)o[2]});
              for (int q = 0; q < 3; q++) o[q] += n3[q];
            }
          for (size_t k = list.size(); k-- > 0;) {  // (backwards: the order must not matter)
            const Cm2Entry e = list[k];
            Cm2Win win; Cm2Out3 O; Cm2Stream t0, t1, t2;
            cm2_win_at(win, sp, &cw[0][0], 66, e.col >> 6, e.col & 63);
            cm2_emit3w(O, zcx, signlut, win);
            cm2_stream_begin(t0, (uint8_t*)sb[0].data(), e.o0); cm2_stream_begin(t1, (uint8_t*)sb[1].data(), e.o1); cm2_stream_begin(t2, (uint8_t*)sb[2].data(), e.o2);
            cm2_stream_put8(t0, O.p0); cm2_stream_put4(t1, O.p1); cm2_stream_put4(t2, O.hdr); cm2_stream_put8(t2, O.p2);
            cm2_stream_end(t0); cm2_stream_end(t1); cm2_stream_end(t2);
          }
          for (int pass = 0; pass < 3; pass++)
            if (o[pass] != (int)e3[pass].size() || memcmp(sb[pass].data(), e3[pass].data(), e3[pass].size())) return -12;
        }
      }
      for (int pass = 0; pass < 3; pass++) {
        const uint8_t* seg = segs.data() + (size_t)p * T1_SEG_BYTES + (pass == 0 ? 0 : (pass == 1 ? T1_SEG1 : T1_SEG2));
        if (first && pass < 2) { if (!e3[pass].empty()) return -6; continue; }
        if ((int)e3[pass].size() != cnt[p * 3 + pass] || memcmp(e3[pass].data(), seg, e3[pass].size())) return -6;
        if (memcmp((const uint8_t*)wbuf[pass].data() + pass + 1, seg, e3[pass].size())) return -7;
        if (d3[pass] != dist[p * 3 + pass]) return -8;
      }
    }
  }
  static uint64_t mq2[94];
  for (int i = 0; i < 94; i++) mq2[i] = mq2_pair(mqtab, i);
  uint32_t cxs[20];
  std::vector<uint32_t> buf32(((size_t)w * h + 16 + 3) & ~(size_t)3);  // capacity: a multiple of 16 bytes
  uint8_t* bufp = (uint8_t*)buf32.data();
  T1Lane t;
  t.cx = cxs; t.mqtab = mqtab; t.mq2 = mq2; t.zclut = zclut; t.out = bufp;
  t.cs = 1; t.cap = (int)buf32.size() * 4; t.w = w; t.h = h; t.orient = orient; t.nbps = nb;
  T1Result r;
  int pl[T1_MAXPASS]; long long pd[T1_MAXPASS];
  r.pass_len = pl; r.pass_dist = pd;
  auto Seg = [&](int p) { return (const uint8_t*)segs.data() + (size_t)p * T1_SEG_BYTES; };
  auto Cnt = [&](int p, int pass) { return cnt[p * 3 + pass]; };
  auto Dist = [&](int p, int pass) { return dist[p * 3 + pass]; };
  t1_mq_block(t, Seg, Cnt, Dist, r, true);  // the two-wave form of the coder (front / back halves)
  if (r.overflow) return -2;
  memcpy(out, bufp + 1, r.len);
  *npasses = r.npasses; *nbps = r.nbps;
  for (int i = 0; i < r.npasses; i++) { pass_len[i] = pl[i]; if (pass_dist) pass_dist[i] = pd[i]; }
  return r.len;
}
