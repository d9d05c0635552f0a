/*
 * include/qsvc_hip.h -- C ABI of libqsvc_hip.so, the MI355X (gfx950) implementation of the
 * `mctf compress` encode hot path of QSVC (MCTF analysis + JPEG 2000 texture coding).
 *
 * The reference exposes NO library / FFI / plug-in ABI for this path: its boundary is
 * "process + flags + raw files in cwd + exit status" (SURVEY.md section 8b).  Every entry point
 * below therefore replaces one reference *executable* (or one stage inside it) and cites it; the
 * Python-3 host in qsvc_amd/cli/ keeps the executables' names, flags and cwd file layout and
 * binds these symbols through ctypes (see INTEGRATION.md for the stub a maintainer would add).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *   - one qsvc_ctx per GPU; calls on one context are serialised by the caller; contexts are
 *     independent (one host process per GPU).
 *   - every function returns 0 (QSVC_OK) or a negative error code and never throws;
 *     qsvc_last_error() gives the message.  There is NO CPU fallback: without a usable GPU
 *     qsvc_init fails.
 *   - "frames" are planar I420 u8 pictures, F = W*H + 2*(W/2)*(H/2) bytes each, the layout of
 *     the reference's low_t / high_t / even_t / odd_t files (trunk/src/split.cpp:55-56).
 *   - motion fields are int16, per B picture four planes PREV-X, PREV-Y, NEXT-X, NEXT-Y of
 *     BY x BX (trunk/src/motion.cpp:9-15,93-101), the layout of motion_t / motion_filtered_t.
 *   - device-resident "files": a context keeps, per temporal subband t, the pictures the
 *     reference keeps in cwd files.  qsvc_put_* / qsvc_get_* move them across PCIe only when the
 *     host wants them (file I/O, tests).
 */
#ifndef QSVC_HIP_H
#define QSVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qsvc_ctx qsvc_ctx;

enum {
  QSVC_OK = 0,
  QSVC_ERR_INVALID = -1,     /* bad argument                                               */
  QSVC_ERR_UNSUPPORTED = -2, /* geometry/option outside what this path implements           */
  QSVC_ERR_HIP = -3,         /* HIP runtime error (message has the hipError string)         */
  QSVC_ERR_STATE = -4,       /* stage called before its inputs exist                        */
  QSVC_ERR_NOMEM = -5
};

/* Which device-resident picture sequence ("file") a put/get addresses. */
enum {
  QSVC_LOW = 0,  /* low_t   : t = 0 is the input sequence (./low_0)                          */
  QSVC_HIGH = 1, /* high_t  : t >= 1                                                         */
  QSVC_PRED = 2  /* prediction_even_t debug pictures (only kept when enabled)                */
};
enum {
  QSVC_MOTION_RAW = 0,     /* motion_t          (motion_estimate output)                      */
  QSVC_MOTION_FILTERED = 1 /* motion_filtered_t (decorrelate output: zeros for 'I' pictures)  */
};

/* ---- context ------------------------------------------------------------------------------ */
int qsvc_init(int device, qsvc_ctx** out);
void qsvc_destroy(qsvc_ctx* ctx);
const char* qsvc_last_error(qsvc_ctx* ctx);
/* HIP stream the context launches on (hipStream_t as void*), for callers that time with events. */
void* qsvc_stream(qsvc_ctx* ctx);
int qsvc_sync(qsvc_ctx* ctx);

/* Geometry of the sequence.  Frees every device-resident picture of a previous sequence.
 * keep_prediction != 0 keeps the prediction_even_t dump decorrelate writes (decorrelate.cpp:454-472). */
int qsvc_seq_config(qsvc_ctx* ctx, int pixels_in_x, int pixels_in_y, int keep_prediction);

/* ---- host <-> device "files" ---------------------------------------------------------------- */
/* replaces: reading ./low_0 etc. (split.cpp:229-253, texture.cpp:122-129) */
int qsvc_put_frames(qsvc_ctx* ctx, int kind, int t, int first, int count, int total, const uint8_t* i420);
int qsvc_get_frames(qsvc_ctx* ctx, int kind, int t, int first, int count, uint8_t* i420);
/* motion_t / motion_filtered_t (motion.cpp:69-101): [pictures/2][4][BY][BX] int16 */
int qsvc_put_motion(qsvc_ctx* ctx, int kind, int t, int b_pictures, int block_size, const int16_t* mv);
/* get: the caller's buffer holds b_pictures fields at block_size; QSVC_ERR_STATE when that is not what is resident */
int qsvc_get_motion(qsvc_ctx* ctx, int kind, int t, int b_pictures, int block_size, int16_t* mv);
/* frame_types_t: one 'I'/'B' byte per B picture (decorrelate.cpp:982,1005) */
int qsvc_put_frame_types(qsvc_ctx* ctx, int t, int b_pictures, const char* types);
int qsvc_get_frame_types(qsvc_ctx* ctx, int t, int b_pictures, char* types);
/* [b_pictures][3][256] u32: luma histogram of the odd picture, of residue+128, of mv+128
 * (decorrelate.cpp:798-819,945-950) -- what the I/B decision is made from. */
int qsvc_get_histograms(qsvc_ctx* ctx, int t, int b_pictures, uint32_t* hist);

/* Options the reference's stage tools take as flags; they apply to the stage calls that follow.
 *   block_overlaping : decorrelate / correlate --block_overlaping (decorrelate.cpp:84-172): 0, 1 or a power of two
 *   border_size      : motion_estimate --border_size (motion_estimate.cpp:213-217,714): the SAD window grows by it
 *   subpixel_accuracy: only 0 (QSVC_ERR_UNSUPPORTED otherwise) */
int qsvc_set_mctf_options(qsvc_ctx* ctx, int block_overlaping, int border_size, int subpixel_accuracy);

/* ---- MCTF stages: one call = one reference executable over one temporal level -------------- */
/* replaces `motion_estimate` (motion_estimate.cpp:490-912; me_for_image :260-412):
 * low_{t-1} -> motion_t.  pictures = number of pictures in low_{t-1}. */
int qsvc_motion_estimate(qsvc_ctx* ctx, int t, int pictures, int block_size, int search_range);
/* replaces `decorrelate` (decorrelate.cpp:199-1078; predict :69-189; entropy.cpp:20-34):
 * low_{t-1}, motion_t -> high_t, frame_types_t, motion_filtered_t [, prediction_even_t]. */
int qsvc_decorrelate(qsvc_ctx* ctx, int t, int pictures, int block_size, int search_range, int always_B);
/* replaces `update` (update.cpp:158-684; update() :50-148):
 * low_{t-1}, high_t, motion_filtered_t, frame_types_t -> low_t. */
int qsvc_update(qsvc_ctx* ctx, int t, int pictures, int block_size, float update_factor);
/* replaces `analyze_step` (analyze_step.py:117-232): split + the three stages above, nothing
 * leaving the device. */
int qsvc_analyze_step(qsvc_ctx* ctx, int t, int pictures, int block_size, int search_range, int always_B,
                      float update_factor);

/* ---- inverse temporal level (synthesize_step.py:96-151) ---------- */
/* Inputs resident for subband t: low_t (pictures/2+1), high_t, frame_types_t and the decoded motion_t, put as
 * QSVC_MOTION_FILTERED.  "merge" (split.cpp without ANALYZE) is the layout: both stages write low_{t-1} in place.
 * replaces `un_update` (update.cpp without ANALYZE, :100-104,:466-478): low_t -> the even pictures of low_{t-1}. */
int qsvc_un_update(qsvc_ctx* ctx, int t, int pictures, int block_size, float update_factor);
/* replaces `correlate` (decorrelate.cpp without ANALYZE, :1029-1068): even pictures of low_{t-1}, high_t ->
 * the odd pictures of low_{t-1} ('I': the stored picture; 'B': clip(residue + prediction)). */
int qsvc_correlate(qsvc_ctx* ctx, int t, int pictures, int block_size, int search_range);
/* replaces `synthesize_step` (synthesize_step.py): the two stages above. */
int qsvc_synthesize_step(qsvc_ctx* ctx, int t, int pictures, int block_size, int search_range, float update_factor);

/* ---- open-GOP boundary exchange (multi-GPU) -------------------------- */
/* GOPs are open: the last even picture of one GPU's shard is the first of the next one's, and its
 * update needs the neighbour's boundary B picture (update.cpp:605-614,649-653).  A halo record =
 * that B picture of high_t (F bytes) + its filtered motion field (4*BY*BX int16) + its frame type
 * (1 byte, padded to 16).  Records live in device buffers owned by the context. */
size_t qsvc_halo_bytes(qsvc_ctx* ctx, int block_size);
/* Host-staged transport (any byte transport, e.g. torch.distributed/gloo):
 * export side 0 = OUR first B picture (for the left neighbour), side 1 = our last (for the right one);
 * import side 0 = record received FROM the left neighbour, side 1 = from the right; NULL clears. */
int qsvc_halo_export_host(qsvc_ctx* ctx, int t, int side, void* dst_host);
int qsvc_halo_import_host(qsvc_ctx* ctx, int t, int side, const void* src_host);
/* RCCL transport over xGMI, device to device: ranks form a line (no wrap-around), rank r owns the
 * r-th shard.  id is an ncclUniqueId (128 bytes) made by rank 0 and distributed by the caller. */
int qsvc_comm_unique_id(void* id_out, size_t cap);
int qsvc_comm_init(qsvc_ctx* ctx, int rank, int nranks, const void* id, size_t id_len);
int qsvc_comm_destroy(qsvc_ctx* ctx);
/* ncclGroupStart; send/recv the halo records of level t with ranks r-1 and r+1; ncclGroupEnd; then
 * import what arrived.  Call between qsvc_decorrelate and qsvc_update. */
int qsvc_halo_exchange(qsvc_ctx* ctx, int t);
/* The same for the n contexts ONE rank keeps in flight on one GPU (n independent sequences, one context each; same picture size
 * and block size): their records travel as ONE message of n records per neighbour.  ctxs[0] owns the batch buffers and, for the
 * RCCL transport, the rank's only communicator (qsvc_comm_init on ctxs[0] alone): a rank needs one communicator however many
 * sequences it codes at once.  Streams are ordered with events: every context packs on its own stream, the message leaves on
 * ctxs[0]'s stream after all of them, and every context's update waits for the arrival.
 * Host-staged: export side s of all n contexts into dst (qsvc_halo_batch_bytes bytes), import what the neighbour's export of the
 * other side produced (NULL: no neighbour on that side). */
size_t qsvc_halo_batch_bytes(qsvc_ctx* const* ctxs, int n, int t);
int qsvc_halo_batch_export_host(qsvc_ctx* const* ctxs, int n, int t, int side, void* dst_host);
int qsvc_halo_batch_import_host(qsvc_ctx* const* ctxs, int n, int t, int side, const void* src_host);
int qsvc_halo_exchange_batch(qsvc_ctx* const* ctxs, int n, int t);

/* ---- truncating 5/3 (unit-testable; used by the ME pyramid and the chroma x2 / /2 steps) ---- */
/* replaces dwt2d<short,_5_3>::analyze / ::synthesize (dwt2d.cpp:76-175, 5_3.cpp:39-115) on a HOST
 * int16 plane (copied to the device, transformed there, copied back). */
int qsvc_dwt53_trunc(qsvc_ctx* ctx, int16_t* plane, int H, int W, int stride, int levels, int inverse);

/* ---- JPEG 2000 texture coding (replaces the external `kdu_compress` call,
 *      texture_compress_fb_j2k.py:183-195) -------------------------------------------------- */
typedef struct {
  int reversible;     /* 1: 5/3 reversible (build extension), 0: 9/7 irreversible (the reference's
                         Creversible=no, texture_compress_fb_j2k.py:186)                          */
  int levels;         /* Clevels = SRLs-1 (texture_compress_fb_j2k.py:205-207)                     */
  int layers;         /* Clayers = nLayers                                                         */
  int cblk_w, cblk_h; /* code-block size, 64x64 (Kakadu default Cblk={64,64})                      */
  int use_sop;        /* Cuse_sop=yes (texture_compress_fb_j2k.py:194; header_size.c:46-58)        */
  float base_step;    /* 9/7: quantiser step of the finest subbands relative to 2^Rb (Qstep)       */
  int slopes[16];     /* -slope list (texture_compress.py:140-165); <= 0 entries: keep all passes  */
} qsvc_j2k_params;

/* Encode `count` pictures of a device-resident sequence, one single-component code-stream per
 * picture per colour component (texture_compress_fb_j2k.py:155-195: <file>_{Y,U,V}_NNNN.j2c).
 * out: host buffer of cap bytes receiving the 3*count code-streams back to back in the order
 * (picture, component); offs[3*count+1] receives their byte offsets. */
int qsvc_j2k_encode_frames(qsvc_ctx* ctx, int kind, int t, int first, int count, const qsvc_j2k_params* p,
                           uint8_t* out, size_t cap, size_t* offs);
/* Several sequences in ONE batch (one DWT / Tier-1 launch set for all of them): sequence i is
 * (kinds[i], ts[i], firsts[i], counts[i]) coded with prms[i].  The transform, step and code-block
 * parameters must agree across the batch; layers and slopes may differ (texture_compress.py:183-217 codes
 * high_1..high_{T-1} and low_{T-1} with different slope lists).  offs[3*sum(counts)+1]. */
int qsvc_j2k_encode_multi(qsvc_ctx* ctx, int nseq, const int* kinds, const int* ts, const int* firsts, const int* counts,
                          const qsvc_j2k_params* prms, uint8_t* out, size_t cap, size_t* offs);
/* Same on a host plane (copied to the device first); one code-stream. */
int qsvc_j2k_encode_plane(qsvc_ctx* ctx, const uint8_t* plane, int H, int W, const qsvc_j2k_params* p, uint8_t* out,
                          size_t cap, size_t* len);
/* `nplanes` host planes of 16-bit SIGNED samples (H x W each, back to back), one code-stream per plane: replaces the
 * `kdu_compress ... Creversible=yes Sprecision=16 Ssigned=yes Clevels=0` call on every motion-field component
 * (motion_compress_j2k.py:131-142).  prm->reversible must be 1 and prm->levels 0.  offs[nplanes+1]. */
int qsvc_j2k_encode_planes_s16(qsvc_ctx* ctx, const int16_t* planes, int nplanes, int H, int W, const qsvc_j2k_params* prm,
                               uint8_t* out, size_t cap, size_t* offs);
/* replaces the motion-field coding chain of `mctf compress` (compress.py:197-212 -> motion_compress.py:146-186: one
 * interlevel_motion_decorrelate process per level, bidirectional_motion_decorrelate on the coarsest, then motion_compress_j2k.py:98-142:
 * demux / split and one kdu_compress per component plane) for the levels ts[0] < ts[1] < ... whose motion_filtered_t is resident,
 * fields[k] fields each, one block size: residues on the host (the fields are kilobytes), every component plane of every level coded
 * lossless in one batch.  out: the 4 * sum(fields) code-streams back to back in (level, field, component) order,
 * offs[4 * sum(fields) + 1]; residues (optional): the motion_residue_t files' contents, levels back to back.
 * filtered_host (optional): the fields themselves (levels back to back, the motion_filtered_t files' contents) instead of the
 * context's resident ones -- a second context can then code the motion of a sequence while the first codes its textures. */
int qsvc_motion_encode(qsvc_ctx* ctx, int nlevels, const int* ts, const int* fields, int block_size, const int16_t* filtered_host,
                       int16_t* residues, uint8_t* out, size_t cap, size_t* offs);
/* The two together -- everything `mctf compress` does behind the analysis (compress.py:197-228: motion_compress, then
 * texture_compress) -- in ONE batch on the GPU: the subband pictures of the sequences (arguments as qsvc_j2k_encode_multi) and the
 * motion fields' component planes (arguments as qsvc_motion_encode, resident fields) share the Tier-1 launches.  out: first the
 * 3 * sum(counts) texture streams, then the 4 * sum(mfields) motion streams; offs has one entry more than streams.  The motion
 * planes use the textures' code-block size (64 x 64 by default, Kakadu's default for both). */
int qsvc_compress_code(qsvc_ctx* ctx, int nseq, const int* kinds, const int* ts, const int* firsts, const int* counts,
                       const qsvc_j2k_params* prms, int nlevels, const int* mts, const int* mfields, int block_size, int16_t* residues,
                       uint8_t* out, size_t cap, size_t* offs);
/* Unit-testable transform: level shift + `levels` of forward DWT of a host 8-bit plane, Mallat layout.
 * out receives H*W int32 (reversible 5/3) or H*W float32 (irreversible 9/7, before quantisation). */
int qsvc_j2k_dwt(qsvc_ctx* ctx, const uint8_t* plane, int H, int W, int reversible, int levels, void* out);
/* ---- JPEG 2000 decoding (replaces the external `kdu_expand` call,
 *      texture_expand_fb_j2k.py:153-162 and motion_expand_j2k.py) -------------------------- */
/* `nstreams` single-component code-streams (host memory) -> their sample planes, back to back in `out` (host): W*H bytes for
 * 8-bit unsigned streams, W*H int16 for 16-bit signed ones; offs[nstreams+1] receives the byte offsets, widths / heights
 * (optional) the geometry read from each SIZ marker.  max_layers <= 0 decodes every quality layer.  All streams of one call
 * share the transform (5/3 or 9/7).  Scope: one tile, LRCP, no precinct partition, code-block style 0. */
int qsvc_j2k_decode_planes(qsvc_ctx* ctx, int nstreams, const uint8_t* const* streams, const size_t* lens, int max_layers,
                           void* out, size_t cap, size_t* offs, int* widths, int* heights);
/* The same straight into a device-resident sequence (the decoder's qsvc_put_frames): pictures [first, first + count) of sequence
 * (kind, t) holding `total` pictures, from 3 * count code-streams in (picture, component Y U V) order, as
 * texture_expand_fb_j2k.py:153-181 reads <file>_{Y,U,V}_NNNN.j2c; the planes must have the sequence's geometry (qsvc_seq_config). */
int qsvc_j2k_decode_frames(qsvc_ctx* ctx, int kind, int t, int first, int count, int total, const uint8_t* const* streams,
                           const size_t* lens, int max_layers);
/* Several sequences in ONE decoding batch (what `texture_expand` does for high_{T-1} .. high_1 and low_{T-1}, texture_expand.py:85-131):
 * sequence k receives counts[k] pictures from firsts[k] on; the streams follow each other in (sequence, picture, component) order. */
int qsvc_j2k_decode_multi(qsvc_ctx* ctx, int nseq, const int* kinds, const int* ts, const int* firsts, const int* counts, const int* totals,
                          const uint8_t* const* streams, const size_t* lens, int max_layers);
/* Host threads that write the code-streams of one qsvc_j2k_encode_* call (Tier-2: packet headers, tag trees, markers; the
 * planes of a call are independent).  n >= 1 sets the count for this context; n = 0 returns to the default, which is the
 * environment variable QSVC_T2_THREADS as it was when the context coded its first plane, else 8.  Never more threads than
 * planes are used.  The reference runs one kdu_compress process per picture and component
 * (texture_compress_fb_j2k.py:180-195); this is the knob for sharing a host's cores between several contexts. */
int qsvc_set_tier2_threads(qsvc_ctx* ctx, int n);
/* Statistics of the last qsvc_j2k_encode_* call: code-blocks coded, coding passes, MQ symbols. */
int qsvc_j2k_stats(qsvc_ctx* ctx, uint64_t* codeblocks, uint64_t* passes, uint64_t* symbols);

/* A/B switches of the library's kernel forms (qsvc_amd/csrc/tuning.h: "mq_pipe", "mq_lds_small", "mq_lanes", "mq_sort", "t1_parts",
 * "t1_devlist", "pyr_tile", "me_piece", "pred_run", "upd_run", "upd_scatter", "dwt_pk" (1 = the register-blocked DWT kernels, 2 = 5/3 by the
 * staged packed kernel, 3 = one tile per workgroup on every level, 0 = the general tile kernel), "defer_ib", "dec_lanes", and "sync_block": 1 = host waits query the stream and sleep,
 * 0 = the runtime's spinning waits).  A context starts from the QSVC_* environment variables as they
 * were when it was created (read once, there); results never depend on the switches. */
int qsvc_set_tuning(qsvc_ctx* ctx, const char* name, int value);

/* Per-kernel-group timing with HIP events recorded on the context's own stream.
 * qsvc_set_profiling(ctx, 1) starts accumulating; qsvc_get_timings synchronises and returns, for each
 * named segment, the summed milliseconds and the number of launches since profiling was (re)enabled.
 * Returns the number of segments (<= cap). */
int qsvc_set_profiling(qsvc_ctx* ctx, int on);
int qsvc_get_timings(qsvc_ctx* ctx, const char** names, float* ms, int* launches, int cap);

#ifdef __cplusplus
}
#endif
#endif /* QSVC_HIP_H */
