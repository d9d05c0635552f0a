// temporary: JPEG 2000 entry points (replaced by j2k_api.hip)
#include "qsvc_ctx.h"
void j2k_release(qsvc_ctx*) {}
extern "C" {
int qsvc_j2k_encode_frames(qsvc_ctx* c, int, int, int, int, const qsvc_j2k_params*, uint8_t*, size_t, size_t*) { return c ? c->fail(QSVC_ERR_UNSUPPORTED, "j2k not built") : QSVC_ERR_INVALID; }
int qsvc_j2k_encode_plane(qsvc_ctx* c, const uint8_t*, int, int, const qsvc_j2k_params*, uint8_t*, size_t, size_t*) { return c ? c->fail(QSVC_ERR_UNSUPPORTED, "j2k not built") : QSVC_ERR_INVALID; }
int qsvc_j2k_stats(qsvc_ctx* c, uint64_t*, uint64_t*, uint64_t*) { return c ? c->fail(QSVC_ERR_UNSUPPORTED, "j2k not built") : QSVC_ERR_INVALID; }
}
