// qsvc_amd/csrc/mctf_margin.hip -- Q7: the chroma margin state picture i of a level leaves behind for picture i+1 when the
// block grid does not tile the picture (decorrelate.cpp:508-509,562-567 never predict those rows / columns, the planes are never
// cleared).  One launch per picture, in order; a few microseconds each (the margin is a thin band).
// A translation unit of its own, built at -O1: the fully inlined 5x5 analysis over the interpolated-chroma prediction lambdas
// makes the -O3 pipeline of this clang run for more than half an hour (found by bisection; everything else builds at -O3).
#include <hip/hip_runtime.h>
#include "mctf_kernels.h"

namespace qsvc {

__global__ void __launch_bounds__(256) k_pred_margin(PredArgs a, int i) {
  size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k < margin_geom(a.g).count()) pred_margin_thread(a, k, i, 1 + (int)blockIdx.y);
}
void launch_pred_margin(hipStream_t st, const PredArgs& a) {
  size_t cnt = margin_geom(a.g).count();
  for (int i = 0; i + 1 < a.g.nB; i++)  // the last picture's state has no reader
    hipLaunchKernelGGL(k_pred_margin, dim3((unsigned)((cnt + 255) / 256), 2), dim3(256), 0, st, a, i);
}

}  // namespace qsvc
