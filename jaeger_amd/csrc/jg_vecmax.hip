// JG_OP_VECMAX: out[row][out_off + c] = max over g of in[row][g * width + c] - NMDMerge(mode="max") over the projected NMD
// vectors of a window (nnlib/v2/nmd.py:150-152: tf.reduce_max over the stacked projections).  A few hundred floats per
// window behind the representation learner: one thread per output element.
#include "jg_vecmax.h"

namespace {

__global__ void vecmax_kernel(const float *__restrict__ in, int in_ld, int groups, int width, int64_t n_rows,
                              float *__restrict__ out, int out_ld, int out_off) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * width) return;
  const int64_t row = idx / width;
  const int c = (int)(idx - row * width);
  const float *v = in + row * in_ld + c;
  float m = v[0];
  for (int g = 1; g < groups; ++g) {
    const float x = v[(size_t)g * width];
    m = (x > m || x != x) ? x : m;                  // (tf.reduce_max propagates NaN)
  }
  out[row * out_ld + out_off + c] = m;
}

}  // namespace

int jg_launch_vecmax(const float *in, int in_ld, int groups, int width, int64_t n_rows, float *out, int out_ld, int out_off,
                     hipStream_t s) {
  JG_REQUIRE(in != nullptr && out != nullptr && groups >= 1 && width >= 1 && in_ld >= groups * width && out_ld >= out_off + width,
             JG_ERR_INVALID, "vecmax: bad geometry (%d groups of %d in rows of %d -> offset %d of %d)", groups, width, in_ld,
             out_off, out_ld);
  if (n_rows <= 0) return JG_OK;
  const int64_t n = n_rows * width;
  hipLaunchKernelGGL(vecmax_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, in_ld, groups, width, n_rows, out,
                     out_ld, out_off);
  JG_HIP(hipGetLastError());
  return JG_OK;
}
