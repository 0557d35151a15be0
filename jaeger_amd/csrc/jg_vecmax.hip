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

// Embedding lookup + SinusoidalPositionEmbedding (builder.py:886-892: x = Add()([x, positional])) - one thread per
// (position, 4 channels); the position rows are computed once on the host (program.py) and live in the weight blob
template <typename ID>
__global__ __launch_bounds__(256) void embed_pos_kernel(const ID *__restrict__ ids, int64_t n_pos, int L, const float *__restrict__ table,
                                                        int vocab, int c4, const float *__restrict__ pe, float *__restrict__ out,
                                                        uint8_t *__restrict__ mask) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_pos * c4) return;
  const int64_t pos = q / c4;
  const int g = (int)(q - pos * c4);
  const int id = min((int)ids[pos], vocab - 1);
  float4 v = reinterpret_cast<const float4 *>(table)[(int64_t)id * c4 + g];
  if (pe != nullptr) {
    const float4 p = reinterpret_cast<const float4 *>(pe)[(pos % L) * c4 + g];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  reinterpret_cast<float4 *>(out)[q] = v;
  if (g == 0 && mask != nullptr) mask[pos] = id != 0;
}

}  // namespace

int jg_launch_embed_pos(const void *ids, int id_bytes, int64_t n_pos, int L, const float *table, int vocab, int c, const float *pe,
                        float *out, uint8_t *mask, hipStream_t s) {
  JG_REQUIRE(c > 0 && c % 4 == 0 && vocab > 0 && L > 0 && (id_bytes == 1 || id_bytes == 2), JG_ERR_UNSUPPORTED,
             "embed: %d channels (multiples of 4), vocabulary %d, ids of %d bytes", c, vocab, id_bytes);
  if (n_pos == 0) return JG_OK;
  const int64_t n = n_pos * (c / 4);
  const dim3 grid((unsigned)((n + 255) / 256));
  if (id_bytes == 2)
    hipLaunchKernelGGL(embed_pos_kernel<uint16_t>, grid, dim3(256), 0, s, static_cast<const uint16_t *>(ids), n_pos, L, table, vocab,
                       c / 4, pe, out, mask);
  else
    hipLaunchKernelGGL(embed_pos_kernel<uint8_t>, grid, dim3(256), 0, s, static_cast<const uint8_t *>(ids), n_pos, L, table, vocab,
                       c / 4, pe, out, mask);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

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
