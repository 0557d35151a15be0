// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal INPUTS on gfx950?  (The split-f16 kernels pre-scale their weights
// by a power of two so that the lo plane stays normal; if subnormals are not flushed the scale could be 1 and the
// epilogue affine folded into the weights / the C operand.)   hipcc --offload-arch=gfx950 -O2 mfma_denorm.hip -o mfma_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float *out) {
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float *d, h;
  hipMalloc(&d, 4);
  const float vals[][2] = {{1.0f, 1.0f}, {5.9604645e-8f, 1.0f}, {3.0517578e-5f, 1.0f}, {1.0f, 3.0517578e-5f}, {3.0517578e-5f, 4096.0f}};
  for (auto &v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, v[0], v[1], d);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %.9g (f16 %s) b = %.9g: sum of 16 products = %.9g, expected %.9g\n", v[0], v[0] < 6.1e-5f ? "subnormal" : "normal", v[1], h,
           16.0 * (double)(float)(_Float16)v[0] * (double)(float)(_Float16)v[1]);
  }
  return 0;
}
