// Main-loop PROBE for the round-4 review's item 2 (results are garbage of the right statistics; nothing here is shipped):
// would Toom-Cook F(2,5) - 6 instead of 10 channel-GEMMs per output pair of the k = 5 conv - pay on the matrix cores once
// its input transform has to run on the vector pipe?  scripts/r5_toomcook_study.py says the accuracy survives (narrowly).
//
// One loop iteration of every variant stands for the SAME logical work: a wave's 64 positions x 64 output channels x one
// 16-channel input chunk x 5 taps of the split-f16 conv (three v_mfma_f32_32x32x16_f16 per product).
//   direct : the operand pattern of conv_f16x3_kernel - per tap 4 weight + 4 activation fragments from LDS, 12 MFMAs: 60
//            MFMAs and 40 ds_read_b128 per iteration, no vector arithmetic.
//   tc     : 12 raw activation fragments (six positions of the lane's output pair, hi / lo), hi + lo -> f32 (48
//            v_fma_mix), the input transform B^T of the points {0, +-1, +-2, inf} on packed f32 (56 v_pk_*), the re-split
//            into f16 planes (24 v_cvt_pk + 48 v_fma_mix + 24 v_cvt_pk), then per component 4 weight fragments and 6
//            MFMAs: 36 MFMAs, 36 ds_read_b128, ~200 vector instructions per iteration; twelve accumulator blocks.
//   tc-nox : the same without the transform arithmetic (operands used as loaded): the ceiling of the form.
//   tc-32  : tc with ONE 32-channel block per wave (six accumulator blocks: two waves per SIMD fit), 18 MFMAs per ~200
//            vector instructions - what a wave pays when the transform is not shared.
// Random f16 operands (the chip's clock follows the data).  Prints time per iteration-equivalent and the ratio to direct.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/tc_probe.hip -o /tmp/tc_probe && /tmp/tc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float mix_sum0(unsigned hi2, unsigned lo2) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi2), "v"(lo2));
  return r;
}
__device__ __forceinline__ float mix_sum1(unsigned hi2, unsigned lo2) {
  float r;
  asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi2), "v"(lo2));
  return r;
}
__device__ __forceinline__ float mix_rem0(float v, unsigned hi2) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v));
  return r;
}
__device__ __forceinline__ float mix_rem1(float v, unsigned hi2) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v));
  return r;
}

// MODE 0 direct, 1 tc, 2 tc without the transform arithmetic, 3 tc with one 32-channel block per wave
template <int MODE>
__global__ __launch_bounds__(256) void probe(const uint4 *src, int iters, float *sink, unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  // x image: [2 planes][2 halves][160 rows]; weights: [6][2 ch-blocks][2 planes][64 lanes]
  constexpr int XR = 160, X_ITEMS = 4 * XR, W_ITEMS = 6 * 2 * 2 * 64;
  for (int q = threadIdx.x; q < X_ITEMS + W_ITEMS; q += 256) lds[q] = src[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const uint4 *X = lds + h * XR + i, *W = lds + X_ITEMS + lane;
  constexpr int NCB = MODE == 3 ? 1 : 2;                  // 32-channel blocks per wave
  constexpr int NACC = MODE == 0 ? 2 * NCB : 6 * NCB;
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const int sh = it & 7;                                // (operands move a little from iteration to iteration)
    if constexpr (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        half8 wh[2], wl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const uint4 a = W[((t * 2 + cb) * 2 + 0) * 64], b = W[((t * 2 + cb) * 2 + 1) * 64];
          wh[cb] = *reinterpret_cast<const half8 *>(&a);
          wl[cb] = *reinterpret_cast<const half8 *>(&b);
        }
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const uint4 a = X[pb * 32 + 3 * t + sh], b = X[2 * XR + pb * 32 + 3 * t + sh];
          const half8 xh = *reinterpret_cast<const half8 *>(&a), xl = *reinterpret_cast<const half8 *>(&b);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            f32x16 &c = acc[pb * 2 + cb];
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cb], xl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[cb], xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[cb], xh, c, 0, 0, 0);
          }
        }
      }
    } else {
      // the six input positions of this lane's output pair (phase positions 2i .. 2i + 5: rows 3 apart in the dilated conv)
      uint4 rh[6], rl[6];
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        rh[p] = X[2 * i * 0 + i + 3 * p + sh];            // (a probe: any conflict-free row pattern of the right count)
        rl[p] = X[2 * XR + i + 3 * p + sh];
      }
      uint4 th[6], tl[6];                                 // transformed operands, re-split
      if constexpr (MODE == 2) {
#pragma unroll
        for (int p = 0; p < 6; ++p) { th[p] = rh[p]; tl[p] = rl[p]; }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                     // channel pairs (2q, 2q + 1) of the lane's eight
          f32x2 d[6];
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            const unsigned hw = reinterpret_cast<const unsigned *>(&rh[p])[q], lw = reinterpret_cast<const unsigned *>(&rl[p])[q];
            d[p] = f32x2{mix_sum0(hw, lw), mix_sum1(hw, lw)};
          }
          // B^T of F(2,5) / F(4,3) on the points {0, 1, -1, 2, -2, inf}
          const f32x2 c4 = {4.f, 4.f}, c5 = {5.f, 5.f}, c2 = {2.f, 2.f};
          f32x2 o[6];
          o[0] = c4 * d[0] - c5 * d[2] + d[4];
          const f32x2 a1 = d[4] - c4 * d[2], b1 = d[3] - c4 * d[1];
          o[1] = a1 + b1;
          o[2] = a1 - b1;
          const f32x2 a2 = d[4] - d[2], b2 = c2 * (d[3] - d[1]);
          o[3] = a2 + b2;
          o[4] = a2 - b2;
          o[5] = c4 * d[1] - c5 * d[3] + d[5];
#pragma unroll
          for (int p = 0; p < 6; ++p) {
            const half2_t hh = {(_Float16)o[p].x, (_Float16)o[p].y};
            const unsigned hw = *reinterpret_cast<const unsigned *>(&hh);
            const half2_t ll = {(_Float16)mix_rem0(o[p].x, hw), (_Float16)mix_rem1(o[p].y, hw)};
            reinterpret_cast<unsigned *>(&th[p])[q] = hw;
            reinterpret_cast<unsigned *>(&tl[p])[q] = *reinterpret_cast<const unsigned *>(&ll);
          }
        }
      }
#pragma unroll
      for (int c6 = 0; c6 < 6; ++c6) {
        const half8 xh = *reinterpret_cast<const half8 *>(&th[c6]), xl = *reinterpret_cast<const half8 *>(&tl[c6]);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const uint4 a = W[((c6 * 2 + cb) * 2 + 0) * 64], b = W[((c6 * 2 + cb) * 2 + 1) * 64];
          const half8 wh = *reinterpret_cast<const half8 *>(&a), wl = *reinterpret_cast<const half8 *>(&b);
          f32x16 &c = acc[c6 * NCB + cb];
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
static double run(const char *name, int wg_per_cu, const uint4 *d_src, float *d_sink, unsigned long long *d_clk, int iters,
                  double work_per_iter, double base) {
  const int grid = 256 * wg_per_cu;
  const size_t smem = (size_t)(4 * 160 + 6 * 2 * 2 * 64) * 16;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), smem, 0, d_src, iters, d_sink, d_clk);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), smem, 0, d_src, iters, d_sink, d_clk);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  std::vector<unsigned long long> clk(2 * (size_t)grid);
  CK(hipMemcpy(clk.data(), d_clk, clk.size() * 8, hipMemcpyDeviceToHost));
  double ghz = 0;
  for (int g = 0; g < grid; ++g) ghz += (double)clk[2 * g] / (double)clk[2 * g + 1] * 0.1;
  ghz /= grid;
  // logical work done per second: waves x iterations x work per iteration
  const double per_s = (double)grid * 4.0 * iters * work_per_iter / (best * 1e-3);
  printf("%-44s %d WG/CU  %8.3f ms  clock %.2f GHz  %7.1f logical TFLOP/s (f32-equivalent)%s", name, wg_per_cu, best, ghz,
         per_s / 1e12, base > 0 ? "" : "\n");
  if (base > 0) printf("   x%.3f of direct\n", per_s / base);
  return per_s;
}

int main() {
  const size_t items = 4 * 160 + 6 * 2 * 2 * 64;
  std::vector<unsigned short> h(items * 8);
  unsigned s = 12345u;
  for (auto &v : h) {                                       // random f16 in +-[0.25, 4)
    s = s * 1664525u + 1013904223u;
    v = (unsigned short)(((s >> 31) << 15) | ((13u + ((s >> 20) & 3u)) << 10) | ((s >> 8) & 0x3ffu));
  }
  uint4 *d_src;
  float *d_sink;
  unsigned long long *d_clk;
  CK(hipMalloc(&d_src, items * 16));
  CK(hipMemcpy(d_src, h.data(), items * 16, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_sink, 4));
  CK(hipMalloc(&d_clk, 2 * 512 * 8));
  const int iters = 20000;
  // logical f32-equivalent work of one iteration: 64 positions x 64 channels x 16 input channels x 5 taps x 2
  const double w64 = 2.0 * 64 * 64 * 16 * 5, w32 = w64 / 2;
  const double d2 = run<0>("direct (60 MFMA, 40 LDS reads)", 2, d_src, d_sink, d_clk, iters, w64, 0);
  run<0>("direct", 1, d_src, d_sink, d_clk, iters, w64, d2);
  run<1>("toom-cook (36 MFMA, ~200 vector instr.)", 1, d_src, d_sink, d_clk, iters, w64, d2);
  run<1>("toom-cook", 2, d_src, d_sink, d_clk, iters, w64, d2);
  run<2>("toom-cook without the transform arithmetic", 1, d_src, d_sink, d_clk, iters, w64, d2);
  run<2>("toom-cook without the transform arithmetic", 2, d_src, d_sink, d_clk, iters, w64, d2);
  run<3>("toom-cook, one 32-channel block per wave", 2, d_src, d_sink, d_clk, iters, w32, d2);
  return 0;
}
