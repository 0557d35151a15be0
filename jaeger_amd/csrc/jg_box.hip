// Box calibration: a bare v_mfma_f32_32x32x16_f16 loop on random operands held in registers - no LDS, no global
// traffic inside the loop - timed by HIP events and stamped once around the loop with s_memtime / s_memrealtime
// (MI355X_MICROARCH.md, DVFS item 6: in-kernel clock = d s_memtime / d s_memrealtime x 100 MHz, median over workgroups).
// MI355X devices differ by up to 12 % on MFMA-dense loops (same guide, item 5); bench.py runs this straight behind its
// timed steps so that a headline can be read against the matrix-core rate THIS device holds under load.
// The stamps go to a buffer of their own; nothing else reads them.
#include "jg_common.h"

#include <algorithm>
#include <vector>

typedef _Float16 box_v8h __attribute__((ext_vector_type(8)));
typedef float box_v16f __attribute__((ext_vector_type(16)));

namespace {

constexpr int BOX_SETS = 8;          // operand register sets cycled through: consecutive MFMAs see different A and B
constexpr int BOX_ACCS = 4;          // independent accumulators (back-to-back issue without a dependent stall)
constexpr int BOX_UNROLL = 16;       // MFMAs per loop iteration

__device__ __forceinline__ unsigned box_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// a random f16 in +-[0.5, 2): sign and 10 mantissa bits random, exponent 14 or 15 - the multipliers toggle like on
// real activations, the accumulators random-walk far inside the f32 range
__device__ __forceinline__ _Float16 box_rand_half(unsigned h) {
  const unsigned short bits = (unsigned short)(((h >> 31) << 15) | ((14u + ((h >> 11) & 1u)) << 10) | (h & 0x3ffu));
  return __builtin_bit_cast(_Float16, bits);
}

__global__ __launch_bounds__(256, 2) void box_mfma_kernel(int iters, unsigned seed, unsigned long long *stamps,
                                                          float *sink) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  box_v8h a[BOX_SETS], b[BOX_SETS];
  for (int s = 0; s < BOX_SETS; ++s)
    for (int j = 0; j < 8; ++j) {
      a[s][j] = box_rand_half(box_hash(seed + gid * 131u + s * 17u + j));
      b[s][j] = box_rand_half(box_hash(~seed + gid * 257u + s * 29u + j * 3u));
    }
  box_v16f acc[BOX_ACCS];
  for (int q = 0; q < BOX_ACCS; ++q)
    for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < BOX_UNROLL; ++u)
      acc[u % BOX_ACCS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u % BOX_SETS], b[(u * 3 + u / BOX_SETS) % BOX_SETS],
                                                                 acc[u % BOX_ACCS], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_sched_barrier(0);
  float s = 0.f;
  for (int q = 0; q < BOX_ACCS; ++q)
    for (int j = 0; j < 16; ++j) s += acc[q][j];
  if (s == 123.456f) sink[0] = s;                        // keeps the loop alive; practically never written
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

}  // namespace

extern "C" int jg_box_calibrate(jg_engine *e, double seconds, double *tflops, double *clock_ghz, double *info) {
  JG_REQUIRE(e != nullptr && tflops != nullptr && clock_ghz != nullptr, JG_ERR_INVALID, "jg_box_calibrate: bad arguments");
  JG_REQUIRE(seconds > 0.0 && seconds <= 30.0, JG_ERR_INVALID, "jg_box_calibrate: seconds must be in (0, 30]");
  JG_HIP(hipSetDevice(e->dev));
  const int grid = e->n_cu * 2;                          // two workgroups of four waves per CU = two waves per SIMD
  unsigned long long *d_st = nullptr;
  float *d_sink = nullptr;
  JG_HIP(hipMalloc(&d_st, sizeof(unsigned long long) * 2 * grid));
  JG_HIP(hipMalloc(&d_sink, sizeof(float)));
  auto cleanup = [&]() { (void)hipFree(d_st); (void)hipFree(d_sink); };
  // ~20 ms per launch at the nominal rate: 2 waves x 16 MFMAs x 32 cycles = 1 024 cycles per SIMD and iteration
  const int iters = 40000;
  const double flop_per_launch = (double)grid * 4.0 * iters * BOX_UNROLL * (2.0 * 32 * 32 * 16);
  double total_ms = 0.0, last_ms = 0.0;
  int launches = 0;
  // one untimed launch (code object load), then back-to-back timed launches until `seconds` of kernel time have passed
  hipLaunchKernelGGL(box_mfma_kernel, dim3(grid), dim3(256), 0, e->stream, 256, 1u, d_st, d_sink);
  while (total_ms < seconds * 1e3 && launches < 4096) {
    if (hipEventRecord(e->t0, e->stream) != hipSuccess) break;
    hipLaunchKernelGGL(box_mfma_kernel, dim3(grid), dim3(256), 0, e->stream, iters, 0x9e3779b9u + launches, d_st, d_sink);
    if (hipEventRecord(e->t1, e->stream) != hipSuccess || hipEventSynchronize(e->t1) != hipSuccess) break;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e->t0, e->t1) != hipSuccess) break;
    total_ms += ms;
    last_ms = ms;
    ++launches;
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess || launches == 0) {
    cleanup();
    jg_set_error("jg_box_calibrate: %s", err != hipSuccess ? hipGetErrorString(err) : "no launch completed");
    return JG_ERR_HIP;
  }
  std::vector<unsigned long long> st(2 * (size_t)grid);
  err = hipMemcpy(st.data(), d_st, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost);
  cleanup();
  if (err != hipSuccess) {
    jg_set_error("jg_box_calibrate: %s", hipGetErrorString(err));
    return JG_ERR_HIP;
  }
  std::vector<double> ghz;
  ghz.reserve(grid);
  for (int i = 0; i < grid; ++i)
    if (st[2 * i + 1] > 0) ghz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 0.1);   // cycles per 10 ns tick
  std::sort(ghz.begin(), ghz.end());
  *clock_ghz = ghz.empty() ? 0.0 : ghz[ghz.size() / 2];
  // rate of the LAST launch: the chip has settled under the load by then (info[2] holds the mean over all launches)
  *tflops = flop_per_launch / (last_ms * 1e-3) / 1e12;
  if (info) {
    info[0] = (double)launches;
    info[1] = last_ms;
    info[2] = flop_per_launch * launches / (total_ms * 1e-3) / 1e12;      // mean over all launches
    info[3] = ghz.empty() ? 0.0 : ghz.front();
    info[4] = ghz.empty() ? 0.0 : ghz.back();
  }
  return JG_OK;
}
