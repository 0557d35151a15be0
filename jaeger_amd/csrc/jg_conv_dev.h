// Device helpers shared by the split-f16 convolution kernels (jg_conv_f16_impl.h: the two-workgroup kernel and its
// variants; jg_conv_pc.hip: the producer / consumer kernel of the 128-channel k = 5 residual stacks).
#pragma once
#include "jg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HM = 256;       // tile rows (positions)
constexpr int HN = 128;       // tile cols (output channels)
constexpr int HT = 256;       // threads: 4 waves, 2 (positions) x 2 (channels), 128 x 64 outputs each
constexpr int NT = 1;         // tiles per workgroup pass
constexpr int A_ITERS = 5;    // 16-B activation pieces per thread per chunk (4*rows_a <= 1280)
constexpr int W_ITERS = 2;    // 16-B weight items per thread per slice
constexpr int W_ITEMS = 2 * 2 * HN;           // 16-B items of one weight slice (chunk, tap): 8 KB
constexpr int LUT_RS = 68;    // floats per LDS row of the first-layer table (64 + 4: rows 16 apart share banks)

// GELU (tanh form) as x * sigmoid(2u), u = sqrt(2/pi)(x + 0.044715 x^3): one v_exp_f32 and
// one v_rcp_f32 (~1 ulp each) instead of a libm tanhf; abs error < 1e-6 * |x|.
__device__ __forceinline__ float fast_gelu(float v) {
  const float t = v * (-2.3022082f - 0.10294324f * v * v);   // -2u * log2(e)
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// tanh-GELU of two values with the operation sequence of fast_gelu() - bit-identical results - on the packed-f32 forms:
// five v_pk_* instructions and four transcendentals per pair instead of ten + four
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fast_gelu2(f32x2 v) {
  const f32x2 c2 = {0.10294324f, 0.10294324f}, c1 = {-2.3022082f, -2.3022082f}, one = {1.0f, 1.0f};
  const f32x2 m = c2 * v;
  const f32x2 u = __builtin_elementwise_fma(-v, m, c1);
  const f32x2 t = v * u;
  f32x2 e;
#ifdef JG_PC_NOTRANS           // timing experiment: no transcendental instructions (results are garbage)
  e = t * c2;
  const f32x2 d0 = one + e;
  return v * (d0 * c1);
#endif
  e.x = __builtin_amdgcn_exp2f(t.x);
  e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2 d = one + e;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x);
  r.y = __builtin_amdgcn_rcpf(d.y);
  return v * r;
}
// exact GELU 0.5 x (1 + erf(x / sqrt 2)): tf.nn.gelu's default, used by the legacy tower (nnlib/v1/layers.py:72-79)
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678f)); }
__device__ __forceinline__ float fast_tanh(float v) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853901f * v));
}
__device__ __forceinline__ float jg_act(float v, int act) {
  switch (act) {
    case JG_ACT_GELU_TANH: return fast_gelu(v);
    case JG_ACT_GELU_ERF: return gelu_erf(v);
    case JG_ACT_RELU: return fmaxf(v, 0.0f);
    case JG_ACT_TANH: return fast_tanh(v);
    case JG_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950f * v));
    default: return v;
  }
}

// 16-byte-per-lane global -> LDS DMA: global address = SGPR base + per-lane 32-bit byte offset,
// LDS address = wave-uniform base (M0) + lane*16.  Issued from inline asm on purpose: hipcc
// serialises the builtin form behind vmcnt(0) waits (one per DMA, and again before the first
// ds_read), which forbids any overlap with the matrix cores.  The kernel tracks the DMA queue
// itself with counted s_waitcnt vmcnt(N).
__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
// the same for data that is read once (activation slices): non-temporal, so that the streams do not displace the
// weight slices every workgroup of the XCD keeps re-reading from L2
__device__ __forceinline__ void glds16_nt(const void *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
#ifdef JG_EXP_NO_NT
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
#else
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
#endif
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
// Streamed (read-once / written-once) global accesses of the conv kernels: non-temporal.  -DJG_EXP_NO_NT (experiment build
// only, scripts/r6_mall.sh round 6): the same accesses with the default cache policy, to see whether the nt hint is what
// keeps a small pass's tensors out of the Infinity Cache.
#ifdef JG_EXP_NO_NT
template <typename T> __device__ __forceinline__ void st_stream(T v, T *p) { *p = v; }
template <typename T> __device__ __forceinline__ T ld_stream(const T *p) { return *p; }
#else
template <typename T> __device__ __forceinline__ void st_stream(T v, T *p) { __builtin_nontemporal_store(v, p); }
template <typename T> __device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }
#endif
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Mixed-precision FMAs (v_fma_mix_f32 reads f16 operands out of packed registers, f32 result): the compiler only
// forms them from fma(fpext, fpext, .) and folds a multiplication by one away, so they are written out.
//   mix_sum<H>(hi2, lo2) = f32(hi2.half[H]) + f32(lo2.half[H])      (exact: hi + lo of one split value)
//   mix_rem<H>(v, hi2)   = v - f32(hi2.half[H])                      (the remainder that becomes the lo half)
template <int H>
__device__ __forceinline__ float mix_sum(unsigned hi2, unsigned lo2) {
  float r;
  if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi2), "v"(lo2));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi2), "v"(lo2));
  return r;
}
template <int H>
__device__ __forceinline__ float mix_rem(float v, unsigned hi2) {
  float r;
  if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v));
  return r;
}

struct Tile {
  int rowblk, m0, valid;
  int T;          // tile index: strips (128 positions) 2T and 2T+1 of the launch
};

// v = q * d + r for v < 2^24 (float reciprocal, one correction step)
__device__ __forceinline__ void udivmod24(int v, int d, float inv, int &q, int &r) {
  q = (int)((float)v * inv);
  r = v - q * d;
  if (r < 0) { --q; r += d; }
  else if (r >= d) { ++q; r -= d; }
}

}  // namespace
