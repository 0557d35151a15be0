// Table net on the matrix cores: the first conv of a two-strand nucleotide model straight from the ids, with its bias,
// activation and global pool - ids in, pooled vectors out (the LDS-table form of the same op: jg_kernels.hip,
// tab_conv_pool_kernel).
//
// A conv over one-hot rows is a GEMM whose left operand holds only 0 and 1: out[co][p] = sum_{t, c} W[t][c][co] . [id[p + t] == c + 1].
// Zeros and ones are exact in f16, so with the weights split into two f16 planes (W = hi + lo to 2^-22) two
// v_mfma_f32_32x32x16_f16 per product are f32-accurate - no third MFMA, no activation planes.  One k-step of 16 covers four
// taps x four nucleotides.  Layout: output channels on the MFMA's M axis - the weight fragments (A operand) sit in registers
// for a workgroup's life, 96 VGPRs for ten taps x 128 channels per wave - positions on N: a lane builds its B operand from
// two id bytes with a shift (one-hot of a = 0x3C00 << 16 (a - 1)).  Four waves own 128 channels each; a row of 491 positions
// is 16 blocks of 32 x 24 MFMAs per wave.  The pool runs in the accumulator layout (a lane holds 16 channels of ONE position):
// per block an elementwise max / sum, per row one butterfly over the 32 lanes of a half-wave.  With a max pool and a monotone
// activation the bias and the activation are applied once per row, behind the max.
#include <algorithm>

#include "jg_common.h"
#include "jg_conv_dev.h"

namespace {

constexpr int TM_CT = 4;       // 32-channel tiles per wave (4 waves: up to 512 channels)

__device__ __forceinline__ float tabm_act(float v, int act) {
  switch (act) {
    case JG_ACT_GELU_TANH: {
      const float t = v * (-2.3022082f - 0.10294324f * v * v);
      return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
    }
    case JG_ACT_GELU_ERF: return 0.5f * v * erfcf(-v * 0.70710678118654752f);
    case JG_ACT_RELU: return fmaxf(v, 0.0f);
    case JG_ACT_TANH: return tanhf(v);
    case JG_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950f * v));
    default: return v;
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// (the accumulators and the pool never hold a signalling NaN the pool would have to quiet: plain v_max_f32, no canonicalising copy)
__device__ __forceinline__ float vmax(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// one block of 32 positions: acc[c] = sum over the taps of W (hi + lo) x one-hot(ids); lane = position p + n, half hh = which two
// taps of a k-step it supplies
template <int KS>
__device__ __forceinline__ void tab_block(const half8 (&A)[TM_CT][KS][2], const uint8_t *sp0, int dil, f32x16 (&acc)[TM_CT]) {
#pragma unroll
  for (int c = 0; c < TM_CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const uint8_t *sp = sp0 + 4 * s * dil;
    const unsigned ia = sp[0], ib = sp[dil];
    union { unsigned long long u64[2]; half8 h; } b;
    b.u64[0] = ia ? 0x3C00ull << (16 * (ia - 1)) : 0ull;        // one-hot over A, G, C, T (ids 1..4; 0 = no base)
    b.u64[1] = ib ? 0x3C00ull << (16 * (ib - 1)) : 0ull;
#pragma unroll
    for (int c = 0; c < TM_CT; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][s][0], b.h, acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][s][1], b.h, acc[c], 0, 0, 0);
    }
  }
}

// a block's accumulators into the running pool; `partial`: positions behind L_out exist in this block (lane flag `valid`)
template <bool LATE, bool MAXP>
__device__ __forceinline__ void tab_fold(f32x16 (&pool)[TM_CT], const f32x16 (&acc)[TM_CT], bool partial, bool valid,
                                         const float *bias_s, int act, int ch0) {
#pragma unroll
  for (int c = 0; c < TM_CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = acc[c][r];
      if (!LATE) x = tabm_act(x + bias_s[ch0 + c * 32 + (r >> 2) * 8 + (r & 3)], act);
      if (partial) x = valid ? x : (MAXP ? -INFINITY : 0.f);
      pool[c][r] = MAXP ? vmax(pool[c][r], x) : pool[c][r] + x;
    }
}

// MFMAs of one block interleaved with the vector work of the previous block's fold (an MFMA leaves 24 of its 32 cycles to
// the vector pipe: MI355X guide, cycle constants).  Vector instructions per MFMA, measured on the DVF branch (200 000 windows,
// whole call): 4 -> 33.7 ms, 6 -> 34.0, 9 -> 31.0, 12 -> 31.4, 16 -> 31.5; without the software pipeline 52.9
template <int KS>
__device__ __forceinline__ void tab_interleave() {
#pragma unroll
  for (int i = 0; i < 2 * TM_CT * KS; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
  }
}

// KS = k-steps of four taps; LATE = max pool + monotone activation: bias and activation behind the pool; MAXP = max pool
template <int KS, bool LATE, bool MAXP>
__global__ __launch_bounds__(256) void tab_mfma_kernel(JgTabMArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tm_smem[];
  float *bias_s = reinterpret_cast<float *>(tm_smem);              // 512 floats
  float *out_s = bias_s + 128 * TM_CT;                             // the row's pooled channels, before bias / activation
  uint8_t *tm_sid = reinterpret_cast<uint8_t *>(out_s + 128 * TM_CT);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, hh = lane >> 5;
  // weight fragments of this wave's four channel tiles
  half8 A[TM_CT][KS][2];
  const half8 *wf = reinterpret_cast<const half8 *>(a.wfrag);
#pragma unroll
  for (int c = 0; c < TM_CT; ++c)
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) A[c][s][h] = wf[((((size_t)(w * TM_CT + c) * KS + s) * 2 + h) * 64) + lane];
  for (int i = tid; i < 128 * TM_CT; i += 256) bias_s[i] = a.bias[i];
  const int nb = (a.L_out + 31) >> 5;                             // blocks of 32 positions
  const int n_sid = nb * 32 + 4 * KS * a.dil;                     // entry i = sequence position i - pad_left
  const int ch0 = w * TM_CT * 32 + hh * 4;
  const bool tail = (a.L_out & 31) != 0, valid_last = ((nb - 1) * 32 + n) < a.L_out;
  for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
    __syncthreads();
    const uint8_t *src = a.ids + (size_t)row * a.L;
    for (int i = tid; i < n_sid; i += 256) {
      const int pos = i - a.pad_left;
      tm_sid[i] = pos >= 0 && pos < a.L ? src[pos] : (uint8_t)0;
    }
    __syncthreads();
    f32x16 pool[TM_CT];
#pragma unroll
    for (int c = 0; c < TM_CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) pool[c][r] = MAXP ? -INFINITY : 0.f;
    // software pipeline over the blocks: block b's MFMAs run while block b - 1's accumulators fold into the pool
    const uint8_t *sp = tm_sid + n + 2 * hh * a.dil;
    f32x16 accA[TM_CT], accB[TM_CT];
    tab_block<KS>(A, sp, a.dil, accA);
    int b = 1;
    for (; b + 1 < nb; b += 2) {
      tab_block<KS>(A, sp + 32 * b, a.dil, accB);
      tab_fold<LATE, MAXP>(pool, accA, false, true, bias_s, a.act, ch0);
      tab_interleave<KS>();
      tab_block<KS>(A, sp + 32 * (b + 1), a.dil, accA);
      tab_fold<LATE, MAXP>(pool, accB, false, true, bias_s, a.act, ch0);
      tab_interleave<KS>();
    }
    if (b < nb) {                                                 // (nb even: one more block, then it is the last)
      tab_block<KS>(A, sp + 32 * b, a.dil, accB);
      tab_fold<LATE, MAXP>(pool, accA, false, true, bias_s, a.act, ch0);
      tab_interleave<KS>();
      tab_fold<LATE, MAXP>(pool, accB, tail, valid_last, bias_s, a.act, ch0);
    } else {
      tab_fold<LATE, MAXP>(pool, accA, tail, valid_last, bias_s, a.act, ch0);
    }
    // positions sit on the 32 lanes of a half-wave: quads, half rows and rows by DPP, the two rows of a half by one shuffle;
    // then every lane holds the row's pooled channels of its half
#pragma unroll
    for (int c = 0; c < TM_CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = pool[c][r], o;
        o = dpp_f32<0xB1>(v); v = MAXP ? vmax(v, o) : v + o;        // quad_perm [1, 0, 3, 2]
        o = dpp_f32<0x4E>(v); v = MAXP ? vmax(v, o) : v + o;        // quad_perm [2, 3, 0, 1]
        o = dpp_f32<0x141>(v); v = MAXP ? vmax(v, o) : v + o;       // row_half_mirror: the other quad of the 8
        o = dpp_f32<0x140>(v); v = MAXP ? vmax(v, o) : v + o;       // row_mirror: the other 8 of the 16
        o = __shfl_xor(v, 16, 64); v = MAXP ? vmax(v, o) : v + o;   // the other row of the half-wave
        pool[c][r] = v;
      }
    // (the two lanes that hold a tile's channels hand them over through LDS: the row's bias / activation / store then is one
    // coalesced pass of the workgroup - sixty-four dependent global loads and scattered stores per lane cost 4 x the MFMAs)
    if (n == 0) {
#pragma unroll
      for (int c = 0; c < TM_CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) out_s[ch0 + c * 32 + (r >> 2) * 8 + (r & 3)] = pool[c][r];
    }
    __syncthreads();
    for (int ch = tid; ch < a.cout; ch += 256) {
      float v = out_s[ch];
      if (LATE) v = tabm_act(v + bias_s[ch], a.act);
      else if (!MAXP) v = v / (float)a.L_out;
      a.out[(size_t)row * a.out_ld + ch] = v;
    }
  }
}

template <int KS>
int launch_ks(jg_engine *e, const JgTabMArgs &a, bool late, hipStream_t s) {
  const size_t smem = (size_t)2 * 128 * TM_CT * sizeof(float) + (size_t)((((a.L_out + 31) & ~31) + 4 * KS * a.dil + 15) & ~15);
  JG_REQUIRE(smem <= 64 * 1024, JG_ERR_UNSUPPORTED, "table net: row of %d positions does not fit the id image", a.L_out);
  const int grid = (int)std::min<int64_t>(a.rows, (int64_t)2 * e->n_cu);
  if (late) hipLaunchKernelGGL((tab_mfma_kernel<KS, true, true>), dim3((unsigned)grid), dim3(256), smem, s, a);
  else if (a.pool_kind != JG_POOL_AVG) hipLaunchKernelGGL((tab_mfma_kernel<KS, false, true>), dim3((unsigned)grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL((tab_mfma_kernel<KS, false, false>), dim3((unsigned)grid), dim3(256), smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

}  // namespace

bool jg_tab_mfma_supports(int k, int vocab, int cout, int dil) {
  return vocab == 5 && k >= 1 && k <= 16 && cout >= 1 && cout <= 128 * TM_CT && dil >= 1 && dil <= 64;
}

// weight fragments in the MFMA A-operand layout: [tile of 32 channels][k-step][hi | lo][lane][8 halves], lane = channel
// (lane & 31), k = 8 (lane >> 5) + j = tap 4 s + k / 4, nucleotide k % 4; `table` = (k, 5, cq) float4 quads as the LDS kernel takes
int64_t jg_tab_mfma_frag_halves(int k) { return (int64_t)(4 * TM_CT) * ((k + 3) / 4) * 2 * 64 * 8; }

int jg_launch_tab_mfma(jg_engine *e, const JgTabMArgs &a, hipStream_t s) {
  if (a.rows == 0) return JG_OK;
  JG_REQUIRE(jg_tab_mfma_supports(a.k, 5, a.cout, a.dil) && a.L_out >= 1, JG_ERR_UNSUPPORTED, "table net (matrix cores): k=%d cout=%d",
             a.k, a.cout);
  // bias + activation behind the pool: only where the activation is monotone and the pool a max
  const bool late = a.pool_kind != JG_POOL_AVG &&
                    (a.act == JG_ACT_NONE || a.act == JG_ACT_RELU || a.act == JG_ACT_TANH || a.act == JG_ACT_SIGMOID);
  switch ((a.k + 3) / 4) {
    case 1: return launch_ks<1>(e, a, late, s);
    case 2: return launch_ks<2>(e, a, late, s);
    case 3: return launch_ks<3>(e, a, late, s);
    default: return launch_ks<4>(e, a, late, s);
  }
}
