// Table net on the matrix cores: the first conv of a two-strand nucleotide model straight from the ids, with its bias,
// activation and global pool - ids in, pooled vectors out (the LDS-table form of the same op: jg_kernels.hip,
// tab_conv_pool_kernel).
//
// A conv over one-hot rows is a GEMM whose left operand holds only 0 and 1: out[co][p] = sum_{t, c} W[t][c][co] . [id[p + t] == c + 1].
// Zeros and ones are exact in f16, so with the weights split into two f16 planes (W = hi + lo to 2^-22) two
// v_mfma_f32_32x32x16_f16 per product are f32-accurate - no third MFMA, no activation planes.  One k-step of 16 covers four
// taps x four nucleotides.  Layout: output channels on the MFMA's N axis - the weight fragments (B operand) sit in registers
// for a workgroup's life, 96 VGPRs for ten taps x 128 channels per wave - positions on M: a lane builds its A operand from
// two id bytes with a shift (one-hot of a = 0x3C00 << 16 (a - 1)).  Four waves own 128 channels each; a row of 491 positions
// is 16 blocks of 32 x 24 MFMAs per wave.  Positions are the M axis, so a lane's 16 accumulator elements are 16 positions of
// ONE channel: the pool over positions is a reduction inside the lane (a running max / sum per tile), one exchange between
// the wave's halves per row, and the lower half writes 32 consecutive channels per tile.  (First form: channels on M - a lane
// held 16 channels of one position, the pool was 64 registers and every row paid a 320-instruction butterfly.)  With a max
// pool and a monotone activation the bias and the activation are applied once per row, behind the max.
#include <algorithm>

#include "jg_common.h"
#include "jg_conv_dev.h"

bool jg_tab_mfma_row_fits(int L_out, int k, int dil);

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

// (the accumulators and the pool never hold a signalling NaN the pool would have to quiet: plain v_max_f32, no canonicalising copy)
__device__ __forceinline__ float vmax(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float vmax3(float x, float y, float z) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

// one block of 32 positions: acc[c][r] = sum over the taps of one-hot(ids) x W (hi + lo) for position (r >> 2) * 8 + 4 hh + (r & 3)
// of the block and channel n of tile c.  Positions are the MFMA's M axis (A operand, built here from two id bytes per k-step:
// lane = position, half hh = which two taps it supplies), channels its N axis (B operand: the weight fragments in registers) - so a
// lane's 16 accumulator elements are 16 POSITIONS of one channel and the pool over positions is a reduction inside the lane.
template <int KS>
__device__ __forceinline__ void tab_block(const half8 (&W)[TM_CT][KS][2], const uint8_t *sp0, int dil, f32x16 (&acc)[TM_CT]) {
#pragma unroll
  for (int c = 0; c < TM_CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const uint8_t *sp = sp0 + 4 * s * dil;
    const unsigned ia = sp[0], ib = sp[dil];
    union { unsigned long long u64[2]; half8 h; } x;
    x.u64[0] = ia ? 0x3C00ull << (16 * (ia - 1)) : 0ull;        // one-hot over A, G, C, T (ids 1..4; 0 = no base)
    x.u64[1] = ib ? 0x3C00ull << (16 * (ib - 1)) : 0ull;
#pragma unroll
    for (int c = 0; c < TM_CT; ++c) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.h, W[c][s][0], acc[c], 0, 0, 0);
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.h, W[c][s][1], acc[c], 0, 0, 0);
    }
  }
}

// a block's accumulators into the running pool (one value per lane and tile); `left` = positions of this block in front of
// L_out (>= 32: all of them)
template <bool LATE, bool MAXP>
__device__ __forceinline__ void tab_fold(float (&pool)[TM_CT], const f32x16 (&acc)[TM_CT], int left, int hh, const float (&bias)[TM_CT],
                                         int act) {
#pragma unroll
  for (int c = 0; c < TM_CT; ++c) {
    float v = pool[c];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float x = acc[c][r], y = acc[c][r + 1];
      if (!LATE) {
        x = tabm_act(x + bias[c], act);
        y = tabm_act(y + bias[c], act);
      }
      if (left < 32) {
        x = (r >> 2) * 8 + hh * 4 + (r & 3) < left ? x : (MAXP ? -INFINITY : 0.f);
        y = (r >> 2) * 8 + hh * 4 + (r & 3) + 1 < left ? y : (MAXP ? -INFINITY : 0.f);
      }
      v = MAXP ? vmax3(v, x, y) : v + x + y;                      // (two positions per instruction: a third of the fold's issue slots)
    }
    pool[c] = v;
  }
}

// MFMAs of one block interleaved with the vector work of the previous block's fold (an MFMA leaves 24 of its 32 cycles to
// the vector pipe: MI355X guide, cycle constants).  Vector instructions per MFMA, measured on the DVF branch (200 000 windows,
// whole call): 3 -> 26.3 ms, 4 -> 25.8 (25.2 with two positions per v_max3_f32), 5 -> 26.2, 6 -> 26.5, 9 -> 26.8; without the
// software pipeline (first form) 52.9
template <int KS>
__device__ __forceinline__ void tab_interleave() {
#pragma unroll
  for (int i = 0; i < 2 * TM_CT * KS; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
  }
}

// KS = k-steps of four taps; LATE = max pool + monotone activation: bias and activation behind the pool; MAXP = max pool
template <int KS, bool LATE, bool MAXP>
__global__ __launch_bounds__(256) void tab_mfma_kernel(JgTabMArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tm_sid[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, hh = lane >> 5;
  // weight fragments of this wave's four channel tiles, and the bias of this lane's channel in each
  half8 W[TM_CT][KS][2];
  float bias[TM_CT];
  const half8 *wf = reinterpret_cast<const half8 *>(a.wfrag);
#pragma unroll
  for (int c = 0; c < TM_CT; ++c) {
    bias[c] = a.bias[(w * TM_CT + c) * 32 + n];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) W[c][s][h] = wf[((((size_t)(w * TM_CT + c) * KS + s) * 2 + h) * 64) + lane];
  }
  const int nb = (a.L_out + 31) >> 5;                             // blocks of 32 positions
  const int n_sid = nb * 32 + 4 * KS * a.dil;                     // entry i = sequence position i - pad_left
  const int left_last = a.L_out - (nb - 1) * 32;
  // the next row's ids travel from HBM while this row is computed: a thread keeps its (up to four) bytes in registers and
  // drops them into the other half of the LDS image behind the row's last MFMA (one barrier per row, no load in its way)
  const int sid_pitch = (n_sid + 15) & ~15;
  auto fetch = [&](int row, uint8_t (&v)[4]) {
    const uint8_t *src = a.ids + (size_t)row * a.L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 256 * q, pos = i - a.pad_left;
      v[q] = row < a.rows && i < n_sid && pos >= 0 && pos < a.L ? src[pos] : (uint8_t)0;
    }
  };
  auto stash = [&](uint8_t *dst, const uint8_t (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tid + 256 * q < n_sid) dst[tid + 256 * q] = v[q];
  };
  uint8_t nxt[4];
  fetch(blockIdx.x, nxt);
  stash(tm_sid, nxt);
  int buf = 0;
  for (int row = blockIdx.x; row < a.rows; row += gridDim.x, buf ^= 1) {
    __syncthreads();                                              // this row's image is complete; the other half is free
    fetch(row + gridDim.x, nxt);
    const uint8_t *sid_row = tm_sid + buf * sid_pitch;
    float pool[TM_CT];
#pragma unroll
    for (int c = 0; c < TM_CT; ++c) pool[c] = MAXP ? -INFINITY : 0.f;
    // software pipeline over the blocks: block b's MFMAs run while block b - 1's accumulators fold into the pool
    const uint8_t *sp = sid_row + n + 2 * hh * a.dil;
    f32x16 accA[TM_CT], accB[TM_CT];
    tab_block<KS>(W, sp, a.dil, accA);
    int b = 1;
    for (; b + 1 < nb; b += 2) {
      tab_block<KS>(W, sp + 32 * b, a.dil, accB);
      tab_fold<LATE, MAXP>(pool, accA, 32, hh, bias, a.act);
      tab_interleave<KS>();
      tab_block<KS>(W, sp + 32 * (b + 1), a.dil, accA);
      tab_fold<LATE, MAXP>(pool, accB, 32, hh, bias, a.act);
      tab_interleave<KS>();
    }
    if (b < nb) {                                                 // (nb even: one more block, then it is the last)
      tab_block<KS>(W, sp + 32 * b, a.dil, accB);
      tab_fold<LATE, MAXP>(pool, accA, 32, hh, bias, a.act);
      tab_interleave<KS>();
      tab_fold<LATE, MAXP>(pool, accB, left_last, hh, bias, a.act);
    } else {
      tab_fold<LATE, MAXP>(pool, accA, left_last, hh, bias, a.act);
    }
    // the two halves of the wave hold the two halves of every block's positions: one exchange per tile, then the lower
    // half writes channel n of each tile - 32 consecutive floats per tile
#pragma unroll
    for (int c = 0; c < TM_CT; ++c) {
      const float o = __shfl_xor(pool[c], 32, 64);
      float v = MAXP ? vmax(pool[c], o) : pool[c] + o;
      const int ch = (w * TM_CT + c) * 32 + n;
      if (LATE) v = tabm_act(v + bias[c], a.act);
      else if (!MAXP) v = v / (float)a.L_out;
      if (hh == 0 && ch < a.cout) a.out[(size_t)row * a.out_ld + ch] = v;
    }
    stash(tm_sid + (buf ^ 1) * sid_pitch, nxt);
  }
}

template <int KS>
int launch_ks(jg_engine *e, const JgTabMArgs &a, bool late, hipStream_t s) {
  const size_t smem = 2 * (size_t)((((a.L_out + 31) & ~31) + 4 * KS * a.dil + 15) & ~15);      // two rows' id images
  JG_REQUIRE(jg_tab_mfma_row_fits(a.L_out, a.k, a.dil), JG_ERR_UNSUPPORTED, "table net: row of %d positions does not fit the id image", a.L_out);
  const int grid = (int)std::min<int64_t>(a.rows, (int64_t)2 * e->n_cu);
  if (late) hipLaunchKernelGGL((tab_mfma_kernel<KS, true, true>), dim3((unsigned)grid), dim3(256), smem, s, a);
  else if (a.pool_kind != JG_POOL_AVG) hipLaunchKernelGGL((tab_mfma_kernel<KS, false, true>), dim3((unsigned)grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL((tab_mfma_kernel<KS, false, false>), dim3((unsigned)grid), dim3(256), smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

}  // namespace

// a row's id image (positions rounded up to blocks of 32 + the taps' reach) must fit the four bytes a thread prefetches
bool jg_tab_mfma_row_fits(int L_out, int k, int dil) { return ((L_out + 31) & ~31) + 4 * ((k + 3) / 4) * dil <= 1024; }

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
