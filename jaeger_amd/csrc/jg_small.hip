// Small-window fused network kernel (BASELINE.json configs[3]: nn_config_500bp_baseline and its relatives).
//
// The 32-channel family - Embedding -> MaskedConv1D(k0, E -> 32) -> BN -> GELU -> residual blocks of k = 3, 32 -> 32
// convolutions -> BN -> GELU -> masked global pool (builder.py:982-1193; layers.py:1128-1332, 1774-1915, 455-538) -
// is 0.1 MFLOP per base: run layer by layer through HBM it is launch- and bandwidth-bound (every 32-channel
// activation tensor makes a round trip).  Here ONE kernel carries a (window, frame) row from codon ids to pooled
// channel sums and nothing but ids (1 B per codon) and 33 floats per row ever touch HBM:
//
//   * one wave owns one row (<= 160 output positions = five 32-position blocks); a workgroup is 4 waves, one per
//     SIMD, up to 512 registers each: the current layer's k = 3 weights are held as MFMA A-fragments (48 VGPRs,
//     refetched from L2 one layer ahead), the accumulators of a whole layer (80) and the residual shortcut (80) too;
//   * first conv = table lookups: conv(embedding(ids)) is linear in one-hot ids, y[p] = sum_t T_t[id[p + t]] with
//     T_t = E . W_t (f64 on the host); the k0 x (vocab + 1) x 32 table sits in LDS (66 KB for k0 = 7);
//   * k = 3 convs on the matrix cores in the split-f16 scheme of jg_conv_f16_impl.h (x = hi + lo, three
//     v_mfma_f32_32x32x16_f16 per product, f32 accumulate): weights are the A operand, so a lane holds one
//     position and 16 channels, the B operand is read from the wave's own LDS row image [position][hi 32ch | lo 32ch]
//     (144-byte rows: conflict-free ds_read_b128) at positions p - 1, p, p + 1;
//   * epilogue in registers: folded bias / batch-norm affine, + shortcut, GELU, second affine + GELU, mask, re-split
//     to hi / lo and written back IN PLACE (a layer's MFMAs are all issued before its first store);
//   * masks are wave-uniform 192-bit words (ballot of ids != 0; "any" rule = shifted ORs), the last layer's epilogue
//     accumulates the masked pool instead of storing.
//
// Waves never exchange data, so there is no barrier after the tables are loaded.
#include <algorithm>
#include <vector>

#include "jg_common.h"
#include "jg_small.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// keep the unrolled blocks in program order: a compiler-only fence for the IR passes (no hoisting of a later block's
// LDS reads; an inline asm here would stop the full unroll and push the accumulator arrays into scratch) plus a
// scheduling barrier for the machine scheduler - otherwise every block's loads are issued up front and spill
#define JG_FENCE()                              \
  do {                                          \
    __atomic_signal_fence(__ATOMIC_SEQ_CST);    \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)

namespace {

constexpr int C = 32;                  // channels of the family
constexpr int NB = 5;                  // 32-position blocks per row
constexpr int POS = NB * 32;           // 160 positions
constexpr int ROWB = 144;              // bytes per position in the row image: 64 hi + 64 lo + 16 pad
constexpr int LUTS = 36;               // floats per table row (32 + 4: random rows spread over 16 bank classes)
constexpr int ACT_BYTES = (POS + 2) * ROWB;
constexpr int IDS_BYTES = 224;         // 8 margin + 192 ids + margin
constexpr int WAVE_BYTES = ACT_BYTES + IDS_BYTES;
constexpr int IDM = 8;                 // left margin of the id buffer
constexpr int PARTW = JG_SMALL_PARTW;  // floats per row in the partial-pool buffer: 32 channels + count

struct M192 {
  unsigned long long w[3];
};
__device__ __forceinline__ M192 m_or(M192 a, M192 b) { return M192{{a.w[0] | b.w[0], a.w[1] | b.w[1], a.w[2] | b.w[2]}}; }
__device__ __forceinline__ M192 m_and(M192 a, M192 b) { return M192{{a.w[0] & b.w[0], a.w[1] & b.w[1], a.w[2] & b.w[2]}}; }
// bit p of the result = bit p + s of m (s in 0..63)
__device__ __forceinline__ M192 m_shr(M192 m, int s) {
  if (s == 0) return m;
  return M192{{(m.w[0] >> s) | (m.w[1] << (64 - s)), (m.w[1] >> s) | (m.w[2] << (64 - s)), m.w[2] >> s}};
}
// bit p of the result = bit p - s of m
__device__ __forceinline__ M192 m_shl(M192 m, int s) {
  if (s == 0) return m;
  return M192{{m.w[0] << s, (m.w[1] << s) | (m.w[0] >> (64 - s)), (m.w[2] << s) | (m.w[1] >> (64 - s))}};
}
__device__ __forceinline__ M192 m_first(int n) {     // bits [0, n)
  M192 r;
  for (int i = 0; i < 3; ++i) {
    const int k = n - 64 * i;
    r.w[i] = k >= 64 ? ~0ull : (k <= 0 ? 0ull : ((1ull << k) - 1ull));
  }
  return r;
}
__device__ __forceinline__ int m_count(M192 m) { return __popcll(m.w[0]) + __popcll(m.w[1]) + __popcll(m.w[2]); }

__device__ __forceinline__ float gelu_tanh(float v) {
  const float t = v * (-2.3022082f - 0.10294324f * v * v);   // -2u * log2(e), u = sqrt(2/pi)(v + 0.044715 v^3)
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// One layer's epilogue over the five accumulator blocks of a row.  Lane (n, h) holds position 32 b + n and channels
// 8 g + 4 h + i in register 4 g + i of block b.  Compiled per pattern (the activation is always the tanh-GELU):
//   P2 = false:  x = gelu(acc * s1 + t1 + addf * shortcut)              (addf = 0 or 1: exact either way)
//   P2 = true :  x = gelu(gelu(...) * s2 + t2)                          (the norm + GELU behind a residual stack)
// LAST: accumulate the masked pool instead of storing the row.
template <bool LAST, bool P2>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[NB], f32x16 (&sc)[NB], const float addf, const int save,
                                         const float *epi, M192 mout, char *act, int n, int h, float &vmax,
                                         float (&pool)[16], int pool_kind) {
  // (the per-channel parameters are re-read from LDS - broadcast reads - for every block: holding them would cost
  // 64 registers next to the accumulators and the shortcut)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const bool keep = (mout.w[b >> 1] >> ((b & 1) * 32 + n)) & 1ull;
    float v[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 s1 = *reinterpret_cast<const f32x4 *>(epi + 0 * C + g * 8 + h * 4);
      const f32x4 t1 = *reinterpret_cast<const f32x4 *>(epi + 1 * C + g * 8 + h * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = gelu_tanh(acc[b][g * 4 + i] * s1[i] + t1[i] + addf * sc[b][g * 4 + i]);
        if constexpr (P2) {
          const float s2 = epi[2 * C + g * 8 + h * 4 + i], t2 = epi[3 * C + g * 8 + h * 4 + i];
          x = gelu_tanh(x * s2 + t2);
        }
        v[g * 4 + i] = x;
      }
    }
    if (save) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[b][r] = v[r];
    }
    if constexpr (LAST) {
      if (pool_kind == JG_POOL_AVG) {
#pragma unroll
        for (int r = 0; r < 16; ++r) pool[r] += keep ? v[r] : 0.0f;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) pool[r] = fmaxf(pool[r], keep ? v[r] : -1.0e9f);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x = keep ? v[g * 4 + i] : 0.0f;
          vmax = fmaxf(vmax, fabsf(x));
          const _Float16 hh = (_Float16)x;
          hi[i] = hh;
          lo[i] = (_Float16)(x - (float)hh);
        }
        char *p = act + (1 + b * 32 + n) * ROWB + (g * 8 + h * 4) * 2;
        *reinterpret_cast<half4 *>(p) = hi;
        *reinterpret_cast<half4 *>(p + 64) = lo;
      }
    }
    JG_FENCE();      // one block at a time: hoisted loads of later blocks would spill
  }
}

template <int NC, int K0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void small_net_kernel(JgSmallArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int VR = a.vocab + 1;
  float *lut = reinterpret_cast<float *>(smem);
  float *epi = lut + K0 * VR * LUTS;
  char *wbase = reinterpret_cast<char *>(epi + (NC + 1) * 4 * C) + wave * WAVE_BYTES;
  char *act = wbase;
  unsigned char *idbuf = reinterpret_cast<unsigned char *>(wbase + ACT_BYTES);

  // tables -> LDS (once per workgroup)
  for (int i = threadIdx.x; i < K0 * VR * 8; i += 256) {          // 8 float4 per table row
    const int row = i >> 3, q = i & 7;
    *reinterpret_cast<f32x4 *>(lut + row * LUTS + q * 4) = *reinterpret_cast<const f32x4 *>(a.lut + row * C + q * 4);
  }
  for (int i = threadIdx.x; i < (NC + 1) * 4 * C; i += 256) epi[i] = a.epi[i];
  // zero halo rows of the row image (positions -1 and 160) and the id margins
  for (int i = lane; i < ROWB / 4; i += 64) {
    reinterpret_cast<unsigned *>(act)[i] = 0u;
    reinterpret_cast<unsigned *>(act + (POS + 1) * ROWB)[i] = 0u;
  }
  for (int i = lane; i < IDS_BYTES; i += 64) idbuf[i] = (unsigned char)a.vocab;
  // k = 3 weights as MFMA A-fragments: the CURRENT layer's twelve (48 registers), fetched from L2 one layer ahead
  // - right after the previous layer's last MFMA, so the loads land under that layer's epilogue
  const half8 *wsrc = reinterpret_cast<const half8 *>(a.wfrag) + lane;
  half8 w[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int p = 0; p < 2; ++p) w[t][c][p] = wsrc[((t * 2 + c) * 2 + p) << 6];
  __syncthreads();

  const int L = a.L, L0 = a.L0, pad0 = a.pad0;
  const M192 valid_in = m_first(L), valid0 = m_first(L0);
  float vmax = 0.0f;
  for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += (long)gridDim.x * 4) {
    // ---- ids of the row -> LDS, input mask by ballot --------------------------------------------------
    M192 m;
    const unsigned char *src = a.ids + row * L;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = j * 64 + lane;
      const int id = q < L ? (int)src[q] : 0;
      m.w[j] = __ballot(a.use_mask ? id != 0 : q < L);
      idbuf[IDM + q] = q < L ? (unsigned char)id : (unsigned char)a.vocab;
    }
    m = m_and(m, valid_in);
    // ---- first conv: k0 table rows per output position --------------------------------------------------
    f32x16 acc[NB], sc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      acc[b] = z;
      sc[b] = z;
    }
    // (a real loop over the taps: fully unrolled, the compiler issues every table read of the row first and adds
    // last - 560 live registers; one tap of the five blocks per iteration keeps 80 in flight)
#pragma unroll 1
    for (int t = 0; t < K0; ++t) {
      const float *lt = lut + t * VR * LUTS + h * 4;
      const unsigned char *it = idbuf + IDM + n + t - pad0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float *rp = lt + (int)it[b * 32] * LUTS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(rp + g * 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[b][g * 4 + i] += v[i];
        }
      }
    }
    // output mask of the first conv ("any" rule over its k0 taps), positions [0, L0)
    M192 mo = m;
    if (a.use_mask) {
      mo = M192{{0, 0, 0}};
#pragma unroll
      for (int t = 0; t < K0; ++t) mo = m_or(mo, t >= pad0 ? m_shr(m, t - pad0) : m_shl(m, pad0 - t));
    }
    mo = m_and(mo, valid0);
    float pool[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pool[i] = a.pool_kind == JG_POOL_AVG ? 0.0f : -1.0e9f;
    epilogue<false, false>(acc, sc, 0.0f, a.layer[0].save, epi, mo, act, n, h, vmax, pool, a.pool_kind);
    // ---- k = 3 convolutions on the matrix cores -----------------------------------------------------------
#pragma unroll 1
    for (int j = 0; j < NC; ++j) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const char *p = act + (b * 32 + n + t) * ROWB + (cc * 16 + h * 8) * 2;
            const half8 xh = *reinterpret_cast<const half8 *>(p);
            const half8 xl = *reinterpret_cast<const half8 *>(p + 64);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t][cc][0], xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t][cc][0], xl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t][cc][1], xh, c, 0, 0, 0);
          }
        acc[b] = c;
        JG_FENCE();  // one block's twelve fragments in flight at a time
      }
      {                                     // next layer's weights (layer 0 of the next row after the last one)
        const half8 *wn = wsrc + ((j + 1 == NC ? 0 : j + 1) * 12 << 6);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int p = 0; p < 2; ++p) w[t][cc][p] = wn[((t * 2 + cc) * 2 + p) << 6];
      }
      if (a.use_mask) mo = m_and(m_or(m_or(m_shl(mo, 1), mo), m_shr(mo, 1)), valid0);
      const float *ep = epi + (j + 1) * 4 * C;
      const float addf = a.layer[j + 1].add ? 1.0f : 0.0f;
      const int save = a.layer[j + 1].save, p2 = a.layer[j + 1].aff2;
      if (j == NC - 1) {
        if (p2) epilogue<true, true>(acc, sc, addf, save, ep, mo, act, n, h, vmax, pool, a.pool_kind);
        else epilogue<true, false>(acc, sc, addf, save, ep, mo, act, n, h, vmax, pool, a.pool_kind);
      } else {
        if (p2) epilogue<false, true>(acc, sc, addf, save, ep, mo, act, n, h, vmax, pool, a.pool_kind);
        else epilogue<false, false>(acc, sc, addf, save, ep, mo, act, n, h, vmax, pool, a.pool_kind);
      }
    }
    // ---- pooled channel sums / maxima of the row: reduce over the 32 lanes that share h --------------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = pool[i];
#pragma unroll
      for (int s = 1; s < 32; s <<= 1) {
        const float o = __shfl_xor(v, s, 64);
        v = a.pool_kind == JG_POOL_AVG ? v + o : fmaxf(v, o);
      }
      pool[i] = v;
    }
    if (n == 0) {
      float *dst = a.part + row * PARTW;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4 *>(dst + g * 8 + h * 4) = f32x4{pool[g * 4], pool[g * 4 + 1], pool[g * 4 + 2], pool[g * 4 + 3]};
      if (h == 0) dst[C] = (float)m_count(mo);
    }
  }
  if (__any(!(vmax <= 65000.0f)) && lane == 0) atomicOr(a.overflow, 1);
}

// pooled vector of a window from the partial rows of its frames: masked average (layers.py:460-480) or masked
// maximum with the all-masked -> 0 rule (layers.py:517-529)
__global__ void small_pool_final_kernel(const float *part, int frames, int n_win, int kind, float *out, int out_ld) {
  const int w = blockIdx.x * (blockDim.x / C) + threadIdx.x / C, c = threadIdx.x % C;
  if (w >= n_win) return;
  float acc = kind == JG_POOL_AVG ? 0.0f : -1.0e9f, cnt = 0.0f;
  for (int f = 0; f < frames; ++f) {
    const float *p = part + ((long)w * frames + f) * PARTW;
    acc = kind == JG_POOL_AVG ? acc + p[c] : fmaxf(acc, p[c]);
    cnt += p[C];
  }
  out[(long)w * out_ld + c] = kind == JG_POOL_AVG ? acc / fmaxf(cnt, 1e-7f) : (cnt > 0.0f ? acc : 0.0f);
}

template <int NC, int K0>
int launch(jg_engine *e, const JgSmallArgs &a, int smem, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_net_kernel<NC, K0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int grid = (int)std::min<long>(e->n_cu, (a.rows + 3) / 4);
  hipLaunchKernelGGL((small_net_kernel<NC, K0>), dim3((unsigned)grid), dim3(256), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

}  // namespace

int jg_small_lds_bytes(int n_conv, int k0, int vocab) {
  return k0 * (vocab + 1) * LUTS * 4 + (n_conv + 1) * 4 * C * 4 + 4 * WAVE_BYTES;
}

bool jg_small_supports(int n_conv, int k0, int vocab) {
  return (n_conv == 2 || n_conv == 4) && (k0 == 3 || k0 == 5 || k0 == 7 || k0 == 9) && vocab <= 254 &&
         jg_small_lds_bytes(n_conv, k0, vocab) <= 160 * 1024;
}

int jg_small_max_positions(void) { return POS; }

int jg_launch_small_net(jg_engine *e, const JgSmallArgs &a, int n_conv, int k0, hipStream_t s) {
  JG_REQUIRE(jg_small_supports(n_conv, k0, a.vocab) && a.L0 >= 1 && a.L0 <= POS && a.L <= 192 && a.pad0 <= IDM,
             JG_ERR_UNSUPPORTED, "small-window kernel: %d convs, k0 = %d, %d output positions are outside its limits", n_conv,
             k0, a.L0);
  if (a.rows == 0) return JG_OK;
  const int smem = jg_small_lds_bytes(n_conv, k0, a.vocab);
#define JG_SMALL_CASE(NC, K0) \
  if (n_conv == NC && k0 == K0) return launch<NC, K0>(e, a, smem, s);
  JG_SMALL_CASE(4, 7) JG_SMALL_CASE(2, 7) JG_SMALL_CASE(4, 5) JG_SMALL_CASE(2, 5) JG_SMALL_CASE(4, 3) JG_SMALL_CASE(2, 3)
  JG_SMALL_CASE(4, 9) JG_SMALL_CASE(2, 9)
#undef JG_SMALL_CASE
  jg_set_error("small-window kernel: no instantiation for %d convs, k0 = %d", n_conv, k0);
  return JG_ERR_UNSUPPORTED;
}

int jg_launch_small_pool_final(const float *part, int frames, int n_win, int kind, float *out, int out_ld, hipStream_t s) {
  if (n_win == 0) return JG_OK;
  const int per = 256 / C;
  hipLaunchKernelGGL(small_pool_final_kernel, dim3((unsigned)((n_win + per - 1) / per)), dim3(256), 0, s, part, frames,
                     n_win, kind, out, out_ld);
  JG_HIP(hipGetLastError());
  return JG_OK;
}
