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
//
// (A variant with TWO waves per row image - waves w and w + 4 on one SIMD, 80 positions each on 16x16x32 MFMAs, two
// workgroup barriers per layer - was built and measured in round 2: bit-identical results, 4 % slower, and the same
// cycles per element in every vector-bound phase with half the elements per wave.  The epilogue is bound by the SIMD's
// vector THROUGHPUT - about 4 cycles per wave64 instruction and 16 per exp2 / rcp - not by a lone wave's issue rate,
// so a second wave per SIMD buys nothing here; the variant was dropped, DESIGN.md section 3.2 has its cycle stamps.)
#include <stdio.h>

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

// packed f16 pair of two floats (v_cvt_pk_f16_f32, round to nearest even)
__device__ __forceinline__ unsigned pk_f16(float a, float b) {
  typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
  const half2_t r = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, r);
}
// v - f32(hi2.half[H]) in one mixed-precision FMA (as in jg_conv_f16_impl.h)
template <int H>
__device__ __forceinline__ float mix_rem(float v, unsigned hi2) {
  float r;
  if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hi2), "v"(v));
  return r;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One layer's epilogue over the five accumulator blocks of a row.  Lane (n, h) holds position 32 b + n and channels
// 8 g + 4 h + i in register 4 g + i of block b.  Compiled per pattern (the activation is always the tanh-GELU):
//   x = gelu(acc * s1 + t1 [+ shortcut: ADD]);  P2: x = gelu(x * s2 + t2) (the norm + GELU behind a residual stack);
//   SAVE: the result becomes the shortcut of a later layer;  LAST: accumulate the masked pool instead of storing the row.
// (Run-time flags instead of ADD / SAVE cost 1.5 more vector instructions per element: a "+ 0 * shortcut" and a select.)
#ifdef JG_SMALL_ABLATE                   /* (make exp EXPFLAGS=-DJG_SMALL_ABLATE: the branches perturb the code they sit in) */
#define JG_SDBG(bit) (dbg & (bit))      /* JG_SMALL_DBG: 1 no table phase, 2 no MFMA, 4 GELU = identity, 8 no LDS stores */
#else
#define JG_SDBG(bit) false
#endif

// Since round 4 a layer's first affine (bias / batch norm in front of the GELU) is not an epilogue stage any more: its
// scale is folded into the layer's weights (f64 on the host, then split into hi / lo - gfx950's MFMA honours f16
// subnormal inputs, scripts/ubench/mfma_denorm.hip, so the weights need no power-of-two pre-scale whose undoing would cost
// the instruction the fold saves) and its shift is the INITIAL VALUE of the accumulators (the MFMA's C operand): sixteen registers per layer,
// read from LDS a layer ahead, copied where the block used to be cleared (reading LDS right in front of a block's first MFMA
// measured 0.9 % slower than the unfolded kernel: a lone wave waits out the LDS latency).  The first layer's table carries scale and
// shift in its rows.  Left per layer: the second affine of a stack end (P2), read once per layer (broadcast LDS reads).
template <bool P2>
struct EpiParams {
  f32x4 s2[P2 ? 4 : 1], t2[P2 ? 4 : 1];
  __device__ __forceinline__ void load(const float *epi, int h) {
    if constexpr (P2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        s2[g] = *reinterpret_cast<const f32x4 *>(epi + 2 * C + g * 8 + h * 4);
        t2[g] = *reinterpret_cast<const f32x4 *>(epi + 3 * C + g * 8 + h * 4);
      }
    }
  }
};
// accumulators of a block start from the layer's shift (t1 row of its epilogue table: channel 8 g + 4 h + i in register 4 g + i)
__device__ __forceinline__ void acc_init(f32x16 &c, const float *epi, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 t = *reinterpret_cast<const f32x4 *>(epi + 1 * C + g * 8 + h * 4);
    c[4 * g + 0] = t[0]; c[4 * g + 1] = t[1]; c[4 * g + 2] = t[2]; c[4 * g + 3] = t[3];
  }
}

// A stage boundary: nothing is scheduled across it.  The epilogue below is written stage by stage - the same
// instruction for the eight channel pairs of a block, then the next instruction for all of them - because a lone wave per
// SIMD pays a full issue slot (4 cycles) for every s_nop, and the compiler separates a packed-f32 or transcendental
// instruction from a consumer right behind it by one (scripts/ubench/valu_rates.hip); left to its own scheduling it
// walks the pairs one after the other under this kernel's register pressure, a nop behind every instruction.
#define JG_STAGE() __builtin_amdgcn_sched_barrier(0)

// tanh-GELU of the eight channel pairs of a block, stage by stage (the arithmetic of gelu_tanh2)
__device__ __forceinline__ void gelu_stages(f32x2 (&x)[8]) {
  const f32x2 c0 = {-2.3022082f, -2.3022082f}, c1 = {-0.10294324f, -0.10294324f}, one = {1.0f, 1.0f};
  f32x2 t[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) t[p] = x[p] * x[p];
  JG_STAGE();
#pragma unroll
  for (int p = 0; p < 8; ++p) t[p] = c0 + c1 * t[p];
  JG_STAGE();
#pragma unroll
  for (int p = 0; p < 8; ++p) t[p] = x[p] * t[p];
  JG_STAGE();
#pragma unroll
  for (int p = 0; p < 8; ++p) t[p] = f32x2{__builtin_amdgcn_exp2f(t[p][0]), __builtin_amdgcn_exp2f(t[p][1])};
  JG_STAGE();
#pragma unroll
  for (int p = 0; p < 8; ++p) t[p] = t[p] + one;
  JG_STAGE();
#pragma unroll
  for (int p = 0; p < 8; ++p) t[p] = f32x2{__builtin_amdgcn_rcpf(t[p][0]), __builtin_amdgcn_rcpf(t[p][1])};
  JG_STAGE();
#pragma unroll
  for (int p = 0; p < 8; ++p) x[p] = x[p] * t[p];
  JG_STAGE();
}

// epilogue of ONE 32-position block in two steps.  epi_math: accumulators c (+ shortcut block scb) -> the layer's output
// values v (packed-f32 arithmetic and transcendentals);  epi_out: v -> masked hi / lo halves into the row image, or the
// pool / tap sums - plain vector instructions only, cut into chunks of 4-8: those issue for free beside a running MFMA
// (scripts/ubench/valu_rates.hip: five per MFMA; packed-f32 instructions wait for the MFMA to finish)
template <bool P2, bool ADD, bool SAVE>
__device__ __forceinline__ void epi_math(const f32x16 &c, f32x16 &scb, const EpiParams<P2> &q, int dbg, float (&v)[16]) {
  (void)dbg;
  f32x2 x[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) x[p] = f32x2{c[2 * p], c[2 * p + 1]};       // (scale in the weights, shift in the C operand)
  if constexpr (ADD) {
#pragma unroll
    for (int p = 0; p < 8; ++p) x[p] = x[p] + f32x2{scb[2 * p], scb[2 * p + 1]};
    JG_STAGE();
  }
  if (!JG_SDBG(4)) gelu_stages(x);
  if constexpr (P2) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int g = p >> 1, i = (p & 1) * 2;
      x[p] = x[p] * f32x2{q.s2[g][i], q.s2[g][i + 1]} + f32x2{q.t2[g][i], q.t2[g][i + 1]};
    }
    JG_STAGE();
    if (!JG_SDBG(4)) gelu_stages(x);
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    v[2 * p] = x[p][0];
    v[2 * p + 1] = x[p][1];
  }
  if constexpr (SAVE) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scb[r] = v[r];
    JG_STAGE();
  }
}

// the output step of a block in chunks.  Row-image layers: per channel group g the chunks  A: range guard + hi halves,
// B: the four remainders, C: lo halves + mask + the two stores;  tap layers add  T: masked tap sums;  the last layer has
// the pool sums only (P).
template <bool LAST, bool TAP>
constexpr int out_chunks() { return LAST ? (TAP ? 8 : 4) : (TAP ? 16 : 12); }
struct OutRegs {
  unsigned h01, h23;
  float r0, r1, r2, r3;
};
template <bool LAST, bool TAP, bool PMAX>
__device__ __forceinline__ void epi_out_chunk(const int k, const float (&v)[16], OutRegs (&o)[4], const int b, const bool keep,
                                              char *act, int n, int h, float &vmax, float (&pool)[16], float (&tapv)[16],
                                              int dbg) {
  (void)dbg;
  constexpr int PER = LAST ? (TAP ? 2 : 1) : (TAP ? 4 : 3);
  const int g = k / PER, kind = k % PER;            // kind: row image 0 A, 1 B, 2 C, 3 T;  last layer 0 P, 1 T
  const float *x = &v[g * 4];
  if (kind == (LAST ? 1 : 3)) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tapv[g * 4 + r] += keep ? x[r] : 0.0f;
    return;
  }
  if constexpr (LAST) {
    if constexpr (!PMAX) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pool[g * 4 + r] += keep ? x[r] : 0.0f;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) pool[g * 4 + r] = fmaxf(pool[g * 4 + r], keep ? x[r] : -1.0e9f);
    }
  } else {
    // x = hi + lo: hi = f16(x) by packed converts, lo = f16(x - hi) with the remainder from v_fma_mix_f32 (reads the
    // f16 half straight out of the packed register); the range guard watches the unmasked values (one v_max3 per two
    // elements); the mask is applied to the packed halves
    if (kind == 0) {
      vmax = fmaxf(fmaxf(vmax, fabsf(x[0])), fabsf(x[1]));
      vmax = fmaxf(fmaxf(vmax, fabsf(x[2])), fabsf(x[3]));
      o[g].h01 = pk_f16(x[0], x[1]);
      o[g].h23 = pk_f16(x[2], x[3]);
    } else if (kind == 1) {
      o[g].r0 = mix_rem<0>(x[0], o[g].h01);
      o[g].r1 = mix_rem<1>(x[1], o[g].h01);
      o[g].r2 = mix_rem<0>(x[2], o[g].h23);
      o[g].r3 = mix_rem<1>(x[3], o[g].h23);
    } else {
      const unsigned km = keep ? 0xffffffffu : 0u;
      const unsigned l01 = pk_f16(o[g].r0, o[g].r1) & km, l23 = pk_f16(o[g].r2, o[g].r3) & km;
      char *p = act + (1 + b * 32 + n) * ROWB + (g * 8 + h * 4) * 2;
      if (!JG_SDBG(8)) {
        *reinterpret_cast<uint2 *>(p) = make_uint2(o[g].h01 & km, o[g].h23 & km);
        *reinterpret_cast<uint2 *>(p + 64) = make_uint2(l01, l23);
      }
    }
  }
}

__device__ __forceinline__ void read_frags(half8 (&fr)[12], const char *fp, const int b) {
#pragma unroll
  for (int q = 0; q < 12; ++q)
    fr[q] = *reinterpret_cast<const half8 *>(fp + (b * 32 + (q >> 2)) * ROWB + ((q >> 1) & 1) * 32 + (q & 1) * 64);
}
// MFMA i of a block's 18: tap i / 6, channel chunk (i / 3) % 2, product i % 3 (hi.hi, hi.lo, lo.hi)
__device__ __forceinline__ void mfma_one(const int i, f32x16 &c, const half8 (&w)[3][2][2], const half8 (&fr)[12], int dbg) {
  (void)dbg;
  const int t = i / 6, cc = (i / 3) % 2, k = i % 3;
  const half8 xh = fr[(t * 2 + cc) * 2], xl = fr[(t * 2 + cc) * 2 + 1];
  if (JG_SDBG(2)) {
    if (k == 0) c[0] += (float)xh[0] + (float)xl[1];
    return;
  }
  if (k == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t][cc][0], xh, c, 0, 0, 0);
  else if (k == 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t][cc][0], xl, c, 0, 0, 0);
  else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t][cc][1], xh, c, 0, 0, 0);
}
// experiment build: per-phase shader cycles of every wave (ids, table phase, first epilogue, MFMA, epilogues, pool)
#ifdef JG_EXPERIMENT
static __device__ unsigned long long jg_small_stamp[12];
#define JG_SST_DECL unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime()
#define JG_SST(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#define JG_SST_END do { if (lane == 0) for (int q_ = 0; q_ < 12; ++q_) atomicAdd(&jg_small_stamp[q_], st_[q_]); } while (0)
#define JG_SST_PARAMS , unsigned long long (&st_)[12], unsigned long long &st_t
#define JG_SST_PASS , st_, st_t
#else
#define JG_SST_DECL
#define JG_SST(i)
#define JG_SST_END
#define JG_SST_PARAMS
#define JG_SST_PASS
#endif

// the 18 MFMAs of a block into cc (fragments fr), one per slot, with the chunks of block b's output step dealt out
// behind them
template <bool LAST, bool TAP, bool PMAX>
__device__ __forceinline__ void mfma_slots(f32x16 &cc, const half8 (&w)[3][2][2], const half8 (&fr)[12], const float (&v)[16],
                                           OutRegs (&o)[4], const int b, const bool keep, char *act, int n, int h,
                                           float &vmax, float (&pool)[16], float (&tapv)[16], int dbg, const f32x16 &c0) {
  constexpr int K = out_chunks<LAST, TAP>();
  cc = c0;                                   // the shift of the layer these MFMAs belong to (read from LDS a layer / a block ahead)
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    mfma_one(i, cc, w, fr, dbg);
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (k * 18 / K == i) epi_out_chunk<LAST, TAP, PMAX>(k, v, o, b, keep, act, n, h, vmax, pool, tapv, dbg);
    JG_STAGE();
  }
}
__device__ __forceinline__ void load_weights(half8 (&w)[3][2][2], const half8 *wn) {
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int p = 0; p < 2; ++p) w[t][c2][p] = wn[((t * 2 + c2) * 2 + p) << 6];
}

// The first layer's epilogue over the five accumulator blocks of a row (its accumulators come out of the table phase).
// The MFMAs of the first k = 3 layer's block 0 run between the chunks of the last block's output step (rows -1 .. 32 of
// the image are complete by then): on return cc holds that block's accumulators.
template <bool SAVE, bool TAP>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[NB], f32x16 (&sc)[NB], const float *epi, M192 mout, char *act,
                                         int n, int h, float &vmax, float (&pool)[16], int dbg, float (&tapv)[16],
                                         const half8 (&w)[3][2][2], f32x16 &cc) {
  constexpr int K = out_chunks<false, TAP>();
  half8 fr[12];
  EpiParams<false> q;
  q.load(epi, h);
  f32x16 tnext;
  acc_init(tnext, epi + 4 * C, h);                      // the first k = 3 layer's shift (its block 0 rides in the last slots)
  const char *fp = act + n * ROWB + h * 16;
  JG_FENCE();
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float v[16];
    OutRegs o[4];
    if (b == NB - 1) read_frags(fr, fp, 0);
    epi_math<false, false, SAVE>(acc[b], sc[b], q, dbg, v);
    const bool keep = (mout.w[b >> 1] >> ((b & 1) * 32 + n)) & 1ull;
    if (b < NB - 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) epi_out_chunk<false, TAP, false>(k, v, o, b, keep, act, n, h, vmax, pool, tapv, dbg);
    } else {
      mfma_slots<false, TAP, false>(cc, w, fr, v, o, b, keep, act, n, h, vmax, pool, tapv, dbg, tnext);   // the next layer's block 0
    }
    JG_FENCE();      // one block at a time: hoisted loads of later blocks would spill
  }
}

// One k = 3 layer of a row, software-pipelined over its five position blocks: the 18 MFMAs of block b + 1 are issued
// one by one BETWEEN the chunks of block b's output step (plain vector instructions: they issue while the matrix pipe
// works), block b + 1's arithmetic follows once its accumulators are complete; the slots of the last block's output step
// take the NEXT layer's block 0 (cc on entry: block 0's accumulators; the same for the next layer on return).  The in-place update of the row image stays safe: block b + 2's fragments (rows 32 b + 63 ..
// 32 b + 96) are read - in program order, i.e. in LDS order - before block b + 1's output step stores rows 32 b + 32 ..
// 32 b + 63, block b's stores (rows 32 b .. 32 b + 31) come behind block b + 1's own reads, issued a step earlier, and
// the next layer's blocks 0 and 1 read rows -1 .. 64, final since this layer's third output step.
template <bool LAST, bool P2, bool ADD, bool SAVE, bool TAP, bool PMAX>
__device__ __forceinline__ void conv_layer(half8 (&w)[3][2][2], const half8 *wn, f32x16 &cc, f32x16 (&sc)[NB],
                                           const float *epi, M192 mout, char *act, int n, int h, float &vmax,
                                           float (&pool)[16], int dbg, float (&tapv)[16] JG_SST_PARAMS) {
  constexpr int K = out_chunks<LAST, TAP>();
  JG_SST(8);
  EpiParams<P2> q;
  q.load(epi, h);
  f32x16 tcur, tnext;
  acc_init(tcur, epi, h);
  if constexpr (!LAST) acc_init(tnext, epi + 4 * C, h);
  JG_SST(9);
  const char *fp = act + n * ROWB + h * 16;
  float v[16];
  half8 fr[12];
  read_frags(fr, fp, 1);
  JG_SST(6);
  epi_math<P2, ADD, SAVE>(cc, sc[0], q, dbg, v);
  JG_SST(4);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    OutRegs o[4];
    const bool keep = (mout.w[b >> 1] >> ((b & 1) * 32 + n)) & 1ull;
    if (b + 1 < NB) {
      mfma_slots<LAST, TAP, PMAX>(cc, w, fr, v, o, b, keep, act, n, h, vmax, pool, tapv, dbg, tcur);
      if (b + 2 < NB) {
        read_frags(fr, fp, b + 2);                      // rows 32 b + 63 ..: in front of block b + 1's stores
      } else {
        load_weights(w, wn);                            // behind this layer's last MFMA: the next layer's weights
        if constexpr (!LAST) read_frags(fr, fp, 0);     // and its block 0
      }
      JG_STAGE();
      JG_SST(3);
      epi_math<P2, ADD, SAVE>(cc, sc[b + 1], q, dbg, v);
      JG_SST(4);
    } else if constexpr (!LAST) {
      mfma_slots<LAST, TAP, PMAX>(cc, w, fr, v, o, b, keep, act, n, h, vmax, pool, tapv, dbg, tnext);   // next layer's block 0
      JG_STAGE();
      JG_SST(3);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) epi_out_chunk<LAST, TAP, PMAX>(k, v, o, b, keep, act, n, h, vmax, pool, tapv, dbg);
      JG_STAGE();
      JG_SST(7);
    }
  }
}


// (a run-time tap flag inside the block loop cost 8 % of the kernel - the sixteen tap registers stayed allocated in
// every variant - so the tap is a template parameter like the other stage flags)
#define JG_EPI_CALL(SAVEV)                                                                                     \
  do {                                                                                                        \
    float tapv[16];                                                                                           \
    if (TAPS && tap) {                                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) tapv[i_] = 0.0f;                                      \
      epilogue<SAVEV, true>(acc, sc, ep, mo, act, n, h, vmax, pool, dbg, tapv, w, cc);                    \
      row_reduce_store(tapv, true, prow + tap * PARTW, m_count(mo), n, h, lane);                              \
    } else {                                                                                                  \
      epilogue<SAVEV, false>(acc, sc, ep, mo, act, n, h, vmax, pool, dbg, tapv, w, cc);                   \
    }                                                                                                         \
  } while (0)

// (the pool kind of the last layer is a template parameter: a run-time branch in the block would end the scheduling
// region that interleaves its MFMAs and vector instructions)
#define JG_LAYER_CALL2(LASTV, P2V, ADDV, SAVEV, PMAXV)                                                        \
  do {                                                                                                        \
    float tapv[16];                                                                                           \
    if (TAPS && tap) {                                                                                        \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) tapv[i_] = 0.0f;                                      \
      conv_layer<LASTV, P2V, ADDV, SAVEV, true, PMAXV>(w, wn, cc, sc, ep, mo, act, n, h, vmax, pool, dbg, tapv JG_SST_PASS); \
      row_reduce_store(tapv, true, prow + tap * PARTW, m_count(mo), n, h, lane);                              \
    } else {                                                                                                  \
      conv_layer<LASTV, P2V, ADDV, SAVEV, false, PMAXV>(w, wn, cc, sc, ep, mo, act, n, h, vmax, pool, dbg, tapv JG_SST_PASS); \
    }                                                                                                         \
  } while (0)
#define JG_LAYER_CALL(P2V, ADDV, SAVEV) JG_LAYER_CALL2(false, P2V, ADDV, SAVEV, false)
#define JG_LAYER_CALL_LAST(P2V, ADDV)                              \
  do {                                                             \
    if (a.pool_kind == JG_POOL_AVG) JG_LAYER_CALL2(true, P2V, ADDV, false, false); \
    else JG_LAYER_CALL2(true, P2V, ADDV, false, true);             \
  } while (0)

// Reduce a lane's 16 channel values over the 32 lanes that share h and store the row's 32 channel totals + the mask
// count.  Register-halving butterfly: at each step a lane hands half of its registers to its partner and keeps the
// sums of the other half (16 exchanges instead of 16 registers x 5 steps); lane bits 4..1 end up selecting the
// register, i.e. the channel 8 g + 4 h + i with 4 g + i = r.
__device__ __forceinline__ void row_reduce_store(const float (&p)[16], const bool avg, float *dst, const int count,
                                                 const int n, const int h, const int lane) {
  float v8[8], v4[4], v2[2], v1;
  const bool b16 = n & 16, b8 = n & 8, b4 = n & 4, b2 = n & 2;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float o = __shfl_xor(b16 ? p[i] : p[i + 8], 16, 64), k = b16 ? p[i + 8] : p[i];
    v8[i] = avg ? k + o : fmaxf(k, o);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float o = __shfl_xor(b8 ? v8[i] : v8[i + 4], 8, 64), k = b8 ? v8[i + 4] : v8[i];
    v4[i] = avg ? k + o : fmaxf(k, o);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float o = __shfl_xor(b4 ? v4[i] : v4[i + 2], 4, 64), k = b4 ? v4[i + 2] : v4[i];
    v2[i] = avg ? k + o : fmaxf(k, o);
  }
  {
    const float o = __shfl_xor(b2 ? v2[0] : v2[1], 2, 64), k = b2 ? v2[1] : v2[0];
    v1 = avg ? k + o : fmaxf(k, o);
  }
  {
    const float o = __shfl_xor(v1, 1, 64);
    v1 = avg ? v1 + o : fmaxf(v1, o);
  }
  const int r = (b16 ? 8 : 0) | (b8 ? 4 : 0) | (b4 ? 2 : 0) | (b2 ? 1 : 0);
  if ((n & 1) == 0) dst[(r >> 2) * 8 + h * 4 + (r & 3)] = v1;
  if (lane == 0) dst[C] = (float)count;
}

// Round 4, measured beside the affine fold and dropped: the residual shortcut folded into the accumulators' initial value as
// well (shift + shortcut block by eight packed adds where the block is initialised, no add in the epilogue): bit-compatible,
// and it gave back the fold's whole 2 % (0.742 vs 0.722 ms per launch, three interleaved pairs) - the packed adds sit in
// front of a block's first MFMA, where a lone wave has nothing to run beside them.
// Round-3 experiments that did NOT pay and are not in this file (git history has them):
//   * the layer program as a compile-time constant (layer loop unrolled, every layer's variant chosen at compile time,
//     nothing carried around a loop through the 16-way switch): bit-identical, 3 990 vs 3 990 Mbp/s interleaved;
//   * the first layer's table reads dealt out between the stages of the previous block's arithmetic (one block of sums
//     alive, LDS latency under vector work): bit-identical, 1.5 - 3 % slower - the compiler keeps every tap's rows in
//     flight whatever the source order, the wave then runs into the 16-deep LDS queue, and pinning the sums step by
//     step costs as much as it hides.
//   * a two-block-deep pipeline with the block's 32 transcendentals in the MFMA slots as well (exp2 in slot p, + 1 in
//     slot p + 5, 1 / x in slot p + 10; two accumulator sets across the layer boundary): bit-identical, 5.5 % slower
//     (3 800 vs 4 023 Mbp/s) - 41 spilled registers, and the tap variant crashes the compiler's AGPR-copy rewrite pass;
//     it would need the 32 affine parameter registers back first (scale folded into the weights, shift into the
//     MFMA's C operand).
// TAPS = false: the model has no NMD taps - the tap paths (24 more inlined epilogues, 17 more registers) are compiled out
template <bool TAPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void small_net_kernel(JgSmallArgs a) {
  const int NC = a.n_conv, K0 = a.k0;
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
  const int dbg = a.dbg;
  (void)dbg;
  const M192 valid_in = m_first(L), valid0 = m_first(L0);
  float vmax = 0.0f;
  JG_SST_DECL;
  // the ids of a wave's NEXT row are requested while it works on the current one (a lone wave per SIMD has nothing else
  // to cover the HBM latency with)
  int idn[3];
  {
    const long row0 = (long)blockIdx.x * 4 + wave;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = j * 64 + lane;
      idn[j] = (row0 < a.rows && q < L) ? (int)a.ids[row0 * L + q] : 0;
    }
  }
  for (long row = (long)blockIdx.x * 4 + wave; row < a.rows; row += (long)gridDim.x * 4) {
    // ---- ids of the row -> LDS, input mask by ballot --------------------------------------------------
    M192 m;
    const long nrow = row + (long)gridDim.x * 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = j * 64 + lane;
      const int id = idn[j];
      idn[j] = (nrow < a.rows && q < L) ? (int)a.ids[nrow * L + q] : 0;
      m.w[j] = __ballot(a.use_mask ? id != 0 : q < L);
      idbuf[IDM + q] = q < L ? (unsigned char)id : (unsigned char)a.vocab;
    }
    m = m_and(m, valid_in);
    JG_SST(0);
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
    {
      const unsigned char *it = idbuf + IDM + n - pad0;
      int idc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) idc[b] = it[b * 32];
#pragma unroll 1
      for (int t = 0; t < (JG_SDBG(1) ? 1 : K0); ++t) {
        const float *lt = lut + t * VR * LUTS + h * 4;
        // all twenty 16-byte reads of this tap first, the next tap's ids behind them, the adds last
        f32x4 v[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float *rp = lt + idc[b] * LUTS;
#pragma unroll
          for (int g = 0; g < 4; ++g) v[b][g] = *reinterpret_cast<const f32x4 *>(rp + g * 8);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) idc[b] = it[b * 32 + t + 1];      // (reads the zero-row margin behind the last tap)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[b][g * 4 + i] += v[b][g][i];
      }
    }
    JG_SST(1);
    // output mask of the first conv ("any" rule over its k0 taps), positions [0, L0)
    M192 mo = m;
    if (a.use_mask) {
      mo = M192{{0, 0, 0}};
#pragma unroll
      for (int t = 0; t < K0; ++t) mo = m_or(mo, t >= pad0 ? m_shr(m, t - pad0) : m_shl(m, pad0 - t));
    }
    mo = m_and(mo, valid0);
    float pool[16];                 // (initialised right in front of the last layer: live there only)
    f32x16 cc;                      // accumulators of block 0 of the upcoming k = 3 layer
    float *prow = a.part + row * (long)a.n_slots * PARTW;
    {
      const float *ep = epi;
      const int tap = TAPS ? a.layer[0].tap : 0;
      if (a.layer[0].save) JG_EPI_CALL(true);
      else JG_EPI_CALL(false);
    }
    JG_SST(2);
    // ---- k = 3 convolutions on the matrix cores -----------------------------------------------------------
#pragma unroll 1
    for (int j = 0; j < NC; ++j) {
      // next layer's weights (layer 0 of the next row after the last one): requested by conv_layer behind its last MFMA
      const half8 *wn = wsrc + ((j + 1 == NC ? 0 : j + 1) * 12 << 6);
      if (a.use_mask) mo = m_and(m_or(m_or(m_shl(mo, 1), mo), m_shr(mo, 1)), valid0);
      const float *ep = epi + (j + 1) * 4 * C;
      {
        const int add = a.layer[j + 1].add, save = a.layer[j + 1].save, p2 = a.layer[j + 1].aff2;
        const int tap = TAPS ? a.layer[j + 1].tap : 0;
        if (j == NC - 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) pool[i] = a.pool_kind == JG_POOL_AVG ? 0.0f : -1.0e9f;
        }
        const int code = (j == NC - 1 ? 8 : 0) | (p2 ? 4 : 0) | (add ? 2 : 0) | (save ? 1 : 0);
        switch (code) {               // wave-uniform: one compiled epilogue per (last, second norm, add, save)
          case 0: JG_LAYER_CALL(false, false, false); break;
          case 1: JG_LAYER_CALL(false, false, true); break;
          case 2: JG_LAYER_CALL(false, true, false); break;
          case 3: JG_LAYER_CALL(false, true, true); break;
          case 4: JG_LAYER_CALL(true, false, false); break;
          case 5: JG_LAYER_CALL(true, false, true); break;
          case 6: JG_LAYER_CALL(true, true, false); break;
          case 7: JG_LAYER_CALL(true, true, true); break;
          case 8: case 9: JG_LAYER_CALL_LAST(false, false); break;
          case 10: case 11: JG_LAYER_CALL_LAST(false, true); break;
          case 12: case 13: JG_LAYER_CALL_LAST(true, false); break;
          default: JG_LAYER_CALL_LAST(true, true); break;
        }
      }
      JG_SST(10);
    }
    // ---- pooled channel sums / maxima of the row -----------------------------------------------------------
    row_reduce_store(pool, a.pool_kind == JG_POOL_AVG, prow, m_count(mo), n, h, lane);
    JG_SST(5);
  }
  JG_SST_END;
  if (__any(!(vmax <= 65000.0f)) && lane == 0) atomicOr(a.overflow, 1);
}

// pooled vector of a window from the partial rows of its frames: masked average (layers.py:460-480) or masked
// maximum with the all-masked -> 0 rule (layers.py:517-529)
// kind JG_POOL_AVG / JG_POOL_MAX as above; kind 2 = NMD vector: sum / (count + eps) - moving_mean (nmd.py:52-77)
__global__ void small_pool_final_kernel(const float *part, int frames, int n_slots, int slot, int n_win, int kind,
                                        const float *moving_mean, float eps, float *out, int out_ld) {
  const int w = blockIdx.x * (blockDim.x / C) + threadIdx.x / C, c = threadIdx.x % C;
  if (w >= n_win) return;
  float acc = kind == JG_POOL_MAX ? -1.0e9f : 0.0f, cnt = 0.0f;
  for (int f = 0; f < frames; ++f) {
    const float *p = part + (((long)w * frames + f) * n_slots + slot) * PARTW;
    acc = kind == JG_POOL_MAX ? fmaxf(acc, p[c]) : acc + p[c];
    cnt += p[C];
  }
  float r;
  if (kind == JG_POOL_AVG) r = acc / fmaxf(cnt, 1e-7f);
  else if (kind == JG_POOL_MAX) r = cnt > 0.0f ? acc : 0.0f;
  else r = acc / (cnt + eps) - moving_mean[c];
  out[(long)w * out_ld + c] = r;
}

int launch(jg_engine *e, const JgSmallArgs &a, int smem, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_net_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_net_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int grid = (int)std::min<long>(e->n_cu, (a.rows + 3) / 4);
  if (a.n_slots > 1) hipLaunchKernelGGL(small_net_kernel<true>, dim3((unsigned)grid), dim3(256), (size_t)smem, s, a);
  else hipLaunchKernelGGL(small_net_kernel<false>, dim3((unsigned)grid), dim3(256), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
#ifdef JG_EXPERIMENT
  if (a.dbg & 16) {
    unsigned long long hst[12], z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    JG_HIP(hipStreamSynchronize(s));
    JG_HIP(hipMemcpyFromSymbol(hst, HIP_SYMBOL(jg_small_stamp), sizeof(hst)));
    JG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(jg_small_stamp), z, sizeof(z)));
    const double w = (double)a.rows;
    fprintf(stderr, "SMALL STAMP rows=%ld cycles/row: ids %.0f table %.0f epi0 %.0f slots %.0f math %.0f pool %.0f frag1 %.0f last-out %.0f dispatch %.0f params %.0f\n",
            a.rows, hst[0] / w, hst[1] / w, hst[2] / w, hst[3] / w, hst[4] / w, hst[5] / w, hst[6] / w, hst[7] / w,
            (hst[8] + hst[10]) / w, hst[9] / w);
  }
#endif
  return JG_OK;
}

}  // namespace

int jg_small_lds_bytes(int n_conv, int k0, int vocab) {
  return k0 * (vocab + 1) * LUTS * 4 + (n_conv + 1) * 4 * C * 4 + 4 * WAVE_BYTES;
}

bool jg_small_supports(int n_conv, int k0, int vocab) {
  return n_conv >= 1 && n_conv <= JG_SMALL_MAX_LAYERS - 1 && k0 >= 1 && k0 <= 9 && vocab <= 254 &&
         jg_small_lds_bytes(n_conv, k0, vocab) <= 160 * 1024;
}

int jg_small_max_positions(void) { return POS; }

int jg_launch_small_net(jg_engine *e, const JgSmallArgs &a, int n_conv, int k0, hipStream_t s) {
  JG_REQUIRE(jg_small_supports(n_conv, k0, a.vocab) && a.L0 >= 1 && a.L0 <= POS && a.L <= 192 && a.pad0 <= IDM,
             JG_ERR_UNSUPPORTED, "small-window kernel: %d convs, k0 = %d, %d output positions are outside its limits", n_conv,
             k0, a.L0);
  if (a.rows == 0) return JG_OK;
  JgSmallArgs b = a;
  b.n_conv = n_conv;
  b.k0 = k0;
  return launch(e, b, jg_small_lds_bytes(n_conv, k0, a.vocab), s);
}

int jg_launch_small_pool_final(const float *part, int frames, int n_slots, int slot, int n_win, int kind,
                               const float *moving_mean, float eps, float *out, int out_ld, hipStream_t s) {
  if (n_win == 0) return JG_OK;
  const int per = 256 / C;
  hipLaunchKernelGGL(small_pool_final_kernel, dim3((unsigned)((n_win + per - 1) / per)), dim3(256), 0, s, part, frames,
                     n_slots, slot, n_win, kind, moving_mean, eps, out, out_ld);
  JG_HIP(hipGetLastError());
  return JG_OK;
}
