// gfx950 (MI355X / CDNA4) kernels of the Jaeger predict hot path.
//
//   encode_kernel   window bytes -> 6-frame codon ids + G/C/A/T counts
//                   (seqops/io.py:119-133, seqops/encode.py:228-302)
//   mask_kernel     MaskedConv1D output-mask rule (nnlib/v2/layers.py:1226-1255)
//   conv_f32_kernel MaskedConv1D as an implicit-im2col GEMM on the exact-f32
//                   matrix cores (v_mfma_f32_32x32x2_f32) with the following
//                   norm / activation / residual-add / NMD tap fused behind it
//                   (layers.py:1217-1280, :918-938, :431-444, :1882-1915; nmd.py:52-77)
//   pool/dense/nmd  masked global pools, Dense heads, NMD finalisation
//                   (layers.py:455-538, builder.py:589-613, nmd.py:66-77)
//
// Written for 64-lane wavefronts; no other target is supported.
#include <math.h>

#include <stdio.h>

#include <type_traits>

#include "jg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// (tanh-GELU and sigmoid through v_exp_f32 / v_rcp_f32 - the formulas of the split-f16 kernels, jg_conv_dev.h -
// instead of libm's tanhf / expf: 8 instead of ~40 instructions per element; the exact-f32 conv's epilogue was a third
// of its tile time.  Saturates correctly: 2^t -> 0 or inf gives x or -0.)
__device__ __forceinline__ float jg_apply_act(float v, int act) {
  switch (act) {
    case JG_ACT_GELU_TANH: {
      // tf.nn.gelu(approximate=True): 0.5x(1+tanh(u)) = x / (1 + e^(-2u)), u = sqrt(2/pi)(x+0.044715x^3)
      const float t = v * (-2.3022082f - 0.10294324f * v * v);   // -2u * log2(e)
      return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
    }
    case JG_ACT_GELU_ERF:
      return 0.5f * v * erfcf(-v * 0.70710678118654752f);
    case JG_ACT_RELU:
      return fmaxf(v, 0.0f);
    case JG_ACT_TANH:
      // libm: 1 - 2 / (1 + e^(2v)) cancels for small |v| (relative error 1e-3 at |v| = 1e-4), and this kernel is the safe
      // path the range guard falls back to; a bare tanh activation is not on any hot path
      return tanhf(v);
    case JG_ACT_SIGMOID:
      return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950f * v));
    default:
      return v;
  }
}

// Apply the fused stage list to 4 consecutive channels n..n+3 of one position.
// `o` = flat element offset of channel n in the (rows, L_out, cout) output,
// `mk` = output mask of the position (1 when the op carries no mask).
// Up to two NMD taps per stage list (a return_nmd norm's tap in front of the norm and an nmd layer behind the block):
// the first accumulates into nmd_acc, the second into nmd_acc2.
__device__ __forceinline__ float4 jg_apply_stages(float4 v, const StageArg *st, int n_stages, int n,
                                                  size_t o, float mk, float4 *nmd_acc, float4 *nmd_acc2 = nullptr) {
  bool tapped = false;
  for (int s = 0; s < n_stages; ++s) {
    const StageArg &g = st[s];
    switch (g.kind) {
      case JG_ST_BIAS: {
        const float4 b = *reinterpret_cast<const float4 *>(g.p0 + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      } break;
      case JG_ST_BN: {
        // MaskedBatchNorm inference (layers.py:928-935): gamma*((x-mu)*inv_std)+beta
        const float4 mu = *reinterpret_cast<const float4 *>(g.p0 + n);
        const float4 is = *reinterpret_cast<const float4 *>(g.p1 + n);
        const float4 ga = *reinterpret_cast<const float4 *>(g.p2 + n);
        const float4 be = *reinterpret_cast<const float4 *>(g.p3 + n);
        v.x = ga.x * ((v.x - mu.x) * is.x) + be.x;
        v.y = ga.y * ((v.y - mu.y) * is.y) + be.y;
        v.z = ga.z * ((v.z - mu.z) * is.z) + be.z;
        v.w = ga.w * ((v.w - mu.w) * is.w) + be.w;
      } break;
      case JG_ST_DYT: {
        // MaskedDYT (layers.py:431-444): tanh(alpha*x)*gamma+beta, then *mask
        const float4 ga = *reinterpret_cast<const float4 *>(g.p2 + n);
        const float4 be = *reinterpret_cast<const float4 *>(g.p3 + n);
        const float al = g.f0;
        const float mm = g.arg ? mk : 1.0f;
        v.x = (tanhf(al * v.x) * ga.x + be.x) * mm;
        v.y = (tanhf(al * v.y) * ga.y + be.y) * mm;
        v.z = (tanhf(al * v.z) * ga.z + be.z) * mm;
        v.w = (tanhf(al * v.w) * ga.w + be.w) * mm;
      } break;
      case JG_ST_ADD: {
        const float4 r = *reinterpret_cast<const float4 *>(g.p0 + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      } break;
      case JG_ST_ACT:
        v.x = jg_apply_act(v.x, g.arg);
        v.y = jg_apply_act(v.y, g.arg);
        v.z = jg_apply_act(v.z, g.arg);
        v.w = jg_apply_act(v.w, g.arg);
        break;
      case JG_ST_NMD: {
        float4 *acc = (tapped && nmd_acc2 != nullptr) ? nmd_acc2 : nmd_acc;
        acc->x += v.x * mk; acc->y += v.y * mk;
        acc->z += v.z * mk; acc->w += v.w * mk;
        tapped = true;
      } break;
      case JG_ST_MASKMUL:
        v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
        break;
      default:
        break;
    }
  }
  return v;
}


// The same stage list applied to the RPT positions one thread of the exact-f32 conv owns (same channel quad n, rows
// r0, r0 + RSTEP, ...), STAGE-MAJOR: a stage's per-channel parameters are loaded once instead of once per position, and
// the residual shortcut's RPT loads are issued together.  Position-major (one jg_apply_stages call per row) every row
// paid its own chain of dependent loads - bias, four batch-norm vectors, the shortcut from HBM - and the epilogue was a
// quarter to two fifths of a workgroup's life (cycle stamps, round 4).  Element-wise operations and the NMD sums run in
// the same order per element as before: bit-identical results.  (Measured beside it and dropped: the parameter vectors of a
// tile's channel block staged in the LDS left beside the accumulator exchange, -3 %; the shortcut of all of a thread's rows
// requested at the top of the epilogue, -12 %, or at the top of each row group, -0.5 %; a leading [bias, batch norm] pair's vectors loaded once per thread, -2 % -
// under the 128-register cap of four waves per SIMD both cost the main loop
// more than the epilogue gains.)
template <int RPT>
__device__ __forceinline__ void jg_apply_stages_rows(float4 (&v)[RPT], const bool (&ok)[RPT], const StageArg *st, int n_stages,
                                                     int n, const size_t (&o)[RPT], const float (&mk)[RPT], float4 *nmd_acc,
                                                     float4 *nmd_acc2) {
  bool tapped = false;
  for (int s = 0; s < n_stages; ++s) {
    const StageArg &g = st[s];
    const auto vec = [&](const float *gp, int) { return *reinterpret_cast<const float4 *>(gp + n); };
    switch (g.kind) {
      case JG_ST_BIAS: {
        const float4 b = vec(g.p0, 0);
#pragma unroll
        for (int i = 0; i < RPT; ++i) { v[i].x += b.x; v[i].y += b.y; v[i].z += b.z; v[i].w += b.w; }
      } break;
      case JG_ST_BN: {
        const float4 mu = vec(g.p0, 0), is = vec(g.p1, 1), ga = vec(g.p2, 2), be = vec(g.p3, 3);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          v[i].x = ga.x * ((v[i].x - mu.x) * is.x) + be.x;
          v[i].y = ga.y * ((v[i].y - mu.y) * is.y) + be.y;
          v[i].z = ga.z * ((v[i].z - mu.z) * is.z) + be.z;
          v[i].w = ga.w * ((v[i].w - mu.w) * is.w) + be.w;
        }
      } break;
      case JG_ST_DYT: {
        const float4 ga = vec(g.p2, 0), be = vec(g.p3, 1);
        const float al = g.f0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const float mm = g.arg ? mk[i] : 1.0f;
          v[i].x = (tanhf(al * v[i].x) * ga.x + be.x) * mm;
          v[i].y = (tanhf(al * v[i].y) * ga.y + be.y) * mm;
          v[i].z = (tanhf(al * v[i].z) * ga.z + be.z) * mm;
          v[i].w = (tanhf(al * v[i].w) * ga.w + be.w) * mm;
        }
      } break;
      case JG_ST_ADD: {
        float4 r[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) r[i] = *reinterpret_cast<const float4 *>(g.p0 + (ok[i] ? o[i] : o[0]));   // (o[0] is always a real row)
#pragma unroll
        for (int i = 0; i < RPT; ++i) { v[i].x += r[i].x; v[i].y += r[i].y; v[i].z += r[i].z; v[i].w += r[i].w; }
      } break;
      case JG_ST_ACT:
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          v[i].x = jg_apply_act(v[i].x, g.arg);
          v[i].y = jg_apply_act(v[i].y, g.arg);
          v[i].z = jg_apply_act(v[i].z, g.arg);
          v[i].w = jg_apply_act(v[i].w, g.arg);
        }
        break;
      case JG_ST_NMD: {
        float4 *acc = (tapped && nmd_acc2 != nullptr) ? nmd_acc2 : nmd_acc;
#pragma unroll
        for (int i = 0; i < RPT; ++i)
          if (ok[i]) {
            acc->x += v[i].x * mk[i]; acc->y += v[i].y * mk[i];
            acc->z += v[i].z * mk[i]; acc->w += v[i].w * mk[i];
          }
        tapped = true;
      } break;
      case JG_ST_MASKMUL:
#pragma unroll
        for (int i = 0; i < RPT; ++i) { v[i].x *= mk[i]; v[i].y *= mk[i]; v[i].z *= mk[i]; v[i].w *= mk[i]; }
        break;
      default:
        break;
    }
  }
}


// ---------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------
// One workgroup per window.  The window's bytes are scanned once from HBM
// (coalesced), reduced to 2-bit codes (T,C,A,G = 0..3, anything else = 4) in
// LDS, and every (frame, codon slot) is then a 3-code LDS read + one lookup in
// the LDS-resident 64-entry codon table.  flags bit0: bytes are already cased
// (lower case = soft-masked: not counted); bit1: ids are case sensitive
// (string_processor.masking = true, encode.py:259-261); bit2: nucleotide ids (n_win, 2, l_pad) instead of codon ids.
__global__ __launch_bounds__(256) void encode_kernel(const uint8_t *__restrict__ bases,
                                                     const int64_t *__restrict__ win_start,
                                                     const int32_t *__restrict__ win_len, int fsize,
                                                     const uint8_t *__restrict__ lut, int flags,
                                                     int l_pad, uint8_t *__restrict__ ids,
                                                     int32_t *__restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sm[];
  uint8_t *code = sm;                 // fsize bytes (padded to 16)
  uint8_t *slut = sm + ((fsize + 15) & ~15);
  int *scnt = reinterpret_cast<int *>(slut + 64);
  const int64_t w = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t start = win_start[w];
  int n = win_len[w];
  n = n < fsize ? n : fsize;           // forward_strand[:crop_size], encode.py:234
  if (tid < 64) slut[tid] = lut[tid];
  if (tid < 4) scnt[tid] = 0;
  __syncthreads();
  int cg = 0, cc = 0, ca = 0, ct = 0;
  for (int i = tid; i < n; i += 256) {
    uint8_t ch = bases[start + i];
    uint8_t up = (ch >= 'a' && ch <= 'z') ? (uint8_t)(ch - 32) : ch;
    const uint8_t cnt_ch = (flags & 1) ? ch : up;  // io.py:104 upper() unless pre-cased
    const uint8_t id_ch = (flags & 2) && !(flags & 4) ? cnt_ch : up;   // the nucleotide map holds both cases (encode.py:36-41)
    cg += cnt_ch == 'G'; cc += cnt_ch == 'C'; ca += cnt_ch == 'A'; ct += cnt_ch == 'T';
    uint8_t cd = 4;
    if (id_ch == 'T') cd = 0;
    else if (id_ch == 'C') cd = 1;
    else if (id_ch == 'A') cd = 2;
    else if (id_ch == 'G') cd = 3;
    code[i] = cd;
  }
  // wave reduction of the counts, then one LDS atomic per wave
  for (int off = 32; off > 0; off >>= 1) {
    cg += __shfl_down(cg, off); cc += __shfl_down(cc, off);
    ca += __shfl_down(ca, off); ct += __shfl_down(ct, off);
  }
  if ((tid & 63) == 0) {
    atomicAdd(&scnt[0], cg); atomicAdd(&scnt[1], cc);
    atomicAdd(&scnt[2], ca); atomicAdd(&scnt[3], ct);
  }
  __syncthreads();
  if (counts != nullptr && tid < 4) counts[w * 4 + tid] = scnt[tid];
  if (flags & 4) {
    // input_type "nucleotide" (encode.py:265-271): rows = forward strand, reverse complement of the SAME cropped bases;
    // A,G,C,T -> 0,1,2,3 (+1 here, 0 = the all-zero one-hot row of any other byte and of the padding)
    uint8_t *out2 = ids + w * 2 * (int64_t)l_pad;
    for (int idx = tid; idx < 2 * l_pad; idx += 256) {
      const int f = idx >= l_pad, i = idx - f * l_pad;
      uint8_t v = 0;
      if (i < n) {
        uint8_t cd = f ? code[n - 1 - i] : code[i];
        if (cd < 4) {
          if (f) cd ^= 2;                                   // complement: T <-> A, C <-> G
          v = (uint8_t)((0x02010304u >> (8 * cd)) & 0xff);  // codes T,C,A,G = 0..3 -> ids 4,3,1,2
        }
      }
      out2[idx] = v;
    }
    return;
  }
  // frame length: ceil((n-5+off)/3), off from crop_size % 3 (encode.py:232-236, :279-284)
  const int off3 = (fsize % 3 == 0) ? -2 : ((fsize % 3 == 1) ? -1 : 0);
  if (flags & 8) {
    // codon = DICODON (ngram_width 6): ngrams() leaves n - 5 six-grams, frame j keeps every sixth one from j below the
    // codon frames' own stop (-3 + j + off): ceil((n - 8 + off) / 6) entries; id = 64 * codon(first half) +
    // codon(second half) + 1 in the reference's codon order (seqops/maps.py:544-546), 16 bits wide
    const int usable6 = n - 8 + off3;
    const int lw6 = usable6 > 0 ? (usable6 + 5) / 6 : 0;
    uint16_t *out16 = reinterpret_cast<uint16_t *>(ids) + w * 6 * (int64_t)l_pad;
    for (int idx = tid; idx < 6 * l_pad; idx += 256) {
      const int f = idx / l_pad, i = idx - f * l_pad;
      uint16_t v = 0;
      if (i < lw6) {
        const int j = f >= 3 ? f - 3 : f;
        int b[6];
        if (f < 3) {
          const int p = j + 6 * i;
#pragma unroll
          for (int q = 0; q < 6; ++q) b[q] = code[p + q];
        } else {
          const int p = n - 1 - j - 6 * i;              // rev[q] = complement(fwd[n - 1 - q])
#pragma unroll
          for (int q = 0; q < 6; ++q) b[q] = code[p - q] ^ 2;
        }
        if ((b[0] | b[1] | b[2] | b[3] | b[4] | b[5]) < 4) {
          const int ca_ = slut[16 * b[0] + 4 * b[1] + b[2]], cb_ = slut[16 * b[3] + 4 * b[4] + b[5]];
          if (ca_ != 0 && cb_ != 0) v = (uint16_t)(64 * (ca_ - 1) + (cb_ - 1) + 1);
        }
      }
      out16[idx] = v;
    }
    return;
  }
  const int usable = n - 5 + off3;
  const int lw = usable > 0 ? (usable + 2) / 3 : 0;
  uint8_t *out = ids + w * 6 * (int64_t)l_pad;
  for (int idx = tid; idx < 6 * l_pad; idx += 256) {
    const int f = idx / l_pad, i = idx - f * l_pad;
    uint8_t v = 0;
    if (i < lw) {
      const int j = f >= 3 ? f - 3 : f;
      int b0, b1, b2;
      if (f < 3) {
        const int p = j + 3 * i;
        b0 = code[p]; b1 = code[p + 1]; b2 = code[p + 2];
      } else {
        // reverse strand: rev[q] = complement(fwd[n-1-q]); T<->A, C<->G is code^2
        const int p = n - 1 - j - 3 * i;
        b0 = code[p] ^ 2; b1 = code[p - 1] ^ 2; b2 = code[p - 2] ^ 2;
      }
      if ((b0 | b1 | b2) < 4) v = slut[16 * b0 + 4 * b1 + b2];
    }
    out[idx] = v;
  }
}

int jg_launch_encode(const uint8_t *bases, const int64_t *win_start, const int32_t *win_len,
                     int64_t n_win, int fsize, const uint8_t *lut, int flags, int l_pad,
                     uint8_t *ids, int32_t *counts, hipStream_t s) {
  if (n_win == 0) return JG_OK;
  const size_t smem = ((fsize + 15) & ~15) + 64 + 16;
  JG_REQUIRE(smem <= 160 * 1024, JG_ERR_UNSUPPORTED, "encode: fsize %d exceeds the LDS window", fsize);
  if (smem > 48 * 1024)
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encode_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(encode_kernel, dim3((unsigned)n_win), dim3(256), smem, s, bases, win_start,
                     win_len, fsize, lut, flags, l_pad, ids, counts);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// Embedding lookup of 16-bit ids (JG_OP_EMBED: dicodon models) - one thread per (position, 4 channels)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_kernel(const uint16_t *__restrict__ ids, int64_t n_pos,
                                                    const float *__restrict__ table, int vocab, int c4,
                                                    float *__restrict__ out, uint8_t *__restrict__ mask) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= n_pos * c4) return;
  const int64_t pos = q / c4;
  const int g = (int)(q - pos * c4);
  const int id = min((int)ids[pos], vocab - 1);
  reinterpret_cast<float4 *>(out)[q] = reinterpret_cast<const float4 *>(table)[(int64_t)id * c4 + g];
  if (g == 0 && mask != nullptr) mask[pos] = id != 0;
}

int jg_launch_embed(const uint16_t *ids, int64_t n_pos, const float *table, int vocab, int c, float *out, uint8_t *mask,
                    hipStream_t s) {
  JG_REQUIRE(c > 0 && c % 4 == 0 && vocab > 0, JG_ERR_UNSUPPORTED, "embed: %d channels (multiples of 4), vocabulary %d", c, vocab);
  if (n_pos == 0) return JG_OK;
  const int64_t n = n_pos * (c / 4);
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, n_pos, table, vocab, c / 4, out,
                     mask);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// conv output mask
// ---------------------------------------------------------------------------
// One thread = 4 consecutive output positions of one row.  The launch is bound by memory INSTRUCTIONS, not bytes
// (a byte per lane and tap): for stride 1 a tap of four neighbouring outputs is one (unaligned) 32-bit load and the
// result one 32-bit store - a quarter of the instructions, 14.7 -> ~5 us per launch, 13 launches per chunk.
__global__ __launch_bounds__(256) void mask_kernel(const uint8_t *__restrict__ in, int64_t n_groups,
                                                   int groups_per_row, int L_in, int L_out, int k, int stride,
                                                   int dil, int pad_left, int mode,
                                                   uint8_t *__restrict__ out) {
  const int64_t gidx = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (gidx >= n_groups) return;
  const int64_t row = gidx / groups_per_row;
  const int m0 = (int)(gidx - row * groups_per_row) * 4;
  const uint8_t *r = in + row * L_in;
  unsigned cnt = 0;                         // four byte-wide counters (k <= 255)
  for (int t = 0; t < k; ++t) {
    const int p = m0 * stride - pad_left + t * dil;
    if (stride == 1 && p >= 0 && p + 3 < L_in) {
      unsigned v;
      __builtin_memcpy(&v, r + p, 4);
      // per byte: != 0 -> 1  (set bit 7 of every non-zero byte, move it to bit 0)
      cnt += (((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) >> 7 & 0x01010101u;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = (m0 + j) * stride - pad_left + t * dil;
        if (q >= 0 && q < L_in && r[q] != 0) cnt += 1u << (8 * j);
      }
    }
  }
  unsigned res = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned c = (cnt >> (8 * j)) & 0xffu;
    unsigned v;
    if (mode == JG_MASK_ANY) v = c > 0;
    else if (mode == JG_MASK_MAJORITY) v = c >= (unsigned)((k + 1) / 2);
    else v = c == (unsigned)k;
    res |= v << (8 * j);
  }
  uint8_t *o = out + row * L_out + m0;
  if (m0 + 3 < L_out) {
    __builtin_memcpy(o, &res, 4);
  } else {
    for (int j = 0; m0 + j < L_out; ++j) o[j] = (uint8_t)(res >> (8 * j));
  }
}

int jg_launch_mask(const uint8_t *in, int rows, int L_in, int L_out, int k, int stride, int dil,
                   int pad_left, int mode, uint8_t *out, hipStream_t s) {
  if (rows == 0 || L_out <= 0) return JG_OK;
  JG_REQUIRE(k >= 1 && k <= 255, JG_ERR_UNSUPPORTED, "mask: k=%d outside the byte-wide tap counters", k);
  const int groups_per_row = (L_out + 3) / 4;
  const int64_t n_groups = (int64_t)rows * groups_per_row;
  hipLaunchKernelGGL(mask_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, s, in, n_groups,
                     groups_per_row, L_in, L_out, k, stride, dil, pad_left, mode, out);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// conv: implicit-im2col GEMM on exact-f32 MFMA
// ---------------------------------------------------------------------------
// One workgroup computes a BM x BN tile of one (window, frame) row:
//   out[m, n] = sum_{t, c} x[m*stride - pad_left + t*dil, c] * w[t, c, n]
// The (BM-1)*stride + (k-1)*dil + 1 input positions it touches are staged once
// in LDS (mask multiply / embedding gather fused into the staging), row pitch
// cin8+4 floats: 16-byte aligned rows whose 32 A-fragment rows spread over all banks.
// The contraction runs in groups of 8 input channels = four 32x32x2 MFMA steps: lane half h
// takes channels 8q+4h..8q+4h+3 of its row with ONE ds_read_b128 and the matching weights
// with ONE 16-byte load from the re-packed blob [tap][cin/8][cout_pad][8] (L2-resident),
// fetched two groups ahead of its use (with 64-position tiles a group is eight MFMAs = 512 cycles, less than an L2
// round trip: + 1.5 % over one group ahead; the A fragment is read from LDS one group ahead).
// Accumulators go through LDS once so the epilogue runs on float4 channel
// quads with fully coalesced stores.
#ifdef JG_F32_STAMP      /* experiment: per-phase shader cycles of wave 0 of every workgroup (staging, main loop, exchange, epilogue) */
static __device__ unsigned long long jg_f32_stamp[8];
#define JG_FST(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&jg_f32_stamp[i], n_ - fst_t); fst_t = n_; } } while (0)
#else
#define JG_FST(i)
#endif
template <int WM, int WN, int TM, int TN>
// (64-position tiles are 40 KB of LDS: four workgroups per CU only if a wave stays within 128 registers)
__global__ __launch_bounds__(WM *WN * 64) __attribute__((amdgpu_waves_per_eu(TM == 1 ? 4 : 2)))
void conv_f32_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef JG_F32_STAMP
  unsigned long long fst_t = __builtin_amdgcn_s_memtime();
#endif
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  constexpr int LDC = BN + 4;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid - wm * WN;
  const int row = blockIdx.x / a.tiles_m;
  const int tile = blockIdx.x - row * a.tiles_m;
  const int n_blk = blockIdx.y * BN;
  const int m0 = tile * BM;
  const int cch = a.cchunk;                // input channels staged per pass (a multiple of 8; = cin_pad when all fit)
  const int ldr = cch + 4;
  const int rows_in = (BM - 1) * a.stride + (a.k - 1) * a.dil + 1;
  const int p0 = m0 * a.stride - a.pad_left;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int i = lane & 31, h = lane >> 5;
  const int groups = a.cin_pad >> 3;
  // this lane's weight quads: [tap][group][cout_pad][8], channels 4h..4h+3 of output column n
  const float4 *wq = reinterpret_cast<const float4 *>(a.w8) + ((size_t)n_blk + wn * (TN * 32) + i) * 2 + h;
  const size_t w_group = (size_t)a.cout_pad * 2;                 // float4 units per (tap, group)
  // Input channels go through LDS in passes of `cch` (one pass unless the rows of a wide input under a long halo or a
  // stride exceed the 160 KB: 256 channels x 175 rows at k = 7, dilation 8, stride 2)
  for (int c0 = 0; c0 < a.cin_pad; c0 += cch) {
  const int cw = min(cch, a.cin_pad - c0);                        // multiple of 8
  if (c0 != 0) __syncthreads();                                   // every wave is done with the previous pass's rows
  // ---- stage the input rows (mask multiply / embedding gather / zero K padding fused) -----------------------------
  // In batches of SU pieces per thread, phase by phase: the position bytes (mask or codon id) of a batch are requested
  // together, then its 16-byte data loads, then the LDS stores.  (Round 4: written piece by piece the loop compiled to
  // byte load -> wait -> data load -> wait -> store, ten times per thread and tile - two dependent memory round trips per
  // piece, ~20 us of pure latency in front of a tile's 17 us of matrix-core work.)
  {
    constexpr int SU = 5;               // (ten = the whole tile in one batch: 19.9 vs 24.5 Mbp/s - 40 more live registers)
    const int c4 = cw >> 2;
    const int total = rows_in * c4;
    for (int base = tid; base < total; base += NT * SU) {
      int dst[SU];                 // LDS float offset of the piece, -1: no piece
      size_t src[SU];              // element offset of its source (activation tensor or embedding table)
      unsigned char byte[SU];
      bool inr[SU];
#pragma unroll
      for (int u = 0; u < SU; ++u) {
        const int idx = base + u * NT;
        const int r = idx / c4, cq = idx - r * c4;
        const int p = p0 + r;
        const int c = c0 + cq * 4;
        dst[u] = idx < total ? r * ldr + cq * 4 : -1;
        inr[u] = idx < total && p >= 0 && p < a.L_in && c < a.cin;          // (cin is a multiple of 4)
        const size_t pos = (size_t)row * a.L_in + (inr[u] ? p : 0);
        const int cs = inr[u] ? c : 0;                                       // (a valid address for pieces that are padding)
        src[u] = a.ids != nullptr ? (size_t)cs : pos * a.cin + cs;
        byte[u] = 1;
        if (inr[u]) {
          if (a.ids != nullptr) byte[u] = a.ids[pos];
          else if (a.mask_in != nullptr) byte[u] = a.mask_in[pos];
        }
      }
      // (every load is issued - at a clamped, valid address where the piece is padding or masked - and the zero is
      // selected afterwards: loads under divergent branches are waited for one by one)
      float4 v[SU];
      if (a.ids != nullptr) {
#pragma unroll
        for (int u = 0; u < SU; ++u) v[u] = *reinterpret_cast<const float4 *>(a.emb + (size_t)byte[u] * a.cin + src[u]);
#pragma unroll
        for (int u = 0; u < SU; ++u)
          if (!inr[u] || (byte[u] == 0 && a.mask_from_ids)) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 t[SU];
        // (activations are read once per launch: non-temporal, so that they do not displace the weights in L2)
#pragma unroll
        for (int u = 0; u < SU; ++u) t[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.x + src[u]));
#pragma unroll
        for (int u = 0; u < SU; ++u)
          v[u] = (inr[u] && byte[u] != 0) ? make_float4(t[u][0], t[u][1], t[u][2], t[u][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < SU; ++u)
        if (dst[u] >= 0) *reinterpret_cast<float4 *>(smem + dst[u]) = v[u];
    }
  }
  JG_FST(0);
  __syncthreads();
  JG_FST(1);
  // ---- MFMA main loop over (tap, 8-channel group of this pass) ---------------------------------------------------
  const int gpass = cw >> 3, g0 = c0 >> 3;
  const int steps = a.k * gpass;
  // operands in flight: the weight quads of steps s + 1 and s + 2 (L2 round trips of several hundred ns against a step
  // of sixteen MFMAs = 1 024 cycles), the activation fragments of step s + 1 (LDS)
  const auto w_at = [&](int tt, int gg, int tn) { return wq[(size_t)(tt * groups + g0 + gg) * w_group + (size_t)tn * 64]; };
  const float *abase = smem + ((wm * (TM * 32) + i) * a.stride) * ldr + 4 * h;
  const auto a_at = [&](int tt, int gg, int tm) {
    return *reinterpret_cast<const float4 *>(abase + (tt * a.dil + tm * 32 * a.stride) * ldr + gg * 8);
  };
#ifndef JG_F32_WPF
#define JG_F32_WPF 2            /* weight quads in flight ahead of the step that uses them */
#endif
  constexpr int WPF = JG_F32_WPF;
  float4 bq[WPF][TN], a0[TM];
  int tb = 0, gb = 0, ta = 0, ga = 0;
#pragma unroll
  for (int d = 0; d < WPF; ++d) {
    if (d < steps) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bq[d][tn] = w_at(tb, gb, tn);
      if (++gb == gpass) { gb = 0; ++tb; }
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) a0[tm] = a_at(0, 0, tm);
  if (++ga == gpass) { ga = 0; ++ta; }
  for (int sidx = 0; sidx < steps; ++sidx) {
    float4 bv[TN], av[TM];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      bv[tn] = bq[0][tn];
#pragma unroll
      for (int d = 0; d + 1 < WPF; ++d) bq[d][tn] = bq[d + 1][tn];
    }
    if (sidx + WPF < steps) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bq[WPF - 1][tn] = w_at(tb, gb, tn);
      if (++gb == gpass) { gb = 0; ++tb; }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) av[tm] = a0[tm];
    if (sidx + 1 < steps) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a0[tm] = a_at(ta, ga, tm);
      if (++ga == gpass) { ga = 0; ++ta; }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bv[tn].z, acc[tm][tn], 0, 0, 0);
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bv[tn].w, acc[tm][tn], 0, 0, 0);
      }
  }
  }
  JG_FST(2);
  __syncthreads();  // every wave is done reading the input rows
  JG_FST(3);
  // ---- accumulators -> LDS C tile ---------------------------------------------
  // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ml = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int nl = (wn * TN + tn) * 32 + i;
        smem[ml * LDC + nl] = acc[tm][tn][r];
      }
  JG_FST(4);
  __syncthreads();
  JG_FST(5);

  // ---- fused epilogue on channel quads ------------------------------------------
  constexpr int QN = BN / 4;          // channel quads per tile row
  constexpr int RSTEP = NT / QN;      // rows advanced per iteration
  const int q = tid % QN, r0 = tid / QN;
  const int n = n_blk + q * 4;
  float4 nmd_acc = make_float4(0.f, 0.f, 0.f, 0.f), nmd_acc2 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < a.cout) {
    // positions per thread: rows r0, r0 + RSTEP, ... of the tile, one channel quad; handled HR at a time (eight at once cost
    // 196 registers = two waves per SIMD instead of four)
    constexpr int RPT = BM / RSTEP, HR = RPT < 4 ? RPT : 4;
    static_assert(RPT % HR == 0, "rows per thread");
#pragma unroll 1
    for (int i0 = 0; i0 < RPT; i0 += HR) {
      float4 v[HR];
      size_t o[HR];
      float mk[HR];
      bool ok[HR];
#pragma unroll
      for (int i = 0; i < HR; ++i) {
        const int ml = r0 + (i0 + i) * RSTEP, m = m0 + ml;
        ok[i] = m < a.L_out;
        const size_t pos = (size_t)row * a.L_out + (ok[i] ? m : m0);
        o[i] = pos * a.cout + n;
        mk[i] = (a.mask_out != nullptr && ok[i]) ? (a.mask_out[pos] != 0 ? 1.f : 0.f) : 1.f;
        v[i] = *reinterpret_cast<const float4 *>(smem + ml * LDC + q * 4);
      }
      if (!ok[0]) break;                  // rows ascend: the rest of the thread's rows lie beyond the frame as well
      jg_apply_stages_rows<HR>(v, ok, a.st, a.n_stages, n, o, mk, &nmd_acc, &nmd_acc2);
      typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int i = 0; i < HR; ++i)
        if (ok[i]) {
          const f32x4 t = {v[i].x, v[i].y, v[i].z, v[i].w};
          __builtin_nontemporal_store(t, reinterpret_cast<f32x4 *>(a.y + o[i]));
        }
    }
  }
  JG_FST(6);
  // NMD taps (up to two per conv): deterministic in-block reduction over the RSTEP row groups, one partial per
  // (row, tile) written to each stage's partial-sum buffer.
  float *nmd_out[2] = {nullptr, nullptr};
  int n_taps = 0;
  for (int s = 0; s < a.n_stages; ++s)
    if (a.st[s].kind == JG_ST_NMD && n_taps < 2) nmd_out[n_taps++] = const_cast<float *>(a.st[s].p0);
  for (int tp = 0; tp < n_taps; ++tp) {
    __syncthreads();
    float4 *red = reinterpret_cast<float4 *>(smem);
    red[r0 * QN + q] = tp == 0 ? nmd_acc : nmd_acc2;
    __syncthreads();
    if (r0 == 0 && n < a.cout) {
      float4 sacc = red[q];
      for (int g = 1; g < RSTEP; ++g) {
        const float4 t4 = red[g * QN + q];
        sacc.x += t4.x; sacc.y += t4.y; sacc.z += t4.z; sacc.w += t4.w;
      }
      *reinterpret_cast<float4 *>(nmd_out[tp] + ((size_t)row * a.tiles_m + tile) * a.cout + n) = sacc;
    }
  }
}

// positions per workgroup tile of the exact-f32 conv: 64.  (Rounds 1 - 2 took 128 unless that wasted a tenth of the
// last tile.  Round 3, bench.py --precision f32 on the brain stack, interleaved on one box: 64-position tiles
// 22.5 Mbp/s against 18.9 - a 40 KB tile lets FOUR workgroups share a CU instead of two, and what the kernel lacked was
// other workgroups' MFMAs under a workgroup's staging, barriers and epilogue (matrix cores 55 % busy, waves parked 29 %
// of their life at barriers).  Measured beside it: the weight quads two steps ahead and the activation fragments one
// step ahead: 18.6 with 128-position tiles, + 1.5 % with 64 (kept); the epilogue's stage parameters preloaded into
// registers: 18.4 with 64-position tiles - 158 registers, two waves per SIMD again; the epilogue straight from the
// accumulators, stage by stage (a lane = one channel x 16 positions, no LDS round trip, parameters as scalars):
// bit-compatible results, 21.5 vs 22.7 - 119 registers, three workgroups per CU; exp2 / rcp activations instead of
// libm's: + 5 %, kept; the input channels in two passes + the accumulator exchange one wave row at a time (21 KB of LDS)
// with 80 registers per lane = SIX workgroups per CU: 22.85 vs 22.58 - beyond four the overlap is bought, dropped.)
int jg_conv_tile_m(int l_out) {
  (void)l_out;
  return 64;
}

// LDS bytes of a BM-position tile: `cch` channels of the staged input rows, or the accumulator exchange of the epilogue
static size_t conv_f32_lds(int bm, int bn, int k, int cch, int stride, int dil) {
  const size_t rows_in = (size_t)(bm - 1) * stride + (size_t)(k - 1) * dil + 1;
  const size_t lds_a = rows_in * (cch + 4) * sizeof(float), lds_c = (size_t)bm * (bn + 4) * sizeof(float);
  return ((lds_a > lds_c ? lds_a : lds_c) + 15) & ~(size_t)15;
}

// ... of one conv: 64 as well when the input rows of a 128-position tile do not fit the 160 KB of LDS (wide inputs
// under a long halo or a stride: 256 channels at dilation 8 need 166 KB)
int jg_conv_tile_m_for(int l_out, int k, int cin, int stride, int dil) {
  const int cin_pad = (cin + 7) / 8 * 8;
  if (conv_f32_lds(128, 128, k, cin_pad, stride, dil) > 160 * 1024) return 64;
  return jg_conv_tile_m(l_out);
}

// input channels staged per pass of that tile: all of them when they fit, else the largest multiple of 8 that does
// (0: not even 8 channels fit - taps x dilation beyond any model family here)
static int conv_f32_cchunk(int bm, int k, int cin_pad, int stride, int dil) {
  const size_t rows_in = (size_t)(bm - 1) * stride + (size_t)(k - 1) * dil + 1;
  const long fit = (long)((160 * 1024) / (rows_in * sizeof(float))) - 4;
  if (fit >= cin_pad) return cin_pad;
  return fit < 8 ? 0 : (int)(fit / 8 * 8);
}

template <int WM, int WN, int TM, int TN>
static int launch_conv_t(const ConvArgs &a, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  ConvArgs b = a;
  b.cchunk = conv_f32_cchunk(BM, a.k, a.cin_pad, a.stride, a.dil);
  JG_REQUIRE(b.cchunk >= 8, JG_ERR_UNSUPPORTED, "conv: k=%d stride=%d dil=%d: not even 8 input channels of a tile's %d rows fit LDS",
             a.k, a.stride, a.dil, (BM - 1) * a.stride + (a.k - 1) * a.dil + 1);
  const size_t smem = conv_f32_lds(BM, BN, a.k, b.cchunk, a.stride, a.dil);
  auto kern = conv_f32_kernel<WM, WN, TM, TN>;
  static size_t attr_set = 0;
  if (smem > attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = 160 * 1024;
  }
  dim3 grid((unsigned)((size_t)a.rows * a.tiles_m), (unsigned)((a.cout + BN - 1) / BN));
  hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), smem, s, b);
  JG_HIP(hipGetLastError());
#ifdef JG_F32_STAMP
  {
    unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    JG_HIP(hipStreamSynchronize(s));
    JG_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(jg_f32_stamp), sizeof(h)));
    JG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(jg_f32_stamp), z, sizeof(z)));
    const double w = (double)grid.x * grid.y;
    fprintf(stderr, "F32STAMP k=%d cin=%d wgs=%.0f cycles/wg (100 MHz ticks x ?): stage %.0f barrier %.0f main %.0f barrier %.0f exchange %.0f barrier %.0f epilogue %.0f\n",
            a.k, a.cin, w, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w);
  }
#endif
  return JG_OK;
}

int jg_launch_conv(jg_engine *e, const ConvArgs &a, hipStream_t s) {
  (void)e;
  JG_REQUIRE(a.cin % 4 == 0 && a.cout % 4 == 0, JG_ERR_UNSUPPORTED,
             "conv: cin=%d / cout=%d must be multiples of 4", a.cin, a.cout);
  if (a.rows == 0 || a.L_out <= 0) return JG_OK;
  const bool bm64 = jg_conv_tile_m_for(a.L_out, a.k, a.cin, a.stride, a.dil) == 64;
  JG_REQUIRE(a.tiles_m == (a.L_out + (bm64 ? 63 : 127)) / (bm64 ? 64 : 128), JG_ERR_INVALID,
             "conv: tiles_m=%d does not match the tile size chosen for L_out=%d", a.tiles_m, a.L_out);
#ifndef JG_F32_SHAPE
#define JG_F32_SHAPE 0
#endif
  // 64 x 128 tiles: wave layout (WM x WN waves, TM x TN 32 x 32 blocks each)
#if JG_F32_SHAPE == 1
  if (a.cout_pad % 128 == 0) return bm64 ? launch_conv_t<1, 4, 2, 1>(a, s) : launch_conv_t<2, 2, 2, 2>(a, s);
#elif JG_F32_SHAPE == 2
  if (a.cout_pad % 128 == 0) return bm64 ? launch_conv_t<1, 2, 2, 2>(a, s) : launch_conv_t<2, 2, 2, 2>(a, s);
#else
  if (a.cout_pad % 128 == 0) return bm64 ? launch_conv_t<2, 2, 1, 2>(a, s) : launch_conv_t<2, 2, 2, 2>(a, s);
#endif
  if (a.cout_pad % 64 == 0) return bm64 ? launch_conv_t<2, 2, 1, 1>(a, s) : launch_conv_t<2, 2, 2, 1>(a, s);
  return bm64 ? launch_conv_t<2, 1, 1, 1>(a, s) : launch_conv_t<4, 1, 1, 1>(a, s);
}

// ---------------------------------------------------------------------------
// standalone elementwise stages (norm / activation not preceded by a conv)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eltwise_kernel(EltArgs a) {
  const int cq = a.c >> 2;
  const int64_t total = a.n_pos * cq;
  float4 dummy = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t pos = idx / cq;
    const int n = (int)(idx - pos * cq) * 4;
    const float mk = a.mask != nullptr ? (a.mask[pos] != 0 ? 1.f : 0.f) : 1.f;
    const size_t o = (size_t)pos * a.c + n;
    float4 v = *reinterpret_cast<const float4 *>(a.x + o);
    v = jg_apply_stages(v, a.st, a.n_stages, n, o, mk, &dummy);
    *reinterpret_cast<float4 *>(a.y + o) = v;
  }
}

int jg_launch_eltwise(const EltArgs &a, hipStream_t s) {
  JG_REQUIRE(a.c % 4 == 0, JG_ERR_UNSUPPORTED, "eltwise: c=%d must be a multiple of 4", a.c);
  const int64_t total = a.n_pos * (a.c >> 2);
  if (total == 0) return JG_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(eltwise_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// MaskedLayerNormalization (nnlib/v2/layers.py:293-382) + the stages that follow it: one workgroup per
// (window, frame) row, a wave per position at a time, lanes over channel quads.  Input zeroed at masked positions,
// moments over the channel axis, (x - mean) / sqrt(var + eps) * gamma + beta, times the mask; then the rest of the
// stage list (shortcut add, activation, NMD tap ...).  An NMD tap leaves one partial row per (row, tile) like
// the conv epilogue does: tile 0 carries the row's sums, the other tiles zeros.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(EltArgs a, int L, int tiles_m) {
  __shared__ float4 red[4 * 64 * 2];                     // NMD partials: 4 waves x up to 128 quads (c <= 512)
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int nq = a.c >> 2;                               // channel quads
  const StageArg ln = a.st[0];
  const float inv_c = 1.0f / (float)a.c;
  float4 nmd_acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  for (int p = wid; p < L; p += 4) {
    const int64_t pos = (int64_t)row * L + p;
    const float mk = a.mask != nullptr ? (a.mask[pos] != 0 ? 1.f : 0.f) : 1.f;
    const float lmk = ln.arg ? mk : 1.f;                 // the layer's own mask multiply (use_masking)
    float4 v[2];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = lane + 64 * j;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < nq) {
        v[j] = *reinterpret_cast<const float4 *>(a.x + (size_t)pos * a.c + q * 4);
        v[j].x *= lmk; v[j].y *= lmk; v[j].z *= lmk; v[j].w *= lmk;
        sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    const float mean = sum * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (lane + 64 * j < nq) {
        const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
        sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d, 64);
    const float denom = sqrtf(sq * inv_c + ln.f0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = lane + 64 * j;
      if (q >= nq) continue;
      const int n = q * 4;
      const size_t o = (size_t)pos * a.c + n;
      const float4 ga = *reinterpret_cast<const float4 *>(ln.p2 + n);
      const float4 be = *reinterpret_cast<const float4 *>(ln.p3 + n);
      float4 y;
      y.x = ((v[j].x - mean) / denom * ga.x + be.x) * lmk;
      y.y = ((v[j].y - mean) / denom * ga.y + be.y) * lmk;
      y.z = ((v[j].z - mean) / denom * ga.z + be.z) * lmk;
      y.w = ((v[j].w - mean) / denom * ga.w + be.w) * lmk;
      y = jg_apply_stages(y, a.st + 1, a.n_stages - 1, n, o, mk, &nmd_acc[j]);
      *reinterpret_cast<float4 *>(a.y + o) = y;
    }
  }
  float *nmd_out = nullptr;
  for (int s = 1; s < a.n_stages; ++s)
    if (a.st[s].kind == JG_ST_NMD) nmd_out = const_cast<float *>(a.st[s].p0);
  if (nmd_out != nullptr) {
#pragma unroll
    for (int j = 0; j < 2; ++j) red[(wid * 2 + j) * 64 + lane] = nmd_acc[j];
    __syncthreads();
    for (int q = tid; q < nq; q += 256) {
      const int j = q >> 6, ln_ = q & 63;
      float4 t = red[(0 * 2 + j) * 64 + ln_];
      for (int w = 1; w < 4; ++w) {
        const float4 u = red[(w * 2 + j) * 64 + ln_];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      float *dst = nmd_out + ((size_t)row * tiles_m) * a.c + q * 4;
      *reinterpret_cast<float4 *>(dst) = t;
      for (int tl = 1; tl < tiles_m; ++tl)
        *reinterpret_cast<float4 *>(dst + (size_t)tl * a.c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

int jg_launch_layernorm(const EltArgs &a, int rows, int L, int tiles_m, hipStream_t s) {
  JG_REQUIRE(a.c % 4 == 0 && a.c <= 512, JG_ERR_UNSUPPORTED, "layernorm: c=%d must be a multiple of 4 up to 512", a.c);
  JG_REQUIRE(a.n_stages >= 1 && a.st[0].kind == JG_ST_LN, JG_ERR_INVALID, "layernorm: the op's first stage must be LN");
  for (int q = 1; q < a.n_stages; ++q)
    JG_REQUIRE(a.st[q].kind != JG_ST_LN, JG_ERR_UNSUPPORTED, "layernorm: a second LN in one op");
  if (rows == 0 || L == 0) return JG_OK;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)rows), dim3(256), 0, s, a, L, tiles_m);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// masked global pooling over (frames, length): one workgroup per window
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_kernel(const float *__restrict__ x,
                                                   const uint8_t *__restrict__ mask, int positions,
                                                   int c, int kind, float *__restrict__ out,
                                                   int out_ld) {
  __shared__ float4 red[256];
  __shared__ float redc[256];
  const int w = blockIdx.x, tid = threadIdx.x;
  const int cq = c >> 2;
  const int groups = 256 / cq;
  const int q = tid % cq, g = tid / cq;
  const bool is_max = kind != JG_POOL_AVG;
  float4 acc = is_max ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
  float cnt = 0.f;
  if (g < groups) {
    const float *xb = x + (size_t)w * positions * c + q * 4;
    const uint8_t *mb = mask != nullptr ? mask + (size_t)w * positions : nullptr;
    for (int p = g; p < positions; p += groups) {
      const bool keep = mb == nullptr || mb[p] != 0;
      if (!keep) continue;
      const float4 v = *reinterpret_cast<const float4 *>(xb + (size_t)p * c);
      cnt += 1.f;
      if (is_max) {
        acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y);
        acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w);
      } else {
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  }
  red[tid] = acc;
  redc[tid] = cnt;
  __syncthreads();
  if (g == 0) {
    for (int gg = 1; gg < groups; ++gg) {
      const float4 t = red[gg * cq + q];
      cnt += redc[gg * cq + q];
      if (is_max) {
        acc.x = fmaxf(acc.x, t.x); acc.y = fmaxf(acc.y, t.y);
        acc.z = fmaxf(acc.z, t.z); acc.w = fmaxf(acc.w, t.w);
      } else {
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
    }
    float4 r;
    if (is_max) {
      // layers.py:517-529: masked positions lose to the -1e9 sentinel; a window
      // with no valid position pools to zeros.  Without a mask: plain max.
      if (mask == nullptr) {
        r = acc;
      } else if (cnt <= 0.f) {
        r = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (cnt < (float)positions) {
        r = make_float4(fmaxf(acc.x, -1.0e9f), fmaxf(acc.y, -1.0e9f), fmaxf(acc.z, -1.0e9f),
                        fmaxf(acc.w, -1.0e9f));
      } else {
        r = acc;
      }
    } else {
      // layers.py:460-480: sum(x*m) / max(sum(m), 1e-7); plain mean without a mask
      const float d = mask != nullptr ? fmaxf(cnt, 1e-7f) : (float)positions;
      r = make_float4(acc.x / d, acc.y / d, acc.z / d, acc.w / d);
    }
    *reinterpret_cast<float4 *>(out + (size_t)w * out_ld + q * 4) = r;
  }
}

int jg_launch_pool(const float *x, const uint8_t *mask, int n_win, int positions, int c, int kind,
                   float *out, int out_ld, hipStream_t s) {
  JG_REQUIRE(c % 4 == 0 && c / 4 <= 256, JG_ERR_UNSUPPORTED, "pool: c=%d unsupported", c);
  JG_REQUIRE(out_ld % 4 == 0, JG_ERR_UNSUPPORTED, "pool: out_ld=%d must be a multiple of 4", out_ld);
  if (n_win == 0) return JG_OK;
  hipLaunchKernelGGL(pool_kernel, dim3((unsigned)n_win), dim3(256), 0, s, x, mask, positions, c,
                     kind, out, out_ld);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// finish of the max pool fused into the last conv's epilogue: max over a window's partial rows
// (-inf = no valid position in that strip); a window without any valid position pools to zeros
__global__ __launch_bounds__(128) void pool_final_kernel(const float *__restrict__ part, int rows_per_win, int c,
                                                         float *__restrict__ out, int out_ld) {
  const int w = blockIdx.x;
  for (int ch = threadIdx.x; ch < c; ch += 128) {
    const float *p = part + (size_t)w * rows_per_win * c + ch;
    float m = -INFINITY;
    for (int r = 0; r < rows_per_win; ++r) m = fmaxf(m, p[(size_t)r * c]);
    out[(size_t)w * out_ld + ch] = m == -INFINITY ? 0.f : m;
  }
}

int jg_launch_pool_final(const float *part, int rows_per_win, int n_win, int c, float *out, int out_ld,
                         hipStream_t s) {
  if (n_win == 0) return JG_OK;
  hipLaunchKernelGGL(pool_final_kernel, dim3((unsigned)n_win), dim3(128), 0, s, part, rows_per_win, c, out, out_ld);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// Dense head layer: out[w, o] = act(sum_i in[w, i] * W[i, o] + b[o])
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_kernel(const float *__restrict__ in, int in_ld,
                                                    const float *__restrict__ w,
                                                    const float *__restrict__ b, int64_t total,
                                                    int cin, int cout, int act,
                                                    float *__restrict__ out, int out_ld) {
  const int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x;
  if (idx >= total) return;
  const int64_t win = idx / cout;
  const int o = (int)(idx - win * cout);
  const float *xi = in + win * in_ld;
  float acc = 0.f;
  for (int i = 0; i < cin; ++i) acc = fmaf(xi[i], w[(size_t)i * cout + o], acc);
  if (b != nullptr) acc += b[o];
  out[win * out_ld + o] = jg_apply_act(acc, act);
}

// Narrow heads (cout <= 8, e.g. 128 -> 6 logits, 512 -> 8): one WAVE per window, lanes stride over the inputs and
// reduce with DPP-free shuffles - a thread per output would walk all cin inputs alone (70 us for 512 -> 8)
template <int CO>
__global__ __launch_bounds__(256) void dense_narrow_kernel(const float *__restrict__ in, int in_ld,
                                                           const float *__restrict__ w,
                                                           const float *__restrict__ b, int n_win, int cin,
                                                           int cout, int act, float *__restrict__ out, int out_ld) {
  const int win = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (win >= n_win) return;
  const float *xi = in + (size_t)win * in_ld;
  float acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) acc[o] = 0.f;
  for (int i = lane; i < cin; i += 64) {
    const float x = xi[i];
    const float *wr = w + (size_t)i * cout;
#pragma unroll
    for (int o = 0; o < CO; ++o)
      if (o < cout) acc[o] = fmaf(x, wr[o], acc[o]);
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    float v = acc[o];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    acc[o] = v;
  }
  if (lane < cout) {
    float v = 0.f;
#pragma unroll
    for (int o = 0; o < CO; ++o)
      if (o == lane) v = acc[o];
    if (b != nullptr) v += b[lane];
    out[(size_t)win * out_ld + lane] = jg_apply_act(v, act);
  }
}

// Wide heads (the strand model's 500 -> 500 layer): a workgroup takes WT windows x 256 outputs, the windows' inputs sit in
// LDS, and every weight a thread loads serves WT windows (the kernel above loads one weight per multiply-add: 5 TFLOP/s).
// Same sums in the same order as dense_kernel - bit-identical results.
template <int WT>
__global__ __launch_bounds__(256) void dense_tiled_kernel(const float *__restrict__ in, int in_ld, const float *__restrict__ w,
                                                          const float *__restrict__ b, int n_win, int cin, int cout, int act,
                                                          float *__restrict__ out, int out_ld) {
  extern __shared__ __attribute__((aligned(16))) float dense_xs[];
  const int cin_pad = (cin + 3) & ~3, tid = threadIdx.x;
  const int w0 = blockIdx.x * WT, o = blockIdx.y * 256 + tid;
  const int nw = n_win - w0 < WT ? n_win - w0 : WT;
  for (int idx = tid; idx < WT * cin_pad; idx += 256) {
    const int wi = idx / cin_pad, i = idx - wi * cin_pad;
    dense_xs[idx] = wi < nw && i < cin ? in[(size_t)(w0 + wi) * in_ld + i] : 0.f;
  }
  __syncthreads();
  if (o >= cout) return;
  float acc[WT];
#pragma unroll
  for (int wi = 0; wi < WT; ++wi) acc[wi] = 0.f;
  for (int i = 0; i < cin_pad; i += 4) {
    float wv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) wv[c] = i + c < cin ? w[(size_t)(i + c) * cout + o] : 0.f;
#pragma unroll
    for (int wi = 0; wi < WT; ++wi) {
      const float4 x = *reinterpret_cast<const float4 *>(dense_xs + wi * cin_pad + i);
      float a = acc[wi];
      a = fmaf(x.x, wv[0], a);
      a = fmaf(x.y, wv[1], a);
      a = fmaf(x.z, wv[2], a);
      a = fmaf(x.w, wv[3], a);
      acc[wi] = a;
    }
  }
  const float bias = b != nullptr ? b[o] : 0.f;
#pragma unroll
  for (int wi = 0; wi < WT; ++wi)
    if (wi < nw) out[(size_t)(w0 + wi) * out_ld + o] = jg_apply_act(b != nullptr ? acc[wi] + bias : acc[wi], act);
}

int jg_launch_dense(const float *in, int in_ld, const float *w, const float *b, int n_win, int cin,
                    int cout, int act, float *out, int out_ld, hipStream_t s) {
  const int64_t total = (int64_t)n_win * cout;
  if (total == 0) return JG_OK;
  constexpr int WT = 8;
  if (cout >= 64 && cin >= 64 && n_win >= WT && (size_t)WT * ((cin + 3) & ~3) * sizeof(float) <= 48 * 1024) {
    const size_t smem = (size_t)WT * ((cin + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(dense_tiled_kernel<WT>, dim3((unsigned)((n_win + WT - 1) / WT), (unsigned)((cout + 255) / 256)), dim3(256),
                       smem, s, in, in_ld, w, b, n_win, cin, cout, act, out, out_ld);
    JG_HIP(hipGetLastError());
    return JG_OK;
  }
  if (cout <= 8 && cin >= 64) {
    hipLaunchKernelGGL(dense_narrow_kernel<8>, dim3((unsigned)((n_win + 3) / 4)), dim3(256), 0, s, in, in_ld, w, b,
                       n_win, cin, cout, act, out, out_ld);
    JG_HIP(hipGetLastError());
    return JG_OK;
  }
  hipLaunchKernelGGL(dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_ld,
                     w, b, total, cin, cout, act, out, out_ld);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// NMD finalisation: sum partials, divide by (sum(mask)+eps), subtract moving mean
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nmd_final_kernel(const float *__restrict__ part,
                                                        int parts_per_win,
                                                        const uint8_t *__restrict__ mask,
                                                        int positions,
                                                        const float *__restrict__ moving_mean,
                                                        float eps, int c, float *__restrict__ out,
                                                        int out_ld, int out_off) {
  __shared__ float scount[256];
  const int w = blockIdx.x, tid = threadIdx.x;
  float cnt = 0.f;
  if (mask != nullptr) {
    const uint8_t *mb = mask + (size_t)w * positions;
    for (int p = tid; p < positions; p += 256) cnt += mb[p] != 0 ? 1.f : 0.f;
  }
  scount[tid] = cnt;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) scount[tid] += scount[tid + st];
    __syncthreads();
  }
  // nmd.py:60-69: masked -> sum/(count+eps); unmasked -> reduce_mean
  const float denom = mask != nullptr ? scount[0] + eps : (float)positions;
  for (int ch = tid; ch < c; ch += 256) {
    const float *pb = part + (size_t)w * parts_per_win * c + ch;
    float sacc = 0.f;
    for (int j = 0; j < parts_per_win; ++j) sacc += pb[(size_t)j * c];
    out[(size_t)w * out_ld + out_off + ch] = sacc / denom - moving_mean[ch];
  }
}

int jg_launch_nmd_final(const float *part, int parts_per_win, const uint8_t *mask, int positions,
                        const float *moving_mean, float eps, int n_win, int c, float *out,
                        int out_ld, int out_off, hipStream_t s) {
  if (n_win == 0) return JG_OK;
  hipLaunchKernelGGL(nmd_final_kernel, dim3((unsigned)n_win), dim3(256), 0, s, part, parts_per_win,
                     mask, positions, moving_mean, eps, c, out, out_ld, out_off);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// OODSignalLayer (layers.py:1632-1667): one thread per window
// bits: 1 max_prob, 2 entropy, 4 energy, 8 margin, 16 nmd_norm (emitted in the
// order of the model's `signals` list, which the host encodes in `order`).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void oodsig_kernel(const float *__restrict__ logits, int logits_ld, int n_cls,
                                                     const float *__restrict__ nmd, int nmd_ld, int nmd_w,
                                                     int n_win, unsigned order, float eps,
                                                     float *__restrict__ out, int out_ld,
                                                     int out_off) {
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_win) return;
  const float *lg = logits + (size_t)w * logits_ld;      // vector slots are padded to float4 rows: pitch != width
  float mx = -INFINITY;
  for (int i = 0; i < n_cls; ++i) mx = fmaxf(mx, lg[i]);
  float se = 0.f;
  for (int i = 0; i < n_cls; ++i) se += expf(lg[i] - mx);
  float p1 = 0.f, p2 = 0.f, ent = 0.f;
  for (int i = 0; i < n_cls; ++i) {
    const float p = expf(lg[i] - mx) / se;
    if (p > p1) { p2 = p1; p1 = p; } else if (p > p2) p2 = p;
    const float sp = fmaxf(p, eps);
    ent -= sp * logf(sp);
  }
  int col = 0;
  for (int sidx = 0; sidx < 5; ++sidx) {
    const unsigned code = (order >> (4 * sidx)) & 0xF;
    if (code == 0) break;
    float v = 0.f;
    if (code == 1) v = p1;
    else if (code == 2) v = ent;
    else if (code == 3) v = mx + logf(se);
    else if (code == 4) v = p1 - p2;
    else if (code == 5) {
      float ss = 0.f;
      for (int i = 0; i < nmd_w; ++i) { const float t = nmd[(size_t)w * nmd_ld + i]; ss += t * t; }
      v = sqrtf(ss);
    }
    out[(size_t)w * out_ld + out_off + col] = v;
    ++col;
  }
}

int jg_launch_oodsig(const float *logits, int logits_ld, int n_cls, const float *nmd, int nmd_ld, int nmd_w,
                     int n_win, unsigned signal_bits, float eps, float *out, int out_ld, int out_off,
                     hipStream_t s) {
  if (n_win == 0) return JG_OK;
  hipLaunchKernelGGL(oodsig_kernel, dim3((unsigned)((n_win + 255) / 256)), dim3(256), 0, s, logits, logits_ld,
                     n_cls, nmd, nmd_ld, nmd_w, n_win, signal_bits, eps, out, out_ld, out_off);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// legacy tower helpers: MaxPooling1D(pool 2, stride 2, VALID) and the frame sum
// (nnlib/v1/layers.py:154-207, :399-423)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool1d_kernel(const float *__restrict__ x, int64_t total,
                                                        int L_in, int L_out, int c,
                                                        float *__restrict__ y) {
  const int cq = c >> 2;
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * 256) {
    const int q = (int)(idx % cq);
    const int64_t pos = idx / cq;
    const int64_t row = pos / L_out;
    const int m = (int)(pos - row * L_out);
    const float *s0 = x + ((size_t)(row * L_in + 2 * m)) * c + q * 4;
    const float4 a = *reinterpret_cast<const float4 *>(s0);
    const float4 b = *reinterpret_cast<const float4 *>(s0 + c);
    *reinterpret_cast<float4 *>(y + (size_t)pos * c + q * 4) =
        make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
  }
}

int jg_launch_maxpool1d(const float *x, const uint8_t *mask_in, int rows, int L_in, int L_out, int c,
                        float *y, uint8_t *mask_out, hipStream_t s) {
  (void)mask_in; (void)mask_out;
  JG_REQUIRE(c % 4 == 0, JG_ERR_UNSUPPORTED, "maxpool1d: c=%d must be a multiple of 4", c);
  const int64_t total = (int64_t)rows * L_out * (c >> 2);
  if (total == 0) return JG_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(maxpool1d_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, total, L_in, L_out,
                     c, y);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// MaxPooling1D(2) on an F16S tensor [row][c/16][plane][h][L][8 x f16]: x = hi + lo is rebuilt in f32,
// pooled, and split again (the split is exact for f32 values, so this equals pooling the f32 tensor)
typedef _Float16 jg_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void maxpool1d_f16s_kernel(const uint4 *__restrict__ x, int64_t total, int L_in,
                                                             int L_out, uint4 *__restrict__ y) {
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t run = idx / L_out;                 // (row, chunk, h): two planes per run pair
    const int m = (int)(idx - run * L_out);
    const int64_t rc = run >> 1;                     // row * chunks + chunk
    const int h = (int)(run & 1);
    const uint4 *hi = x + ((rc * 4 + h) * (int64_t)L_in + 2 * m);
    const uint4 *lo = x + ((rc * 4 + 2 + h) * (int64_t)L_in + 2 * m);
    const uint4 h0 = hi[0], h1 = hi[1], l0 = lo[0], l1 = lo[1];
    const jg_half8 a0 = *reinterpret_cast<const jg_half8 *>(&h0), a1 = *reinterpret_cast<const jg_half8 *>(&h1);
    const jg_half8 b0 = *reinterpret_cast<const jg_half8 *>(&l0), b1 = *reinterpret_cast<const jg_half8 *>(&l1);
    jg_half8 oh, ol;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float v = fmaxf((float)a0[q] + (float)b0[q], (float)a1[q] + (float)b1[q]);
      const _Float16 hv = (_Float16)v;
      oh[q] = hv;
      ol[q] = (_Float16)(v - (float)hv);
    }
    y[(rc * 4 + h) * (int64_t)L_out + m] = *reinterpret_cast<const uint4 *>(&oh);
    y[(rc * 4 + 2 + h) * (int64_t)L_out + m] = *reinterpret_cast<const uint4 *>(&ol);
  }
}

// ---- layout conversions between f32 rows (rows, L, c) and F16S items [row][c/16][hi|lo][h][L][8 halfs] ------------
// (mixed programs: a conv the split-f16 tiling does not cover - strided, 1x1, other widths - or a LayerNorm runs on
// f32 tensors between split-f16 convs).  One thread per (row, position, 8-channel group): 32 B of f32 <-> a hi and a
// lo item; 16 consecutive lanes cover one position's 128 channels (coalesced on the f32 side).
// Values beyond the f16 range would become hi = +-Inf, lo = -+Inf (a NaN in the next split-f16 conv): `overflow` is raised
// for any |v| > 65000 or NaN, exactly like the conv epilogue's guard, and the host reruns the chunk in exact f32.
__global__ __launch_bounds__(256) void f32_to_f16s_kernel(const float *__restrict__ x, int64_t total, int L, int groups,
                                                          uint4 *__restrict__ y, int *__restrict__ overflow) {
  bool bad = false;
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int g = (int)(idx % groups);
    const int64_t rp = idx / groups;                 // row * L + pos
    const int64_t row = rp / L;
    const int pos = (int)(rp - row * L);
    const float4 a = reinterpret_cast<const float4 *>(x)[idx * 2], b = reinterpret_cast<const float4 *>(x)[idx * 2 + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    jg_half8 hi, lo;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const _Float16 hv = (_Float16)v[q];
      hi[q] = hv;
      lo[q] = (_Float16)(v[q] - (float)hv);
      bad = bad || !(fabsf(v[q]) <= 65000.0f);       // (NaN compares false: caught too)
    }
    const int64_t rc = row * (groups / 2) + (g >> 1);
    const int h = g & 1;
    y[(rc * 4 + h) * (int64_t)L + pos] = *reinterpret_cast<const uint4 *>(&hi);
    y[(rc * 4 + 2 + h) * (int64_t)L + pos] = *reinterpret_cast<const uint4 *>(&lo);
  }
  if (bad && overflow != nullptr) atomicOr(overflow, 1);
}

__global__ __launch_bounds__(256) void f16s_to_f32_kernel(const uint4 *__restrict__ x, int64_t total, int L, int groups,
                                                          float *__restrict__ y) {
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int g = (int)(idx % groups);
    const int64_t rp = idx / groups;
    const int64_t row = rp / L;
    const int pos = (int)(rp - row * L);
    const int64_t rc = row * (groups / 2) + (g >> 1);
    const int h = g & 1;
    const uint4 uh = x[(rc * 4 + h) * (int64_t)L + pos], ul = x[(rc * 4 + 2 + h) * (int64_t)L + pos];
    const jg_half8 hi = *reinterpret_cast<const jg_half8 *>(&uh), lo = *reinterpret_cast<const jg_half8 *>(&ul);
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (float)hi[q] + (float)lo[q];
    reinterpret_cast<float4 *>(y)[idx * 2] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4 *>(y)[idx * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

static int cvt_launch(bool to_f16s, const void *x, int64_t rows, int L, int c, void *y, hipStream_t s, int *overflow = nullptr) {
  JG_REQUIRE(c % 16 == 0, JG_ERR_UNSUPPORTED, "layout conversion: c=%d must be a multiple of 16", c);
  const int64_t total = rows * L * (c / 8);
  if (total == 0) return JG_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (to_f16s)
    hipLaunchKernelGGL(f32_to_f16s_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float *>(x), total, L,
                       c / 8, static_cast<uint4 *>(y), overflow);
  else
    hipLaunchKernelGGL(f16s_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4 *>(x), total, L,
                       c / 8, static_cast<float *>(y));
  JG_HIP(hipGetLastError());
  return JG_OK;
}
int jg_launch_f32_to_f16s(const float *x, int64_t rows, int L, int c, uint4 *y, hipStream_t s, int *overflow) {
  return cvt_launch(true, x, rows, L, c, y, s, overflow);
}
int jg_launch_f16s_to_f32(const uint4 *x, int64_t rows, int L, int c, float *y, hipStream_t s) {
  return cvt_launch(false, x, rows, L, c, y, s);
}

int jg_launch_maxpool1d_f16s(const uint4 *x, int rows, int L_in, int L_out, int c, uint4 *y, hipStream_t s) {
  JG_REQUIRE(c % 16 == 0, JG_ERR_UNSUPPORTED, "maxpool1d_f16s: c=%d must be a multiple of 16", c);
  const int64_t total = (int64_t)rows * (c / 16) * 2 * L_out;
  if (total == 0) return JG_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(maxpool1d_f16s_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, total, L_in, L_out, y);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---------------------------------------------------------------------------
// table net: ids -> conv (as table rows) -> bias -> activation -> global max / mean, one kernel
// ---------------------------------------------------------------------------
// Persistent grid of 1 024-thread workgroups, one row (a strand) per workgroup at a time.  The (k, vocab, cout) table sits
// in LDS for the workgroup's life; a position's conv output is bias + the k table rows its ids select - one 16-byte LDS
// read per tap and channel quad.  Lane groups of 64 / 128 / 256 lanes own the channel quads and split the positions between
// them in blocks of four CONSECUTIVE positions: four independent id -> row chains per tap and thread and four waves per SIMD
// hide the LDS latency that a single dependent chain pays in full (first form of this kernel: 555 ms per 200 000 windows
// against 321 ms layer by layer).  The id row is stored with a halo of `zero_id` entries - SAME padding, and the overhang of the
// last block - so that the inner loop has no bounds checks.  Bound: LDS bandwidth, 4 * k * cout bytes per position.
#define JG_TAB_PB 4
__global__ __launch_bounds__(1024) void tab_conv_pool_kernel(JgTabArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t tab_sm[];
  const int cq = a.cq, tid = threadIdx.x, V = a.vocab;
  float4 *tab = reinterpret_cast<float4 *>(tab_sm);
  float4 *red = tab + (size_t)a.k * V * cq;
  uint8_t *sid = reinterpret_cast<uint8_t *>(red + 1024);
  const float4 *gt = reinterpret_cast<const float4 *>(a.table);
  for (int i = tid; i < a.k * V * cq; i += 1024) tab[i] = gt[i];
  const int lanes = cq <= 64 ? 64 : (cq <= 128 ? 128 : 256), groups = 1024 / lanes;
  const int q = tid % lanes, g = tid / lanes;
  const bool live = q < cq;
  const float4 b4 = live ? reinterpret_cast<const float4 *>(a.bias)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
  const bool is_max = a.pool_kind != JG_POOL_AVG;
  const int n_sid = a.L_out + JG_TAB_PB - 1 + (a.k - 1) * a.dil;       // entry i = sequence position i - pad_left
  for (int row = blockIdx.x; row < a.rows; row += gridDim.x) {
    __syncthreads();                                        // (table loaded; the previous row's ids and partials are done with)
    const uint8_t *src = a.ids + (size_t)row * a.L;
    for (int i = tid; i < n_sid; i += 1024) {
      const int pos = i - a.pad_left;
      sid[i] = pos >= 0 && pos < a.L ? src[pos] : (uint8_t)a.zero_id;
    }
    __syncthreads();
    float4 pool = is_max ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
      for (int p = g * JG_TAB_PB; p < a.L_out; p += groups * JG_TAB_PB) {
        float4 acc[JG_TAB_PB];
#pragma unroll
        for (int j = 0; j < JG_TAB_PB; ++j) acc[j] = b4;
        if (a.dil == 1) {
          // the ids of a block's taps through a sliding 8-byte register window: one aligned 32-bit LDS read per four taps
          // instead of sixteen byte reads (the byte reads were a fifth of the kernel's LDS instructions)
          const uint32_t *sw = reinterpret_cast<const uint32_t *>(sid + p);
          uint64_t win = (uint64_t)sw[0] | ((uint64_t)sw[1] << 32);
          int nxt = 2;
          for (int t0 = 0; t0 < a.k; t0 += 4) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              if (t0 + tt >= a.k) break;
              const float4 *tr = tab + (size_t)(t0 + tt) * V * cq + q;
#pragma unroll
              for (int j = 0; j < JG_TAB_PB; ++j) {
                const int id = (int)((win >> (8 * (tt + j))) & 0xff);
                const float4 w = tr[id * cq];
                acc[j].x += w.x; acc[j].y += w.y; acc[j].z += w.z; acc[j].w += w.w;
              }
            }
            win = (win >> 32) | ((uint64_t)sw[nxt++] << 32);
          }
        } else {
          const uint8_t *sp = sid + p;
#pragma unroll 2
          for (int t = 0; t < a.k; ++t) {
            const float4 *tr = tab + (size_t)t * V * cq + q;
#pragma unroll
            for (int j = 0; j < JG_TAB_PB; ++j) {
              const float4 w = tr[(int)sp[j] * cq];
              acc[j].x += w.x; acc[j].y += w.y; acc[j].z += w.z; acc[j].w += w.w;
            }
            sp += a.dil;
          }
        }
#pragma unroll
        for (int j = 0; j < JG_TAB_PB; ++j) {
          if (p + j >= a.L_out) break;
          float4 v = acc[j];
          v.x = jg_apply_act(v.x, a.act); v.y = jg_apply_act(v.y, a.act);
          v.z = jg_apply_act(v.z, a.act); v.w = jg_apply_act(v.w, a.act);
          if (is_max) {
            pool.x = fmaxf(pool.x, v.x); pool.y = fmaxf(pool.y, v.y); pool.z = fmaxf(pool.z, v.z); pool.w = fmaxf(pool.w, v.w);
          } else {
            pool.x += v.x; pool.y += v.y; pool.z += v.z; pool.w += v.w;
          }
        }
      }
    }
    red[tid] = pool;
    __syncthreads();
    if (g == 0 && live) {
      for (int gg = 1; gg < groups; ++gg) {
        const float4 o = red[gg * lanes + q];
        if (is_max) {
          pool.x = fmaxf(pool.x, o.x); pool.y = fmaxf(pool.y, o.y); pool.z = fmaxf(pool.z, o.z); pool.w = fmaxf(pool.w, o.w);
        } else {
          pool.x += o.x; pool.y += o.y; pool.z += o.z; pool.w += o.w;
        }
      }
      if (!is_max) {
        const float d = (float)a.L_out;
        pool.x /= d; pool.y /= d; pool.z /= d; pool.w /= d;
      }
      float *dst = a.out + (size_t)row * a.out_ld + q * 4;
      const float v[4] = {pool.x, pool.y, pool.z, pool.w};
      for (int c = 0; c < 4; ++c)
        if (q * 4 + c < a.cout) dst[c] = v[c];
    }
  }
}

// LDS image: table, 1 024 partial pools, the id row with its halo (L + k * dil + 3 covers every padding split)
int64_t jg_tab_lds_bytes(int k, int vocab, int cq, int L, int dil) {
  return (int64_t)k * vocab * cq * 16 + 1024 * 16 + ((L + k * dil + JG_TAB_PB + 8 + 15) & ~15);   // (+ 8: the id window reads a word ahead)
}

int jg_launch_tab_conv_pool(jg_engine *e, const JgTabArgs &a, hipStream_t s) {
  if (a.rows == 0) return JG_OK;
  const int64_t smem = jg_tab_lds_bytes(a.k, a.vocab, a.cq, a.L, a.dil);
  JG_REQUIRE(smem <= 160 * 1024 && a.cq <= 256 && a.L_out >= 1 && a.zero_id >= 0 && a.zero_id < a.vocab, JG_ERR_UNSUPPORTED,
             "table net: %lld bytes of LDS / %d channel quads / %d positions", (long long)smem, a.cq, a.L_out);
  if (smem > 48 * 1024)
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tab_conv_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
  const int grid = (int)std::min<int64_t>(a.rows, e->n_cu);
  hipLaunchKernelGGL(tab_conv_pool_kernel, dim3((unsigned)grid), dim3(1024), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// Merge of a branched model's strand outputs (tf.keras.layers.Average / Add / Maximum / Concatenate over the branch outputs,
// builder.py:1251-1265; the embedding output is always their Average, :779-780): x (n_win * strands, x_ld) -> y (n_win, width)
__global__ __launch_bounds__(256) void strand_merge_kernel(const float *__restrict__ x, int x_ld, int64_t total, int strands,
                                                           int width, int kind, float *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  if (kind == JG_MERGE_CONCAT) {                        // y (n_win, strands * width): strand q's vector at columns q width ..
    const int64_t row = i / width;                      // = window * strands + strand
    y[i] = x[row * (int64_t)x_ld + (i - row * width)];
    return;
  }
  const int64_t w = i / width;
  const int c = (int)(i - w * width);
  const float *src = x + w * strands * (int64_t)x_ld + c;
  float acc = src[0];
  for (int q = 1; q < strands; ++q) {
    const float v = src[(int64_t)q * x_ld];
    acc = kind == JG_MERGE_MAX ? fmaxf(acc, v) : acc + v;
  }
  if (kind == JG_MERGE_AVERAGE) acc = acc / (float)strands;
  y[i] = acc;
}

int jg_launch_strand_merge(const float *x, int x_ld, int n_win, int strands, int width, int kind, float *y, hipStream_t s) {
  const int64_t total = (int64_t)n_win * width * (kind == JG_MERGE_CONCAT ? strands : 1);
  if (total == 0) return JG_OK;
  hipLaunchKernelGGL(strand_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x_ld, total, strands,
                     width, kind, y);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

__global__ __launch_bounds__(256) void framesum_kernel(const float *__restrict__ x, int64_t total4,
                                                       int frames, int64_t per_frame4,
                                                       float *__restrict__ y) {
  for (int64_t idx = blockIdx.x * (int64_t)256 + threadIdx.x; idx < total4;
       idx += (int64_t)gridDim.x * 256) {
    const int64_t w = idx / per_frame4, r = idx - w * per_frame4;
    const float4 *src = reinterpret_cast<const float4 *>(x) + w * frames * per_frame4 + r;
    float4 acc = src[0];
    for (int f = 1; f < frames; ++f) {
      const float4 t = src[(size_t)f * per_frame4];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    reinterpret_cast<float4 *>(y)[idx] = acc;
  }
}

int jg_launch_framesum(const float *x, int n_win, int frames, int64_t per_frame, float *y,
                       hipStream_t s) {
  JG_REQUIRE(per_frame % 4 == 0, JG_ERR_UNSUPPORTED, "framesum: L*C must be a multiple of 4");
  const int64_t total4 = (int64_t)n_win * (per_frame / 4);
  if (total4 == 0) return JG_OK;
  int64_t blocks = (total4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(framesum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, total4, frames,
                     per_frame / 4, y);
  JG_HIP(hipGetLastError());
  return JG_OK;
}
