// Split-f16 ("f16x3") MaskedConv1D for gfx950: f32-accurate convolution on the
// f16 matrix cores.
//
// Every f32 operand is carried as an f16 pair  x = hi + lo  (hi = f16(x),
// lo = f16(x - hi), 22 significant bits together) and each logical product is
// three MFMAs accumulating in f32:  a.w ~= hi_a*hi_w + lo_a*hi_w + hi_a*lo_w
// (the dropped lo*lo term is 2^-22 relative).  v_mfma_f32_32x32x16_f16 runs at
// 16x the rate of the exact-f32 MFMA, so the scheme nets ~5.3x the f32 roof
// while keeping the logits inside the 1e-4 gate (tests/test_gpu_parity.py).
//
// Layouts
//   F16S activations, per (window, frame) row block of L positions and C = 16*CC
//     channels:  [cc][plane hi|lo][h][L][8 halfs]   (16-byte items; channel
//     c = 16*cc + 8*h + j).  Same 4 B/element as f32, but a tile's operand slice
//     for one 16-channel chunk is 4 contiguous runs - coalesced 16-B loads in,
//     conflict-free ds_read_b128 MFMA fragments out.
//   weights  [plane][tap][kc = cin/8][cout_pad][8 halfs], pre-scaled by 2^s so
//     the lo parts stay out of the f16 subnormal range (undone in the epilogue).
//
// One persistent workgroup (8 waves, 4(M) x 2(N), 64x64 outputs per wave) walks
// 256 x 128 output tiles.  The K loop is cut into stages = (16-channel chunk,
// group of <= 5 taps); stage s+1's operands are fetched into registers while
// stage s runs on the matrix cores from the other LDS buffer (one barrier per
// stage).  Mask multiply, zero padding and the embedding gather happen while
// staging; bias / norm / residual add / activation / NMD tap run in the epilogue.
#include <stdlib.h>

#include "jg_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int HM = 256;       // tile rows (positions)
constexpr int HN = 128;       // tile cols (output channels)
constexpr int HT = 512;       // threads
constexpr int TGMAX = 5;      // taps per stage
constexpr int A_ITERS = 3;    // 16-B A pieces per thread per stage (<= 4*384/512)
constexpr int B_ITERS = TGMAX;

__device__ __forceinline__ float jg_act(float v, int act) {
  switch (act) {
    case JG_ACT_GELU_TANH: {
      const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return 0.5f * v * (1.0f + tanhf(u));
    }
    case JG_ACT_GELU_ERF: return 0.5f * v * erfcf(-v * 0.70710678118654752f);
    case JG_ACT_RELU: return fmaxf(v, 0.0f);
    case JG_ACT_TANH: return tanhf(v);
    case JG_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// stages on 4 consecutive channels; ADD takes the already-loaded shortcut values
__device__ __forceinline__ float4 apply4(float4 v, const StageArg *st, int n_stages, int n,
                                         float4 addv, float mk, float4 &nmd) {
  for (int s = 0; s < n_stages; ++s) {
    const StageArg &g = st[s];
    switch (g.kind) {
      case JG_ST_BIAS: {
        const float4 b = *reinterpret_cast<const float4 *>(g.p0 + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      } break;
      case JG_ST_BN: {
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
        const float4 ga = *reinterpret_cast<const float4 *>(g.p2 + n);
        const float4 be = *reinterpret_cast<const float4 *>(g.p3 + n);
        const float al = g.f0, mm = g.arg ? mk : 1.0f;
        v.x = (tanhf(al * v.x) * ga.x + be.x) * mm;
        v.y = (tanhf(al * v.y) * ga.y + be.y) * mm;
        v.z = (tanhf(al * v.z) * ga.z + be.z) * mm;
        v.w = (tanhf(al * v.w) * ga.w + be.w) * mm;
      } break;
      case JG_ST_ADD:
        v.x += addv.x; v.y += addv.y; v.z += addv.z; v.w += addv.w;
        break;
      case JG_ST_ACT:
        v.x = jg_act(v.x, g.arg); v.y = jg_act(v.y, g.arg);
        v.z = jg_act(v.z, g.arg); v.w = jg_act(v.w, g.arg);
        break;
      case JG_ST_NMD:
        nmd.x += v.x * mk; nmd.y += v.y * mk; nmd.z += v.z * mk; nmd.w += v.w * mk;
        break;
      case JG_ST_MASKMUL:
        v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
        break;
      default: break;
    }
  }
  return v;
}

__device__ __forceinline__ void split8(const float *v, uint4 &hi, uint4 &lo, bool &ovf) {
  half8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 hh = (_Float16)v[j];
    h[j] = hh;
    l[j] = (_Float16)(v[j] - (float)hh);
    ovf |= !(fabsf(v[j]) <= 65000.0f);
  }
  hi = *reinterpret_cast<uint4 *>(&h);
  lo = *reinterpret_cast<uint4 *>(&l);
}

__device__ __forceinline__ void join8(const uint4 &hi, const uint4 &lo, float *v) {
  const half8 h = *reinterpret_cast<const half8 *>(&hi);
  const half8 l = *reinterpret_cast<const half8 *>(&lo);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)h[j] + (float)l[j];
}

struct Stage {   // one pipeline stage of one tile
  int rowblk, m0, cc, t0, nt;
};

__global__ __launch_bounds__(HT) void conv_f16x3_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int i = lane & 31, h = lane >> 5;
  // LDS carve (16-byte units)
  const int rows_a = HM + (TGMAX - 1) * a.dil;            // rows of one A stage buffer (max)
  uint4 *Abuf = lds;                                       // [2][4][rows_a]
  uint4 *Bbuf = lds + 2 * 4 * rows_a;                      // [2][2 planes][TGMAX][2 h][HN]
  float *scratch = reinterpret_cast<float *>(Bbuf + 2 * 2 * TGMAX * 2 * HN) + wid * (32 * 33);

  const int n_tg = (a.k + TGMAX - 1) / TGMAX;
  const int stages_per_tile = a.cc_in * n_tg;
  const int n_tiles = a.rows * a.tiles_m;                  // BN = cout_pad = 128: one N block
  int my_tiles = 0;
  if ((int)blockIdx.x < n_tiles) my_tiles = (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
  const int total_stages = my_tiles * stages_per_tile;
  if (total_stages == 0) return;

  auto stage_of = [&](int s) {
    Stage st;
    const int tl = s / stages_per_tile, r = s - tl * stages_per_tile;
    const int T = blockIdx.x + tl * gridDim.x;
    st.rowblk = T / a.tiles_m;
    st.m0 = (T - st.rowblk * a.tiles_m) * HM;
    st.cc = r / n_tg;
    const int tg = r - st.cc * n_tg;
    st.t0 = tg * TGMAX;
    st.nt = min(TGMAX, a.k - st.t0);
    return st;
  };

  uint4 ra[A_ITERS], rb[B_ITERS];

  // ---- fetch one stage's operands into registers ---------------------------------
  auto fetch = [&](const Stage &st) {
    const int rows_g = HM + (st.nt - 1) * a.dil;          // rows this tap group touches
    const int pbase = st.m0 - a.pad_left + st.t0 * a.dil; // input position of LDS row 0
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int q = tid + it * HT;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      const int ph = q / rows_g;
      if (ph < 4) {
        const int r = q - ph * rows_g;
        const int p = pbase + r;
        if (p >= 0 && p < a.L_in) {
          const size_t pos = (size_t)st.rowblk * a.L_in + p;
          if (a.ids != nullptr) {
            const int id = a.ids[pos];
            if (id != 0 || !a.mask_from_ids)
              v = a.embh[((size_t)id * a.cc_in + st.cc) * 4 + ph];
          } else if (a.mask_in == nullptr || a.mask_in[pos] != 0) {
            v = a.xh[(((size_t)st.rowblk * a.cc_in + st.cc) * 4 + ph) * a.L_in + p];
          }
        }
      }
      ra[it] = v;
    }
    // weights: [plane][tap][kc][cout_pad][8]; stage slice = planes x taps x h x HN items
    const int kc_total = a.cc_in * 2;
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) {
      const int q = tid + it * HT;               // = ((plane*TGMAX + tl)*2 + hh)*HN + n
      const int n = q & (HN - 1);
      const int blk = q >> 7;                    // HN == 128
      const int hh = blk & 1;
      const int tl = (blk >> 1) % TGMAX;
      const int plane = (blk >> 1) / TGMAX;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (tl < st.nt)
        v = a.wh[(((size_t)plane * a.k + (st.t0 + tl)) * kc_total + (st.cc * 2 + hh)) * a.cout_pad + n];
      rb[it] = v;
    }
  };

  auto commit = [&](const Stage &st, int buf) {
    const int rows_g = HM + (st.nt - 1) * a.dil;
    uint4 *A = Abuf + buf * 4 * rows_a;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const int q = tid + it * HT;
      const int ph = q / rows_g;
      if (ph < 4) A[ph * rows_a + (q - ph * rows_g)] = ra[it];
    }
    uint4 *B = Bbuf + buf * (2 * TGMAX * 2 * HN);
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) B[tid + it * HT] = rb[it];
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
  };
  zero_acc();

  Stage cur = stage_of(0);
  fetch(cur);
  for (int s = 0; s < total_stages; ++s) {
    const int buf = s & 1;
    commit(cur, buf);
    __syncthreads();
    Stage nxt = cur;
    if (s + 1 < total_stages) {
      nxt = stage_of(s + 1);
      if (!(a.dbg & 4)) fetch(nxt);
    }
    // ---- matrix-core work of this stage ---------------------------------------------
    {
      const uint4 *A = Abuf + buf * 4 * rows_a;
      const uint4 *B = Bbuf + buf * (2 * TGMAX * 2 * HN);
      const int arow = wm * 64 + i;
      const int bcol = wn * 64 + i;
      for (int tl = 0; tl < ((a.dbg & 2) ? 0 : cur.nt); ++tl) {
        half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          const int r = arow + tm * 32 + tl * a.dil;
          const uint4 vh = A[(0 * 2 + h) * rows_a + r];
          const uint4 vl = A[(1 * 2 + h) * rows_a + r];
          ah[tm] = *reinterpret_cast<const half8 *>(&vh);
          al[tm] = *reinterpret_cast<const half8 *>(&vl);
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const uint4 vh = B[((0 * TGMAX + tl) * 2 + h) * HN + bcol + tn * 32];
          const uint4 vl = B[((1 * TGMAX + tl) * 2 + h) * HN + bcol + tn * 32];
          bh[tn] = *reinterpret_cast<const half8 *>(&vh);
          bl[tn] = *reinterpret_cast<const half8 *>(&vl);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
      }
    }
    // ---- tile finished: fused epilogue (per wave, through a private LDS transposer) ----
    const bool tile_end = ((s + 1) % stages_per_tile) == 0;
    if (tile_end && (a.dbg & 1)) {
      if (acc[0][0][0] + acc[1][1][3] + acc[0][1][7] + acc[1][0][9] == 12345.678f) a.overflow[0] = 2;
      zero_acc();
    } else if (tile_end) {
      bool ovf = false;
      const int m_l = lane & 31;               // position inside the 32-row block
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        float4 nmd0[2], nmd1[2];               // NMD partials of this lane's channel groups
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          nmd0[g2] = make_float4(0.f, 0.f, 0.f, 0.f);
          nmd1[g2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
          // C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            scratch[((r & 3) + 8 * (r >> 2) + 4 * h) * 33 + i] = acc[tm][tn][r] * a.acc_scale;
          // wave-private scratch: LDS ops of one wave execute in order, so only the
          // compiler has to be kept from reordering across the exchange
          __builtin_amdgcn_wave_barrier();
          const int m = cur.m0 + (wm * 2 + tm) * 32 + m_l;
          const bool mvalid = m < a.L_out;
          const size_t pos = (size_t)cur.rowblk * a.L_out + (mvalid ? m : 0);
          const float mk = (a.mask_out != nullptr && mvalid) ? (a.mask_out[pos] != 0 ? 1.f : 0.f) : 1.f;
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const int g = h + 2 * g2;                          // 8-channel group inside the 32-col block
            const int n = (wn * 2 + tn) * 32 + g * 8;          // first output channel
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = scratch[m_l * 33 + g * 8 + j];
            if (g2 == 1) __builtin_amdgcn_wave_barrier();
            if (mvalid && n < a.cout) {
              const int G = n >> 3;                            // global 8-channel group
              const size_t item = (((size_t)cur.rowblk * (a.cout_pad >> 4) + (G >> 1)) * 4 + (G & 1)) *
                                      a.L_out + m;             // hi plane item; lo = + 2*L_out
              float sc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
              if (a.addh != nullptr) join8(a.addh[item], a.addh[item + 2 * (size_t)a.L_out], sc);
              float4 v0 = make_float4(v[0], v[1], v[2], v[3]), v1 = make_float4(v[4], v[5], v[6], v[7]);
              v0 = apply4(v0, a.st, a.n_stages, n, make_float4(sc[0], sc[1], sc[2], sc[3]), mk, nmd0[g2]);
              v1 = apply4(v1, a.st, a.n_stages, n + 4, make_float4(sc[4], sc[5], sc[6], sc[7]), mk, nmd1[g2]);
              if (a.out_f16s) {
                const float o8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                uint4 hi, lo;
                split8(o8, hi, lo, ovf);
                uint4 *yh = reinterpret_cast<uint4 *>(a.y);
                yh[item] = hi;
                yh[item + 2 * (size_t)a.L_out] = lo;
              } else {
                float *yf = reinterpret_cast<float *>(a.y) + pos * a.cout + n;
                *reinterpret_cast<float4 *>(yf) = v0;
                *reinterpret_cast<float4 *>(yf + 4) = v1;
              }
            }
          }
        }
        // NMD partial: reduce the 32 positions of this lane half, one row per (tile, wm)
        if (a.nmd_out != nullptr) {
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            float vals[8] = {nmd0[g2].x, nmd0[g2].y, nmd0[g2].z, nmd0[g2].w,
                             nmd1[g2].x, nmd1[g2].y, nmd1[g2].z, nmd1[g2].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float x = vals[j];
#pragma unroll
              for (int off = 16; off > 0; off >>= 1) x += __shfl_xor(x, off, 32);
              vals[j] = x;
            }
            const int n = (wn * 2 + tn) * 32 + (h + 2 * g2) * 8;
            if (m_l == 0 && n < a.cout) {
              const int tile = cur.m0 / HM;
              float *dst = a.nmd_out + (((size_t)cur.rowblk * a.tiles_m + tile) * 4 + wm) * a.cout + n;
#pragma unroll
              for (int j = 0; j < 8; ++j) dst[j] = vals[j];
            }
          }
        }
      }
      if (ovf && a.overflow != nullptr) atomicOr(a.overflow, 1);
      zero_acc();
    }
    cur = nxt;
  }
}

}  // namespace

int jg_conv_f16_lds_bytes(int dil) {
  const int rows_a = HM + (TGMAX - 1) * dil;
  return (2 * 4 * rows_a + 2 * 2 * TGMAX * 2 * HN) * 16 + 8 * 32 * 33 * 4;
}

int jg_conv_f16_tile_m(void) { return HM; }

int jg_launch_conv_f16(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  JG_REQUIRE(a.cout_pad == HN, JG_ERR_UNSUPPORTED, "conv_f16x3: cout_pad=%d (needs 128)", a.cout_pad);
  JG_REQUIRE((HM + (TGMAX - 1) * a.dil) * 4 <= A_ITERS * HT, JG_ERR_UNSUPPORTED,
             "conv_f16x3: dilation %d too large", a.dil);
  if (a.rows == 0 || a.L_out <= 0) return JG_OK;
  const int smem = jg_conv_f16_lds_bytes(a.dil);
  JG_REQUIRE(smem <= 160 * 1024, JG_ERR_UNSUPPORTED, "conv_f16x3: needs %d B of LDS", smem);
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_f16x3_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int n_tiles = a.rows * a.tiles_m;
  static int dbg = -1;
  if (dbg < 0) { const char *ev = getenv("JG_DBG"); dbg = ev ? atoi(ev) : 0; }
  const_cast<ConvHArgs &>(a).dbg = dbg;
  int grid = e->n_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(conv_f16x3_kernel, dim3((unsigned)grid), dim3(HT), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}
