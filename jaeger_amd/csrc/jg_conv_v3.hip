// PROTOTYPE (main loop only, timing experiment).  Split-f16 MaskedConv1D, "v3": ONE 8-wave workgroup per CU
// (two waves per SIMD), 512-position tiles, tap-paired K = 32 steps on v_mfma_f32_16x16x32_f16.
// Derived from the "v2" for the k = 5 convolutions of the residual stacks.
//
// Same arithmetic, layouts and epilogue semantics as jg_conv_f16_impl.h; a different decomposition, chosen from
// that kernel's real-data ablations: ONE workgroup of 4 waves per CU, each wave 128 positions x
// 128 channels (256 accumulator registers of the 512 a lone wave per SIMD may hold) on v_mfma_f32_16x16x32_f16:
//   * a third fewer LDS bytes per MAC and half the accumulator traffic per MAC (the clock follows the energy),
//   * the matrix-core shape the chip clocks higher on,
//   * steps of 192 MFMAs (3 072 pipe cycles) between barriers instead of 24 (768), so that a lone wave's LDS
//     latency and barrier skew are a tenth of a step instead of a third.
// K = 32 per MFMA = two taps of one 16-channel chunk: a chunk pair (c0, c1) runs as 5 steps
//   (c0; t0 t1) (c0; t2 t3) (c1; t0 t1) (c1; t2 t3) (c0 t4 | c1 t4)
// and the k-group of a lane (lane >> 4) selects (tap, channel half) purely by LDS address.
// LDS: a ring of three 16-channel activation slices of 512 + 4 dil positions (3 x 33.5 KB) and a ring of three
// weight pair-slots (2 taps x 8 KB), both filled by counted LDS-DMA as in the first kernel.
#include <stdio.h>
#include <stdlib.h>

#include "jg_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int VM = 512;                        // positions per tile
constexpr int VN = 128;                        // output channels per tile
constexpr int VT = 512;                        // threads (8 waves: 4 position quarters x 2 channel halves, two per SIMD)
constexpr int VK = 5;                          // taps
constexpr int SLICE_ITEMS = 2 * 2 * VN;        // one (chunk, tap) weight slice [plane][h][128]: 512 items = 8 KB
constexpr int PAIR_ITEMS = 2 * SLICE_ITEMS;    // a step's two slices
constexpr int WRING = 3;                       // pair-slots: one in use, two in flight
constexpr int XRING = 3;                       // activation slices: the pair in use + one in flight
constexpr int XA = 5;                          // 16-B activation pieces per thread per slice (4 * rows_a <= 2560)
constexpr int WA = PAIR_ITEMS / VT;            // 4 weight items per thread per step
#ifndef V2_ISSUE_MID
#define V2_ISSUE_MID 1
#endif

__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Tile2 {
  int rowblk, m0, valid;
};

template <unsigned EP>
__global__ __launch_bounds__(VT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_v3_kernel(ConvHArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;              // position quarter (128), channel half (64)
  const int li = lane & 15, kg = lane >> 4;           // MFMA row / column index, k-group
  const int kh = kg & 1, ksel = kg >> 1;              // channel half, which slice of the step's pair
  const int rows_a = VM + (VK - 1) * a.dil;
  const int buf_items = 4 * rows_a;                   // [plane][h][rows_a]
  uint4 *Abuf = lds;                                  // [XRING][buf_items]
  uint4 *Wbuf = lds + XRING * buf_items;              // [WRING][2 slices][plane][h][VN]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);
  const unsigned ldsW = __builtin_amdgcn_readfirstlane(lds0 + XRING * buf_items * 16 + wid * 1024);

  const int tiles_m = (a.L_out + VM - 1) / VM;
  const int n_tiles = a.rows * tiles_m;
  int my_tiles = 0;
  if ((int)blockIdx.x < n_tiles) my_tiles = (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
  if (my_tiles == 0) return;
  auto tile_of = [&](int pass) {
    Tile2 t;
    const int T = (int)blockIdx.x + pass * (int)gridDim.x;
    const int Tc = min(T, n_tiles - 1);
    t.rowblk = Tc / tiles_m;
    t.m0 = (Tc - t.rowblk * tiles_m) * VM;
    t.valid = T < n_tiles;
    return t;
  };

  // ---- activation pieces of a thread: q = tid + it * VT -> (plane/half ph, row r); recomputed where a tile's
  // table is built (once per tile) instead of living in registers; x_has bit it = the piece exists
  auto piece = [&](int it, int &ph, int &r) {
    const int q = tid + it * VT;
    ph = q / rows_a;                         // >= 4: no piece
    r = q - ph * rows_a;
  };
  unsigned x_has = 0;
#pragma unroll
  for (int it = 0; it < XA; ++it) {
    int ph, r;
    piece(it, ph, r);
    if (ph < 4) x_has |= 1u << it;
  }
  const bool x_last_wave = __builtin_amdgcn_readfirstlane((int)((XA - 1) * VT + wid * 64 < 4 * rows_a)) != 0;
  // A piece's source = scalar base (row block, chunk) + per-lane offset (plane/half run, clamped position): the
  // per-lane part depends on the tile's first position only, so it survives from tile to tile when a row is one
  // tile (L <= 512), and nothing about the next tile has to be kept in registers.
  unsigned x_voff[XA];
  unsigned x_ok = 0, x_ok_n = 0;             // pieces that are real data (in range, unmasked); the rest is zero-filled
  auto build_voff = [&](int m0) {
#pragma unroll
    for (int it = 0; it < XA; ++it) {
      int ph, r;
      piece(it, ph, r);
      const int pc = min(max(m0 + r - a.pad_left, 0), a.L_in - 1);
      x_voff[it] = (unsigned)((min(ph, 3) * a.L_in + pc) * 16);
    }
  };
  // mask bytes of a tile's pieces: loaded two steps before they are turned into x_ok, all at once and
  // unconditionally (clamped addresses) - a conditional load per piece compiles into nine serial round trips
  unsigned char raw[XA];
  auto load_bytes = [&](const Tile2 &t) {
    if (a.mask_in != nullptr) {
#pragma unroll
      for (int it = 0; it < XA; ++it) {
        int ph, r;
        piece(it, ph, r);
        raw[it] = a.mask_in[(size_t)t.rowblk * a.L_in + min(max(t.m0 + r - a.pad_left, 0), a.L_in - 1)];
      }
    } else {
#pragma unroll
      for (int it = 0; it < XA; ++it) raw[it] = 1;
    }
  };
  auto build_ok = [&](const Tile2 &t) -> unsigned {       // consumes raw[]
    unsigned ok = 0;
#pragma unroll
    for (int it = 0; it < XA; ++it) {
      int ph, r;
      piece(it, ph, r);
      const int p = t.m0 + r - a.pad_left;
      if (p >= 0 && p < a.L_in && t.valid && ph < 4 && raw[it] != 0) ok |= 1u << it;
    }
    return ok;
  };
  const char *x_base = reinterpret_cast<const char *>(a.xh);
  const size_t x_cc_stride = (size_t)4 * a.L_in * 16;              // bytes per (row block, chunk)
  auto issue_x = [&](int rowblk, int cc, int buf) {
    const char *sb = x_base + ((size_t)rowblk * a.cc_in + cc) * x_cc_stride;
    const unsigned dst = ldsA + buf * (buf_items * 16);
#pragma unroll
    for (int it = 0; it < XA - 1; ++it) glds16(sb, x_voff[it], dst + it * (VT * 16));
    if (x_last_wave) {
      if ((x_has >> (XA - 1)) & 1u) glds16(sb, x_voff[XA - 1], dst + (XA - 1) * (VT * 16));
    }
  };
  auto zero_fill = [&](int buf, unsigned ok) {
    uint4 *A = Abuf + buf * buf_items;
#pragma unroll
    for (int it = 0; it < XA; ++it)
      if (((x_has >> it) & 1u) && !((ok >> it) & 1u)) A[tid + it * VT] = make_uint4(0u, 0u, 0u, 0u);
  };
  // ---- weight items of a thread: q = tid + it * VT -> [slice][plane][h][n] -----------------------
  unsigned w_voff[WA];
#pragma unroll
  for (int it = 0; it < WA; ++it) {
    const int q = (tid + it * VT) & (SLICE_ITEMS - 1);
    w_voff[it] = (unsigned)((((q >> 8) * VK * a.cc_in * 2 + ((q >> 7) & 1)) * VN + (q & (VN - 1))) * 16);
  }
  // step k of chunk pair j reads slices (c, t): k 0..3 -> (2j + k/2; 2(k&1), 2(k&1)+1), k 4 -> (2j, 4), (2j+1, 4)
  auto issue_w = [&](int j, int k, int slot) {
    const int ca = k < 4 ? 2 * j + (k >> 1) : 2 * j, cb = k < 4 ? ca : 2 * j + 1;
    const int ta = k < 4 ? 2 * (k & 1) : 4, tb = k < 4 ? ta + 1 : 4;
    const char *sa = reinterpret_cast<const char *>(a.wh) + ((size_t)(ta * a.cc_in * 2 + ca * 2) * VN) * 16;
    const char *sb = reinterpret_cast<const char *>(a.wh) + ((size_t)(tb * a.cc_in * 2 + cb * 2) * VN) * 16;
    const unsigned dst = ldsW + slot * (PAIR_ITEMS * 16);
    glds16(sa, w_voff[0], dst);
    glds16(sb, w_voff[1], dst + VT * 16);
  };

  f32x4 acc[8][4];       // [position block][channel block]
  auto zero_acc = [&]() {
#pragma unroll
    for (int pb = 0; pb < 8; ++pb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[pb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // fragment addresses (16-byte items)
  const int w_frag = ksel * SLICE_ITEMS + kh * VN + wn * 64 + li;              // + plane * 2 * VN + cb * 16 (+ slot * PAIR_ITEMS)
  const int x_frag = kh * rows_a + wm * 128 + li;                   // + plane * 2 * rows_a + pb * 16 + tap * dil (+ buf)

  const int n_pairs = a.cc_in >> 1;          // chunk pairs per tile (cc_in = 8 -> 4)
  Tile2 cur = tile_of(0), nxt = tile_of(1);
  build_voff(cur.m0);
  load_bytes(cur);
  x_ok = build_ok(cur);                      // the only exposed byte-load latency of the launch
  // prologue: chunk 0, the pair-slots of steps 0 and 1
  int G = 0;                                 // running chunk counter: chunk G lives in buffer G % 3
  int g = 0;                                 // running step counter: step g reads pair-slot g % 3
  issue_x(cur.rowblk, 0, 0);
  issue_w(0, 0, 0);
  issue_w(0, 1, 1);

  for (int pass = 0; pass < my_tiles; ++pass) {
    const bool last_tile = pass == my_tiles - 1;
    for (int j = 0; j < n_pairs; ++j) {
      const int b0 = G % XRING, b1 = (G + 1) % XRING;
      const bool last_pair = j == n_pairs - 1;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        // -- wait for this step's operands (counted: what may stay in flight was issued after them) --
        const bool tail = last_tile && last_pair && k >= 3;        // nothing is issued behind these steps
        if (tail) wait_vm<0>();
        else if (k == 1 || k == 3) {
          if (x_last_wave) wait_vm<WA + XA>(); else wait_vm<WA + XA - 1>();
        } else wait_vm<WA>();
        if (k == 0) zero_fill(b0, x_ok);
        if (k == 2) zero_fill(b1, x_ok);
        __syncthreads();
        // -- keep the DMA queues full: issued from inside the matrix-core stream, after the first half of the
        // step's MFMAs are queued, so that the issue cost runs under matrix-core time -------------------
        auto issue_step = [&]() {
          if (k == 2 && last_pair && !last_tile) x_ok_n = build_ok(nxt);      // bytes loaded at k = 0
          // weights of step g + 2 into the slot step g - 1 has just released
          int j2 = j, k2 = k + 2;
          if (k2 >= 5) { k2 -= 5; j2 = last_pair ? 0 : j + 1; }
          const bool beyond = last_tile && last_pair && k + 2 >= 5;
          if (!beyond) issue_w(j2, k2, (g + 2) % WRING);
          if (k == 0) {
            issue_x(cur.rowblk, 2 * j + 1, b1);                        // the pair's second chunk
            if (last_pair && !last_tile) load_bytes(nxt);
          }
          if (k == 2) {                                                // the next pair's first chunk
            if (!last_pair) issue_x(cur.rowblk, 2 * j + 2, (G + 2) % XRING);
            else if (!last_tile) {
              // the current tile has issued its last slice: the piece table moves on to the next tile
              if (nxt.m0 != cur.m0) build_voff(nxt.m0);
              issue_x(nxt.rowblk, 0, (G + 2) % XRING);
            }
          }
        };
        if (!V2_ISSUE_MID) issue_step();
        // -- matrix-core work ---------------------------------------------------------------------
        {
          const uint4 *Wp = Wbuf + (g % WRING) * PAIR_ITEMS + w_frag;
          const int tap = k < 4 ? 2 * (k & 1) + ksel : 4;
          const int xb = k < 2 ? b0 : (k < 4 ? b1 : (ksel ? b1 : b0));
          const uint4 *Xp = Abuf + xb * buf_items + x_frag + tap * a.dil;
          half8 wh[4], wl[4];
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) {
            const uint4 vh = Wp[cb * 16];
            const uint4 vl = Wp[2 * VN + cb * 16];
            wh[cb] = *reinterpret_cast<const half8 *>(&vh);
            wl[cb] = *reinterpret_cast<const half8 *>(&vl);
          }
#pragma unroll
          for (int ph = 0; ph < 2; ++ph) {             // four position blocks at a time
            half8 xh[4], xl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint4 vh = Xp[(ph * 4 + q) * 16];
              const uint4 vl = Xp[2 * rows_a + (ph * 4 + q) * 16];
              xh[q] = *reinterpret_cast<const half8 *>(&vh);
              xl[q] = *reinterpret_cast<const half8 *>(&vl);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int cb = 0; cb < 4; ++cb) {
                f32x4 &c = acc[ph * 4 + q][cb];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cb], xl[q], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[cb], xh[q], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[cb], xh[q], c, 0, 0, 0);
              }
            if (ph == 0 && !V2_ISSUE_MID) __builtin_amdgcn_sched_barrier(0);   // keep the second half's fragment reads behind the first half's MFMAs (register pressure)
            if (ph == 0 && V2_ISSUE_MID) {
              __builtin_amdgcn_sched_barrier(0);
              issue_step();
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        ++g;
      }
      G += 2;
    }
    // ---- tile finished -----------------------------------------------------------------------
    {
      float s = 0.f;
#pragma unroll
      for (int pb = 0; pb < 8; ++pb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) s += acc[pb][cb][0] + acc[pb][cb][1] + acc[pb][cb][2] + acc[pb][cb][3];
      if (s == 12345.678f) a.overflow[0] = 2;
    }
    zero_acc();
    cur = nxt;
    x_ok = x_ok_n;
    nxt = tile_of(pass + 2);
  }
}

template <unsigned EP>
int launch_v3(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  const int rows_a = VM + (VK - 1) * a.dil;
  const int smem = (XRING * 4 * rows_a + WRING * PAIR_ITEMS) * 16 + JG_EPI_ROWS * 2 * VN * 4;
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_v3_kernel<EP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int n_tiles = a.rows * ((a.L_out + VM - 1) / VM);
  int grid = e->n_cu;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((conv_v3_kernel<EP>), dim3((unsigned)grid), dim3(VT), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

}  // namespace

bool jg_conv_v3_supports(const ConvHArgs &a) {
  const int rows_a = VM + (VK - 1) * a.dil;
  const int smem = (XRING * 4 * rows_a + WRING * PAIR_ITEMS) * 16 + JG_EPI_ROWS * 2 * VN * 4;
  return a.k == 5 && !a.flat && a.ids == nullptr && a.lut == nullptr && (a.cc_in & 1) == 0 && a.cc_in >= 2 &&
         a.cout_pad == VN && smem <= 160 * 1024 && 4 * rows_a <= XA * VT && 4 * rows_a > (XA - 1) * VT;
}

int jg_launch_conv_v3(jg_engine *e, const ConvHArgs &a, hipStream_t s) { return launch_v3<0u>(e, a, s); }
