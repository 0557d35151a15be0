// Producer / consumer form of the split-f16 MaskedConv1D for the 128 -> 128 channel, 5-tap convs of the residual stacks
// (layers.py:1217-1280 MaskedConv1D.call, layers.py:1882-1915 ResidualBlock.call): the same arithmetic as
// conv_f16x3_kernel<5, ...> (jg_conv_f16_impl.h) - same operand layouts, same MFMA order per accumulator, same epilogue
// expressions, bit-identical outputs - with the work of a 256-position x 128-channel tile split by ROLE instead of by
// time:
//
//   * ONE workgroup of 8 waves per CU.  Waves 0-3 are MATH waves (one per SIMD): they read MFMA fragments from the LDS
//     operand ring and issue matrix-core instructions, nothing else - no DMA issue, no zero-fill, no epilogue.  Fragments
//     of the next 12-MFMA group are requested before the current group is issued, and a step's barrier is taken BEFORE
//     the step's last group is issued (its operands are already in registers), so the LDS latency at a step boundary is
//     covered by 12 MFMAs instead of idling the pipe.
//   * Waves 4-7 are HELPER waves (the SIMD partners of waves 0-3): they own the DMA ring (global_load_lds, counted
//     s_waitcnt vmcnt - the two-workgroup kernel's pipeline, unchanged), apply zero padding / input masks in LDS, and run
//     the fused epilogue of the PREVIOUS tile while the math waves work on the current one: one 32 x 32 accumulator
//     block per 16-channel chunk (8 blocks, 8 chunks), cut into two halves that sit behind the chunk's first two step
//     barriers, its stores behind the third.  The epilogue's vector instructions therefore issue beside another wave's
//     MFMAs on every SIMD all the time, instead of taking the matrix cores away from a whole workgroup for 20 - 40 %
//     of its life.
//   * At a tile boundary a math wave hands its 128 accumulator registers to its partner through a 16 KB LDS slot, in two
//     halves (two extra barriers per tile; the helper keeps them in registers - the slot is a transit buffer).
//   * Every vector-memory operation a helper issues in steady state is either an LDS-DMA (operands, residual shortcut
//     items, output-mask bytes) or a store, and sits in front of the step's operand DMAs in issue order, so the counted
//     waits of the operand ring stay exact and no compiler-inserted vmcnt wait can drain the ring.
//
// LDS: operand ring 75 264 B (dilation 3) + epilogue table 4 096 + accumulator transit 65 536 + shortcut staging 17 408.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "jg_common.h"
#include "jg_conv_dev.h"

namespace {

constexpr int PK = 5;                 // taps
constexpr int PCC = 8;                // 16-channel input chunks: Cin = 128
constexpr int PT = 512;               // threads: 4 math + 4 helper waves
constexpr int X_ITEMS = 1024;         // 16-byte items of one pair's transit slot: 64 registers x 64 lanes x 4 B
constexpr int S_ITEMS = 4 * 64 + 16;  // 16-byte items of one helper's staging area: 4 shortcut items per lane + 64 mask dwords

#ifndef JG_PC_LATE
#define JG_PC_LATE 1                  // take a step's barrier before the previous step's last MFMA group is issued
#endif
#ifndef JG_PC_PRIO
#define JG_PC_PRIO 3                  // s_setprio of the math waves
#endif
#ifndef JG_PC_RINGFENCE
#define JG_PC_RINGFENCE 1             // scheduling fences around the ring's DMA pieces inside a group's MFMAs
#endif
#ifndef JG_PC_HPRIO
#define JG_PC_HPRIO 0                 // s_setprio of the helper waves
#endif

#ifdef JG_STAMP
static __device__ unsigned long long jg_pc_stamp_acc[16];
#define PC_ST_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime(); const unsigned long long st_t0 = st_t
#define PC_ST(idx) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[idx] += n_ - st_t; st_t = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PC_ST_END(base) do { st_[7] = __builtin_amdgcn_s_memtime() - st_t0; if (lane == 0) { for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&jg_pc_stamp_acc[(base) + q_], st_[q_]); } } while (0)
#else
#define PC_ST_DECL
#define PC_ST(idx)
#define PC_ST_END(base)
#endif
#if defined(JG_STAMP) && defined(JG_STAMP_M)
#define PC_STM(idx) PC_ST(idx)      // math-wave step stamps: every s_memtime drains lgkmcnt, i.e. the fragment lookahead
#else
#define PC_STM(idx)
#endif

__device__ __forceinline__ void lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void bar() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// 4-byte-per-lane global -> LDS DMA of one unsigned byte per lane (zero-extended dword at M0 + lane*4)
__device__ __forceinline__ void glds_ubyte(const void *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_ubyte %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}

// DIL: the dilation as a compile-time constant - every LDS offset of the fragment reads and of the DMA destinations folds
// into an instruction immediate (the math waves have no registers to spare for address arithmetic)
// F32OUT: the conv's output stays f32 (the last conv of a stack: masked max pool fused, or an f32 reader behind it) - no
// F16S re-split.
template <unsigned EP, bool FLAT, int DIL, bool F32OUT>
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_pc_kernel(ConvHArgs a) {
  constexpr bool HAS_ADD = (EP & JG_EP_ADD) != 0;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_math = wid < 4;
  const int w4 = wid & 3;                               // pair index: math wave w4 and helper wave w4 + 4 share a SIMD
  const int wm = w4 >> 1, wn = w4 & 1;                  // the pair's 128-position x 64-channel quarter of the tile
  const int i = lane & 31, h = lane >> 5;
  const int ptid = tid & 255;                           // thread index inside the role (0..255)
  const int vgrid = (int)gridDim.x;
  int vb = (int)blockIdx.x;
  if ((vgrid & 7) == 0) vb = (vb & 7) * (vgrid >> 3) + (vb >> 3);     // XCD-aware tile order (see conv_f16x3_kernel)
  // LDS carve (16-byte units)
  constexpr int rows_a = HM + (PK - 1) * DIL;
  constexpr int a_items = 4 * rows_a;                    // [4 ph][rows_a]
  uint4 *Abuf = lds;                                     // [2 bufs][a_items]
  uint4 *Wbuf = lds + 2 * a_items;                       // [5 slots][2 planes][2 h][HN]
  float *epiL = reinterpret_cast<float *>(Wbuf + PK * W_ITEMS);      // [JG_EPI_ROWS][2][HN]
  uint4 *Xbuf = Wbuf + PK * W_ITEMS + JG_EPI_ROWS * 2 * HN / 4;      // [4 pairs][X_ITEMS]
  uint4 *Sbuf = Xbuf + 4 * X_ITEMS;                                   // [4 helpers][S_ITEMS]
  unsigned char *Bbuf = reinterpret_cast<unsigned char *>(Sbuf + 4 * S_ITEMS);   // [A_ITERS][256] input-mask bytes of the next tile
  for (int q = tid; q < a.n_epi_rows * 2 * HN; q += PT) epiL[q] = a.epi[q];   // visible after the first barrier
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  const int n_tiles = FLAT ? a.flat_tiles : a.rows * a.tiles_m;
  int my_pairs = 0;
  if (vb < n_tiles) my_pairs = (n_tiles - 1 - vb) / vgrid + 1;
  if (my_pairs == 0) return;
  uint4 *Xp = Xbuf + w4 * X_ITEMS + lane;                // this pair's transit slot, lane column
  f32x16 acc[4][2];                                      // [tm: position block][tn: channel block]; both roles

  auto resolve = [&](const Tile &tile, int local, int len, int &row, int &p) -> bool {
    if constexpr (!FLAT) {
      row = tile.rowblk;
      p = tile.m0 + local;
      return tile.valid && p >= 0 && p < len;
    }
    const int v = tile.T * HM + local;
    if (v < 0 || !tile.valid) { row = 0; p = 0; return false; }
    int g, u, f;
    udivmod24(v, a.flat_wp, a.flat_inv_wp, g, u);
    udivmod24(u, a.flat_p, a.flat_inv_p, f, p);
    row = g * a.flat_frames + f;
    return f < a.flat_frames && p < len && row < a.rows;
  };
  auto tile_of = [&](int pass, Tile &t) {
    const int T = vb + pass * vgrid;
    const int Tc = min(T, n_tiles - 1);
    t.rowblk = Tc / a.tiles_m;
    t.m0 = (Tc - t.rowblk * a.tiles_m) * HM;
    t.valid = T < n_tiles;
    t.T = Tc;
  };
  // per-thread activation piece coordinates (as in conv_f16x3_kernel): piece q = ptid + it*256 -> (plane/half ph, row r);
  // ph >= 4: no piece (only the last iteration can run past the slice).  Recomputed where needed: divisions by constants.
  auto piece_ph = [&](int it) -> int { return (ptid + it * HT) / rows_a; };
  auto piece_row = [&](int it) -> int { return (ptid + it * HT) % rows_a; };
  const uint8_t *bsrc = a.mask_in;
  auto piece_pos = [&](const Tile &tl, int it, int &pc, bool &inr) -> int {
    int rb, p;
    inr = resolve(tl, piece_row(it) - a.pad_left, a.L_in, rb, p) && piece_ph(it) < 4;
    pc = min(max(p, 0), a.L_in - 1);
    if constexpr (FLAT) rb = min(rb, a.rows - 1);
    return rb;
  };

  if (is_math) {
    // =========================================== MATH WAVE ===============================================
    // MFMA stream + the operand ring of conv_f16x3_kernel (same slots, same issue points, same counted waits); no
    // epilogue.  Nothing else this wave issues touches vector memory, so the counts are exact.
    lgkm0();                                             // the epilogue-table writes above
    __builtin_amdgcn_s_setprio(JG_PC_PRIO);
    PC_ST_DECL;
    const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);                       // + buf*a_items*16 + it*4096
    const unsigned ldsW = __builtin_amdgcn_readfirstlane(lds0 + 2 * a_items * 16 + wid * 1024);   // + slot*8192 + it*4096
    unsigned w_voff[W_ITERS];
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
      const int q = ptid + it * HT;          // [plane][h][n]
      w_voff[it] = (unsigned)((((q >> 8) * PK * PCC * 2 + ((q >> 7) & 1)) * HN + (q & (HN - 1))) * 16);
    }
    unsigned raw[A_ITERS];
    unsigned x_voff[A_ITERS];
    unsigned x_ok = 0;
    auto build_pieces = [&](const Tile &tl) {      // consumes raw[]
      x_ok = 0;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        int pc; bool inr;
        const int rb = piece_pos(tl, it, pc, inr);
        const int ph = piece_ph(it) & 3;
        x_voff[it] = (unsigned)(((rb * PCC * 4 + ph) * a.L_in + pc) * 16);
        if (inr && raw[it] != 0) x_ok |= 1u << it;
      }
    };
    const char *x_base = reinterpret_cast<const char *>(a.xh);
    const unsigned x_cc_stride = 4u * (unsigned)a.L_in * 16u;   // bytes per chunk
    const bool x_last_wave = __builtin_amdgcn_readfirstlane((int)((A_ITERS - 1) * HT + wid * 64 < 4 * rows_a)) != 0;
    auto issue_w = [&](int cc, int t) {        // weight slice (cc, t) -> ring slot t
      const char *sb = reinterpret_cast<const char *>(a.wh) + ((size_t)(t * PCC * 2 + cc * 2) * HN) * 16;
#pragma unroll
      for (int it = 0; it < W_ITERS; ++it) glds16(sb, w_voff[it], ldsW + t * (W_ITEMS * 16) + it * (HT * 16));
    };
    auto issue_x = [&](int cc, int buf) {      // the tile's activation slice of chunk cc
      const char *sb = x_base + (size_t)cc * x_cc_stride;
      const unsigned dst = ldsA + buf * (a_items * 16);
#pragma unroll
      for (int it = 0; it < A_ITERS - 1; ++it) glds16_nt(sb, x_voff[it], dst + it * (HT * 16));
      if (x_last_wave) {                        // wave-uniform: the counted waits must know how many DMAs are in flight
        if (piece_ph(A_ITERS - 1) < 4) glds16_nt(sb, x_voff[A_ITERS - 1], dst + (A_ITERS - 1) * (HT * 16));
      }
    };
    auto zero_fill = [&](int buf) {
      uint4 *A = Abuf + buf * a_items;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it)
        if (piece_ph(it) < 4 && !((x_ok >> it) & 1u)) A[ptid + it * HT] = make_uint4(0u, 0u, 0u, 0u);
    };
    const uint4 *Wb = Wbuf + h * HN + wn * 64 + i;       // + t*W_ITEMS + plane*2*HN + tn*32
    const int x_frag = h * rows_a + wm * 128 + i;        // + plane*2*rows_a + tm*32 + t*dil
    constexpr int dil = DIL;
    struct XF { uint4 h[2], l[2]; };
    struct WF { uint4 h[2], l[2]; };
    XF xf[2];
    WF wf[2];
    auto ldx = [&](XF &f, const uint4 *A, int t, int tp) {
#pragma unroll
      for (int tq = 0; tq < 2; ++tq) {
        f.h[tq] = A[(tp * 2 + tq) * 32 + t * dil];
        f.l[tq] = A[2 * rows_a + (tp * 2 + tq) * 32 + t * dil];
      }
    };
    auto ldw = [&](WF &f, int t) {
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        f.h[tn] = Wb[t * W_ITEMS + tn * 32];
        f.l[tn] = Wb[t * W_ITEMS + 2 * HN + tn * 32];
      }
    };
    // 12 MFMAs: two position blocks x two channel blocks x (hi.lo, lo.hi, hi.hi), in the order of conv_f16x3_kernel;
    // ZERO: the block's first product of the tile starts from C = 0 (no accumulator clearing between tiles)
    auto mm = [&](auto zero_c, const WF &w, const XF &x, int tp, auto &&between) {
      constexpr bool ZERO = decltype(zero_c)::value;
#ifdef JG_PC_NOMFMA            // timing experiment: fragments are still read, the matrix cores stay idle (results are garbage)
      acc[tp * 2][0][0] += __uint_as_float(w.h[0].x ^ w.l[1].y ^ x.h[0].z ^ x.l[1].w ^ w.h[1].x ^ w.l[0].y ^ x.h[1].z ^ x.l[0].w);
      for (int k = 0; k < 4; ++k) between(k);
      return;
#endif
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          f32x16 &c = acc[tp * 2 + tq][tn];
          const half8 wh = *reinterpret_cast<const half8 *>(&w.h[tn]), wl = *reinterpret_cast<const half8 *>(&w.l[tn]);
          const half8 xh = *reinterpret_cast<const half8 *>(&x.h[tq]), xl = *reinterpret_cast<const half8 *>(&x.l[tq]);
          if constexpr (ZERO) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, z, 0, 0, 0);
          } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
          between(tq * 2 + tn);
        }
    };
    // accumulator hand-off: blocks tm = 2*half, 2*half + 1 -> the pair's transit slot (register r of lane l lands where
    // the helper's register r of lane l reads it back)
    auto xwrite = [&](int half) {
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const f32x16 &c = acc[half * 2 + tq][tn];
            Xp[((tq * 2 + tn) * 4 + r4) * 64] = make_uint4(__float_as_uint(c[4 * r4]), __float_as_uint(c[4 * r4 + 1]),
                                                          __float_as_uint(c[4 * r4 + 2]), __float_as_uint(c[4 * r4 + 3]));
          }
    };
    // ---- prologue: the pipeline of pass 0 ----
    Tile cur, np;
    tile_of(0, cur);
    tile_of(1, np);
    if (bsrc != nullptr) {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        int pc; bool inr;
        const int rb = piece_pos(cur, it, pc, inr);
        raw[it] = bsrc[(size_t)rb * a.L_in + pc];
      }
    } else {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) raw[it] = 1;
    }
    build_pieces(cur);                 // the only exposed byte-load latency of the launch
    issue_x(0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) issue_w(0, t);
    wait_vm<2 * W_ITERS>();
    zero_fill(0);
    lgkm0();
    bar();                                               // step A of (pass 0, chunk 0)
    ldw(wf[0], 0);
    ldx(xf[0], Abuf + x_frag, 0, 0);
    for (int pass = 0; pass < my_pairs; ++pass) {
      const bool more = pass + 1 < my_pairs;
      for (int cp = 0; cp < PCC / 2; ++cp) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {           // chunk cc = 2*cp + half reads activation buffer `half`
          const int cc = 2 * cp + half;
          const bool last_chunk = half == 1 && cp == PCC / 2 - 1;
          const bool tail = last_chunk && !more;          // nothing is issued behind this chunk
          const int ncc = last_chunk ? 0 : cc + 1;
          const uint4 *A = Abuf + half * a_items + x_frag;
          const uint4 *An = Abuf + (half ^ 1) * a_items + x_frag;
#pragma unroll
          for (int g = 0; g < 10; ++g) {                 // group g: tap g/2, position-block pair g%2
            const int t = g >> 1, tp = g & 1;
            const bool step_end = g == 3 || g == 7 || g == 9;
            const bool pass_end = g == 9 && last_chunk;
            if (step_end && !pass_end) {
              // the next step's operands: counted wait, padding / mask zeros, publish.  Taken BEFORE this step's last
              // group is issued - its fragments are in registers (lgkmcnt 0), so the slots may be refilled - and the
              // first fragments of the next step are requested under those 12 MFMAs.
              if (g == 3) {
                if (tail) wait_vm<W_ITERS>();
                else if (x_last_wave) wait_vm<W_ITERS + A_ITERS>();
                else wait_vm<W_ITERS + A_ITERS - 1>();
              } else if (g == 7) {
                if (tail) wait_vm<0>();
                else if (x_last_wave) wait_vm<2 * W_ITERS + A_ITERS>();
                else wait_vm<2 * W_ITERS + A_ITERS - 1>();
              } else {
                wait_vm<2 * W_ITERS>();
                zero_fill(half ^ 1);
              }
              lgkm0();
              PC_STM(0);
              bar();
              PC_STM(1);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!pass_end) {                             // fragments of the group after g
              if (g < 9) {
                ldx(xf[(g + 1) & 1], A, (g + 1) >> 1, (g + 1) & 1);
                if (tp == 1) ldw(wf[(half + t + 1) & 1], t + 1);
              } else {
                ldx(xf[0], An, 0, 0);
                ldw(wf[(half ^ 1) & 1], 0);
              }
            }
            // (no fence here: the fragment reads above are independent of this group's MFMAs - the group barriers below
            // deal them out one per MFMA, so that their issue runs in the matrix cores' shadow instead of in front of it)
            // the ring's DMA pieces of this step ride in the same shadow, one or two behind each accumulator block's three MFMAs
            auto ring = [&](int k) {
              if (g != 0 && g != 4 && g != 8) return;
#if JG_PC_RINGFENCE
              __builtin_amdgcn_sched_barrier(0);
#endif
              if (g == 0) {
                const char *sbw = reinterpret_cast<const char *>(a.wh) + ((size_t)(4 * PCC * 2 + cc * 2) * HN) * 16;
                if (k == 0) glds16(sbw, w_voff[0], ldsW + 4 * (W_ITEMS * 16));
                if (k == 1) glds16(sbw, w_voff[1], ldsW + 4 * (W_ITEMS * 16) + HT * 16);
                if (!tail) {
                  if (k == 0 && last_chunk) {
                    if (bsrc != nullptr) {
#pragma unroll
                      for (int it = 0; it < A_ITERS; ++it) raw[it] = Bbuf[it * 256 + ptid];     // fetched by the helpers
                    }
                    build_pieces(np);
                  }
                  const char *sbx = x_base + (size_t)ncc * x_cc_stride;
                  const unsigned dst = ldsA + (half ^ 1) * (a_items * 16);
                  if (k == 1) glds16_nt(sbx, x_voff[0], dst);
                  if (k == 2) { glds16_nt(sbx, x_voff[1], dst + HT * 16); glds16_nt(sbx, x_voff[2], dst + 2 * (HT * 16)); }
                  if (k == 3) {
                    glds16_nt(sbx, x_voff[3], dst + 3 * (HT * 16));
                    if (x_last_wave) {
                      if (piece_ph(A_ITERS - 1) < 4) glds16_nt(sbx, x_voff[A_ITERS - 1], dst + (A_ITERS - 1) * (HT * 16));
                    }
                  }
                }
              } else if (!tail) {
                const int t0 = g == 4 ? 0 : 2;           // slices (ncc, t0), (ncc, t0 + 1): one piece per slot
                const int tt = t0 + (k >> 1);
                const char *sbw = reinterpret_cast<const char *>(a.wh) + ((size_t)(tt * PCC * 2 + ncc * 2) * HN) * 16;
                glds16(sbw, w_voff[k & 1], ldsW + tt * (W_ITEMS * 16) + (k & 1) * (HT * 16));
              }
#if JG_PC_RINGFENCE
              __builtin_amdgcn_sched_barrier(0);
#endif
            };
            if (half == 0 && g < 2) {
              if (cp == 0) mm(std::true_type{}, wf[(half + t) & 1], xf[g & 1], tp, ring);
              else mm(std::false_type{}, wf[(half + t) & 1], xf[g & 1], tp, ring);
            } else {
              mm(std::false_type{}, wf[(half + t) & 1], xf[g & 1], tp, ring);
            }
            if (g != 0 && g != 4 && g != 8) {
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // ---- tile finished: hand the accumulators to the helper ----
      PC_ST(0);
      xwrite(0);
      lgkm0();
      bar();                                             // X1: first half published
      if (more) { wait_vm<2 * W_ITERS>(); zero_fill(0); }        // (the next tile's first operands, while the helper reads)
      bar();                                             // X2: the helper has it in registers
      xwrite(1);
      lgkm0();
      bar();                                             // step A of the next pass / X3 after the last one
      PC_ST(2);
      if (more) {
        ldw(wf[0], 0);
        ldx(xf[0], Abuf + x_frag, 0, 0);
      }
      cur = np;
      tile_of(pass + 2, np);
    }
    PC_ST_END(0);
    return;
  }

  // ============================================= HELPER WAVE =================================================
  const int hw = wid - 4;
  __builtin_amdgcn_s_setprio(JG_PC_HPRIO);
  PC_ST_DECL;
  uint4 *Sp = Sbuf + hw * S_ITEMS;                       // this wave's staging: [4 items][64 lanes] + 64 mask dwords
  const unsigned ldsS = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((Sbuf - lds) + hw * S_ITEMS) * 16u);
  unsigned char nb_raw[A_ITERS] = {1, 1, 1, 1, 1};

  // ---- epilogue (expressions of conv_f16x3_kernel's compiled patterns, tanh-GELU) ------------------------
  // Per tile and position block tm this lane's output position is resolved ONCE: ob[tm] = item offset (16-byte units) of
  // its hi-plane item of channel group 8*wn + h... (tn, j) and the lo plane are constant strides away; om[tm] = offset of
  // its output-mask byte; bit tm of olive = the position exists.  Dead lanes point at position 0 of their row.
  const unsigned L4 = 4u * (unsigned)a.L_out;
  unsigned ob[4], om[4], olive = 0;
  auto tile_positions = [&](const Tile &tile) {
    olive = 0;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      int row, p;
      bool live = resolve(tile, (wm * 4 + tm) * 32 + i, a.L_out, row, p);
      const int mc = live ? p : 0;
      if constexpr (FLAT) {
        if (!live) row = min(max(row, 0), a.rows - 1);
      }
      ob[tm] = (unsigned)(((row * 8 + 4 * wn) * 4 + h) * a.L_out + mc);
      om[tm] = (unsigned)(row * a.L_out + mc);
      if (live) olive |= 1u << tm;
    }
  };
  // request what block (tm, tn) needs from memory: NL LDS-DMAs into the wave's staging area
  auto epi_request = [&](unsigned obase, unsigned omask, int tn) {
    if constexpr (HAS_ADD) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned it4 = obase + (unsigned)(2 * tn + j) * L4;
        glds16_nt(a.addh, it4 * 16u, ldsS + (2 * j) * 1024);
        glds16_nt(a.addh, (it4 + 2u * (unsigned)a.L_out) * 16u, ldsS + (2 * j + 1) * 1024);
      }
    }
    // (no output mask: any readable byte - the value is ignored)
    const void *mb = a.mask_out != nullptr ? static_cast<const void *>(a.mask_out) : static_cast<const void *>(a.wh);
    glds_ubyte(mb, a.mask_out != nullptr ? omask : 0u, ldsS + 4 * 1024);
  };
  uint2 sh[4], sl[4];                // residual shortcut of the block in hand: this lane's 4 channels of groups 0..3, hi / lo
  float mk = 0.f;                    // its output-mask value (1 / 0), 0 on dead positions
  auto epi_collect = [&](int tm) {   // ... out of the staging area (after the wait that covers those DMAs)
    const unsigned mbyte = reinterpret_cast<const unsigned *>(Sp + 4 * 64)[lane];
    mk = (((olive >> tm) & 1u) != 0u && (a.mask_out == nullptr || mbyte != 0u)) ? 1.f : 0.f;
    if constexpr (HAS_ADD) {
      // lane (i, h) staged the whole item of group 2j + h at its position; it needs channels 4h..4h+3 of groups 2j and 2j+1
      const uint2 *S2 = reinterpret_cast<const uint2 *>(Sp);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          sh[2 * j + gg] = S2[((2 * j) * 64 + gg * 32 + i) * 2 + h];
          sl[2 * j + gg] = S2[((2 * j + 1) * 64 + gg * 32 + i) * 2 + h];
        }
    }
  };
  float vmax = 0.f;                  // f16-range guard: running max |output| (drops NaN) and "an output is NaN"
  bool vnan = false;
  f32x2 nmd2[8];                     // NMD sums of the channel block in hand (pairs of channels)
  uint4 outv[4];                     // converted outputs of the block in hand: [j][hi | lo] items, stored a step later
  auto swap32 = [](unsigned &lo_half_keeps, unsigned &hi_half_keeps) {
    const auto r = __builtin_amdgcn_permlane32_swap(lo_half_keeps, hi_half_keeps, false, false);
    lo_half_keeps = r[0];
    hi_half_keeps = r[1];
  };
  // half j (channel groups 2j, 2j+1: accumulator registers 8j .. 8j+7) of one 32 x 32 block.  Stage-major over the four
  // channel pairs, so that the independent chains interleave; F16S outputs go to outv[2j], outv[2j+1] (hi, lo item of group
  // 2j + h), f32 outputs stay in the accumulator registers.
  auto epi_half = [&](f32x16 &x, int tn, int j) {
#ifdef JG_PC_NOEPI             // timing experiment: no epilogue arithmetic (results are garbage)
    return;
#endif
    const int nb = (wn * 2 + tn) * 32;
    f32x2 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = f32x2{x[8 * j + 2 * q], x[8 * j + 2 * q + 1]};
    auto st_affine = [&](int row) {
      const float *pr = epiL + (row * 2) * HN + nb + 4 * h + 16 * j;
#ifdef JG_PC_NOPARAM           // timing experiment: no LDS reads of the epilogue parameters (results are garbage)
      const float4 sc0 = make_float4(1.f, 1.5f, 0.5f, 0.25f), sc1 = sc0, of0 = make_float4(0.5f, 0.25f, 0.125f, 1.f), of1 = of0;
      (void)pr;
#else
      const float4 sc0 = *reinterpret_cast<const float4 *>(pr), sc1 = *reinterpret_cast<const float4 *>(pr + 8);
      const float4 of0 = *reinterpret_cast<const float4 *>(pr + HN), of1 = *reinterpret_cast<const float4 *>(pr + HN + 8);
#endif
      v[0] = __builtin_elementwise_fma(v[0], f32x2{sc0.x, sc0.y}, f32x2{of0.x, of0.y});
      v[1] = __builtin_elementwise_fma(v[1], f32x2{sc0.z, sc0.w}, f32x2{of0.z, of0.w});
      v[2] = __builtin_elementwise_fma(v[2], f32x2{sc1.x, sc1.y}, f32x2{of1.x, of1.y});
      v[3] = __builtin_elementwise_fma(v[3], f32x2{sc1.z, sc1.w}, f32x2{of1.z, of1.w});
    };
    auto st_add = [&]() {
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * j + gg;
        v[2 * gg] += f32x2{mix_sum<0>(sh[g].x, sl[g].x), mix_sum<1>(sh[g].x, sl[g].x)};
        v[2 * gg + 1] += f32x2{mix_sum<0>(sh[g].y, sl[g].y), mix_sum<1>(sh[g].y, sl[g].y)};
      }
    };
    auto st_gelu = [&]() {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = fast_gelu2(v[q]);
    };
    auto st_nmd = [&]() {
      const f32x2 mk2 = {mk, mk};
#pragma unroll
      for (int q = 0; q < 4; ++q) nmd2[4 * j + q] = __builtin_elementwise_fma(v[q], mk2, nmd2[4 * j + q]);
    };
    constexpr int N1 = (EP >> 1) & 3, N2 = (EP >> 6) & 3;
    static_assert(N1 != 2 && N2 != 2, "DyT patterns are not built into the producer / consumer kernel");
    st_affine(0);
    if constexpr (EP & JG_EP_NMD1) st_nmd();
    if constexpr (N1 == 1) st_affine(1);
    if constexpr (EP & JG_EP_ADD) st_add();
    if constexpr (EP & JG_EP_ACT1) st_gelu();
    if constexpr (EP & JG_EP_NMD2) st_nmd();
    if constexpr (N2 == 1) st_affine(N1 ? 2 : 1);
    if constexpr (EP & JG_EP_ACT2) st_gelu();
    if constexpr (F32OUT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { x[8 * j + 2 * q] = v[q].x; x[8 * j + 2 * q + 1] = v[q].y; }
    } else {
      unsigned hp[4], lp[4];             // packed hi / lo halves of the four pairs
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        const half2_t hh = {(_Float16)v[q].x, (_Float16)v[q].y};
        hp[q] = *reinterpret_cast<const unsigned *>(&hh);
        const half2_t ll = {(_Float16)mix_rem<0>(v[q].x, hp[q]), (_Float16)mix_rem<1>(v[q].y, hp[q])};
        lp[q] = *reinterpret_cast<const unsigned *>(&ll);
      }
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0].x), fabsf(v[0].y))), fmaxf(fabsf(v[1].x), fabsf(v[1].y)));
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[2].x), fabsf(v[2].y))), fmaxf(fabsf(v[3].x), fabsf(v[3].y)));
      vnan = vnan || __builtin_isunordered(v[0].x, v[0].y) || __builtin_isunordered(v[1].x, v[1].y) ||
             __builtin_isunordered(v[2].x, v[2].y) || __builtin_isunordered(v[3].x, v[3].y);
      // pairs 0, 1 = group 2j (this lane's channels 4h..4h+3), pairs 2, 3 = group 2j+1: lane half h keeps group 2j+h whole
      swap32(hp[0], hp[2]); swap32(hp[1], hp[3]);
      swap32(lp[0], lp[2]); swap32(lp[1], lp[3]);
      outv[2 * j] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
      outv[2 * j + 1] = make_uint4(lp[0], lp[1], lp[2], lp[3]);
    }
  };
  // stores of half j of a block in channel block tn whose position data is (obase, omask, live)
  auto store_half = [&](const f32x16 &x, unsigned obase, unsigned omask, bool live, int tn, int j) {
#ifdef JG_PC_NOSTORE           // timing experiment: outputs are not stored (results are garbage)
    if (outv[2 * j].x == 0x12345678u && outv[2 * j + 1].y == 0x9abcdef0u) a.overflow[0] = 8;
    return;
#endif
    if (live) {
      if constexpr (!F32OUT) {
        uint4 *yh = reinterpret_cast<uint4 *>(a.y);
#ifdef JG_PC_SAMESTORE          // timing experiment: every store of a wave lands in one 2 KB window of the output (results are garbage)
        const unsigned it4 = (unsigned)(wid * 128 + lane);
        (void)obase;
#else
        const unsigned it4 = obase + (unsigned)(2 * tn + j) * L4;
#endif
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 vhi = {outv[2 * j].x, outv[2 * j].y, outv[2 * j].z, outv[2 * j].w};
        const u32x4 vlo = {outv[2 * j + 1].x, outv[2 * j + 1].y, outv[2 * j + 1].z, outv[2 * j + 1].w};
#ifdef JG_PC_PLAINSTORE
        *reinterpret_cast<u32x4 *>(yh + it4) = vhi;
        *reinterpret_cast<u32x4 *>(yh + it4 + 2u * (unsigned)a.L_out) = vlo;
#else
        __builtin_nontemporal_store(vhi, reinterpret_cast<u32x4 *>(yh + it4));
        __builtin_nontemporal_store(vlo, reinterpret_cast<u32x4 *>(yh + it4 + 2u * (unsigned)a.L_out));
#endif
      } else if (a.pool_out == nullptr) {
        float *yf = reinterpret_cast<float *>(a.y) + (size_t)omask * a.cout + (wn * 2 + tn) * 32 + 4 * h + 16 * j;
        *reinterpret_cast<float4 *>(yf) = make_float4(x[8 * j], x[8 * j + 1], x[8 * j + 2], x[8 * j + 3]);
        *reinterpret_cast<float4 *>(yf + 8) = make_float4(x[8 * j + 4], x[8 * j + 5], x[8 * j + 6], x[8 * j + 7]);
      }
    }
  };
#define JG_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), (ctrl), 0xf, 0xf, false))
  auto lane_reduce = [&](const float (&in)[16], auto op) -> float {
    const bool b2 = (i & 4) != 0, b1 = (i & 2) != 0, b0 = (i & 1) != 0, b3 = (i & 8) != 0;
    float s8[8], s4[4], s2[2];
#pragma unroll
    for (int q = 0; q < 8; ++q) {        // partner i ^ 7 (row_half_mirror)
      const float keep = b2 ? in[8 + q] : in[q], send = b2 ? in[q] : in[8 + q];
      s8[q] = op(keep, JG_DPP(send, 0x141));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {        // partner i ^ 2 (quad_perm [2,3,0,1])
      const float keep = b1 ? s8[4 + q] : s8[q], send = b1 ? s8[q] : s8[4 + q];
      s4[q] = op(keep, JG_DPP(send, 0x4e));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {        // partner i ^ 1 (quad_perm [1,0,3,2])
      const float keep = b0 ? s4[2 + q] : s4[q], send = b0 ? s4[q] : s4[2 + q];
      s2[q] = op(keep, JG_DPP(send, 0xb1));
    }
    const float keep1 = b3 ? s2[1] : s2[0], send1 = b3 ? s2[0] : s2[1];
    const float v = op(keep1, JG_DPP(send1, 0x128));      // partner i ^ 8 (row_ror:8)
    return op(v, __shfl_xor(v, 16, 32));                  // partner i ^ 16
  };
#undef JG_DPP
  auto reduced_slot = [&](const Tile &tile, int tn, int &ch) -> size_t {
    const int r = 8 * (int)((i & 4) != 0) + 4 * (int)((i & 2) != 0) + 2 * (int)((i & 1) != 0) + (int)((i & 8) != 0);
    ch = (wn * 2 + tn) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
    return ((size_t)tile.T * 2 + wm) * a.cout + ch;        // one partial row per wave strip (128 positions)
  };
  constexpr bool HAS_NMD = (EP & (JG_EP_NMD1 | JG_EP_NMD2)) != 0;
  auto nmd_flush = [&](const Tile &tile, int tn) {
    float na[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) { na[2 * q] = nmd2[q].x; na[2 * q + 1] = nmd2[q].y; }
    const float v = lane_reduce(na, [](float x, float y) { return x + y; });
    int ch;
    const size_t slot = reduced_slot(tile, tn, ch);
    if (i < 16 && tile.valid) a.nmd_out[slot] = v;
  };
  // fused masked global max pool (layers.py:496-538): running max of the channel block in hand over its position blocks
  // (the block outputs are not stored), reduced over the lanes once per channel block like the NMD sums
  float pool_acc[16];
  auto pool_block = [&](const f32x16 &x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) pool_acc[r] = mk != 0.f ? fmaxf(pool_acc[r], x[r]) : pool_acc[r];
  };
  auto pool_flush = [&](const Tile &tile, int tn) {
    const float v = lane_reduce(pool_acc, [](float x, float y) { return fmaxf(x, y); });
    int ch;
    const size_t slot = reduced_slot(tile, tn, ch);
    if (i < 16 && tile.valid) a.pool_out[slot] = v;
  };
  // accumulator hand-off, helper side.  The first half (blocks tm = 0, 1) is taken into registers between the X1 / X2
  // barriers; the second half stays in the transit slot for the whole pass and is read block by block when its turn
  // comes (tm = 2, 3 at chunks 2, 3, 6, 7): the helper never holds more than 80 accumulator registers.
  f32x16 hacc[2][2];
  f32x16 blk;
  auto xread_block = [&](f32x16 &c, int tq, int tn) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const uint4 v = Xp[((tq * 2 + tn) * 4 + r4) * 64];
      c[4 * r4] = __uint_as_float(v.x); c[4 * r4 + 1] = __uint_as_float(v.y);
      c[4 * r4 + 2] = __uint_as_float(v.z); c[4 * r4 + 3] = __uint_as_float(v.w);
    }
  };

  Tile cur, np, et;                  // tile the math waves work on / the next one / the one whose accumulators this wave holds
  tile_of(0, cur);
  tile_of(1, np);
  et = cur;
  tile_positions(cur);
#pragma unroll
  for (int g = 0; g < 4; ++g) sh[g] = sl[g] = make_uint2(0u, 0u);
#pragma unroll
  for (int q = 0; q < 4; ++q) outv[q] = make_uint4(0u, 0u, 0u, 0u);

  // Pass `my_pairs` is the drain: no barriers - only the last tile's epilogue, through the same code.  The helper owns no
  // part of the operand ring: besides the epilogue it fetches the next tile's input-mask bytes for the math waves' piece
  // table (plain loads - nothing here counts vmcnt) and joins the step barriers.
  for (int pass = 0; pass <= my_pairs; ++pass) {
    const bool drain = pass == my_pairs;
    const bool epi = pass > 0;                         // this wave holds a tile's accumulators
    if (epi) {
      PC_ST(4);
      bar();                                           // X1: first half of the finished tile's accumulators
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
#pragma unroll
        for (int tn2 = 0; tn2 < 2; ++tn2) xread_block(hacc[tq][tn2], tq, tn2);
      lgkm0();
      bar();                                           // X2
      PC_ST(2);
    }
#pragma unroll
    for (int cc = 0; cc < PCC; ++cc) {
      const bool last_chunk = cc == PCC - 1;
      const int tm = cc & 3, tn = cc >> 2;             // the accumulator block this chunk's steps carry
      f32x16 &xb = tm < 2 ? hacc[tm & 1][tn] : blk;    // the block in hand (cc is a compile-time value)
      // ---- step A ----
      if (!drain || cc == 0) {
        PC_ST(4);
        bar();                                         // A (after a pass: the second half is in the slot; drain: X3)
        PC_ST(1);
      }
      const unsigned ob_c = ob[tm], om_c = om[tm];
      const bool live_c = ((olive >> tm) & 1u) != 0u;  // (kept: the last chunk re-resolves the positions for the next tile)
      if (epi) {
        wait_vm<0>();                                  // this block's inputs (requested a chunk ago; the stores in the queue are older still)
        PC_ST(0);
        // second half of the previous block's stores (converted during the previous chunk's step B; F32OUT blocks read `blk`
        // before it is refilled below)
        if (cc > 0) store_half((((cc - 1) & 3) < 2 ? hacc[(cc - 1) & 1][(cc - 1) >> 2] : blk), ob[(cc - 1) & 3], om[(cc - 1) & 3], ((olive >> ((cc - 1) & 3)) & 1u) != 0u, (cc - 1) >> 2, 1);
        if (HAS_NMD && cc == 4) nmd_flush(et, 0);
        PC_ST(5);
        if (tm >= 2) xread_block(blk, tm - 2, tn);      // (published by barrier A of this pass's first chunk)
        epi_collect(tm);
        lgkm0();
        PC_ST(6);
        if (HAS_NMD && (cc == 0 || cc == 4)) {
#pragma unroll
          for (int q = 0; q < 8; ++q) nmd2[q] = f32x2{0.f, 0.f};
        }
        if (F32OUT && (cc == 0 || cc == 4)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) pool_acc[r] = -INFINITY;
        }
        epi_half(xb, tn, 0);
      }
      if (!drain && cc == 2 && bsrc != nullptr) {      // the next tile's input-mask bytes, for the math waves' piece table
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) {
          int pc; bool inr;
          const int rb = piece_pos(np, it, pc, inr);
          nb_raw[it] = bsrc[(size_t)rb * a.L_in + pc];
        }
      }
      PC_ST(4);
      // ---- step B ----
      if (!drain) {
        bar();
        PC_ST(1);
      }
      if (epi) store_half(xb, ob_c, om_c, live_c, tn, 0);     // (converted during step A)
      // inputs of the next block: block cc + 1 of the tile in hand, or block 0 of the tile the math waves are finishing
      // (the staging area was read out at step A)
      if (epi && !last_chunk) epi_request(ob[(cc + 1) & 3], om[(cc + 1) & 3], (cc + 1) >> 2);
      if (!drain && last_chunk) {
        tile_positions(cur);                            // (block 7's remaining stores use ob_c / om_c / live_c)
        epi_request(ob[0], om[0], 0);
      }
      PC_ST(5);
      if (epi) {
        epi_half(xb, tn, 1);
        if (F32OUT && a.pool_out != nullptr) pool_block(xb);
      }
      if (!drain && cc == 4 && bsrc != nullptr) {
#pragma unroll
        for (int it = 0; it < A_ITERS; ++it) Bbuf[it * 256 + ptid] = nb_raw[it];
      }
      PC_ST(4);
      // ---- step C ----
      if (!drain) {
        bar();
        PC_ST(1);
      }
      if (epi && last_chunk) {
        store_half(xb, ob_c, om_c, live_c, tn, 1);
        if (HAS_NMD) nmd_flush(et, 1);
      }
      if (epi && F32OUT && (cc == 3 || cc == 7) && a.pool_out != nullptr) pool_flush(et, tn);
      PC_ST(5);
    }
    if (drain) break;
    et = cur;
    cur = np;
    tile_of(pass + 2, np);
  }
  if ((!(vmax <= 65000.0f) || vnan) && a.overflow != nullptr) atomicOr(a.overflow, 1);
  PC_ST_END(8);
}

int pc_lds_bytes(int dil) {
  const int rows_a = HM + (PK - 1) * dil;
  return (2 * 4 * rows_a + PK * W_ITEMS + 4 * X_ITEMS + 4 * S_ITEMS) * 16 + JG_EPI_ROWS * 2 * HN * 4 + A_ITERS * 256;
}

template <unsigned EP, bool FLAT, int DIL, bool F32OUT>
int launch_pc(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  const int smem = pc_lds_bytes(a.dil);
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_pc_kernel<EP, FLAT, DIL, F32OUT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int n_tiles = FLAT ? a.flat_tiles : a.rows * a.tiles_m;
  int grid = e->n_cu;                                   // one 8-wave workgroup per CU
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL((conv_pc_kernel<EP, FLAT, DIL, F32OUT>), dim3((unsigned)grid), dim3(PT), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
#ifdef JG_STAMP
  {
    unsigned long long hh[16], z[16] = {0};
    JG_HIP(hipStreamSynchronize(s));
    JG_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(jg_pc_stamp_acc), sizeof(hh)));
    JG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(jg_pc_stamp_acc), z, sizeof(z)));
    const double tm = (double)hh[7], th = (double)hh[15];
    fprintf(stderr, "PCSTAMP ep=0x%x rows=%d grid=%d math: cyc/wave=%.0f mfma+lds=%.3f barrier=%.3f handoff=%.3f | helper: cyc/wave=%.0f "
            "wait=%.3f barrier=%.3f handoff=%.3f dma_issue=%.3f epi_math=%.3f stores+requests=%.3f lds_reads=%.3f\n", EP, a.rows, grid, tm / (grid * 4.0), hh[0] / tm, hh[1] / tm, hh[2] / tm,
            th / (grid * 4.0), hh[8] / th, hh[9] / th, hh[10] / th, hh[11] / th, hh[12] / th, hh[13] / th, hh[14] / th);
  }
#endif
  return JG_OK;
}

}  // namespace

// the stage patterns of the residual stacks (tanh-GELU): plain / + shortcut / stack end with NMD tap, norm and second GELU
bool jg_conv_pc_supports(const ConvHArgs &a) {
  if (a.dil != 3) return false;                  // the instantiated dilation (the residual stacks of the in-tree 128-channel models)
  if (a.k != PK || a.cc_in != PCC || a.cout != HN || a.cout_pad != HN || a.cw != HN || a.ch0 != 0 || a.ostride != 1 ||
      a.tap_lo != 0 || a.tap_hi != PK - 1 || a.lut != nullptr || a.ids != nullptr || a.act_kind != JG_ACT_GELU_TANH)
    return false;
  if (a.ep != JG_EP_ACT1 && a.ep != (JG_EP_ADD | JG_EP_ACT1) &&
      a.ep != (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2))
    return false;
  if (((a.ep & JG_EP_ADD) != 0) != (a.addh != nullptr)) return false;
  if (((a.ep & (JG_EP_NMD1 | JG_EP_NMD2)) != 0) != (a.nmd_out != nullptr)) return false;
  if (4 * (HM + (PK - 1) * a.dil) > A_ITERS * HT || pc_lds_bytes(a.dil) > 160 * 1024) return false;
  // 32-bit DMA offsets of the shortcut tensor
  if ((double)a.rows * (a.cout_pad / 16) * 4.0 * a.L_out * 16.0 >= 4.0e9) return false;
  const int n_tiles = a.flat ? a.flat_tiles : a.rows * a.tiles_m;
  return n_tiles >= 1;
}

int jg_conv_pc_launch(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  switch (a.ep) {
#define JG_CASE(ep)                                                                                                   \
  case (ep):                                                                                                          \
    if (a.out_f16s) return a.flat ? launch_pc<(ep), true, 3, false>(e, a, s) : launch_pc<(ep), false, 3, false>(e, a, s); \
    return a.flat ? launch_pc<(ep), true, 3, true>(e, a, s) : launch_pc<(ep), false, 3, true>(e, a, s);
    JG_CASE(JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2)
#undef JG_CASE
    default: break;
  }
  jg_set_error("conv_pc: stage pattern 0x%x has no compiled epilogue", a.ep);
  return JG_ERR_UNSUPPORTED;
}
