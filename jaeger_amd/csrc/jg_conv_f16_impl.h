// Split-f16 ("f16x3") MaskedConv1D for gfx950: f32-accurate convolution on the
// f16 matrix cores.
//
// Every f32 operand is carried as an f16 pair  x = hi + lo  (hi = f16(x),
// lo = f16(x - hi), 22 significant bits together) and each logical product is
// three MFMAs accumulating in f32:  x.w ~= hi_x*hi_w + lo_x*hi_w + hi_x*lo_w
// (the dropped lo*lo term is 2^-22 relative).  v_mfma_f32_32x32x16_f16 runs at
// 16x the rate of the exact-f32 MFMA, so the scheme nets ~5.3x the f32 roof
// while keeping the logits inside the 1e-4 gate (tests/test_gpu_parity.py).
//
// Layouts
//   F16S activations, per (window, frame) row block of L positions and C = 16*CC
//     channels:  [cc][plane hi|lo][h][L][8 halfs]   (16-byte items; channel
//     c = 16*cc + 8*h + j).  Same 4 B/element as f32, but a tile's operand slice
//     for one 16-channel chunk is 4 contiguous runs - coalesced 16-B DMA in,
//     conflict-free ds_read_b128 MFMA fragments out.
//   weights  [plane][tap][kc = cin/8][cout_pad][8 halfs], pre-scaled by 2^s so
//     the lo parts stay out of the f16 subnormal range (undone in the epilogue).
//
// Work decomposition.  Persistent workgroups of 4 waves (2 x 2, each wave 128 positions x 64
// channels = 8 accumulator blocks of 32 x 32) walk 256-position x 128-channel output tiles; two
// workgroups are resident per CU for k = 5.  The K loop runs over (16-channel chunk, tap): a tap
// needs one 8 KB weight slice and the chunk's activation slice; k = 5 synchronises once per TWO
// taps - steps (t0 t1)(t2 t3)(t4) - k = 7 / 9 once per tap.  Everything moves global->LDS by DMA
// (global_load_lds with an SGPR base + per-lane 32-bit offset, no register staging): weight slices
// two steps (k = 5) / K-2 taps ahead into a ring with one slot per tap, activation slices one chunk
// ahead into a double buffer.  Waves run at priority 2 in this loop and 0 in the epilogue.  Steps wait with COUNTED s_waitcnt vmcnt, so the DMA queue
// never drains; zero padding / mask multiply are applied by zero-filling the affected 16-byte
// pieces after the DMA has landed (rare).  The MFMAs take the WEIGHTS as their A operand, so an
// accumulator register holds one channel and a lane holds one position: the fused epilogue (folded
// bias / batch-norm affine, DyT, residual add, GELU, NMD tap, max pool, f16 re-split) needs no
// cross-lane transpose and reads its per-channel parameters from LDS.
// Variants of the same kernel (template parameters): LUT - the first layer as table lookups
// instead of matrix-core work; FLAT - window-packed position tiling;
// CW - the width- / stride-general tiles (64- / 32-channel workgroup tiles, a channel base for convs wider than 128,
// stride 2, tap ranges for 1- to 4-tap convs).  The instantiations are compiled in seven translation units
// (JG_CONV_PART, see the bottom of this file).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "jg_common.h"

#include "jg_conv_dev.h"

namespace {


// -DJG_STAMP: experiment build that accumulates per-phase shader cycles of every wave
// (wait / barrier / DMA issue / LDS+MFMA / epilogue / whole kernel) and prints them per launch.
#ifdef JG_STAMP
static __device__ unsigned long long jg_stamp_acc[8];
#define JG_ST_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = __builtin_amdgcn_s_memtime(); const unsigned long long st_t0 = st_t
#define JG_ST(idx) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[idx] += n_ - st_t; st_t = n_; } while (0)
#define JG_ST_END do { st_[5] = __builtin_amdgcn_s_memtime() - st_t0; if (lane == 0) { for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&jg_stamp_acc[q_], st_[q_]); } } while (0)
#else
#define JG_ST_DECL
#define JG_ST(idx)
#define JG_ST_END
#endif

// Wave priority: a wave in its main loop runs at priority 2, in its epilogue at 0, so that the matrix-core
// stream of one resident workgroup is not held up by the other's epilogue arithmetic (+1.4 % measured A/B;
// the reverse, or raising the priority only around each step's MFMA cluster, gains less).  -DJG_PRIO=0: off.
#ifndef JG_PRIO
#define JG_PRIO 1
#endif
#ifndef JG_LUT_WAVES
#define JG_LUT_WAVES 8  // waves of a first-layer table workgroup: 8 = two per SIMD share one table half (4 tiles per pass)
#endif
#ifndef JG_FASTEP
#define JG_FASTEP 0     // 1: the hot tanh-GELU patterns run the packed-f32, store-as-you-go epilogue (bit-identical; measured 2 % slower - the epilogue is bound by store-issue stalls, not by its instruction count: DESIGN 3.1)
#endif
#ifndef JG_PAIRED
#define JG_PAIRED 1     // k = 5: one barrier per two taps (+1.3 % measured A/B); 0 = one per tap
#endif
#define JG_PRIO_MAIN() do { if (JG_PRIO) __builtin_amdgcn_s_setprio(2); } while (0)
#define JG_PRIO_EPI() do { if (JG_PRIO) __builtin_amdgcn_s_setprio(0); } while (0)

// LUT = true is the first-layer variant: a convolution whose input is an embedding gather is linear
// in one-hot ids, so y[p] = sum_t T_t[id[p + t]] with T_t = E . W_t (vocab x Cout, exact f32, built
// on the host in f64).  The matrix-core loop is replaced by LDS row lookups (k rows of 64 channels per
// output position); the epilogue is the same code.  A workgroup owns one 64-channel half of the
// table (k * (vocab + 1) rows, row `vocab` = zeros for padding) and its JG_LUT_WAVES (8) waves cover
// four 256-position tiles per pass: waves {0,1} the first, {2,3} the second, ... - two waves per SIMD
// share the table, so one's row lookups run under the other's epilogue.  K is unused (taps come from a.k).
// CW = channel width of the workgroup tile.  128: waves 2 x 2, 128 positions x 64 channels each, exactly 128 output
// channels at stride 1 (the residual stacks: no run-time geometry in the epilogue).  129: the same tile with run-time
// geometry - a channel base (convs wider than 128 run one launch per 128 channels), fewer than 128 real channels,
// stride 2 evaluated at stride 1 with the odd outputs dropped.  64 / 32 (narrow convs of pyramid-shaped models): waves
// 4 x 1, 64 positions x 64 / 32 channels each, run-time geometry too - the weight slices and the epilogue table keep
// their 128-wide (zero-padded) layout, only the matrix-core work and the outputs shrink.
// TANH: compiled for the tanh-GELU alone (the residual stacks' hot patterns: half the code, inside the instruction cache).
// PIPE (k = 5, 128 channels, dilation 3, an even number of input chunks): the main loop of the producer / consumer
// experiment's math wave (jg_conv_pc.hip) in this kernel - MFMA fragments of the next 12-MFMA group are requested while the
// current group is issued, dealt out one LDS read per MFMA; a step's barrier is taken before the step's last group
// is issued; the ring's DMA pieces ride behind single accumulator blocks; the tile's first products start from C = 0.
template <int K, unsigned EP, bool LUT = false, bool FLAT = false, int CW = 128, bool TANH = false, bool PIPE = false>
// K = 5 fits two workgroups per CU in LDS (<= 80 KB each): hold the register file to 256 per
// lane so that both are really resident (without the bound hipcc takes ~340 and the second
// workgroup of a CU only starts when the first has finished).
__global__ __launch_bounds__(LUT ? JG_LUT_WAVES * 64 : HT) __attribute__((amdgpu_waves_per_eu(K == 5 ? 2 : 1, 2)))
void conv_f16x3_kernel(ConvHArgs a) {
  constexpr int NTHR = LUT ? JG_LUT_WAVES * 64 : HT;      // threads of this variant's workgroup
  constexpr int WA = K - 2;                  // weight slices in flight ahead of the matrix cores
  constexpr bool NARROW = !LUT && (CW == 64 || CW == 32);
  constexpr bool GEN = CW != 128;            // run-time output geometry (channel base, real width, out-stride)
  const int ch0 = GEN ? a.ch0 : 0;
  const int L_res = GEN ? a.L_res : a.L_out;
  constexpr int TM = NARROW ? 2 : 4;         // 32-position blocks per wave
  constexpr int TN = (NARROW && CW == 32) ? 1 : 2;   // 32-channel blocks per wave
  constexpr int STRIPS = HM / (TM * 32);     // wave strips (partial NMD / pool rows) per tile: 2, or 4 when narrow
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = LUT ? (wid & 1) : (NARROW ? wid : (wid >> 1));
  const int wn = LUT ? ((GEN && a.lut_one_half) ? 0 : (int)(blockIdx.x & 1u)) : (NARROW ? 0 : (wid & 1));
  const int i = lane & 31, h = lane >> 5;
  // virtual block index / grid / tiles per pass (LUT: two blocks = the two channel halves share one index)
  const int lut_sh = (LUT && !(GEN && a.lut_one_half)) ? 1 : 0;
  const int vgrid = (int)(gridDim.x >> lut_sh);
  // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by index, each XCD has its own L2.  Giving the
  // workgroups of one XCD a contiguous run of tiles keeps neighbouring tiles of a row (shared halo rows, mask bytes) and,
  // in the table variant, the two channel halves of a tile on one L2.
  int vb = (int)(blockIdx.x >> lut_sh);
  if ((vgrid & 7) == 0) vb = (vb & 7) * (vgrid >> 3) + (vb >> 3);
  constexpr int TPER = LUT ? JG_LUT_WAVES / 2 : NT;       // tiles per pass (LUT: two waves per 256-position tile)
  const int tsub = LUT ? (wid >> 1) : 0;
  // LDS carve (16-byte units)
  const int rows_a = HM + (K - 1) * a.dil;                 // rows of one activation slice
  const int a_items = NT * 4 * rows_a;                      // [NT][4 ph][rows_a]
  uint4 *Abuf = lds;                                        // [2 bufs][a_items]
  uint4 *Wbuf = lds + 2 * a_items;                          // [K slots][2 planes][2 h][HN]
  const int lut_rows = a.k * (a.lut_vocab + 1);
  float *epiL = LUT ? reinterpret_cast<float *>(lds) + lut_rows * LUT_RS
                    : reinterpret_cast<float *>(Wbuf + K * W_ITEMS);
  for (int q = tid; q < a.n_epi_rows * 2 * HN; q += NTHR) epiL[q] = a.epi[q];   // visible after the first barrier
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)lds;
  // wave-uniform LDS byte addresses of this wave's DMA destinations
  const unsigned ldsA = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);                       // + buf*a_items*16 + it*4096
  const unsigned ldsW = __builtin_amdgcn_readfirstlane(lds0 + 2 * a_items * 16 + wid * 1024);   // + slot*8192 + it*4096

  const int n_tiles = FLAT ? a.flat_tiles : a.rows * a.tiles_m;
  const int n_pairs = (n_tiles + TPER - 1) / TPER;
  // Position of a lane.  Row-tiled launches cut every (window, frame) row into 256-position tiles of its
  // own.  Window-packed ("flat") launches lay a window's frames end to end on one axis, each followed by a
  // gap >= the conv's halo (row pitch P), the window padded to a multiple of 128 (WP), and tile that axis:
  // frames whose length is an awkward fraction of 256 (665 codons at 2000 bp) no longer waste a third of
  // their last tile.  A tile may then span frames, so row and position are per lane.
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
  int my_pairs = 0;
  if (vb < n_pairs) my_pairs = (n_pairs - 1 - vb) / vgrid + 1;
  if (my_pairs == 0) return;

  // per-thread activation piece coordinates: q -> (tile u, plane/half ph, row r)
  // packed as (u << 20) | (ph << 16) | r to keep the register footprint small
  unsigned a_pk[A_ITERS];
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int q = tid + it * HT;
    const int ph = q / rows_a;          // >= 4: no piece
    a_pk[it] = ((unsigned)(ph >> 2) << 20) | ((unsigned)(ph & 3) << 16) | (unsigned)(q - ph * rows_a);
  }
  // weight slice: one item per thread, [plane][h][n] -> byte offset inside the tap-major blob
  unsigned w_voff[W_ITERS];
#pragma unroll
  for (int it = 0; it < W_ITERS; ++it) {
    const int q = tid + it * HT;          // [plane][h][n]
    w_voff[it] = (unsigned)((((q >> 8) * K * a.cc_in * 2 + ((q >> 7) & 1)) * HN + (q & (HN - 1))) * 16);
  }

  auto tiles_of = [&](int pass, Tile *t) {
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int T = (vb + pass * vgrid) * TPER + tsub + u;
      const int Tc = min(T, n_tiles - 1);
      t[u].rowblk = Tc / a.tiles_m;
      t[u].m0 = (Tc - t[u].rowblk * a.tiles_m) * HM;
      t[u].valid = T < n_tiles;
      t[u].T = Tc;
    }
  };

  // ---- per-pass piece table --------------------------------------------------------------
  // A piece's source offset and validity depend on its input position only (in range, mask
  // byte / codon id non-zero), not on the channel chunk: per pass each thread keeps 5 byte
  // offsets (relative to the chunk's SGPR base) and 5 validity bits.  The position bytes are
  // read one pass ahead (raw[]), so no step ever waits on a dependent load.
  const uint8_t *bsrc = a.ids != nullptr ? a.ids : a.mask_in;
  unsigned char raw[A_ITERS];
  unsigned x_voff[A_ITERS];
  unsigned x_ok = 0;
  auto piece_pos = [&](const Tile *tl2, int it, int &pc, bool &inr) -> int {
    const int a_u = (int)(a_pk[it] >> 20), a_r = (int)(a_pk[it] & 0xffff);
    int rb, p;
    inr = resolve(tl2[0], a_r - a.pad_left, a.L_in, rb, p) && a_u < NT;
    pc = min(max(p, 0), a.L_in - 1);
    if constexpr (FLAT) rb = min(rb, a.rows - 1);
    return rb;
  };
  auto load_bytes = [&](const Tile *tl2) {
    if (bsrc != nullptr) {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        int pc; bool inr;
        const int rb = piece_pos(tl2, it, pc, inr);
        raw[it] = bsrc[(size_t)rb * a.L_in + pc];
      }
    } else {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) raw[it] = 1;
    }
  };
  auto build_pieces = [&](const Tile *tl2) {     // consumes raw[] (loaded a pass ago)
    x_ok = 0;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      int pc; bool inr;
      const int rb = piece_pos(tl2, it, pc, inr);
      const unsigned ph = (a_pk[it] >> 16) & 3;
      const unsigned byte = raw[it];
      bool keep;
      if (a.ids != nullptr) {       // embedding gather (first conv): byte = codon id
        x_voff[it] = (byte * a.cc_in * 4 + ph) * 16;
        keep = byte != 0 || !a.mask_from_ids;
      } else {
        x_voff[it] = (unsigned)(((rb * a.cc_row * 4 + (int)ph) * a.L_in + pc) * 16);
        keep = byte != 0;
      }
      if (inr && keep) x_ok |= 1u << it;
    }
  };
  const char *x_base = a.ids != nullptr ? reinterpret_cast<const char *>(a.embh)
                                        : reinterpret_cast<const char *>(a.xh);
  const unsigned x_cc_stride = a.ids != nullptr ? 4u * 16u : 4u * (unsigned)a.L_in * 16u;   // bytes per chunk

  // ---- DMA issue ---------------------------------------------------------------------------
  // does this wave own pieces in the last (partial) activation iteration?
  const bool x_last_wave = __builtin_amdgcn_readfirstlane((int)((A_ITERS - 1) * HT + wid * 64 < 4 * rows_a)) != 0;
  auto issue_w = [&](int cc, int t) {        // weight slice (cc, t) -> ring slot t
    if (a.dbg & 16) return;
    const char *sb = reinterpret_cast<const char *>(a.wh) + ((size_t)(t * a.cc_in * 2 + cc * 2) * HN) * 16;
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) glds16(sb, w_voff[it], ldsW + t * (W_ITEMS * 16) + it * (HT * 16));
  };
  auto issue_x = [&](int cc, int buf) {      // both tiles' activation slices of chunk cc
    if (a.dbg & 8) return;
    const char *sb = x_base + (size_t)cc * x_cc_stride;
    const unsigned dst = ldsA + buf * (a_items * 16);
    // Iterations 0..A_ITERS-2 are full (4*rows_a > 4*HT); the last one covers the halo remainder
    // and exists only in the leading wave(s).  The branch around it is wave-uniform on purpose:
    // the counted waits below must know exactly how many DMAs each wave has in flight.
#pragma unroll
    for (int it = 0; it < A_ITERS - 1; ++it) glds16_nt(sb, x_voff[it], dst + it * (HT * 16));
    if (x_last_wave) {
      if ((a_pk[A_ITERS - 1] >> 20) < NT) glds16_nt(sb, x_voff[A_ITERS - 1], dst + (A_ITERS - 1) * (HT * 16));
    }
  };
  // after a slice's DMA has landed: overwrite padding / masked-out pieces with zeros (by the
  // lanes that fetched them); the step barrier then publishes the slice
  auto zero_fill = [&](int buf) {
    uint4 *A = Abuf + buf * a_items;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it)
      if ((a_pk[it] >> 20) < NT && !((x_ok >> it) & 1u)) A[tid + it * HT] = make_uint4(0u, 0u, 0u, 0u);
  };

  f32x16 acc[TM][TN];       // [tm: position block][tn: channel block]
  auto zero_acc = [&]() {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
  };
  zero_acc();

  // LDS fragment addresses (16-byte items): weights [plane][h][n], activations [u][plane][h][row]
  const int w_frag = h * HN + wn * 64 + i;                  // + plane*2*HN + tn*32 (+ slot*W_ITEMS)
  const int x_frag = h * rows_a + wm * (TM * 32) + i;       // + plane*2*rows_a + tm*32 + t*dil

  // ---- software pipeline -----------------------------------------------------------------
  // step (cc, t): weights of step +WA are issued into slot (t+WA)%K, whose last reader was step
  // -2; the next chunk's activations are issued at tap 0 into the other buffer.  A step may
  // leave outstanding only what was issued after its own weight slice: WA-1 weight DMAs, plus
  // the A_ITERS activation DMAs when those went out in between (taps 1..WA).
  Tile cur[NT], np[NT];            // tiles of this pass / of the next pass
  JG_ST_DECL;
  tiles_of(0, cur);
  tiles_of(1, np);
  // ---- LUT variant: table half -> LDS once; per pass the wave stages the table-row index of
  // each of its 128 + (k-1)*dil input positions in LDS (fetched one pass ahead) ----------------
  unsigned char *stage = reinterpret_cast<unsigned char *>(epiL + JG_EPI_ROWS * 2 * HN) + wid * 256;
  unsigned char nxt[4] = {0, 0, 0, 0};
  auto lut_fetch = [&](const Tile &tl) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int p = tl.m0 + wm * (TM * 32) - a.pad_left + lane + 64 * q4;
      const int pc = min(max(p, 0), a.L_in - 1);
      const unsigned char b = a.ids[(size_t)tl.rowblk * a.L_in + pc];
      nxt[q4] = (tl.valid && p >= 0 && p < a.L_in) ? b : (unsigned char)a.lut_vocab;   // row `vocab` = zeros
    }
  };
  if constexpr (LUT) {
    float4 *T4 = reinterpret_cast<float4 *>(lds);
    const float4 *src = reinterpret_cast<const float4 *>(a.lut) + (size_t)wn * lut_rows * 16;
    for (int q = tid; q < lut_rows * 16; q += NTHR) T4[(q >> 4) * (LUT_RS / 4) + (q & 15)] = src[q];
    lut_fetch(cur[0]);
    __syncthreads();
  } else {
    load_bytes(cur);
    build_pieces(cur);               // the only exposed byte-load latency of the launch
    load_bytes(np);
    issue_x(0, 0);
#pragma unroll
    for (int t = 0; t < ((K == 5 && JG_PAIRED) ? 4 : WA); ++t) issue_w(0, t);
  }
  int xc = 0;                      // running chunk count: activation buffer parity
#ifdef JG_VALU_PROBE
  float probe[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) probe[q] = (float)(lane + q);
#endif
  JG_PRIO_MAIN();
  for (int pass = 0; pass < my_pairs; ++pass) {
    if constexpr (LUT) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) stage[lane + 64 * q4] = nxt[q4];
      lut_fetch(np[0]);                                   // next pass's indices fly under this pass
      __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the wave's own staging writes
      __builtin_amdgcn_wave_barrier();
      const float *Tl = reinterpret_cast<const float *>(lds) + 4 * h;
      for (int t = 0; t < a.k; ++t) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const int rowi = stage[tm * 32 + i + t * a.dil];
          const float *r = Tl + (t * (a.lut_vocab + 1) + rowi) * LUT_RS;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = *reinterpret_cast<const float4 *>(r + tn * 32 + 8 * g);
              acc[tm][tn][4 * g + 0] += v.x; acc[tm][tn][4 * g + 1] += v.y;
              acc[tm][tn][4 * g + 2] += v.z; acc[tm][tn][4 * g + 3] += v.w;
            }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (PIPE) {
      static_assert(K == 5 && !LUT && CW == 128, "the pipelined main loop is built for the 128-channel five-tap convs");
      constexpr int DIL = 3;                       // (launch_ke checks a.dil)
      struct XF { uint4 h[2], l[2]; };
      struct WF { uint4 h[2], l[2]; };
      XF xf[2];
      WF wf[2];
      const uint4 *Wb = Wbuf + w_frag;
      auto ldx = [&](XF &f, const uint4 *A, int t, int tp) {
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
          f.h[tq] = A[(tp * 2 + tq) * 32 + t * DIL];
          f.l[tq] = A[2 * rows_a + (tp * 2 + tq) * 32 + t * DIL];
        }
      };
      auto ldw = [&](WF &f, int t) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          f.h[tn] = Wb[t * W_ITEMS + tn * 32];
          f.l[tn] = Wb[t * W_ITEMS + 2 * HN + tn * 32];
        }
      };
      auto mm = [&](auto zero_c, const WF &w, const XF &x, int tp, auto &&between) {
        constexpr bool ZERO = decltype(zero_c)::value;
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
      const bool last_pass_ = pass == my_pairs - 1;
      // step A of the pass's first chunk: its operands were requested by the prologue / the previous pass
      wait_vm<2 * W_ITERS>();
      zero_fill(0);
      __syncthreads();
      ldw(wf[0], 0);
      ldx(xf[0], Abuf + x_frag, 0, 0);
      for (int cp = 0; cp < a.cc_in / 2; ++cp) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int cc = 2 * cp + half;
          const bool last_chunk = cc == a.cc_in - 1;
          const bool tail = last_chunk && last_pass_;
          const int ncc = last_chunk ? 0 : cc + 1;
          const uint4 *A = Abuf + half * a_items + x_frag;
          const uint4 *An = Abuf + (half ^ 1) * a_items + x_frag;
#pragma unroll
          for (int g = 0; g < 10; ++g) {
            const int t = g >> 1, tp = g & 1;
            const bool step_end = g == 3 || g == 7 || g == 9;
            const bool pass_end = g == 9 && last_chunk;
            if (step_end && !pass_end) {
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
              __syncthreads();       // (lgkmcnt 0: this step's last fragments are in registers, the slots may be refilled)
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!pass_end) {
              if (g < 9) {
                ldx(xf[(g + 1) & 1], A, (g + 1) >> 1, (g + 1) & 1);
                if (tp == 1) ldw(wf[(half + t + 1) & 1], t + 1);
              } else {
                ldx(xf[0], An, 0, 0);
                ldw(wf[(half ^ 1) & 1], 0);
              }
            }
            auto ring = [&](int k) {
              if (g != 0 && g != 4 && g != 8) return;
              __builtin_amdgcn_sched_barrier(0);
              if (g == 0) {
                const char *sbw = reinterpret_cast<const char *>(a.wh) + ((size_t)(4 * a.cc_in * 2 + cc * 2) * HN) * 16;
                if (k == 0) glds16(sbw, w_voff[0], ldsW + 4 * (W_ITEMS * 16));
                if (k == 1) glds16(sbw, w_voff[1], ldsW + 4 * (W_ITEMS * 16) + HT * 16);
                if (!tail) {
                  if (k == 0 && last_chunk) build_pieces(np);
                  const char *sbx = x_base + (size_t)ncc * x_cc_stride;
                  const unsigned dst = ldsA + (half ^ 1) * (a_items * 16);
                  if (k == 1) glds16_nt(sbx, x_voff[0], dst);
                  if (k == 2) { glds16_nt(sbx, x_voff[1], dst + HT * 16); glds16_nt(sbx, x_voff[2], dst + 2 * (HT * 16)); }
                  if (k == 3) {
                    glds16_nt(sbx, x_voff[3], dst + 3 * (HT * 16));
                    if (x_last_wave) {
                      if ((a_pk[A_ITERS - 1] >> 20) < NT) glds16_nt(sbx, x_voff[A_ITERS - 1], dst + (A_ITERS - 1) * (HT * 16));
                    }
                  }
                }
              } else if (!tail) {
                const int tt = (g == 4 ? 0 : 2) + (k >> 1);
                const char *sbw = reinterpret_cast<const char *>(a.wh) + ((size_t)(tt * a.cc_in * 2 + ncc * 2) * HN) * 16;
                glds16(sbw, w_voff[k & 1], ldsW + tt * (W_ITEMS * 16) + (k & 1) * (HT * 16));
              }
              __builtin_amdgcn_sched_barrier(0);
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
      xc += a.cc_in;
    }
    if constexpr (!LUT && !PIPE) for (int cc = 0; cc < a.cc_in; ++cc) {
      const int abuf = xc & 1;
      const bool last_chunk = cc == a.cc_in - 1;
      const bool tail = last_chunk && pass == my_pairs - 1;     // nothing is issued behind this chunk
      const uint4 *A = Abuf + abuf * a_items + x_frag;
      if constexpr (K == 5 && JG_PAIRED) {
        // Paired steps: one barrier per TWO taps - (t0 t1) (t2 t3) (t4) - so a chunk costs three barriers and
        // three counted waits instead of five.  Ring slots stay one per tap; the slices of a step are issued two
        // steps ahead: (cc, t4) and the next chunk's activations at step A, the next chunk's (t0 t1) at B,
        // (t2 t3) at C.
        const int ncc = last_chunk ? 0 : cc + 1;
        auto taps = [&](int t, auto &&mid, bool has_mid = true) {
          if constexpr (GEN) {
            if (a.tap_lo != 0 && (t < a.tap_lo || t > a.tap_hi)) {   // 1x1 / 3-tap conv on the 5-tap pipeline: this tap's weights are zero
              if (has_mid) mid();
              return;
            }
          }
          const uint4 *B = Wbuf + t * W_ITEMS + w_frag;
          half8 wh[TN], wl[TN];
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const uint4 vh = B[tn * 32];
            const uint4 vl = B[2 * HN + tn * 32];
            wh[tn] = *reinterpret_cast<const half8 *>(&vh);
            wl[tn] = *reinterpret_cast<const half8 *>(&vl);
          }
#pragma unroll
          for (int tp = 0; tp < TM / 2; ++tp) {
            half8 xh[2], xl[2];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
              const uint4 vh = A[(tp * 2 + tq) * 32 + t * a.dil];
              const uint4 vl = A[2 * rows_a + (tp * 2 + tq) * 32 + t * a.dil];
              xh[tq] = *reinterpret_cast<const half8 *>(&vh);
              xl[tq] = *reinterpret_cast<const half8 *>(&vl);
            }
#pragma unroll
            for (int tq = 0; tq < 2; ++tq)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) {
                f32x16 &c = acc[tp * 2 + tq][tn];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tn], xl[tq], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[tn], xh[tq], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tn], xh[tq], c, 0, 0, 0);
              }
            if (tp == 0 && has_mid) {
              __builtin_amdgcn_sched_barrier(0);
              mid();
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        };
        auto nop = []() {};
        // ---- step A: taps 0, 1 ----
        JG_ST(6);
        wait_vm<2 * W_ITERS>();
        JG_ST(0);
        zero_fill(abuf);
        __syncthreads();
        JG_ST(1);
        taps(0, [&]() {
          issue_w(cc, 4);
          if (!tail) {
            if (last_chunk) build_pieces(np);
            issue_x(ncc, abuf ^ 1);
          }
        });
        taps(1, nop, false);
        // ---- step B: taps 2, 3 ----
        JG_ST(6);
        if (tail) wait_vm<W_ITERS>();
        else if (x_last_wave) wait_vm<W_ITERS + A_ITERS>();
        else wait_vm<W_ITERS + A_ITERS - 1>();
        JG_ST(0);
        __syncthreads();
        JG_ST(1);
        taps(2, [&]() {
          if (!tail) { issue_w(ncc, 0); issue_w(ncc, 1); }
        });
        taps(3, nop, false);
        // ---- step C: tap 4 ----
        JG_ST(6);
        if (tail) wait_vm<0>();
        else if (x_last_wave) wait_vm<2 * W_ITERS + A_ITERS>();
        else wait_vm<2 * W_ITERS + A_ITERS - 1>();
        JG_ST(0);
        __syncthreads();
        JG_ST(1);
        taps(4, [&]() {
          if (!tail) { issue_w(ncc, 2); issue_w(ncc, 3); }
        });
        ++xc;
        continue;
      }
#pragma unroll
      for (int t = 0; t < K; ++t) {
        // -- wait for this step's operands, publish them ---------------------------------------
        JG_ST(6);
        if (tail) wait_vm<0>();
        else if (t >= 1 && t <= WA) {
          if (x_last_wave) wait_vm<(WA - 1) * W_ITERS + A_ITERS>();
          else wait_vm<(WA - 1) * W_ITERS + A_ITERS - 1>();
        }
        else wait_vm<(WA - 1) * W_ITERS>();
        JG_ST(0);
        if (t == 0) zero_fill(abuf);
        __syncthreads();
        JG_ST(1);
        // -- keep the DMA queue full: issued from inside the matrix-core stream (after the first half
        // of the step's MFMAs are queued) so that the DMA issue cost runs under matrix-core time ------
        auto issue_step = [&]() {
          if (t + WA < K) {
            issue_w(cc, t + WA);
          } else if (!tail) {
            issue_w(last_chunk ? 0 : cc + 1, t + WA - K);
          }
          if (t == 0 && !tail) {
            if (last_chunk) build_pieces(np);        // the next pass's pieces take over from here
            issue_x(last_chunk ? 0 : cc + 1, abuf ^ 1);
          }
        };
        if (a.dbg & 2) issue_step();
        JG_ST(2);
        // -- matrix-core work: one tap of one 16-channel chunk ----------------------------------
        if (!(a.dbg & 2)) {
          const uint4 *B = Wbuf + t * W_ITEMS + w_frag;
          half8 wh[TN], wl[TN];
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) {
            const uint4 vh = B[tn * 32];
            const uint4 vl = B[2 * HN + tn * 32];
            wh[tn] = *reinterpret_cast<const half8 *>(&vh);
            wl[tn] = *reinterpret_cast<const half8 *>(&vl);
          }
#pragma unroll
          for (int tp = 0; tp < TM / 2; ++tp) {           // two position blocks at a time
            half8 xh[2], xl[2];
#pragma unroll
            for (int tq = 0; tq < 2; ++tq) {
              const uint4 vh = A[(tp * 2 + tq) * 32 + t * a.dil];
              const uint4 vl = A[2 * rows_a + (tp * 2 + tq) * 32 + t * a.dil];
              xh[tq] = *reinterpret_cast<const half8 *>(&vh);
              xl[tq] = *reinterpret_cast<const half8 *>(&vl);
            }
#pragma unroll
            for (int tq = 0; tq < 2; ++tq)
#pragma unroll
              for (int tn = 0; tn < TN; ++tn) {
                // weights are the MFMA A operand: acc rows = channels, cols = positions
                f32x16 &c = acc[tp * 2 + tq][tn];
#ifdef JG_MFMA16_PROBE
                // experiment build (timing only, results are garbage): the same operand registers and LDS
                // reads, but every 32x32x16 MFMA replaced by two 16x16x32 ones (same MACs) on quarter slices
                // of the accumulator block - does the chip hold a higher clock on that shape in THIS loop?
                {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                const half8 *ops[3][2] = {{&wh[tn], &xl[tq]}, {&wl[tn], &xh[tq]}, {&wh[tn], &xh[tq]}};
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                  for (int s2 = 0; s2 < 2; ++s2) {
                    const int q0 = 4 * (2 * (t & 1) + s2);
                    f32x4 d = {c[q0], c[q0 + 1], c[q0 + 2], c[q0 + 3]};
                    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(*ops[m][0], *ops[m][1], d, 0, 0, 0);
                    c[q0] = d[0]; c[q0 + 1] = d[1]; c[q0 + 2] = d[2]; c[q0 + 3] = d[3];
                  }
                continue;
                }
#endif
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tn], xl[tq], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[tn], xh[tq], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[tn], xh[tq], c, 0, 0, 0);
              }
#ifdef JG_VALU_PROBE
            // experiment: independent VALU work next to the MFMAs of a step (does the scheduler hide it
            // in the matrix cores' shadow?)  JG_VALU_PROBE = number of dummy FMAs per 12-MFMA group
#pragma unroll
            for (int q = 0; q < JG_VALU_PROBE; ++q) probe[q & 7] = fmaf(probe[q & 7], 1.0001f, 0.5f);
#if JG_VALU_PROBE_SCHED
#pragma unroll
            for (int q = 0; q < 12; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, JG_VALU_PROBE / 12, 0);         // then its share of VALU
            }
#endif
#endif
            if (tp == 0) {
              __builtin_amdgcn_sched_barrier(0);
              issue_step();
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      ++xc;
    }
    JG_ST(3);

    // ---- pass finished: fused epilogue straight from the accumulators ----------------------
    if (a.dbg & 1) {
      if (acc[0][0][0] + acc[1][TN - 1][3] + acc[TM - 1][TN - 1][7] + acc[TM - 1][0][9] == 12345.678f) a.overflow[0] = 2;
    } else if constexpr (JG_FASTEP && TANH && !LUT && CW == 128 && EP != JG_EP_GENERIC && K == 5) {
      // ---- the residual stacks' hot patterns (128 channels, tanh-GELU): the same expressions on packed-f32 forms ----
      // Every instruction a wave issues besides its MFMAs costs the SIMD matrix-core time (about 6 cycles each, whichever
      // of the two resident waves issues it - profiles/r3_pc_*), so this epilogue is written for instruction count:
      // v_pk_* arithmetic on channel pairs (the operation sequence per element is unchanged: results are bit-identical),
      // output positions resolved once per position block, every block converted and stored as soon as it is finished
      // (no write-back into the accumulator tuple: no register moves), its inputs fetched two blocks ahead so that no
      // load's wait meets a fresh store.
      JG_PRIO_EPI();
      constexpr bool HAS_ADD = (EP & JG_EP_ADD) != 0;
      constexpr bool HAS_NMD = (EP & (JG_EP_NMD1 | JG_EP_NMD2)) != 0;
      constexpr int N1 = (EP >> 1) & 3, N2 = (EP >> 6) & 3;
      static_assert(N1 != 2 && N2 != 2, "the tanh-GELU builds carry no DyT stage");
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const Tile &tile = cur[0];
      float vmax = 0.f;
      bool vnan = false;
      const unsigned L4 = 4u * (unsigned)a.L_out;
      unsigned ob[4], om[4], olive = 0;          // per position block: hi-plane item base, mask-byte offset, "position exists"
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        int row, p;
        const bool live = resolve(tile, (wm * 4 + tm) * 32 + i, a.L_out, row, p);
        const int mc = live ? p : 0;
        if constexpr (FLAT) {
          if (!live) row = min(max(row, 0), a.rows - 1);
        }
        ob[tm] = (unsigned)(((row * 8 + 4 * wn) * 4 + h) * a.L_out + mc);
        om[tm] = (unsigned)(row * a.L_out + mc);
        if (live) olive |= 1u << tm;
      }
      struct In {                                  // what a block needs from memory
        u32x4 sh0, sh1, sl0, sl1;                  // whole shortcut items of groups h and 2 + h, hi / lo plane (separate
        unsigned char mkb;                         // fields, not arrays: register promotion wants static indices)
      };
      auto fetch = [&](In &q, int b) {
        const int tm = b & 3, tn = b >> 2;
        q.mkb = a.mask_out != nullptr ? a.mask_out[om[tm]] : (unsigned char)1;
        if constexpr (HAS_ADD) {
          if (!(a.dbg & 128)) {
            const unsigned it0 = ob[tm] + (unsigned)(2 * tn) * L4, it1 = it0 + L4;
            q.sh0 = ld_stream(reinterpret_cast<const u32x4 *>(a.addh + it0));
            q.sl0 = ld_stream(reinterpret_cast<const u32x4 *>(a.addh + it0 + 2u * (unsigned)a.L_out));
            q.sh1 = ld_stream(reinterpret_cast<const u32x4 *>(a.addh + it1));
            q.sl1 = ld_stream(reinterpret_cast<const u32x4 *>(a.addh + it1 + 2u * (unsigned)a.L_out));
          } else {
            q.sh0 = q.sh1 = q.sl0 = q.sl1 = u32x4{0u, 0u, 0u, 0u};
          }
        }
      };
      auto swap32 = [](unsigned &lo_half_keeps, unsigned &hi_half_keeps) {
        const auto r = __builtin_amdgcn_permlane32_swap(lo_half_keeps, hi_half_keeps, false, false);
        lo_half_keeps = r[0];
        hi_half_keeps = r[1];
      };
#define JG_DPP(v, ctrl) __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), (ctrl), 0xf, 0xf, false))
      auto lane_reduce = [&](const float (&in)[16], auto op) -> float {      // (see the general epilogue below)
        const bool b2 = (i & 4) != 0, b1 = (i & 2) != 0, b0 = (i & 1) != 0, b3 = (i & 8) != 0;
        float s8[8], s4[4], s2[2];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float keep = b2 ? in[8 + q] : in[q], send = b2 ? in[q] : in[8 + q];
          s8[q] = op(keep, JG_DPP(send, 0x141));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float keep = b1 ? s8[4 + q] : s8[q], send = b1 ? s8[q] : s8[4 + q];
          s4[q] = op(keep, JG_DPP(send, 0x4e));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float keep = b0 ? s4[2 + q] : s4[q], send = b0 ? s4[q] : s4[2 + q];
          s2[q] = op(keep, JG_DPP(send, 0xb1));
        }
        const float keep1 = b3 ? s2[1] : s2[0], send1 = b3 ? s2[0] : s2[1];
        const float v = op(keep1, JG_DPP(send1, 0x128));
        return op(v, __shfl_xor(v, 16, 32));
      };
#undef JG_DPP
      auto reduced_slot = [&](int tn, int &ch) -> size_t {
        const int r = 8 * (int)((i & 4) != 0) + 4 * (int)((i & 2) != 0) + 2 * (int)((i & 1) != 0) + (int)((i & 8) != 0);
        ch = (wn * 2 + tn) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        return ((size_t)tile.T * STRIPS + wm) * a.cout + ch;
      };
      f32x2 nmd2[8];
      float pool_acc[16];
      In in0, in1;                                 // (two objects, not an array: the register promotion wants static indices)
      fetch(in0, 0);
      fetch(in1, 1);
      auto do_block = [&](int b, In &q) {
        const int tm = b & 3, tn = b >> 2;
        f32x16 &x = acc[tm][tn];
        const bool live = ((olive >> tm) & 1u) != 0u;
        const float mk = (live && q.mkb != 0) ? 1.f : 0.f;
        const int nb = (wn * 2 + tn) * 32;
        if (tm == 0) {
          if constexpr (HAS_NMD) {
#pragma unroll
            for (int r = 0; r < 8; ++r) nmd2[r] = f32x2{0.f, 0.f};
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) pool_acc[r] = -INFINITY;
        }
        if (!(a.dbg & 32)) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {              // half j: channel groups 2j, 2j + 1 (accumulator registers 8j .. 8j + 7)
            f32x2 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = f32x2{x[8 * j + 2 * r], x[8 * j + 2 * r + 1]};
            auto st_affine = [&](int row) {
              const float *pr = epiL + (row * 2) * HN + nb + 4 * h + 16 * j;
              const float4 sc0 = *reinterpret_cast<const float4 *>(pr), sc1 = *reinterpret_cast<const float4 *>(pr + 8);
              const float4 of0 = *reinterpret_cast<const float4 *>(pr + HN), of1 = *reinterpret_cast<const float4 *>(pr + HN + 8);
              v[0] = __builtin_elementwise_fma(v[0], f32x2{sc0.x, sc0.y}, f32x2{of0.x, of0.y});
              v[1] = __builtin_elementwise_fma(v[1], f32x2{sc0.z, sc0.w}, f32x2{of0.z, of0.w});
              v[2] = __builtin_elementwise_fma(v[2], f32x2{sc1.x, sc1.y}, f32x2{of1.x, of1.y});
              v[3] = __builtin_elementwise_fma(v[3], f32x2{sc1.z, sc1.w}, f32x2{of1.z, of1.w});
            };
            auto st_add = [&]() {
              // lanes i and i + 32 loaded the whole items of groups 2j and 2j + 1: one permlane32 swap per dword pair gives
              // every lane its own four channels of both groups
              const u32x4 qh = j == 0 ? q.sh0 : q.sh1, ql = j == 0 ? q.sl0 : q.sl1;
              unsigned hx = qh[0], hy = qh[1], hz = qh[2], hw_ = qh[3];
              unsigned lx = ql[0], ly = ql[1], lz = ql[2], lw_ = ql[3];
              swap32(hx, hz); swap32(hy, hw_);
              swap32(lx, lz); swap32(ly, lw_);
              v[0] += f32x2{mix_sum<0>(hx, lx), mix_sum<1>(hx, lx)};
              v[1] += f32x2{mix_sum<0>(hy, ly), mix_sum<1>(hy, ly)};
              v[2] += f32x2{mix_sum<0>(hz, lz), mix_sum<1>(hz, lz)};
              v[3] += f32x2{mix_sum<0>(hw_, lw_), mix_sum<1>(hw_, lw_)};
            };
            auto st_gelu = [&]() {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = fast_gelu2(v[r]);
            };
            auto st_nmd = [&]() {
              const f32x2 mk2 = {mk, mk};
#pragma unroll
              for (int r = 0; r < 4; ++r) nmd2[4 * j + r] = __builtin_elementwise_fma(v[r], mk2, nmd2[4 * j + r]);
            };
            st_affine(0);
            if constexpr (EP & JG_EP_NMD1) st_nmd();
            if constexpr (N1 == 1) st_affine(1);
            if constexpr (HAS_ADD) st_add();
            if constexpr (EP & JG_EP_ACT1) st_gelu();
            if constexpr (EP & JG_EP_NMD2) st_nmd();
            if constexpr (N2 == 1) st_affine(N1 ? 2 : 1);
            if constexpr (EP & JG_EP_ACT2) st_gelu();
            if (a.out_f16s) {
              unsigned hp[4], lp[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                const half2_t hh = {(_Float16)v[r].x, (_Float16)v[r].y};
                hp[r] = *reinterpret_cast<const unsigned *>(&hh);
                const half2_t ll = {(_Float16)mix_rem<0>(v[r].x, hp[r]), (_Float16)mix_rem<1>(v[r].y, hp[r])};
                lp[r] = *reinterpret_cast<const unsigned *>(&ll);
              }
              vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0].x), fabsf(v[0].y))), fmaxf(fabsf(v[1].x), fabsf(v[1].y)));
              vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[2].x), fabsf(v[2].y))), fmaxf(fabsf(v[3].x), fabsf(v[3].y)));
              vnan = vnan || __builtin_isunordered(v[0].x, v[0].y) || __builtin_isunordered(v[1].x, v[1].y) ||
                     __builtin_isunordered(v[2].x, v[2].y) || __builtin_isunordered(v[3].x, v[3].y);
              swap32(hp[0], hp[2]); swap32(hp[1], hp[3]);      // -> whole item of group 2j + h
              swap32(lp[0], lp[2]); swap32(lp[1], lp[3]);
              if (live && !(a.dbg & 64)) {
                uint4 *yh = reinterpret_cast<uint4 *>(a.y);
                const unsigned it4 = ob[tm] + (unsigned)(2 * tn + j) * L4;
                const u32x4 vhi = {hp[0], hp[1], hp[2], hp[3]}, vlo = {lp[0], lp[1], lp[2], lp[3]};
                st_stream(vhi, reinterpret_cast<u32x4 *>(yh + it4));
                st_stream(vlo, reinterpret_cast<u32x4 *>(yh + it4 + 2u * (unsigned)a.L_out));
              }
            } else if (a.pool_out != nullptr) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                pool_acc[8 * j + 2 * r] = mk != 0.f ? fmaxf(pool_acc[8 * j + 2 * r], v[r].x) : pool_acc[8 * j + 2 * r];
                pool_acc[8 * j + 2 * r + 1] = mk != 0.f ? fmaxf(pool_acc[8 * j + 2 * r + 1], v[r].y) : pool_acc[8 * j + 2 * r + 1];
              }
            } else if (live && !(a.dbg & 64)) {
              float *yf = reinterpret_cast<float *>(a.y) + (size_t)om[tm] * a.cout + nb + 4 * h + 16 * j;
              *reinterpret_cast<float4 *>(yf) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
              *reinterpret_cast<float4 *>(yf + 8) = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
            }
          }
        }
        if (b + 2 < 8) fetch(q, b + 2);
        if (tm == 3) {
          if constexpr (HAS_NMD) {
            float na[16];
#pragma unroll
            for (int r = 0; r < 8; ++r) { na[2 * r] = nmd2[r].x; na[2 * r + 1] = nmd2[r].y; }
            const float rv = lane_reduce(na, [](float x_, float y_) { return x_ + y_; });
            int ch;
            const size_t slot = reduced_slot(tn, ch);
            if (i < 16 && tile.valid) a.nmd_out[slot] = rv;
          }
          if (a.pool_out != nullptr) {
            const float rv = lane_reduce(pool_acc, [](float x_, float y_) { return fmaxf(x_, y_); });
            int ch;
            const size_t slot = reduced_slot(tn, ch);
            if (i < 16 && tile.valid) a.pool_out[slot] = rv;
          }
        }
      };
#pragma unroll
      for (int bp = 0; bp < 4; ++bp) {
        do_block(2 * bp, in0);
        do_block(2 * bp + 1, in1);
      }
      if ((!(vmax <= 65000.0f) || vnan) && a.overflow != nullptr && a.dbg == 0) atomicOr(a.overflow, 1);
      JG_PRIO_MAIN();
    } else {
      JG_PRIO_EPI();
      float vmax = 0.f;             // running max |output|: f16-range guard
      bool vnan = false;            // ... and "an output is NaN"
      float nmd_acc[16];            // per-lane NMD sums of the current channel block
      float nmd_acc2[16];           // ... of a second tap in the same conv (JG_EP_RUNTIME)
      // what a block needs from memory, fetched one block ahead so the loads of block b+1
      // fly under the arithmetic of block b
      struct Pre {
        uint2 sh[4], sl[4];
        unsigned char mkb;
      };
      // lanes i and i+32 hold channels 0-3 / 4-7 of the same 8-channel F16S item: a
      // v_permlane32_swap per dword turns two 8-byte accesses per lane into one 16-byte access
      // (lane half h then owns the whole item of group 2j+h)
      auto swap32 = [](unsigned &lo_half_keeps, unsigned &hi_half_keeps) {
        const auto r = __builtin_amdgcn_permlane32_swap(lo_half_keeps, hi_half_keeps, false, false);
        lo_half_keeps = r[0];
        hi_half_keeps = r[1];
      };
      auto item4 = [&](int row, int mc, int nb, int j) -> unsigned {
        // hi-plane item of group 2j + h (uint4 units); lo plane = + 2*L_out
        const int G = ((nb + ch0) >> 3) + 2 * j + h;
        return (unsigned)(((row * (a.cout_pad >> 4) + (G >> 1)) * 4 + (G & 1)) * a.L_out + mc);
      };
      // where an output item goes: as item4, or - phase-split store - position p of chunk cc at chunk (p & 1) * CC + cc,
      // position p >> 1 of rows that hold 2 * CC chunks of (L_out + 1) / 2 positions
      // (run-time-geometry variants only: the 128-channel instantiations of the residual stacks stay free of it - a
      // 128-wide producer of a phase-split tensor is dispatched to the general tile, CW = 129)
      const bool psplit = GEN && a.psplit != 0;
      const int L_st = psplit ? ((a.L_out + 1) >> 1) : a.L_out;      // positions per chunk row of the stored tensor
      auto item4_out = [&](int row, int mc, int nb, int j) -> unsigned {
        if constexpr (GEN) {
          if (psplit) {
            const int G = ((nb + ch0) >> 3) + 2 * j + h, CC = a.cout_pad >> 4;
            return (unsigned)(((row * 2 * CC + (mc & 1) * CC + (G >> 1)) * 4 + (G & 1)) * L_st + (mc >> 1));
          }
        }
        return item4(row, mc, nb, j);
      };
      // this lane's output position in block tm: row, clamped position, alive
      auto out_pos = [&](const Tile &tile, int tm, int &row, int &mc) -> bool {
        int p;
        bool live = resolve(tile, (wm * TM + tm) * 32 + i, L_res, row, p);
        if constexpr (GEN) {
          if (a.ostride == 2) {           // stride-2 conv evaluated at stride 1: odd positions are dropped
            live = live && !(p & 1);
            p >>= 1;
          }
        }
        mc = live ? p : 0;
        if constexpr (FLAT) {
          if (!live) row = min(max(row, 0), a.rows - 1);
        }
        return live;
      };
      auto prefetch = [&](Pre &p, const Tile &tile, int tm, int tn) {
        const int nb = (wn * 2 + tn) * 32;
        int orow, mc;
        out_pos(tile, tm, orow, mc);
        p.mkb = a.mask_out != nullptr ? a.mask_out[(size_t)orow * a.L_out + mc] : (unsigned char)1;
        if (a.addh != nullptr && !(a.dbg & 128)) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            // (a 16-channel chunk past the tensor's width - zero-padded weights - reads the last real chunk: unused)
            const unsigned it4 = item4(orow, mc, (!GEN || nb + ch0 + 16 * j < a.cout) ? nb : 0, j);
            // (read once, like the activation slices: non-temporal)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 nh = ld_stream(reinterpret_cast<const u32x4 *>(a.addh + it4));   // whole item of group 2j+h
            const u32x4 nl = ld_stream(reinterpret_cast<const u32x4 *>(a.addh + it4 + 2u * (unsigned)a.L_out));
            uint4 vh = make_uint4(nh[0], nh[1], nh[2], nh[3]);
            uint4 vl = make_uint4(nl[0], nl[1], nl[2], nl[3]);
            // give each lane back its own 4 channels of groups 2j and 2j+1
            swap32(vh.x, vh.z); swap32(vh.y, vh.w);
            swap32(vl.x, vl.z); swap32(vl.y, vl.w);
            p.sh[2 * j] = make_uint2(vh.x, vh.y);  p.sh[2 * j + 1] = make_uint2(vh.z, vh.w);
            p.sl[2 * j] = make_uint2(vl.x, vl.y);  p.sl[2 * j + 1] = make_uint2(vl.z, vl.w);
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) p.sh[g] = p.sl[g] = make_uint2(0u, 0u);
        }
      };
      // one 32-channel x 32-position accumulator block, processed in place (static register
      // indices only).  C/D layout: col = lane&31 -> position, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
      // -> channel
      auto epi_block = [&](f32x16 &x, const Tile &tile, int tm, int tn) {
        Pre p;
        prefetch(p, tile, tm, tn);
        const int nb = (wn * 2 + tn) * 32;
        int orow_, mc_;
        const bool live = out_pos(tile, tm, orow_, mc_);
        const float mk = p.mkb != 0 ? 1.f : 0.f;
        // ---- stage primitives on this lane's 16 channels of one position -----------------
        auto st_affine = [&](int row) {
          const float *pr = epiL + (row * 2) * HN + nb + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 sc = *reinterpret_cast<const float4 *>(pr + 8 * g);
            const float4 of = *reinterpret_cast<const float4 *>(pr + HN + 8 * g);
            x[4 * g + 0] = fmaf(x[4 * g + 0], sc.x, of.x);
            x[4 * g + 1] = fmaf(x[4 * g + 1], sc.y, of.y);
            x[4 * g + 2] = fmaf(x[4 * g + 2], sc.z, of.z);
            x[4 * g + 3] = fmaf(x[4 * g + 3], sc.w, of.w);
          }
        };
        auto st_dyt = [&](int row, float alpha, int use_mask) {
          const float *pr = epiL + (row * 2) * HN + nb + 4 * h;
          const float mm = use_mask ? mk : 1.0f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 ga = *reinterpret_cast<const float4 *>(pr + 8 * g);
            const float4 be = *reinterpret_cast<const float4 *>(pr + HN + 8 * g);
            x[4 * g + 0] = (fast_tanh(alpha * x[4 * g + 0]) * ga.x + be.x) * mm;
            x[4 * g + 1] = (fast_tanh(alpha * x[4 * g + 1]) * ga.y + be.y) * mm;
            x[4 * g + 2] = (fast_tanh(alpha * x[4 * g + 2]) * ga.z + be.z) * mm;
            x[4 * g + 3] = (fast_tanh(alpha * x[4 * g + 3]) * ga.w + be.w) * mm;
          }
        };
        auto st_add = [&]() {
#pragma unroll
          for (int g = 0; g < 4; ++g) {          // hi + lo is exact in f32: one mixed-precision instruction each
            x[4 * g + 0] += mix_sum<0>(p.sh[g].x, p.sl[g].x);
            x[4 * g + 1] += mix_sum<1>(p.sh[g].x, p.sl[g].x);
            x[4 * g + 2] += mix_sum<0>(p.sh[g].y, p.sl[g].y);
            x[4 * g + 3] += mix_sum<1>(p.sh[g].y, p.sl[g].y);
          }
        };
        auto st_gelu = [&]() {
          // DyT patterns (two tanh norms + two activations at a stack end) are compiled for the tanh-GELU only: with
          // the erf / ReLU alternatives beside it the stack-end pattern spilled 576 bytes per lane and ran 10x slower
          // (the run-time-flag epilogue carries every stage kind at once: tanh-GELU only as well - the planner sees to it)
          constexpr bool TANH_ONLY = TANH || EP == JG_EP_RUNTIME ||
                                     (EP != JG_EP_GENERIC && ((((EP >> 1) & 3) == 2) || (((EP >> 6) & 3) == 2)));
          if (TANH_ONLY || a.act_kind == JG_ACT_GELU_TANH) {
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = fast_gelu(x[r]);
          } else if (a.act_kind == JG_ACT_GELU_ERF) {    // one activation kind per op: uniform branches
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = gelu_erf(x[r]);
          } else {                                        // ReLU
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = fmaxf(x[r], 0.0f);
          }
        };
        auto st_nmd = [&](int which = 0) {
          // masked channel sums: accumulated per lane over the wave's four position blocks of this
          // channel block, reduced across lanes once (nmd_flush) - one partial row per 128 positions
          const float mkl = live ? mk : 0.f;
          if constexpr (EP == JG_EP_RUNTIME) {
            if (which) {
#pragma unroll
              for (int r = 0; r < 16; ++r) nmd_acc2[r] = fmaf(x[r], mkl, nmd_acc2[r]);
              return;
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) nmd_acc[r] = fmaf(x[r], mkl, nmd_acc[r]);
        };
        if constexpr (EP == JG_EP_GENERIC) {
          // any stage list: interpreted at run time (slow path: the accumulators bounce through
          // the interpreter's switch); the model families in-tree all hit a compiled pattern
          for (int q = 0; q < ((a.dbg & 32) ? 0 : a.n_hst); ++q) {
            const HStageArg st = a.hst[q];
            switch (st.kind) {
              case JG_HST_AFFINE: st_affine(st.pad_); break;
              case JG_HST_DYT: st_dyt(st.pad_, st.f0, st.arg); break;
              case JG_HST_ADD: st_add(); break;
              case JG_HST_ACT:
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = jg_act(x[r], st.arg);
                break;
              case JG_HST_NMD: st_nmd(); break;
              case JG_HST_MASKMUL:
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] *= mk;
                break;
              default: break;
            }
          }
        } else if constexpr (EP == JG_EP_RUNTIME) {
          // the canonical order affine [nmd] [norm1] [add] [gelu] [nmd] [norm2] [gelu] with every optional stage behind a
          // wave-uniform flag: one instantiation for all the canonical stage lists that have none of their own (a few
          // scalar branches per block; before, such convs fell to the exact-f32 kernel, 4 x slower)
          const unsigned e_ = a.ep_rt;
          const int n1 = (int)((e_ >> 1) & 3u), n2 = (int)((e_ >> 6) & 3u);
          st_affine(0);
          if (e_ & JG_EP_NMD1) st_nmd(0);
          if (n1 == 1) st_affine(1);
          else if (n1 == 2) st_dyt(1, a.alpha1, a.dytmask1);
          if (e_ & JG_EP_ADD) st_add();
          if (e_ & JG_EP_ACT1) st_gelu();
          if (e_ & JG_EP_NMD2) st_nmd((e_ & JG_EP_NMD1) ? 1 : 0);
          if (n2 == 1) st_affine(n1 ? 2 : 1);
          else if (n2 == 2) st_dyt(n1 ? 2 : 1, a.alpha2, a.dytmask2);
          if (e_ & JG_EP_ACT2) st_gelu();
        } else if (!(a.dbg & 32)) {
          // compiled pattern: affine [nmd] [norm1] [add] [gelu] [nmd] [norm2] [gelu], straight line
          constexpr int N1 = (EP >> 1) & 3, N2 = (EP >> 6) & 3;
          st_affine(0);
          if constexpr (EP & JG_EP_NMD1) st_nmd();
          if constexpr (N1 == 1) st_affine(1);
          if constexpr (N1 == 2) st_dyt(1, a.alpha1, a.dytmask1);
          if constexpr (EP & JG_EP_ADD) st_add();
          if constexpr (EP & JG_EP_ACT1) st_gelu();
          if constexpr (EP & JG_EP_NMD2) st_nmd();
          if constexpr (N2 == 1) st_affine(N1 ? 2 : 1);
          if constexpr (N2 == 2) st_dyt(N1 ? 2 : 1, a.alpha2, a.dytmask2);
          if constexpr (EP & JG_EP_ACT2) st_gelu();
        }
        if constexpr (GEN) {
          if (psplit) {            // the stride-2 readers take their input mask from here (they read x * mask)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] *= mk;
          }
        }
        // results stay in the block's registers (F16S: re-split, lane-pair swapped and bit-cast,
        // dword 4j..4j+3 = hi item, 8+4j.. = lo item of group 2j+h); stored by store_block() once
        // every block's loads are done - no load ever queues behind a store
        if (a.out_f16s) {
          uint2 ph[4], pl[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4 hh4, ll4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              hh4[j] = (_Float16)x[4 * g + j];
              vmax = fmaxf(vmax, fabsf(x[4 * g + j]));      // dead lanes hold finite values too (zero-filled inputs)
            }
            // (the running max drops a NaN - v_max returns the other operand: one unordered compare per two outputs)
            vnan = vnan || __builtin_isunordered(x[4 * g + 0], x[4 * g + 1]) || __builtin_isunordered(x[4 * g + 2], x[4 * g + 3]);
            ph[g] = *reinterpret_cast<uint2 *>(&hh4);
            // lo = f16(v - hi): the remainder straight from the packed hi halves (no conversion back)
            ll4[0] = (_Float16)mix_rem<0>(x[4 * g + 0], ph[g].x);
            ll4[1] = (_Float16)mix_rem<1>(x[4 * g + 1], ph[g].x);
            ll4[2] = (_Float16)mix_rem<0>(x[4 * g + 2], ph[g].y);
            ll4[3] = (_Float16)mix_rem<1>(x[4 * g + 3], ph[g].y);
            pl[g] = *reinterpret_cast<uint2 *>(&ll4);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            unsigned h0 = ph[2 * j].x, h1 = ph[2 * j].y, h2 = ph[2 * j + 1].x, h3 = ph[2 * j + 1].y;
            unsigned l0 = pl[2 * j].x, l1 = pl[2 * j].y, l2 = pl[2 * j + 1].x, l3 = pl[2 * j + 1].y;
            swap32(h0, h2); swap32(h1, h3);      // -> whole item of group 2j+h
            swap32(l0, l2); swap32(l1, l3);
            x[4 * j + 0] = __uint_as_float(h0); x[4 * j + 1] = __uint_as_float(h1);
            x[4 * j + 2] = __uint_as_float(h2); x[4 * j + 3] = __uint_as_float(h3);
            x[8 + 4 * j + 0] = __uint_as_float(l0); x[8 + 4 * j + 1] = __uint_as_float(l1);
            x[8 + 4 * j + 2] = __uint_as_float(l2); x[8 + 4 * j + 3] = __uint_as_float(l3);
          }
        }
      };
      auto store_block = [&](const f32x16 &x, const Tile &tile, int tm, int tn) {
        const int nb = (wn * 2 + tn) * 32;
        int orow, mc;
        const bool live = out_pos(tile, tm, orow, mc);
        if (live && !(a.dbg & 64)) {
          if (a.out_f16s) {
            uint4 *yh = reinterpret_cast<uint4 *>(a.y);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (GEN && nb + ch0 + 16 * j >= a.cout) continue;     // zero-padded channels of a narrow conv
              const unsigned it4 = item4_out(orow, mc, nb, j);
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              const u32x4 vhi = {__float_as_uint(x[4 * j]), __float_as_uint(x[4 * j + 1]),
                                 __float_as_uint(x[4 * j + 2]), __float_as_uint(x[4 * j + 3])};
              const u32x4 vlo = {__float_as_uint(x[8 + 4 * j]), __float_as_uint(x[8 + 4 * j + 1]),
                                 __float_as_uint(x[8 + 4 * j + 2]), __float_as_uint(x[8 + 4 * j + 3])};
              // the output (1.5 GB per launch) is read next by another launch, far beyond any cache:
              // streamed (nt) rather than write-allocated in L2 (+1.3 % measured)
              st_stream(vhi, reinterpret_cast<u32x4 *>(yh + it4));
              st_stream(vlo, reinterpret_cast<u32x4 *>(yh + it4 + 2u * (unsigned)L_st));
              if (GEN && psplit && (a.L_out & 1) && mc == a.L_out - 1) {
                // an odd row length: the odd phase is one position short - its last slot (position L_out) reads as zero
                const unsigned itz = it4 + (unsigned)((a.cout_pad >> 4) * 4 * L_st);
                const u32x4 z = {0u, 0u, 0u, 0u};
                st_stream(z, reinterpret_cast<u32x4 *>(yh + itz));
                st_stream(z, reinterpret_cast<u32x4 *>(yh + itz + 2u * (unsigned)L_st));
              }
            }
          } else {
            float *yf = reinterpret_cast<float *>(a.y) + ((size_t)orow * a.L_out + mc) * a.cout + ch0 + nb + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              if (ch0 + nb + 8 * g + 4 * h < a.cout)
                *reinterpret_cast<float4 *>(yf + 8 * g) =
                    make_float4(x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]);
          }
        }
      };
      // Cross-lane reduction over a half's 32 lanes by register halving: at each step a lane keeps half of
      // its registers and receives the partner's copy of that half (DPP within rows of 16, one bpermute
      // across rows), so 16 registers cost 16 exchanges instead of 80.  Afterwards lane i holds the
      // reduction of channel register r = 8*b2 + 4*b1 + 2*b0 + b3 (bits of i).
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
      // where lane i's reduced channel lives, and the partial row of this wave's 128 positions
      auto reduced_slot = [&](const Tile &tile, int tn, int &ch) -> size_t {
        const int r = 8 * (int)((i & 4) != 0) + 4 * (int)((i & 2) != 0) + 2 * (int)((i & 1) != 0) + (int)((i & 8) != 0);
        ch = ch0 + (wn * 2 + tn) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
        return ((size_t)tile.T * STRIPS + wm) * a.cout + ch;     // one partial row per wave strip (128 / 64 positions)
      };
      auto nmd_flush = [&](const Tile &tile, int tn) {
        const float v = lane_reduce(nmd_acc, [](float x, float y) { return x + y; });
        int ch;
        const size_t slot = reduced_slot(tile, tn, ch);
        if (i < 16 && tile.valid && ch < a.cout) a.nmd_out[slot] = v;
      };
      const bool has_nmd = (EP == JG_EP_GENERIC || EP == JG_EP_RUNTIME) ? a.nmd_out != nullptr : (EP & (JG_EP_NMD1 | JG_EP_NMD2)) != 0;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) nmd_acc[r] = 0.f;
        if constexpr (EP == JG_EP_RUNTIME) {
#pragma unroll
          for (int r = 0; r < 16; ++r) nmd_acc2[r] = 0.f;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) epi_block(acc[tm][tn], cur[0], tm, tn);
        if (has_nmd) nmd_flush(cur[0], tn);
        if constexpr (EP == JG_EP_RUNTIME) {
          if (a.nmd_out2 != nullptr) {
            const float v2 = lane_reduce(nmd_acc2, [](float x, float y) { return x + y; });
            int ch2;
            const size_t slot2 = reduced_slot(cur[0], tn, ch2);
            if (i < 16 && cur[0].valid && ch2 < a.cout) a.nmd_out2[slot2] = v2;
          }
        }
      }
      if (a.pool_out != nullptr) {
        // fused masked global max pool (layers.py:496-538): the block outputs are not stored at all; each
        // wave reduces its 128 positions to one partial row, pool_final takes the max over a window's rows
        float mkv[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          int orow, mc;
          const bool live = out_pos(cur[0], tm, orow, mc);
          const unsigned char mb = a.mask_out != nullptr ? a.mask_out[(size_t)orow * a.L_out + mc] : (unsigned char)1;
          mkv[tm] = (live && mb != 0) ? 1.f : 0.f;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          float pa[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) pa[r] = -INFINITY;
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) pa[r] = mkv[tm] != 0.f ? fmaxf(pa[r], acc[tm][tn][r]) : pa[r];
          const float v = lane_reduce(pa, [](float x, float y) { return fmaxf(x, y); });
          int ch;
          const size_t slot = reduced_slot(cur[0], tn, ch);
          if (i < 16 && cur[0].valid && ch < a.cout) a.pool_out[slot] = v;
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) store_block(acc[tm][tn], cur[0], tm, tn);
        }
      }
      if ((!(vmax <= 65000.0f) || vnan) && a.overflow != nullptr && a.dbg == 0) atomicOr(a.overflow, 1);
      JG_PRIO_MAIN();
    }
    if constexpr (!PIPE) zero_acc();          // (the pipelined loop starts every tile's accumulators from C = 0)
#pragma unroll
    for (int u = 0; u < NT; ++u) cur[u] = np[u];
    tiles_of(pass + 2, np);
    if constexpr (!LUT) load_bytes(np);   // position bytes of the pass after next
    JG_ST(4);
  }
#ifdef JG_VALU_PROBE
  if (probe[0] + probe[1] + probe[2] + probe[3] + probe[4] + probe[5] + probe[6] + probe[7] == 12345.678f) a.overflow[0] = 4;
#endif
  JG_ST_END;
}

template <int K, unsigned EP, bool FLAT = false, int CW = 128, bool TANH = false, bool PIPE = false>
int launch_ke(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  const int smem = jg_conv_f16_lds_bytes(K, a.dil);
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_f16x3_kernel<K, EP, false, FLAT, CW, TANH, PIPE>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int n_tiles = FLAT ? a.flat_tiles : a.rows * a.tiles_m;
  const int n_pairs = (n_tiles + NT - 1) / NT;
  // two 4-wave workgroups per CU when their LDS fits (<= 80 KB each): one's epilogue and stores overlap
  // the other's matrix-core steps.  (Forcing the two out of phase - by dispatch order or by a per-CU
  // arrival ticket - was measured and changes nothing; neither does storing each block early.)
  static const bool one_wg = jg_exp_env("JG_ONE_WG") != nullptr;      // experiment switch: one workgroup per CU
  int grid = ((smem <= 80 * 1024 && !one_wg) ? 2 : 1) * e->n_cu;
  if (grid > n_pairs) grid = n_pairs;
  ConvHArgs b = a;
  hipLaunchKernelGGL((conv_f16x3_kernel<K, EP, false, FLAT, CW, TANH, PIPE>), dim3((unsigned)grid), dim3(HT), (size_t)smem, s, b);
  JG_HIP(hipGetLastError());
#ifdef JG_STAMP
  {
    unsigned long long h[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    JG_HIP(hipStreamSynchronize(s));
    JG_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(jg_stamp_acc), sizeof(h)));
    JG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(jg_stamp_acc), z, sizeof(z)));
    const double tot = (double)h[5];
    fprintf(stderr, "STAMP k=%d ep=%u rows=%d grid=%d total_cyc/wave=%.0f wait=%.3f barrier=%.3f issue=%.3f lds_mfma=%.3f epilogue=%.3f\n",
            K, EP, a.rows, grid, tot / (grid * 4.0), h[0] / tot, h[1] / tot, h[2] / tot, (h[3] + h[6]) / tot, h[4] / tot);
  }
#endif
  return JG_OK;
}

// the residual stacks' hot tanh-GELU patterns (k = 5, 128 channels): the pipelined main loop when the engine asks for it
// (JG_OPT_CONV_PC = 2) and the geometry is the one it is built for
template <int K, unsigned EP, bool FLAT>
int launch_hot(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  if constexpr (K == 5) {
#ifdef JG_EXPERIMENT      /* (the pipelined main loop of the producer / consumer experiment: not in the shipped library) */
    if (e->conv_pc == 2 && a.dil == 3 && a.cc_in % 2 == 0 && a.dbg == 0) return launch_ke<5, EP, FLAT, 128, true, true>(e, a, s);
#endif
    return launch_ke<5, EP, FLAT, 128, true>(e, a, s);
  } else {
    return launch_ke<K, EP, FLAT>(e, a, s);
  }
}

#if JG_CONV_PART == 4
template <unsigned EP, int CW, bool TANH = false>
int launch_lut_e(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  const int smem = jg_conv_lut_lds_bytes(a.k, a.lut_vocab);
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_f16x3_kernel<0, EP, true, false, CW, TANH>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int n_tiles = a.rows * a.tiles_m;
  constexpr int tper = JG_LUT_WAVES / 2;
  const int n_pairs = (n_tiles + tper - 1) / tper;
  // (tile group, channel half), one workgroup per CU; a conv of <= 64 channels has only the first half
  const int grid = a.lut_one_half ? (n_pairs < e->n_cu ? n_pairs : e->n_cu)
                                  : 2 * (n_pairs < e->n_cu / 2 ? n_pairs : e->n_cu / 2);
  hipLaunchKernelGGL((conv_f16x3_kernel<0, EP, true, false, CW, TANH>), dim3((unsigned)grid), dim3(JG_LUT_WAVES * 64), (size_t)smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

int launch_lut(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  switch (a.ep) {
  // 128 output channels: no run-time output geometry; fewer: CW = 129 (channel guards, one table half for <= 64)
#define JG_CASE(ep) case (ep): return a.cout == 128 ? launch_lut_e<(ep), 128>(e, a, s) : launch_lut_e<(ep), 129>(e, a, s);
    JG_CASE(0u)
    JG_CASE(JG_EP_NMD1)
    JG_CASE(JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_AFF | JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ACT1)
    case (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1):       // the brain family's first conv: a tanh-GELU build beside the general one
      if (a.cout == 128 && a.act_kind == JG_ACT_GELU_TANH) return launch_lut_e<(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1), 128, true>(e, a, s);
      return a.cout == 128 ? launch_lut_e<(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1), 128>(e, a, s)
                           : launch_lut_e<(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1), 129>(e, a, s);
    JG_CASE(JG_EP_NMD1 | JG_EP_NORM1_DYT | JG_EP_ACT1)
    JG_CASE(JG_EP_ACT1 | JG_EP_NORM2_AFF)
    JG_CASE(JG_EP_RUNTIME)
#undef JG_CASE
    default: break;
  }
  jg_set_error("conv lut: stage pattern 0x%x has no compiled epilogue", a.ep);
  return JG_ERR_UNSUPPORTED;
}

#endif

// ---- per-translation-unit instantiation sets (JG_CONV_PART selects one; the kernel template is
// instantiated ~200 times, split over seven objects so that they compile in parallel) -------------
#define JG_ROW_CASES(K)                                                                              \
  switch (a.ep) {                                                                                    \
    case 0u: return launch_ke<K, 0u>(e, a, s);           /* plain affine: a LayerNorm follows */      \
    case (JG_EP_NMD1): return launch_ke<K, (JG_EP_NMD1)>(e, a, s);                                   \
    case (JG_EP_ACT1):                                   /* the residual stacks' three hot patterns: tanh-GELU build */ \
      if (K == 5 && a.act_kind == JG_ACT_GELU_TANH) return launch_hot<K, (JG_EP_ACT1), false>(e, a, s); \
      return launch_ke<K, (JG_EP_ACT1)>(e, a, s);                                                     \
    case (JG_EP_NORM1_AFF | JG_EP_ACT1): return launch_ke<K, (JG_EP_NORM1_AFF | JG_EP_ACT1)>(e, a, s); \
    case (JG_EP_NORM1_DYT | JG_EP_ACT1): return launch_ke<K, (JG_EP_NORM1_DYT | JG_EP_ACT1)>(e, a, s); \
    case (JG_EP_ADD | JG_EP_ACT1):                                                                   \
      if (K == 5 && a.act_kind == JG_ACT_GELU_TANH) return launch_hot<K, (JG_EP_ADD | JG_EP_ACT1), false>(e, a, s); \
      return launch_ke<K, (JG_EP_ADD | JG_EP_ACT1)>(e, a, s);                                         \
    case (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1):                                                 \
      return launch_ke<K, (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1)>(e, a, s);                      \
    case (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2):                       \
      if (K == 5 && a.act_kind == JG_ACT_GELU_TANH)                                                  \
        return launch_hot<K, (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2), false>(e, a, s); \
      return launch_ke<K, (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2)>(e, a, s); \
    case (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_DYT | JG_EP_ACT2):     \
      return launch_ke<K, (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_DYT | JG_EP_ACT2)>(e, a, s); \
    case (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1):                                                \
      return launch_ke<K, (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1)>(e, a, s);                     \
    case (JG_EP_NMD1 | JG_EP_NORM1_DYT | JG_EP_ACT1):                                                \
      return launch_ke<K, (JG_EP_NMD1 | JG_EP_NORM1_DYT | JG_EP_ACT1)>(e, a, s);                     \
    case (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2):                                    \
      return launch_ke<K, (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)>(e, a, s);         \
    case (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_DYT | JG_EP_ACT2):                  \
      return launch_ke<K, (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_DYT | JG_EP_ACT2)>(e, a, s); \
    case (JG_EP_ACT1 | JG_EP_NORM2_AFF): return launch_ke<K, (JG_EP_ACT1 | JG_EP_NORM2_AFF)>(e, a, s); \
    case (JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2):                                                \
      return launch_ke<K, (JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)>(e, a, s);                     \
    case (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1):                                    \
      return launch_ke<K, (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1)>(e, a, s);         \
    case (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2):     \
      return launch_ke<K, (JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)>(e, a, s); \
    case (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2):           /* a tap behind a block, no norm after it */ \
      return launch_ke<K, (JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)>(e, a, s);                           \
    case (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2):                                    \
      return launch_ke<K, (JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)>(e, a, s);         \
    case (JG_EP_NORM1_DYT): return launch_ke<K, (JG_EP_NORM1_DYT)>(e, a, s);                         \
    case (JG_EP_RUNTIME): return launch_ke<K, (JG_EP_RUNTIME)>(e, a, s);                             \
    default: break;                                                                                  \
  }                                                                                                  \
  jg_set_error("conv_f16x3: stage pattern 0x%x has no compiled epilogue", a.ep);                     \
  return JG_ERR_UNSUPPORTED;

}  // namespace

#if JG_CONV_PART == 1      // row-tiled, k = 5 (the residual stacks)
int jg_conv_f16_part_k5(jg_engine *e, const ConvHArgs &a, hipStream_t s) { JG_ROW_CASES(5) }
#elif JG_CONV_PART == 2    // row-tiled, k = 7 and 9
int jg_conv_f16_part_k79(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  if (a.k == 7) { JG_ROW_CASES(7) }
  { JG_ROW_CASES(9) }
}
#elif JG_CONV_PART == 3    // window-packed tiling, k = 5: the stage patterns of the residual stacks
int jg_conv_f16_part_flat(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  switch (a.ep) {
#define JG_CASE(ep) case (ep): return launch_ke<5, (ep), true>(e, a, s);
#define JG_CASE_HOT(ep)     /* the residual stacks' hot patterns: a tanh-GELU build beside the general one */ \
  case (ep):                                                                                                  \
    return a.act_kind == JG_ACT_GELU_TANH ? launch_hot<5, (ep), true>(e, a, s) : launch_ke<5, (ep), true>(e, a, s);
    JG_CASE_HOT(JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_AFF | JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ACT1)
    JG_CASE_HOT(JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1)
    JG_CASE_HOT(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_DYT | JG_EP_ACT2)
    JG_CASE(JG_EP_ACT1 | JG_EP_NORM2_AFF)
    JG_CASE(JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_DYT | JG_EP_ACT2)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)
    JG_CASE(JG_EP_RUNTIME)
#undef JG_CASE
#undef JG_CASE_HOT
    default: break;
  }
  jg_set_error("conv_f16x3: stage pattern 0x%x has no compiled window-packed epilogue", a.ep);
  return JG_ERR_UNSUPPORTED;
}
#elif JG_CONV_PART == 4    // first-layer table variant
int jg_conv_f16_part_lut(jg_engine *e, const ConvHArgs &a, hipStream_t s) { return launch_lut(e, a, s); }
#elif JG_CONV_PART >= 5 && JG_CONV_PART <= 7   // run-time output geometry, k = 5, both tilings: narrow convs (64 / 32
                                               // output channels) and the general 128-wide tile (CW = 129)
#if JG_CONV_PART == 5
#define JG_NARROW_CW 64
int jg_conv_f16_part_n64(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
#elif JG_CONV_PART == 6
#define JG_NARROW_CW 32
int jg_conv_f16_part_n32(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
#else
#define JG_NARROW_CW 129
int jg_conv_f16_part_g128(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
#endif
  switch (a.ep) {
#define JG_CASE(ep)                                                          \
  case (ep):                                                                 \
    return a.flat ? launch_ke<5, (ep), true, JG_NARROW_CW>(e, a, s) : launch_ke<5, (ep), false, JG_NARROW_CW>(e, a, s);
    JG_CASE(0u)
    JG_CASE(JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_AFF | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1)
    JG_CASE(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_NMD1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)
    JG_CASE(JG_EP_NORM1_DYT)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)
    JG_CASE(JG_EP_RUNTIME)
#undef JG_CASE
    default: break;
  }
  jg_set_error("conv_f16x3: stage pattern 0x%x has no compiled narrow-conv epilogue", a.ep);
  return JG_ERR_UNSUPPORTED;
}
#elif JG_CONV_PART >= 8 && JG_CONV_PART <= 13  // run-time output geometry, k = 7 and 9, row-tiled: 64 / 32-channel tiles and the
                                               // general 128-wide tile (other widths, more than 128 channels, stride 2)
#define JG_X_CW (((JG_CONV_PART - 8) % 3) == 0 ? 64 : ((JG_CONV_PART - 8) % 3) == 1 ? 32 : 129)
#define JG_X_K ((JG_CONV_PART - 8) / 3 == 0 ? 7 : 9)
#define JG_X_NAME2(p) jg_conv_f16_part_x##p
#define JG_X_NAME(p) JG_X_NAME2(p)
int JG_X_NAME(JG_CONV_PART)(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  switch (a.ep) {
#define JG_CASE(ep) case (ep): return launch_ke<JG_X_K, (ep), false, JG_X_CW>(e, a, s);
    JG_CASE(0u)
    JG_CASE(JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_AFF | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1)
    JG_CASE(JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ACT1)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1)
    JG_CASE(JG_EP_NMD1)
    JG_CASE(JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)
    JG_CASE(JG_EP_NORM1_DYT)
    JG_CASE(JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2)
    JG_CASE(JG_EP_RUNTIME)
#undef JG_CASE
    default: break;
  }
  jg_set_error("conv_f16x3: stage pattern 0x%x has no compiled k = 7 / 9 run-time-geometry epilogue", a.ep);
  return JG_ERR_UNSUPPORTED;
}
#endif
