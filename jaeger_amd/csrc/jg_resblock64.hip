// A whole 64-channel residual block in one launch (ResidualBlock, nnlib/v2/layers.py:1882-1915, stride 1, no bypass):
//     h = gelu(bn1(conv1(x * m0)))          y = gelu(bn2(conv2(h * m1)) + x)
// - the second stage of pyramid-shaped models (train_config/nn_config_baseline.yaml: three five-tap blocks, dilation 2, at
// 330 positions).  Same idea as jg_resblock.hip (the intermediate lives in LDS only, the input is read once and serves
// conv1 AND the shortcut), different division of labour: at 64 channels one conv's weight fragments for one half of the
// output channels already fill 160 VGPRs (5 taps x 4 chunks x 2 planes), so a wave cannot hold both convs.  The eight
// waves of a workgroup (two per SIMD, one workgroup per CU) take ROLES instead - wave w runs on SIMD w % 4, so every SIMD
// hosts one wave of each role:
//     waves 0-3: conv1 of tile s      (output channels 32 ct.., positions 32 blk..) -> intermediate image H[s & 1] in LDS
//     waves 4-7: conv2 of tile s - 1  (same split) from H[(s - 1) & 1]; its epilogue (shortcut from X[(s - 2) % 5], GELU,
//                re-split, store) is DEFERRED to the head of the next step
// so that on each SIMD one wave's matrix-core phase runs under the other's epilogue by construction (conv1: MFMA then
// epilogue, then the fetch of tile s + 2; conv2: epilogue of the previous tile, then MFMA): a three-stage pipeline over
// the workgroup's tiles with ONE pair of barriers per step.  The conv1 waves own everything that is loaded (the input
// image by global_load_lds DMA into a ring of five buffers and the mask bytes, both two tiles ahead and waited for with a
// counted vmcnt; the zero-fill of masked input rows); the conv2 waves only
// store, so nothing they wait on sits behind a store in the memory queue (loads and stores return in order on gfx9): the
// bits they need (output mask, "shortcut readable from the image") reach them through LDS.  Tiles in which a live output
// position is masked on the input side take the shortcut from HBM instead (the image holds conv1's zero there): a
// per-tile flag selects that path, which alone contains loads.
//
// Arithmetic: the split-f16 scheme of conv_f16x3_kernel (three v_mfma_f32_32x32x16_f16 per product, f32 accumulation,
// weights as the A operand); a tile is two blocks of 32 intermediate positions, its outputs the 64 - 4 d positions whose
// taps stay inside.  Kernel size and dilation are template parameters (5 or 3 taps, dilation 1 or 2): with the image
// geometry known at compile time every LDS operand address is one register + an immediate offset - the run-time form needed
// 35 registers more than the 256 a wave of a 512-thread workgroup has.
//
// Measured (MI355X, pyramid bench, 7 644 rows x 330 positions, k 5, d 2): 676 - 683 us per launch against 951 us for the two
// conv launches it replaces.  Per step of 56 output positions the SIMD's matrix pipe is busy 3 840 of ~6 900 cycles: the
// epilogues' vector work (two transcendentals and ~25 other instructions per value pair, SGPR-spill traffic) only partly
// hides beside the partner's dependency-paced MFMAs (phase stamps: -DR6_STAMP).
#include <math.h>

#include <algorithm>
#include <type_traits>

#include "jg_resblock64.h"

#include "jg_conv_dev.h"

namespace {

constexpr int R6_C = 64;
constexpr int R6_CC = R6_C / 16;          // 16-channel chunks
constexpr int R6_PL = R6_CC * 4;          // (chunk, plane, half) item rows per position: 16
constexpr int R6_NB = 2;                  // 32-position blocks of the intermediate per tile: one per wave of a role and channel half
constexpr int R6_RH = 32 * R6_NB;
constexpr int R6_XR = 5;                  // input images in flight: two landing | conv1 | conv2's MFMA | conv2's epilogue (shortcut)
constexpr int R6_OR = 3;                  // per-tile bits for the conv2 waves: written at step s, read at step s + 2
constexpr int R6_DMA = 6;                 // DMA calls per conv1 wave and image at most: 16 x (64 + 32) items / 256

__device__ __forceinline__ void r6_swap32(unsigned &lo_half_keeps, unsigned &hi_half_keeps) {
  const auto r = __builtin_amdgcn_permlane32_swap(lo_half_keeps, hi_half_keeps, false, false);
  lo_half_keeps = r[0];
  hi_half_keeps = r[1];
}
// LDS traffic of this wave done, then the workgroup barrier - WITHOUT the vmcnt(0) __syncthreads() would add (the conv2
// waves' stores drain on their own)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// split four f32 (this lane's channels 4 hh .. 4 hh + 3 of an 8-channel group) into packed hi / lo halfs
__device__ __forceinline__ void r6_split(f32x2 v01, f32x2 v23, unsigned &h0, unsigned &h1, unsigned &l0, unsigned &l1) {
  const half2_t h01 = {(_Float16)v01.x, (_Float16)v01.y}, h23 = {(_Float16)v23.x, (_Float16)v23.y};
  h0 = *reinterpret_cast<const unsigned *>(&h01);
  h1 = *reinterpret_cast<const unsigned *>(&h23);
  const half2_t l01 = {(_Float16)mix_rem<0>(v01.x, h0), (_Float16)mix_rem<1>(v01.y, h0)};
  const half2_t l23 = {(_Float16)mix_rem<0>(v23.x, h1), (_Float16)mix_rem<1>(v23.y, h1)};
  l0 = *reinterpret_cast<const unsigned *>(&l01);
  l1 = *reinterpret_cast<const unsigned *>(&l23);
}

// one block of 32 positions x 32 output channels: K = 64 input channels x RK taps, operands out of an LDS image whose item
// rows are `rs` items apart; `img` points at this lane's first-tap item (every tap of it inside the image).  Operands one
// step ahead of the matrix cores.
template <int RK, int rs, int d>
__device__ __forceinline__ f32x16 r6_block(const uint4 (&wf)[RK][R6_CC][2], const uint4 *img) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  uint4 vh = img[0], vl = img[2 * rs];
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
  for (int st = 0; st < R6_CC * RK; ++st) {
    const int cc = st / RK, t = st % RK;
    uint4 nh = vh, nl = vl;
    if (st + 1 < R6_CC * RK) {
      const int c2 = (st + 1) / RK, t2 = (st + 1) % RK;
      nh = img[(c2 * 4 + 0) * rs + t2 * d];
      nl = img[(c2 * 4 + 2) * rs + t2 * d];
    }
    const half8 wh = *reinterpret_cast<const half8 *>(&wf[t][cc][0]);
    const half8 wl = *reinterpret_cast<const half8 *>(&wf[t][cc][1]);
    const half8 xh = *reinterpret_cast<const half8 *>(&vh), xl = *reinterpret_cast<const half8 *>(&vl);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
    if (st + 1 < R6_CC * RK) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the next step's two LDS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                 // ... then this step's MFMAs
    vh = nh;
    vl = nl;
  }
  return c;
}

#ifdef R6_STAMP
// experiment build: shader cycles per role and phase, summed over the first wave of each role of every workgroup
__device__ unsigned long long r6_stamp[8];
#define R6_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define R6_T(var)
#endif

// A wave's vector phases (epilogues, address arithmetic) run above its SIMD partner's matrix phase: the partner's MFMAs are
// paced by their own dependency chain, the vector instructions fill the issue slots between them (measured on the pyramid's
// blocks: 793 -> 714 us per launch; a static priority for waves 4-7 instead: 718).
#define R6_PRIO_VALU() __builtin_amdgcn_s_setprio(1)
#define R6_PRIO_MFMA() __builtin_amdgcn_s_setprio(0)

// a tile of the launch: (row, tile in the row), advanced by the grid size without dividing
struct R6Tile {
  int row, tile;
};

template <int RK, int RD>
__global__ __launch_bounds__(512, 1) void resblock64_kernel(JgResBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wid >> 2, ct = wid & 1, blk = (wid >> 1) & 1;     // conv1 / conv2; output channels 32 ct ..; positions 32 blk ..
  const int i = lane & 31, hh = lane >> 5;
  // (kernel size and dilation are compile-time: every LDS operand address is one base register + an immediate offset)
  constexpr int d = RD, halo = (RK - 1) * d, pad = halo / 2;
  constexpr int RX = R6_RH + halo;
  constexpr int x_items = R6_PL * RX, x_slot = (x_items + 63) & ~63;   // (a DMA call moves 64 items: whole calls per buffer)
  uint4 *Xbuf = lds;                               // [5 buffers][16][RX] (+ slack up to x_slot)
  uint4 *Hbuf = lds + R6_XR * x_slot;              // [2 buffers][16][RH]
  float *epiL = reinterpret_cast<float *>(Hbuf + 2 * R6_PL * R6_RH);   // [2 convs][scale | shift][64]
  unsigned *obL = reinterpret_cast<unsigned *>(epiL + 4 * R6_C);       // [3][RH] per output position: bit 0 shortcut in the image, bit 1 m2
  unsigned *anyL = obL + R6_OR * R6_RH;            // [3]: some live output of the tile needs the HBM shortcut
  if (tid < 4 * R6_C) epiL[tid] = a.epi[tid];
  // this wave's conv, its half of the output channels: weight fragments in registers for the life of the workgroup
  uint4 wf[RK][R6_CC][2];
#pragma unroll
  for (int t = 0; t < RK; ++t)
#pragma unroll
    for (int cc = 0; cc < R6_CC; ++cc)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        wf[t][cc][p] = a.wfrag[((((((size_t)role * RK + t) * R6_CC + cc) * 2 + p) * 2) + ct) * 64 + lane];
  const unsigned ldsX = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)lds);
  const int n_units = a.rows * a.tiles_per_row;
  const int G = gridDim.x, g_rows = G / a.tiles_per_row, g_tiles = G - g_rows * a.tiles_per_row;
  const int n_local = (n_units - (int)blockIdx.x + G - 1) / G;
  auto advance = [&](R6Tile t) -> R6Tile {
    t.row += g_rows;
    t.tile += g_tiles;
    if (t.tile >= a.tiles_per_row) { t.tile -= a.tiles_per_row; ++t.row; }
    return t;
  };
  float vmax = 0.f;
  bool vnan = false;
  const int L_st = a.psplit ? ((a.L + 1) >> 1) : a.L;

  // (conv1 waves) the DMA calls of an image: call c of wave w moves items 256 c + 64 w + lane; an item is (item row, position
  // in the image) - the same for every tile: call 0's split is kept, the later calls step it by 256 items
  int dma_row0, dma_pos0;
  udivmod24(64 * (wid & 3) + lane, RX, 1.0f / (float)RX, dma_row0, dma_pos0);
  constexpr int step_rows = 256 / RX, step_pos = 256 - step_rows * RX;   // (64 < RX <= 96)
  // (conv1 waves) the mask bytes of a tile, loaded in front of that tile's DMA - every wave a fixed number of loads from
  // clamped addresses (the step's counted wait relies on it):
  //   bit 0      (waves 0, 1: tid = input row of the image) the row is zero: outside the row, or masked (conv1 reads x * m0)
  //   bit 1      m1 at this lane's intermediate position (conv2 reads h * m1; outside the row: SAME padding zeros)
  //   bit 4      (wave 2: lane = output position) the image holds the shortcut value (m0 set)
  //   bit 5      (wave 2: lane = output position) output mask (phase-split store only)
  const uint8_t *m2p = a.psplit ? a.m2 : nullptr;
  const int n_mask_loads = (a.m0 != nullptr) + (a.m1 != nullptr) + (wid == 2 ? (a.m0 != nullptr) + (m2p != nullptr) : 0);
  // The bytes are only REQUESTED here (load_raw) and turned into bits one step later (pack_masks), when they have long
  // landed: arithmetic on them right behind the loads would park the wave for a memory round trip in every step.
  struct R6Raw {
    unsigned b0, b1, b4, b5;
  };
  auto load_raw = [&](R6Tile t) -> R6Raw {
    const int p0 = t.tile * a.tile_out, hp0 = p0 - pad, xp0 = hp0 - pad;
    const unsigned row_off = (unsigned)t.row * (unsigned)a.L;
    auto get = [&](const uint8_t *m, int p) -> unsigned {
      return m != nullptr ? (unsigned)m[row_off + (unsigned)min(max(p, 0), a.L - 1)] : 1u;
    };
    R6Raw r;
    r.b0 = get(a.m0, xp0 + tid);
    r.b1 = get(a.m1, hp0 + 32 * blk + i);
    r.b4 = 1u;
    r.b5 = 1u;
    if (wid == 2) {
      r.b4 = get(a.m0, p0 + lane);
      r.b5 = get(m2p, p0 + lane);
    }
    return r;
  };
  auto pack_masks = [&](const R6Raw &r, R6Tile t) -> unsigned {
    const int p0 = t.tile * a.tile_out, hp0 = p0 - pad, xp0 = hp0 - pad;
    auto bit = [&](unsigned raw, int p, unsigned outside) -> unsigned {
      return (unsigned)p < (unsigned)a.L ? (unsigned)(raw != 0u) : outside;
    };
    return (bit(r.b0, xp0 + tid, 0u) ^ 1u) | (bit(r.b1, hp0 + 32 * blk + i, 0u) << 1) | (bit(r.b4, p0 + lane, 1u) << 4) |
           (bit(r.b5, p0 + lane, 1u) << 5);
  };
  // (conv1 waves) the input image of a tile: DMA, 64 items per wave and call; positions outside the row are fetched from a
  // clamped address and zeroed like masked ones once the image has landed
  auto issue_x = [&](R6Tile t, int buf) {
    const int xp0 = t.tile * a.tile_out - halo;
    const unsigned row_off = (unsigned)t.row * (unsigned)(R6_PL * a.L);
    int cph = dma_row0, j = dma_pos0;
#pragma unroll
    for (int c = 0; c < R6_DMA; ++c) {
      const int q0 = 256 * c + 64 * wid;
      if (q0 < x_items) {
        const int pos = min(max(xp0 + j, 0), a.L - 1);
        // (lanes past the image's end fetch some item of the last row again, into the slack behind the image)
        const unsigned voff = (row_off + __umul24((unsigned)min(cph, R6_PL - 1), (unsigned)a.L) + (unsigned)pos) * 16u;
        glds16_nt(a.xh, voff, __builtin_amdgcn_readfirstlane(ldsX + (unsigned)(buf * x_slot + q0) * 16u));
      }
      cph += step_rows;
      j += step_pos;
      if (j >= RX) { j -= RX; ++cph; }
    }
  };

  // ---- conv1 of a tile -> intermediate image: this wave's block and channel half ----------------------------------
  auto conv1_block = [&](const uint4 *Ximg, uint4 *Himg, unsigned mk) {
    const f32x16 c = r6_block<RK, RX, d>(wf, Ximg + hh * RX + 32 * blk + i);
    const float m1f = (mk & 2u) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {                  // channel groups 2j and 2j + 1 of this wave's 32 channels
      unsigned ph[4], pl[4];                       // [2 groups][2 dwords]: hi / lo halfs of this lane's four channels
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * j + gg;
        const float4 sc = *reinterpret_cast<const float4 *>(epiL + 32 * ct + 8 * g + 4 * hh);
        const float4 of = *reinterpret_cast<const float4 *>(epiL + R6_C + 32 * ct + 8 * g + 4 * hh);
        f32x2 v01 = {fmaf(c[4 * g + 0], sc.x, of.x), fmaf(c[4 * g + 1], sc.y, of.y)};
        f32x2 v23 = {fmaf(c[4 * g + 2], sc.z, of.z), fmaf(c[4 * g + 3], sc.w, of.w)};
        v01 = fast_gelu2(v01) * f32x2{m1f, m1f};
        v23 = fast_gelu2(v23) * f32x2{m1f, m1f};
        r6_split(v01, v23, ph[2 * gg], ph[2 * gg + 1], pl[2 * gg], pl[2 * gg + 1]);
      }
      r6_swap32(ph[0], ph[2]); r6_swap32(ph[1], ph[3]);            // -> the whole 8-channel item of group 2j + hh
      r6_swap32(pl[0], pl[2]); r6_swap32(pl[1], pl[3]);
      const int Gq = 2 * j + hh, chunk = 2 * ct + (Gq >> 1);
      Himg[(chunk * 4 + 0 + (Gq & 1)) * R6_RH + 32 * blk + i] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      Himg[(chunk * 4 + 2 + (Gq & 1)) * R6_RH + 32 * blk + i] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    }
  };

  // ---- conv2's epilogue: affine + shortcut + GELU -> HBM ------------------------------------------------------------
  auto conv2_finish = [&](auto slow_tag, const f32x16 &c, R6Tile t, const uint4 *Ximg, const unsigned *ob) {
    constexpr bool SLOW = decltype(slow_tag)::value;
    const int row = t.row, p0 = t.tile * a.tile_out;
    const int o = 32 * blk + i, p = p0 + o;                        // output index inside the tile, position in the row
    const bool live = o < a.tile_out && p < a.L;
    const unsigned bits = ob[o];
    const bool from_img = !SLOW || !live || (bits & 1u) != 0u;
    const float m2f = (bits & 2u) ? 1.f : 0.f;
    const char *simg = reinterpret_cast<const char *>(Ximg + min(o + halo, RX - 1)) + 8 * hh;
    const char *sgl = reinterpret_cast<const char *>(a.xh) + 8 * hh;
    // byte offset of this lane's first output item (the launcher bounds the tensor to 32-bit byte offsets); the lo plane
    // sits 2 L items behind the hi plane, the next chunk 4 L
    const unsigned out_step = (unsigned)(4 * L_st) * 16u;
    const unsigned out_item = a.psplit ? (unsigned)(((row * 2 * R6_CC + (p & 1) * R6_CC + 2 * ct) * 4 + hh) * L_st + (p >> 1))
                                       : (unsigned)(((row * R6_CC + 2 * ct) * 4 + hh) * a.L + p);
    const unsigned out_off = out_item * 16u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned ph[4], pl[4];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * j + gg, chunk = 2 * ct + (g >> 1);
        uint2 sh, sl;
        if (from_img) {
          sh = *reinterpret_cast<const uint2 *>(simg + (size_t)((chunk * 4 + 0 + (g & 1)) * RX) * 16);
          sl = *reinterpret_cast<const uint2 *>(simg + (size_t)((chunk * 4 + 2 + (g & 1)) * RX) * 16);
        } else {
          sh = *reinterpret_cast<const uint2 *>(sgl + (((size_t)row * R6_PL + chunk * 4 + 0 + (g & 1)) * a.L + p) * 16);
          sl = *reinterpret_cast<const uint2 *>(sgl + (((size_t)row * R6_PL + chunk * 4 + 2 + (g & 1)) * a.L + p) * 16);
        }
        const float4 sc = *reinterpret_cast<const float4 *>(epiL + 2 * R6_C + 32 * ct + 8 * g + 4 * hh);
        const float4 of = *reinterpret_cast<const float4 *>(epiL + 3 * R6_C + 32 * ct + 8 * g + 4 * hh);
        f32x2 v01 = {fmaf(c[4 * g + 0], sc.x, of.x), fmaf(c[4 * g + 1], sc.y, of.y)};
        f32x2 v23 = {fmaf(c[4 * g + 2], sc.z, of.z), fmaf(c[4 * g + 3], sc.w, of.w)};
        v01 += f32x2{mix_sum<0>(sh.x, sl.x), mix_sum<1>(sh.x, sl.x)};
        v23 += f32x2{mix_sum<0>(sh.y, sl.y), mix_sum<1>(sh.y, sl.y)};
        v01 = fast_gelu2(v01) * f32x2{m2f, m2f};
        v23 = fast_gelu2(v23) * f32x2{m2f, m2f};
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v01.x), fabsf(v01.y))), fmaxf(fabsf(v23.x), fabsf(v23.y)));
        vnan = vnan || __builtin_isunordered(v01.x, v01.y) || __builtin_isunordered(v23.x, v23.y);
        r6_split(v01, v23, ph[2 * gg], ph[2 * gg + 1], pl[2 * gg], pl[2 * gg + 1]);
      }
      r6_swap32(ph[0], ph[2]); r6_swap32(ph[1], ph[3]);
      r6_swap32(pl[0], pl[2]); r6_swap32(pl[1], pl[3]);
      if (live) {
        // group 2j + hh of this wave's channels = item row (chunk 2 ct + j, half hh): 4 L items further per j
        const unsigned off = out_off + (unsigned)j * out_step;
        const u32x4 vhi = {ph[0], ph[1], ph[2], ph[3]}, vlo = {pl[0], pl[1], pl[2], pl[3]};
        char *yb = reinterpret_cast<char *>(a.y);
        __builtin_nontemporal_store(vhi, reinterpret_cast<u32x4 *>(yb + (size_t)off));
        __builtin_nontemporal_store(vlo, reinterpret_cast<u32x4 *>(yb + (size_t)(off + out_step / 2)));
        if (a.psplit && (a.L & 1) && p == a.L - 1) {               // the odd phase of an odd row is one position short: zero
          const unsigned offz = off + (unsigned)(R6_CC * 4 * L_st) * 16u;
          const u32x4 z = {0u, 0u, 0u, 0u};
          __builtin_nontemporal_store(z, reinterpret_cast<u32x4 *>(yb + (size_t)offz));
          __builtin_nontemporal_store(z, reinterpret_cast<u32x4 *>(yb + (size_t)(offz + out_step / 2)));
        }
      }
    }
  };

  // tiles of step s + 2 (being fetched) .. s (conv1) .. s - 1 (conv2's MFMA) .. s - 2 (conv2's epilogue), their image slots
  R6Tile t_cur;
  t_cur.row = (int)blockIdx.x / a.tiles_per_row;
  t_cur.tile = (int)blockIdx.x - t_cur.row * a.tiles_per_row;
  R6Tile t_m1 = t_cur, t_m2 = t_cur, t_n1 = advance(t_cur), t_n2 = advance(t_n1);
  int xs = 0, xs_m2 = 0, xs_m1 = 0, xs_n2 = 2;
  // this wave's vector-memory operations per fetched tile (DMA calls + mask loads): the head of a step waits until only the
  // NEXT tile's are outstanding (they return in order)
  const int n_vm = (x_items - 64 * (wid & 3) + 255) / 256 + n_mask_loads;
  auto wait_outstanding = [&](int n) {
    switch (n) {
      case 3: wait_vm<3>(); break;
      case 4: wait_vm<4>(); break;
      case 5: wait_vm<5>(); break;
      case 6: wait_vm<6>(); break;
      case 7: wait_vm<7>(); break;
      case 8: wait_vm<8>(); break;
      case 9: wait_vm<9>(); break;
      case 10: wait_vm<10>(); break;
      default: wait_vm<0>(); break;
    }
  };
  unsigned mk = 0;                                 // bits of tile s
  R6Raw raw_n1 = {1u, 1u, 1u, 1u};                 // bytes of tile s + 1, in flight
  if (role == 0) {
    raw_n1 = load_raw(t_cur);
    issue_x(t_cur, 0);
    mk = pack_masks(raw_n1, t_cur);
    if (n_local > 1) {
      raw_n1 = load_raw(t_n1);
      issue_x(t_n1, 1);
    }
  }
  f32x16 c2;                                       // (conv2 waves) accumulators of tile s - 2 awaiting their epilogue
#pragma unroll
  for (int r = 0; r < 16; ++r) c2[r] = 0.f;
  int ob = 0;                                      // ring slot of tile s's bits
#ifdef R6_STAMP
  unsigned long long st_acc[4] = {0, 0, 0, 0};
#endif
  for (int s = 0; s <= n_local + 1; ++s) {
    const bool have1 = s < n_local;
    uint4 *Ximg = Xbuf + xs * x_slot;
    R6_T(t0);
    if (role == 0) {                               // this wave's share of tile s's image, its mask bytes
      if (s + 1 < n_local) wait_outstanding(n_vm);
      else wait_vm<0>();
    }
    lds_barrier();                                 // ... every share; and step s - 1 is over (its buffers may be reused)
    if (role == 0 && have1) {
      if ((mk & 1u) && tid < RX) {                 // a row is zeroed across all 16 item rows
#pragma unroll
        for (int c = 0; c < R6_PL; ++c) Ximg[c * RX + tid] = make_uint4(0u, 0u, 0u, 0u);
      }
      if (wid == 2) {                              // (lane = output position of the tile)
        obL[ob * R6_RH + lane] = (mk >> 4) & 3u;
        const bool need_hbm = lane < a.tile_out && t_cur.tile * a.tile_out + lane < a.L && (mk & 16u) == 0u;
        const bool any = __ballot(need_hbm) != 0ull;
        if (lane == 0) anyL[ob] = any ? 1u : 0u;
      }
    }
    lds_barrier();                                 // image of tile s complete; conv2's bits published
    R6_T(t1);
    if (role == 0) {
      if (have1) conv1_block(Ximg, Hbuf + (s & 1) * R6_PL * R6_RH, mk);
      R6_T(t2);
      R6_PRIO_VALU();
      mk = pack_masks(raw_n1, t_n1);               // (requested a whole step ago)
      if (s + 2 < n_local) {                       // fetch two tiles ahead: issued behind this step's matrix-core phase, under conv2's
        raw_n1 = load_raw(t_n2);
        issue_x(t_n2, xs_n2);
      }
#ifdef R6_STAMP
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      st_acc[0] += t1 - t0; st_acc[1] += t3 - t2; st_acc[2] += t2 - t1; st_acc[3] += 1ull;
#endif
    } else {
      R6_T(t2);
      R6_PRIO_VALU();
      if (s >= 2) {                                // the epilogue of tile s - 2: runs under the conv1 wave's matrix-core phase
        const int os = ob >= 2 ? ob - 2 : ob + R6_OR - 2;
        const unsigned slow = __builtin_amdgcn_readfirstlane(anyL[os]);
        if (32 * blk < a.tile_out) {
          if (slow) conv2_finish(std::true_type{}, c2, t_m2, Xbuf + xs_m2 * x_slot, obL + os * R6_RH);
          else conv2_finish(std::false_type{}, c2, t_m2, Xbuf + xs_m2 * x_slot, obL + os * R6_RH);
        }
      }
      R6_T(t3);
      R6_PRIO_MFMA();
      if (s >= 1 && s <= n_local && 32 * blk < a.tile_out)         // dead lanes (past the tile's outputs) repeat its last position
        c2 = r6_block<RK, R6_RH, d>(wf, Hbuf + ((s - 1) & 1) * R6_PL * R6_RH + hh * R6_RH + min(32 * blk + i, R6_RH - halo - 1));
#ifdef R6_STAMP
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();
      st_acc[0] += t1 - t0; st_acc[1] += t3 - t2; st_acc[2] += t4 - t3; st_acc[3] += 1ull;
#endif
    }
    if (role == 0) R6_PRIO_MFMA();
    t_m2 = t_m1; t_m1 = t_cur; t_cur = t_n1; t_n1 = t_n2; t_n2 = advance(t_n2);
    xs_m2 = xs_m1; xs_m1 = xs;
    xs = xs == R6_XR - 1 ? 0 : xs + 1;
    xs_n2 = xs_n2 == R6_XR - 1 ? 0 : xs_n2 + 1;
    ob = ob == R6_OR - 1 ? 0 : ob + 1;
  }
#ifdef R6_STAMP
  if (tid == 0 || tid == 256)
    for (int q = 0; q < 4; ++q) atomicAdd(&r6_stamp[(tid >> 8) * 4 + q], st_acc[q]);
#endif
  if ((!(vmax <= 65000.0f) || vnan) && a.overflow != nullptr) atomicOr(a.overflow, 1);
}

size_t r6_smem(int k, int dil) {
  const int RX = R6_RH + (k - 1) * dil;
  const int x_slot = (R6_PL * RX + 63) & ~63;
  return (size_t)(R6_XR * x_slot + 2 * R6_PL * R6_RH) * 16 + 4 * R6_C * sizeof(float) + (R6_OR * R6_RH + R6_OR) * sizeof(unsigned);
}

}  // namespace

// a tile computes two blocks of 32 intermediate positions; its outputs are the 64 - 4 d positions whose taps stay inside
void jg_resblock64_tiling(int L, int k, int dil, int *nb, int *tile_out, int *tiles) {
  *nb = R6_NB;
  *tile_out = R6_RH - (k - 1) * dil;
  *tiles = (L + *tile_out - 1) / *tile_out;
}

bool jg_resblock64_supports(int c, int k, int dil) {
  // (half of the tile at least is output: a longer reach is left to the layer-by-layer kernels)
  return c == R6_C && (k == 5 || k == 3) && (dil == 1 || dil == 2) && r6_smem(k, dil) <= 160 * 1024;
}

int jg_launch_resblock64(jg_engine *e, const JgResBlockArgs &a, hipStream_t s) {
  JG_REQUIRE(a.xh != nullptr && a.y != nullptr && a.wfrag != nullptr && a.epi != nullptr && a.nb == R6_NB &&
                 (a.k == 5 || a.k == 3) && (a.dil == 1 || a.dil == 2) && a.tile_out == R6_RH - (a.k - 1) * a.dil &&
                 a.tiles_per_row * a.tile_out >= a.L,
             JG_ERR_INVALID, "resblock64: bad geometry (nb %d, tile_out %d, tiles %d, L %d, dilation %d)", a.nb, a.tile_out,
             a.tiles_per_row, a.L, a.dil);
  JG_REQUIRE((double)a.rows * R6_PL * (a.L + 1) * 16.0 < 4.2e9, JG_ERR_UNSUPPORTED,
             "resblock64: activation tensor of %d rows exceeds the 32-bit byte offsets of the DMA and the stores", a.rows);
  if (a.rows == 0 || a.L <= 0) return JG_OK;
  const size_t smem = r6_smem(a.k, a.dil);
  const long units = (long)a.rows * a.tiles_per_row;
  const int grid = (int)std::min<long>(units, (long)e->n_cu);
  auto launch = [&](auto kernel) -> int {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), smem, s, a);
    return JG_OK;
  };
  int rc = JG_ERR_UNSUPPORTED;
  if (a.k == 5 && a.dil == 1) rc = launch(resblock64_kernel<5, 1>);
  else if (a.k == 5 && a.dil == 2) rc = launch(resblock64_kernel<5, 2>);
  else if (a.k == 3 && a.dil == 1) rc = launch(resblock64_kernel<3, 1>);
  else if (a.k == 3 && a.dil == 2) rc = launch(resblock64_kernel<3, 2>);
  if (rc != JG_OK) return rc;
  JG_HIP(hipGetLastError());
#ifdef R6_STAMP
  static int n_launch = 0;
  if (++n_launch % 300 == 0) {
    unsigned long long h[8], z[8] = {0};
    JG_HIP(hipDeviceSynchronize());
    JG_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(r6_stamp), sizeof(h)));
    JG_HIP(hipMemcpyToSymbol(HIP_SYMBOL(r6_stamp), z, sizeof(z)));
    fprintf(stderr, "r6 stamp (cycles per step): conv1 wait %.0f issue %.0f compute %.0f | conv2 wait %.0f epilogue %.0f mfma %.0f\n",
            (double)h[0] / (double)h[3], (double)h[1] / (double)h[3], (double)h[2] / (double)h[3], (double)h[4] / (double)h[7],
            (double)h[5] / (double)h[7], (double)h[6] / (double)h[7]);
  }
#endif
  return JG_OK;
}
