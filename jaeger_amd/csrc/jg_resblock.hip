// A whole narrow residual block in one launch (ResidualBlock, nnlib/v2/layers.py:1882-1915, stride 1, no bypass):
//     h = gelu(bn1(conv1(x * m0)))          y = gelu(bn2(conv2(h * m1)) + x)
// for 32-channel stages of pyramid-shaped models (train_config/nn_config_baseline.yaml: eight five-tap convs at 659
// positions) and the three-tap blocks of the small-window family on longer rows (nn_config_500bp_baseline.yaml at 1 500 bp).  Layer by layer those convs are bound by their HBM round trips, not by the matrix cores (1.3 - 1.9 GB moved
// per launch of 10 GFLOP): here a workgroup keeps a position tile of the block input in LDS (F16S items, brought in by
// global_load_lds DMA), runs conv1 on the matrix cores, writes the re-split intermediate back to LDS, runs conv2 from
// there and adds the shortcut out of the SAME input image - the intermediate tensor never exists in HBM, the input is
// read once (plus 4 x dilation halo rows per tile) and the output written once: 2 passes over the tensor instead of 5.
//
// Arithmetic is the split-f16 scheme of conv_f16x3_kernel (jg_conv_f16_impl.h): every f32 operand an f16 hi / lo pair,
// three v_mfma_f32_32x32x16_f16 per product, f32 accumulation, weights as the MFMA A operand (an accumulator register =
// one channel, a lane = one position).  Both convs' weight fragments (2 x 5 taps x 2 chunks x 2 planes = 40 items per
// lane) live in registers for the workgroup's life: 160 VGPRs, two workgroups of four waves per CU (one's epilogue and
// waits run under the other's matrix-core work), the input image double-buffered: the next tile's DMA lands under the
// current tile's work.
//
// Tile: four blocks of 32 intermediate positions, one per wave; the tile's outputs are the 128 - 4 d positions whose taps
// stay inside it (3 % of the input is read twice at d = 1).
#include <math.h>

#include <algorithm>

#include "jg_common.h"

#include "jg_conv_dev.h"

namespace {

constexpr int RB_C = 32;
constexpr int RB_CC = RB_C / 16;          // 16-channel chunks
constexpr int RB_PL = RB_CC * 4;          // (chunk, plane, half) item rows per position: 8
constexpr int RB_NB = 4;                  // 32-position blocks of the intermediate per tile: one per wave

__device__ __forceinline__ void rb_swap32(unsigned &lo_half_keeps, unsigned &hi_half_keeps) {
  const auto r = __builtin_amdgcn_permlane32_swap(lo_half_keeps, hi_half_keeps, false, false);
  lo_half_keeps = r[0];
  hi_half_keeps = r[1];
}

template <int RB_K>
__global__ __launch_bounds__(256, 2) void resblock32_kernel(JgResBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, hh = lane >> 5;
  const int d = a.dil, halo = (RB_K - 1) * d, pad = halo / 2;
  constexpr int RH = 32 * RB_NB;                   // intermediate positions per tile: one block of 32 per wave
  const int RX = RH + halo;
  const int x_items = RB_PL * RX, x_slot = (x_items + 63) & ~63;       // (a DMA call moves 64 items: whole calls per buffer)
  uint4 *Xbuf = lds;                               // [2 buffers][8][RX] (+ slack up to x_slot)
  uint4 *Himg = lds + 2 * x_slot;                  // [8][RH]
  float *epiL = reinterpret_cast<float *>(Himg + RB_PL * RH);      // [2 convs][scale | shift][32]
  if (tid < 4 * RB_C) epiL[tid] = a.epi[tid];
  // both convs' weight fragments: registers for the life of the workgroup
  uint4 wf[2][RB_K][RB_CC][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < RB_K; ++t)
#pragma unroll
      for (int cc = 0; cc < RB_CC; ++cc)
#pragma unroll
        for (int p = 0; p < 2; ++p) wf[c][t][cc][p] = a.wfrag[((((size_t)c * RB_K + t) * RB_CC + cc) * 2 + p) * 64 + lane];
  const unsigned ldsX = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void *)lds);
  const int n_units = a.rows * a.tiles_per_row;
  const float inv_rx = 1.0f / (float)RX;
  float vmax = 0.f;
  bool vnan = false;
  const int L_st = a.psplit ? ((a.L + 1) >> 1) : a.L;

  // Per tile a thread needs four mask bytes (the zero-fill flag of its input row; m1 of its intermediate position; m0 / m2
  // of its output position).  They are loaded ONE TILE AHEAD, in front of that tile's DMA: loads return in order, so a
  // mask byte requested behind a DMA would make its first use wait for the whole image.  Packed into one register.
  auto load_masks = [&](int unit) -> unsigned {
    const int row = unit / a.tiles_per_row, tile = unit - row * a.tiles_per_row;
    const int p0 = tile * a.tile_out, hp0 = p0 - pad, xp0 = hp0 - pad;
    const uint8_t *m0r = a.m0 != nullptr ? a.m0 + (size_t)row * a.L : nullptr;
    const uint8_t *m1r = a.m1 != nullptr ? a.m1 + (size_t)row * a.L : nullptr;
    const uint8_t *m2r = (a.m2 != nullptr && a.psplit) ? a.m2 + (size_t)row * a.L : nullptr;
    auto get = [&](const uint8_t *mr, int p, unsigned outside) -> unsigned {
      return (p >= 0 && p < a.L) ? (mr != nullptr ? (unsigned)(mr[p] != 0) : 1u) : outside;
    };
    unsigned m = 0;
    if (tid < RX) m |= (get(m0r, xp0 + tid, 0u) ^ 1u);            // bit 0: input row `tid` of the image is zero (outside the row, or masked)
    m |= get(m1r, hp0 + 32 * wid + i, 0u) << 1;                     // bit 1: conv2 reads h * m1; outside the row: SAME padding zeros
    m |= get(m0r, p0 + 32 * wid + i, 1u) << 2;                      // bit 2 clear: the image holds conv1's masked zero, not the shortcut value
    m |= get(m2r, p0 + 32 * wid + i, 1u) << 3;                      // bit 3: output mask (phase-split store only)
    return m;
  };
  // the input image of a tile: DMA, 64 items per wave and call; positions outside the row are fetched from a clamped
  // address and zeroed like masked ones once the image has landed
  auto issue_x = [&](int unit, int buf) {
    const int row = unit / a.tiles_per_row, tile = unit - row * a.tiles_per_row;
    const int xp0 = tile * a.tile_out - halo;
    for (int q0 = wid * 64; q0 < x_items; q0 += 256) {
      const int q = q0 + lane;
      int cph, j;
      udivmod24(min(q, x_items - 1), RX, inv_rx, cph, j);
      const int pos = min(max(xp0 + j, 0), a.L - 1);
      const unsigned voff = (unsigned)((((size_t)row * RB_PL + cph) * a.L + pos) * 16);
      // (lanes past the image's end fetch its last item again into the slack behind the image)
      glds16_nt(a.xh, voff, __builtin_amdgcn_readfirstlane(ldsX + (unsigned)(buf * x_slot + q0) * 16u));
    }
  };
  int unit = blockIdx.x;
  if (unit >= n_units) return;
  unsigned nxt = load_masks(unit);
  issue_x(unit, 0);
  for (int n = 0; unit < n_units; unit += gridDim.x, ++n) {
    const int row = unit / a.tiles_per_row, tile = unit - row * a.tiles_per_row;
    const int p0 = tile * a.tile_out;              // first output position of the tile
    uint4 *Ximg = Xbuf + (n & 1) * x_slot;
    wait_vm<0>();                                  // this wave's share of the tile's image, its mask bytes (and the previous tile's stores)
    __syncthreads();                               // ... and every other wave's share: a row is zeroed across all their items
    const unsigned mk = nxt;
    if (mk & 1u) {
#pragma unroll
      for (int c = 0; c < RB_PL; ++c) Ximg[c * RX + tid] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();       // image complete; every wave is done with the previous tile (intermediate image, other buffer)
    if (unit + (int)gridDim.x < n_units) {         // the next tile's bytes, then its image: lands under this tile's work
      nxt = load_masks(unit + gridDim.x);
      issue_x(unit + gridDim.x, (n + 1) & 1);
    }

    // ---- conv1 -> intermediate image: this wave's block of 32 positions ----------------------------------------
    {
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      const uint4 *X = Ximg + hh * RX + 32 * wid + i;
      // operands one step ahead of the matrix cores: a step's two LDS reads are requested before the previous step's three
      // (dependent) MFMAs are issued and land in their shadow
      uint4 vh = X[0], vl = X[2 * RX];
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // (the first step's operands)
#pragma unroll
      for (int st = 0; st < RB_CC * RB_K; ++st) {
        const int cc = st / RB_K, t = st % RB_K;
        uint4 nh = vh, nl = vl;
        if (st + 1 < RB_CC * RB_K) {
          const int c2 = (st + 1) / RB_K, t2 = (st + 1) % RB_K;
          nh = X[(c2 * 4 + 0) * RX + t2 * d];
          nl = X[(c2 * 4 + 2) * RX + t2 * d];
        }
        const half8 wh = *reinterpret_cast<const half8 *>(&wf[0][t][cc][0]);
        const half8 wl = *reinterpret_cast<const half8 *>(&wf[0][t][cc][1]);
        const half8 xh = *reinterpret_cast<const half8 *>(&vh), xl = *reinterpret_cast<const half8 *>(&vl);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
        if (st + 1 < RB_CC * RB_K) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the next step's two LDS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                 // ... then this step's MFMAs
        vh = nh;
        vl = nl;
      }
      const float m1f = (mk & 2u) ? 1.f : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {                // channel groups 2j and 2j + 1 of the block
        unsigned ph[4], pl[4];                     // [2 groups][2 dwords]: hi / lo halfs of this lane's four channels
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * j + gg;
          const float4 sc = *reinterpret_cast<const float4 *>(epiL + 8 * g + 4 * hh);
          const float4 of = *reinterpret_cast<const float4 *>(epiL + RB_C + 8 * g + 4 * hh);
          f32x2 v01 = {fmaf(c[4 * g + 0], sc.x, of.x), fmaf(c[4 * g + 1], sc.y, of.y)};
          f32x2 v23 = {fmaf(c[4 * g + 2], sc.z, of.z), fmaf(c[4 * g + 3], sc.w, of.w)};
          v01 = fast_gelu2(v01) * f32x2{m1f, m1f};
          v23 = fast_gelu2(v23) * f32x2{m1f, m1f};
          typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
          const half2_t h01 = {(_Float16)v01.x, (_Float16)v01.y}, h23 = {(_Float16)v23.x, (_Float16)v23.y};
          ph[2 * gg] = *reinterpret_cast<const unsigned *>(&h01);
          ph[2 * gg + 1] = *reinterpret_cast<const unsigned *>(&h23);
          const half2_t l01 = {(_Float16)mix_rem<0>(v01.x, ph[2 * gg]), (_Float16)mix_rem<1>(v01.y, ph[2 * gg])};
          const half2_t l23 = {(_Float16)mix_rem<0>(v23.x, ph[2 * gg + 1]), (_Float16)mix_rem<1>(v23.y, ph[2 * gg + 1])};
          pl[2 * gg] = *reinterpret_cast<const unsigned *>(&l01);
          pl[2 * gg + 1] = *reinterpret_cast<const unsigned *>(&l23);
        }
        rb_swap32(ph[0], ph[2]); rb_swap32(ph[1], ph[3]);          // -> the whole 8-channel item of group 2j + hh
        rb_swap32(pl[0], pl[2]); rb_swap32(pl[1], pl[3]);
        const int G = 2 * j + hh;
        Himg[((G >> 1) * 4 + 0 + (G & 1)) * RH + 32 * wid + i] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        Himg[((G >> 1) * 4 + 2 + (G & 1)) * RH + 32 * wid + i] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      }
    }
    __syncthreads();

    // ---- conv2 + shortcut + GELU -> HBM -------------------------------------------------------------------------
    if (32 * wid < a.tile_out) {
      f32x16 c;
#pragma unroll
      for (int r = 0; r < 16; ++r) c[r] = 0.f;
      const int o = 32 * wid + i, p = p0 + o;                      // output index inside the tile, position in the row
      const bool live = o < a.tile_out && p < a.L;
      // operand rows of dead lanes (past the tile's outputs) are clamped into the image: their results are dropped
      const uint4 *H = Himg + hh * RH;
      uint4 vh = H[min(o, RH - 1)], vl = H[2 * RH + min(o, RH - 1)];
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int st = 0; st < RB_CC * RB_K; ++st) {
        const int cc = st / RB_K, t = st % RB_K;
        uint4 nh = vh, nl = vl;
        if (st + 1 < RB_CC * RB_K) {
          const int c2 = (st + 1) / RB_K, t2 = (st + 1) % RB_K;
          const int r_ = min(o + t2 * d, RH - 1);
          nh = H[(c2 * 4 + 0) * RH + r_];
          nl = H[(c2 * 4 + 2) * RH + r_];
        }
        const half8 wh = *reinterpret_cast<const half8 *>(&wf[1][t][cc][0]);
        const half8 wl = *reinterpret_cast<const half8 *>(&wf[1][t][cc][1]);
        const half8 xh = *reinterpret_cast<const half8 *>(&vh), xl = *reinterpret_cast<const half8 *>(&vl);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, c, 0, 0, 0);
        if (st + 1 < RB_CC * RB_K) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        vh = nh;
        vl = nl;
      }
      const bool from_img = !live || (mk & 4u) != 0u;
      const float m2f = (mk & 8u) ? 1.f : 0.f;
      // the shortcut: this lane's four channels of every 8-channel group at the output position - out of the input image
      // (row o + halo), or from HBM where the image holds the conv's masked zero instead of the value
      const char *simg = reinterpret_cast<const char *>(Ximg + min(o + halo, RX - 1)) + 8 * hh;
      const char *sgl = reinterpret_cast<const char *>(a.xh) + 8 * hh;
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      // byte offset of this lane's first output item (the launcher bounds the tensor to 32-bit byte offsets); the lo
      // plane sits 2 L items behind the hi plane, the next chunk 4 L
      const unsigned out_step = (unsigned)(4 * L_st) * 16u;
      const unsigned out_item = a.psplit ? (unsigned)(((row * 2 * RB_CC + (p & 1) * RB_CC) * 4 + hh) * L_st + (p >> 1))
                                         : (unsigned)((row * RB_CC * 4 + hh) * a.L + p);
      const unsigned out_off = out_item * 16u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = 2 * j + gg;
          uint2 sh, sl;
          if (from_img) {
            sh = *reinterpret_cast<const uint2 *>(simg + (size_t)(((g >> 1) * 4 + 0 + (g & 1)) * RX) * 16);
            sl = *reinterpret_cast<const uint2 *>(simg + (size_t)(((g >> 1) * 4 + 2 + (g & 1)) * RX) * 16);
          } else {
            sh = *reinterpret_cast<const uint2 *>(sgl + (((size_t)row * RB_PL + (g >> 1) * 4 + 0 + (g & 1)) * a.L + p) * 16);
            sl = *reinterpret_cast<const uint2 *>(sgl + (((size_t)row * RB_PL + (g >> 1) * 4 + 2 + (g & 1)) * a.L + p) * 16);
          }
          const float4 sc = *reinterpret_cast<const float4 *>(epiL + 2 * RB_C + 8 * g + 4 * hh);
          const float4 of = *reinterpret_cast<const float4 *>(epiL + 3 * RB_C + 8 * g + 4 * hh);
          f32x2 v01 = {fmaf(c[4 * g + 0], sc.x, of.x), fmaf(c[4 * g + 1], sc.y, of.y)};
          f32x2 v23 = {fmaf(c[4 * g + 2], sc.z, of.z), fmaf(c[4 * g + 3], sc.w, of.w)};
          v01 += f32x2{mix_sum<0>(sh.x, sl.x), mix_sum<1>(sh.x, sl.x)};
          v23 += f32x2{mix_sum<0>(sh.y, sl.y), mix_sum<1>(sh.y, sl.y)};
          v01 = fast_gelu2(v01) * f32x2{m2f, m2f};
          v23 = fast_gelu2(v23) * f32x2{m2f, m2f};
          vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v01.x), fabsf(v01.y))), fmaxf(fabsf(v23.x), fabsf(v23.y)));
          vnan = vnan || __builtin_isunordered(v01.x, v01.y) || __builtin_isunordered(v23.x, v23.y);
          typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
          const half2_t h01 = {(_Float16)v01.x, (_Float16)v01.y}, h23 = {(_Float16)v23.x, (_Float16)v23.y};
          ph[2 * gg] = *reinterpret_cast<const unsigned *>(&h01);
          ph[2 * gg + 1] = *reinterpret_cast<const unsigned *>(&h23);
          const half2_t l01 = {(_Float16)mix_rem<0>(v01.x, ph[2 * gg]), (_Float16)mix_rem<1>(v01.y, ph[2 * gg])};
          const half2_t l23 = {(_Float16)mix_rem<0>(v23.x, ph[2 * gg + 1]), (_Float16)mix_rem<1>(v23.y, ph[2 * gg + 1])};
          pl[2 * gg] = *reinterpret_cast<const unsigned *>(&l01);
          pl[2 * gg + 1] = *reinterpret_cast<const unsigned *>(&l23);
        }
        rb_swap32(ph[0], ph[2]); rb_swap32(ph[1], ph[3]);
        rb_swap32(pl[0], pl[2]); rb_swap32(pl[1], pl[3]);
        if (live) {
          // group 2j + hh = item row (chunk j, half hh): 4 L items further per j
          const unsigned off = out_off + (unsigned)j * out_step;
          const u32x4 vhi = {ph[0], ph[1], ph[2], ph[3]}, vlo = {pl[0], pl[1], pl[2], pl[3]};
          char *yb = reinterpret_cast<char *>(a.y);
          __builtin_nontemporal_store(vhi, reinterpret_cast<u32x4 *>(yb + (size_t)off));
          __builtin_nontemporal_store(vlo, reinterpret_cast<u32x4 *>(yb + (size_t)(off + out_step / 2)));
          if (a.psplit && (a.L & 1) && p == a.L - 1) {             // the odd phase of an odd row is one position short: zero
            const unsigned offz = off + (unsigned)(RB_CC * 4 * L_st) * 16u;
            const u32x4 z = {0u, 0u, 0u, 0u};
            __builtin_nontemporal_store(z, reinterpret_cast<u32x4 *>(yb + (size_t)offz));
            __builtin_nontemporal_store(z, reinterpret_cast<u32x4 *>(yb + (size_t)(offz + out_step / 2)));
          }
        }
      }
    }
  }
  if ((!(vmax <= 65000.0f) || vnan) && a.overflow != nullptr) atomicOr(a.overflow, 1);
}

}  // namespace

// a tile computes RB_NB x 32 intermediate positions; its outputs are the 128 - 4 d positions whose taps stay inside it
void jg_resblock_tiling(int L, int k, int dil, int *nb, int *tile_out, int *tiles) {
  *nb = RB_NB;
  *tile_out = 32 * RB_NB - (k - 1) * dil;
  *tiles = (L + *tile_out - 1) / *tile_out;
}

bool jg_resblock_supports(int c, int k, int dil) { return c == RB_C && (k == 5 || k == 3) && dil >= 1 && (k - 1) * dil <= 32; }

int jg_launch_resblock(jg_engine *e, const JgResBlockArgs &a, hipStream_t s) {
  JG_REQUIRE(a.xh != nullptr && a.y != nullptr && a.wfrag != nullptr && a.epi != nullptr && a.nb == RB_NB &&
                 (a.k == 5 || a.k == 3) && a.tile_out == 32 * a.nb - (a.k - 1) * a.dil && a.tile_out >= 1 &&
                 a.tiles_per_row * a.tile_out >= a.L,
             JG_ERR_INVALID, "resblock: bad geometry (nb %d, tile_out %d, tiles %d, L %d, dilation %d)", a.nb, a.tile_out,
             a.tiles_per_row, a.L, a.dil);
  JG_REQUIRE((double)a.rows * RB_PL * (a.L + 1) * 16.0 < 4.2e9, JG_ERR_UNSUPPORTED,
             "resblock: activation tensor of %d rows exceeds the 32-bit byte offsets of the DMA and the stores", a.rows);
  if (a.rows == 0 || a.L <= 0) return JG_OK;
  const int RH = 32 * a.nb, RX = RH + (a.k - 1) * a.dil;
  const int x_slot = (RB_PL * RX + 63) & ~63;        // two input images (double buffer), whole 64-item DMA calls each
  const size_t smem = (size_t)(2 * x_slot + RB_PL * RH) * 16 + 4 * RB_C * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(resblock32_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               128 * 1024));
    JG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(resblock32_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               128 * 1024));
    attr_set = true;
  }
  const long units = (long)a.rows * a.tiles_per_row;
  const int grid = (int)std::min<long>(units, 2L * e->n_cu);
  if (a.k == 5) hipLaunchKernelGGL(resblock32_kernel<5>, dim3((unsigned)grid), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(resblock32_kernel<3>, dim3((unsigned)grid), dim3(256), smem, s, a);
  JG_HIP(hipGetLastError());
  return JG_OK;
}
