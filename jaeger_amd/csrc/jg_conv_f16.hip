// Host entry of the split-f16 convolution: argument checks and dispatch to the instantiation sets
// compiled in jg_conv_f16_k5.hip / _k79.hip / _flat.hip / _lut.hip (all from jg_conv_f16_impl.h).
#include <stdlib.h>

#include "jg_common.h"

int jg_conv_f16_part_k5(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_k79(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_flat(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_lut(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_n64(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_n32(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_g128(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_x8(jg_engine *e, const ConvHArgs &a, hipStream_t s);     // k = 7: 64 / 32-channel tiles, general tile
int jg_conv_f16_part_x9(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_x10(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_x11(jg_engine *e, const ConvHArgs &a, hipStream_t s);    // k = 9
int jg_conv_f16_part_x12(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_part_x13(jg_engine *e, const ConvHArgs &a, hipStream_t s);
#ifdef JG_EXPERIMENT                                                           /* the producer / consumer kernel: experiment build only */
bool jg_conv_pc_supports(const ConvHArgs &a);                                  // jg_conv_pc.hip
int jg_conv_pc_launch(jg_engine *e, const ConvHArgs &a, hipStream_t s);
#endif

namespace {
constexpr int HM = 256, HN = 128, HT = 256, NT = 1, A_ITERS = 5, W_ITEMS = 2 * 2 * HN, LUT_RS = 68;   // as in jg_conv_f16_impl.h
}

int jg_conv_lut_lds_bytes(int k, int vocab) {
  return k * (vocab + 1) * LUT_RS * 4 + JG_EPI_ROWS * 2 * HN * 4 + 8 * 256;     // table half, epilogue rows, 256 B of row indices per wave
}

// first-layer table variant: LDS holds k*(vocab+1) rows, a wave stages <= 256 positions
bool jg_conv_lut_supports(int k, int dil, int vocab) {
  return k >= 1 && (k - 1) * dil <= 128 && vocab <= 254 && jg_conv_lut_lds_bytes(k, vocab) <= 160 * 1024;
}

int jg_conv_f16_lds_bytes(int k, int dil) {
  const int rows_a = HM + (k - 1) * dil;
  return (2 * NT * 4 * rows_a + k * W_ITEMS) * 16 + JG_EPI_ROWS * 2 * HN * 4;
}

// taps / dilation the kernel's tiling can hold: instantiated tap counts, activation slice
// <= A_ITERS pieces per thread, LDS <= 160 KiB
bool jg_conv_f16_supports(int k, int dil) {
  return (k == 5 || k == 7 || k == 9) && NT * 4 * (HM + (k - 1) * dil) <= A_ITERS * HT &&
         jg_conv_f16_lds_bytes(k, dil) <= 160 * 1024;
}

int jg_conv_f16_tile_m(void) { return HM; }

// stage patterns with a compiled epilogue (keep in step with the instantiation sets in jg_conv_f16_impl.h);
// a first-layer conv may end up on the table variant, which carries a subset
bool jg_conv_f16_has_pattern(unsigned ep, bool first_layer) {
  const unsigned lut[] = {0u, JG_EP_NMD1, JG_EP_ACT1, JG_EP_NORM1_AFF | JG_EP_ACT1, JG_EP_NORM1_DYT | JG_EP_ACT1,
                          JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1, JG_EP_NMD1 | JG_EP_NORM1_DYT | JG_EP_ACT1,
                          JG_EP_ACT1 | JG_EP_NORM2_AFF, JG_EP_RUNTIME};
  const unsigned all[] = {0u, JG_EP_NMD1, JG_EP_ACT1, JG_EP_NORM1_AFF | JG_EP_ACT1, JG_EP_NORM1_DYT | JG_EP_ACT1,
                          JG_EP_ADD | JG_EP_ACT1, JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1,
                          JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                          JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_DYT | JG_EP_ACT2,
                          JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1, JG_EP_NMD1 | JG_EP_NORM1_DYT | JG_EP_ACT1,
                          JG_EP_ACT1 | JG_EP_NORM2_AFF, JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                          JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                          JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_DYT | JG_EP_ACT2,
                          JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1,
                          JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                          JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2, JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2,
                          JG_EP_NORM1_DYT, JG_EP_RUNTIME};
  if (first_layer) {
    for (unsigned p : lut) if (p == ep) return true;
    return false;      // (a first layer the table variant cannot take also needs one of `all`: same subset)
  }
  for (unsigned p : all) if (p == ep) return true;
  return false;
}

// stage patterns compiled for the window-packed tiling (jg_conv_f16_part_flat)
bool jg_conv_f16_has_flat_pattern(unsigned ep) {
  const unsigned flat[] = {JG_EP_ACT1, JG_EP_NORM1_AFF | JG_EP_ACT1, JG_EP_NORM1_DYT | JG_EP_ACT1, JG_EP_ADD | JG_EP_ACT1,
                           JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1,
                           JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                           JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_DYT | JG_EP_ACT2,
                           JG_EP_ACT1 | JG_EP_NORM2_AFF, JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                           JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                           JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_DYT | JG_EP_ACT2,
                           JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2, JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2, JG_EP_RUNTIME};
  for (unsigned p : flat) if (p == ep) return true;
  return false;
}

// stage patterns compiled for the run-time-geometry variants (narrow convs of 64 / 32 output channels, the general
// 128-wide tile: jg_conv_f16_part_n64 / _n32 / _g128, k = 5, both tilings)
bool jg_conv_f16_has_narrow_pattern(unsigned ep) {
  const unsigned nar[] = {0u, JG_EP_ACT1, JG_EP_NORM1_AFF | JG_EP_ACT1, JG_EP_ADD | JG_EP_ACT1,
                          JG_EP_ADD | JG_EP_ACT1 | JG_EP_NORM2_AFF | JG_EP_ACT2, JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ACT1,
                          JG_EP_NMD1 | JG_EP_NORM1_AFF | JG_EP_ADD | JG_EP_ACT1,
                          JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2 | JG_EP_NORM2_AFF | JG_EP_ACT2,
                          JG_EP_NORM1_DYT | JG_EP_ACT1, JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1, JG_EP_NMD1,
                          JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2, JG_EP_NORM1_DYT, JG_EP_NORM1_DYT | JG_EP_ADD | JG_EP_ACT1 | JG_EP_NMD2,
                          JG_EP_RUNTIME};
  for (unsigned p : nar) if (p == ep) return true;
  return false;
}

// JG_DBG ablation mask: read at every launch, so that an experiment can warm up on real data and then
// switch (bench.py --timed-dbg)
static int jg_dbg_env(void) {
  const char *ev = jg_exp_env("JG_DBG");
  return ev ? atoi(ev) : 0;
}

int jg_launch_conv_f16(jg_engine *e, const ConvHArgs &a, hipStream_t s) {
  JG_REQUIRE(a.cout % 16 == 0 && a.cout_pad == a.cout && a.ch0 % HN == 0 && a.ch0 < a.cout &&
                 (a.cw == HN || ((a.cw == 64 || a.cw == 32) && a.cout <= a.cw && a.ch0 == 0)) &&
                 (a.ostride == 1 || a.ostride == 2) && a.L_res == (a.ostride == 2 ? 2 * a.L_out - 1 : a.L_out),
             JG_ERR_UNSUPPORTED, "conv_f16x3: cout=%d ch0=%d tile width %d out-stride %d (L_res %d, L_out %d) outside the kernel",
             a.cout, a.ch0, a.cw, a.ostride, a.L_res, a.L_out);
  JG_REQUIRE(a.tap_lo >= 0 && a.tap_lo <= a.tap_hi && a.tap_hi < a.k && ((a.tap_lo == 0 && a.tap_hi == a.k - 1) || (a.k == 5 && a.lut == nullptr)),
             JG_ERR_UNSUPPORTED, "conv_f16x3: a tap range (%d..%d of %d) is only built into the 5-tap kernel", a.tap_lo, a.tap_hi, a.k);
  if (a.lut != nullptr) {
    JG_REQUIRE(a.cw == HN && a.ostride == 1 && a.cout <= HN && (!a.lut_one_half || a.cout <= 64), JG_ERR_UNSUPPORTED,
               "conv lut: cout=%d outside the table variant", a.cout);
    JG_REQUIRE(a.ids != nullptr && jg_conv_lut_supports(a.k, a.dil, a.lut_vocab), JG_ERR_UNSUPPORTED,
               "conv lut: k=%d dilation=%d vocab=%d outside the table variant's limits", a.k, a.dil, a.lut_vocab);
    if (a.rows == 0 || a.L_out <= 0) return JG_OK;
    const_cast<ConvHArgs &>(a).dbg = jg_dbg_env();
    return jg_conv_f16_part_lut(e, a, s);
  }
  JG_REQUIRE(jg_conv_f16_supports(a.k, a.dil), JG_ERR_UNSUPPORTED,
             "conv_f16x3: k=%d dilation=%d outside the kernel's tiling", a.k, a.dil);
  // DMA offsets are 32-bit: the activation tensor of one launch must stay below 4 GiB
  if (a.cc_row == 0) const_cast<ConvHArgs &>(a).cc_row = a.cc_in;
  JG_REQUIRE(a.cc_row >= a.cc_in && (!a.psplit || (a.out_f16s && a.pool_out == nullptr && a.ostride == 1)), JG_ERR_INVALID,
             "conv_f16x3: phase-split geometry (cc_row %d, cc_in %d, psplit %d)", a.cc_row, a.cc_in, a.psplit);
  JG_REQUIRE((double)a.rows * a.cc_row * 4.0 * a.L_in * 16.0 < 4.0e9, JG_ERR_UNSUPPORTED,
             "conv_f16x3: activation tensor of %d rows exceeds the 32-bit DMA offset range", a.rows);
  if (a.rows == 0 || a.L_out <= 0) return JG_OK;
  const_cast<ConvHArgs &>(a).dbg = jg_dbg_env();
  JG_REQUIRE(!a.lut_one_half, JG_ERR_INVALID, "conv_f16x3: lut_one_half without a table");
  if (a.cw != HN) {
    if (a.k == 5) return a.cw == 64 ? jg_conv_f16_part_n64(e, a, s) : jg_conv_f16_part_n32(e, a, s);
    JG_REQUIRE(!a.flat, JG_ERR_UNSUPPORTED, "conv_f16x3: window-packed tiling is only built for k = 5");
    if (a.k == 7) return a.cw == 64 ? jg_conv_f16_part_x8(e, a, s) : jg_conv_f16_part_x9(e, a, s);
    return a.cw == 64 ? jg_conv_f16_part_x11(e, a, s) : jg_conv_f16_part_x12(e, a, s);
  }
  if (a.cout != HN || a.ostride != 1 || a.tap_lo != 0 || a.tap_hi != a.k - 1 || a.psplit) {     // (ch0 != 0 implies cout > 128; a
                                                                  // phase-split store is built into the run-time-geometry tiles only)
    if (a.k == 5) return jg_conv_f16_part_g128(e, a, s);
    JG_REQUIRE(!a.flat, JG_ERR_UNSUPPORTED, "conv_f16x3: window-packed tiling is only built for k = 5");
    return a.k == 7 ? jg_conv_f16_part_x10(e, a, s) : jg_conv_f16_part_x13(e, a, s);
  }
#ifdef JG_EXPERIMENT
  // the residual stacks' 128 -> 128 five-tap convs on the producer / consumer kernel (math waves + DMA / epilogue helper
  // waves; round 3's committed negative, profiles/r3_pc_experiments.md): experiment build only, JG_OPT_CONV_PC = 1;
  // same results bit for bit as the two-workgroup kernel below
  static const bool no_pc = jg_exp_env("JG_NO_PC") != nullptr;
  if (e->conv_pc == 1 && !no_pc && a.dbg == 0 && jg_conv_pc_supports(a)) return jg_conv_pc_launch(e, a, s);
#endif
  if (a.flat) {
    JG_REQUIRE(a.k == 5 && a.ostride == 1, JG_ERR_UNSUPPORTED, "conv_f16x3: window-packed tiling is only built for k = 5, stride 1");
    return jg_conv_f16_part_flat(e, a, s);
  }
  if (a.k == 5) return jg_conv_f16_part_k5(e, a, s);
  return jg_conv_f16_part_k79(e, a, s);       // 7 or 9: jg_conv_f16_supports() above
}
