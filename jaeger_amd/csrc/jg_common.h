// Internal declarations shared by the HIP translation units of libjaeger_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#include "jaeger_hip.h"

void jg_set_error(const char *fmt, ...);

// Experiment switches (JG_DBG ablation mask, JG_NO_LUT, JG_NO_FLAT, JG_NO_POOL_FUSE, JG_ONE_WG) exist only in the
// `make exp` build (libjaeger_hip_exp.so, -DJG_EXPERIMENT): the shipped library never reads them, so that a stray
// variable in a user's environment cannot change results.
#include <stdlib.h>
static inline const char *jg_exp_env(const char *name) {
#ifdef JG_EXPERIMENT
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

#define JG_HIP(call)                                                                 \
  do {                                                                               \
    hipError_t err_ = (call);                                                        \
    if (err_ != hipSuccess) {                                                        \
      jg_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(err_)); \
      return JG_ERR_HIP;                                                             \
    }                                                                                \
  } while (0)

#define JG_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      jg_set_error(__VA_ARGS__);     \
      return (code);                 \
    }                                \
  } while (0)

// ---- device-side argument blocks (passed by value as kernargs) -------------
struct StageArg {
  int kind;
  int arg;
  const float *p0, *p1, *p2, *p3;
  float f0;
  int pad_;
};

struct ConvArgs {
  const float *x;          // (rows, L_in, cin) or null when ids != null
  const uint8_t *ids;      // (rows, L_in) embedding-gather source
  const float *emb;        // (vocab, cin)
  const uint8_t *mask_in;  // (rows, L_in) or null
  const uint8_t *mask_out; // (rows, L_out) or null
  const float *w;          // (k, cin_pad2, cout_pad) as handed over in the weight blob
  const float *w8;         // re-packed for the kernel: [k][cin_pad/8][cout_pad][8], cin_pad = cin rounded up to 8
  float *y;                // (rows, L_out, cout)
  int rows, L_in, L_out;
  int cin, cin_pad, cout, cout_pad;
  int k, stride, dil, pad_left;
  int tiles_m;
  int cchunk;        // input channels staged in LDS per pass (set by the launcher)
  int mask_from_ids; // conv input multiplied by (ids != 0)
  int n_stages;
  StageArg st[JG_MAX_STAGES];
};

// compact epilogue of the split-f16 conv: bias / batch-norm chains are folded into one
// per-channel affine on the host (f64), parameters live in a (stage, 2, 128) f32 table
// compiled epilogue pattern of the split-f16 conv: affine [nmd] [norm1] [add] [gelu] [nmd] [norm2] [gelu]
#define JG_EP_NMD1 0x001u
#define JG_EP_NORM1_AFF 0x002u
#define JG_EP_NORM1_DYT 0x004u
#define JG_EP_ADD 0x008u
#define JG_EP_ACT1 0x010u
#define JG_EP_NMD2 0x020u
#define JG_EP_NORM2_AFF 0x040u
#define JG_EP_NORM2_DYT 0x080u
#define JG_EP_ACT2 0x100u
#define JG_EP_GENERIC 0xffffu
#define JG_EP_RUNTIME 0xfffeu  /* the same canonical stage order with the optional stages behind wave-uniform run-time flags
                                  (ConvHArgs.ep_rt): every canonical stage list without an instantiation of its own */
#define JG_EPI_ROWS 4   /* parameter rows (affine / dyt stages) of a split-f16 epilogue */
enum { JG_HST_AFFINE = 1, JG_HST_DYT = 2, JG_HST_ADD = 3, JG_HST_ACT = 4, JG_HST_NMD = 5, JG_HST_MASKMUL = 6 };
struct HStageArg {
  int kind;
  int arg;
  float f0;
  int pad_;
};

// split-f16 conv (jg_conv_f16_impl.h); uint4 = one 16-byte item of 8 halfs
struct ConvHArgs {
  const uint4 *xh;         // F16S input [rows][cc_in][4][L_in] or null when ids != null
  const uint8_t *ids;      // (rows, L_in) embedding-gather source
  const uint4 *embh;       // pre-split table [vocab][cc_in][4]
  const uint8_t *mask_in;  // (rows, L_in) or null
  const uint8_t *mask_out; // (rows, L_out) or null
  const uint4 *wh;         // [2 planes][k][cc_in*2][cout_pad]
  void *y;                 // F16S [rows][cout_pad/16][4][L_out] items, or f32 (rows, L_out, cout)
  const uint4 *addh;       // residual shortcut in F16S (same geometry as y) or null
  float *nmd_out;          // NMD partial sums [rows][tiles_m * 2][cout] (one row per wave strip) or null
  float *nmd_out2;         // ... of a second NMD tap in the same conv (JG_EP_RUNTIME only: a tap before norm1 AND one behind act1)
  unsigned ep_rt;          // JG_EP_RUNTIME: the JG_EP_* bits of the stage list
  float *pool_out;         // fused masked max-pool partials, same geometry; when set the output is not stored
  int *overflow;           // set to 1 when an output leaves the f16 range
  int rows, L_in, L_out;
  int cc_in, cout, cout_pad;   // cout_pad: output channels rounded up to 16 (F16S chunks of y / addh)
  int cc_row;                  // 16-channel chunks per input ROW in memory (0 = cc_in): a 1x1 stride-2 bypass reads only the
                               // even-phase half of a phase-split tensor
  int psplit;                  // 1: the output is stored PHASE-SPLIT and mask-multiplied for the stride-2 convs that read it:
                               // position p of channel chunk cc goes to chunk (p & 1) * cout_pad/16 + cc, position p >> 1 of a
                               // tensor [rows][2 * cout_pad/16][4][(L_out + 1) / 2] - a stride-2 conv over L positions then is a
                               // stride-1 conv over (L + 1) / 2 positions of twice the channels, no output computed and dropped
  int ch0;                     // first output channel of this launch (convs wider than 128 run one launch per 128)
  int cw;                      // channel width of a workgroup tile: 128, 64 or 32 (narrow convs: waves split positions only)
  int ostride;                 // 1, or 2: the conv is evaluated at stride 1 over L_res positions and even ones are kept
  int L_res;                   // positions the tiles resolve (= L_out when ostride == 1, 2*L_out - 1 otherwise)
  int lut_one_half;            // table variant: only the first 64-channel half exists (cout <= 64)
  int tap_lo, tap_hi;          // taps tap_lo .. tap_hi of the kernel's five carry weights (0, 4 normally; a 1x1 / 3-tap
                               // conv rides the 5-tap pipeline with 2, 2 / 1, 3: the matrix-core work of the other taps
                               // is skipped, their zero weight slices still flow through the DMA ring)
  int k, dil, pad_left, tiles_m;
  // window-packed tiling (see jg_conv_f16_impl.h): frames of a window on one axis, row pitch flat_p, window
  // pitch flat_wp (multiple of 128), flat_tiles tiles of 256; 0 = every row tiled on its own
  int flat, flat_p, flat_wp, flat_frames, flat_tiles;
  float flat_inv_p, flat_inv_wp;
  int mask_from_ids, out_f16s;
  int act_kind;            // jg_act of the compiled patterns' activation stages (one kind per op)
  int dbg;                 // ablation switches (JG_DBG env, timing experiments only)
  unsigned ep;             // JG_EP_* pattern of the stage list (JG_EP_GENERIC: interpret hst[])
  float alpha1, alpha2;    // DyT alphas of norm1 / norm2
  int dytmask1, dytmask2;
  int n_hst;
  int n_epi_rows;
  const float *epi;        // [n_epi_rows][2][128] epilogue parameters (HStageArg.pad_ = row)
  const float *lut;        // first-layer table [2 halves][k][lut_vocab + 1][64] f32 or null (matrix-core path)
  int lut_vocab;
  HStageArg hst[JG_MAX_STAGES];
};

// fused narrow residual block (jg_resblock.hip): conv1 -> affine -> GELU -> conv2 -> affine -> + x -> GELU in one launch
struct JgResBlockArgs {
  const uint4 *xh;        // block input, F16S [rows][2][4][L]
  uint4 *y;               // block output, F16S of the same geometry - or phase-split and masked (psplit, see ConvHArgs)
  const uint8_t *m0;      // (rows, L) input mask of conv1 (x * m0), or null
  const uint8_t *m1;      // ... of conv2 (= conv1's output mask), or null
  const uint8_t *m2;      // conv2's output mask: only read for a phase-split store, or null
  const uint4 *wfrag;     // [2 convs][k taps][2 chunks][2 planes][64 lanes] MFMA A-operand fragments (hi / lo f16, pre-scaled)
  const float *epi;       // [2 convs][scale | shift][32] folded bias / batch-norm affines (incl. the weights' un-scale)
  int *overflow;
  int rows, L, k, dil;    // taps (5 or 3), dilation
  int nb, tile_out, tiles_per_row;   // jg_resblock_tiling
  int psplit;
};
bool jg_resblock_supports(int c, int k, int dil);
void jg_resblock_tiling(int L, int k, int dil, int *nb, int *tile_out, int *tiles);
int jg_launch_resblock(jg_engine *e, const JgResBlockArgs &a, hipStream_t s);

struct EltArgs {
  const float *x;
  float *y;
  const uint8_t *mask; // (rows*L) or null
  int64_t n_pos;       // rows * L
  int c;
  int n_stages;
  StageArg st[JG_MAX_STAGES];
};

struct ProfEvent {
  hipEvent_t a, b;
  double flops;
  int cls = 0;      // JG_PROF_*: which kernel family the launch belongs to
};

struct jg_engine {
  int dev = 0;
  hipStream_t stream = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  bool profile = false;
  std::vector<ProfEvent> pending; // conv launches awaiting readout
  std::vector<hipEvent_t> pool;   // recycled events
  double conv_ms = 0.0, conv_flops = 0.0;
  int64_t conv_launches = 0;
  double cls_ms[4] = {}, cls_flops[4] = {};     // the same split by kernel family (JG_PROF_*)
  int64_t cls_launches[4] = {};
  int n_cu = 256;
  int dust_on_copy = 1;           // JG_OPT_DUST_ON_COPY_STREAM: streamed spans are soft-masked on the copy stream (1) or in front of their encoder (0)
  int termini_exact = 0;          // JG_OPT_TERMINI_EXACT: every terminal-repeat alignment through the length / gap carrying kernel
  int termini_report_min = 0;     // JG_OPT_TERMINI_REPORT_MIN: alignments of fewer columns are reported as none (0 = every alignment exactly)
  int tab_lds_only = 0;           // JG_OPT_TABLE_NET_LDS: keep the table net on the LDS-table kernel
  int fuse_resblock = 1;          // JG_OPT_FUSE_RESBLOCK: narrow residual blocks as one launch (jg_resblock.hip)
  int conv_pc = 0;                // JG_OPT_CONV_PC: producer / consumer kernel for the 128-channel five-tap convs
  // streamed ingest of host-resident bases (jg_predict_windows): spans above `stream_bytes` go through two pinned
  // staging buffers and two device buffers on a copy stream, record group by record group
  int64_t stream_bytes = (int64_t)32 << 20;
  hipStream_t copy_stream = nullptr;
  void *pin[2] = {nullptr, nullptr};
  void *dbase[2] = {nullptr, nullptr};
  int64_t pin_cap = 0, dbase_cap = 0;
  hipEvent_t h2d_done[2] = {nullptr, nullptr};
  hipEvent_t enc_done[2] = {nullptr, nullptr};    // group g's encode has read device span g % 2
  hipEvent_t grp_done[2] = {nullptr, nullptr};    // group g's outputs are in pinned staging g % 2
  void *pin_io[2] = {nullptr, nullptr};           // pinned window tables / outputs / counts of the two groups in flight
  int64_t pin_io_cap = 0;
  std::atomic<int64_t> windows_done{0};           // JG_STAT_WINDOWS_DONE: rows [0, windows_done) of the running call are final
  int64_t streamed_groups = 0, streamed_bytes = 0, peak_dev_bases = 0;   // statistics of the last call (jg_engine_get_stat)
  // DUST on the device (jg_engine_set_dust): record offsets of the host base buffer the next jg_predict_windows /
  // jg_encode calls are given; the uploaded bases are soft-masked before they are encoded
  int64_t *d_rec_off = nullptr;
  int64_t rec_cap = 0, n_rec = 0, rec_end = 0;
  int dust_window = 0, dust_threshold = 0;
  unsigned long long *d_dust_cnt = nullptr;
};

struct ConvHPrep {          // per CONV op: split-f16 operands (built at model creation)
  uint4 *d_wh = nullptr;    // weights [n_half][2][k][cin16/8][128]
  int n_half = 1;           // launches per conv: one per 128 output channels
  int cw = 128;             // workgroup tile width (128, or 64 / 32 for narrow convs)
  bool as_k5 = false;       // a 1x1 or 3-tap conv carried by the 5-tap kernel (weights in the middle taps, the others skipped)
  std::string why_f32;      // (CONV ops that are not f16_ok) what keeps this conv on the exact-f32 kernel
  int64_t wh_half_items = 0;   // 16-byte items of one half's weight blob
  uint4 *d_embh = nullptr;  // embedding table [vocab][cin16/16][4] (conv on ids only)
  float acc_scale = 1.f;
  int cc_in = 0;
  bool out_f16s = false;
  float *d_epi = nullptr;   // compact epilogue parameter table
  float *d_w8 = nullptr;    // exact-f32 path: weights re-packed [k][cin8/8][cout_pad32][8]
  float *d_lut = nullptr;   // first-layer table E.W_t (conv on ids) and its epilogue table (no acc un-scale)
  float *d_epi_lut = nullptr;
  int n_hst = 0, n_epi_rows = 0;
  HStageArg hst[JG_MAX_STAGES] = {};
  int add_slot = -1, nmd_slot = -1, nmd_slot2 = -1;     // (nmd_slot2: the second NMD tap of one conv)
  unsigned ep_rt = 0;       // stage bits when ep == JG_EP_RUNTIME
  int pool_op = -1;         // index of the OP_POOL (masked max) fused into this conv's epilogue, or -1
  int act_kind = JG_ACT_GELU_TANH;   // activation of the op's ACT stages (the compiled patterns allow one kind per op)
  bool pool_f16s = false;   // (MAXPOOL1D ops) input and output are F16S tensors
  // stride-2 convs without dropped work (phase-split tensors, ConvHArgs::psplit):
  bool ps_store = false;    // (CONV ops) this conv's output is read by stride-2 convs only: stored phase-split and masked
  int ps_read = 0;          // (CONV ops) 1: a 5-tap stride-2 conv run as a 3-tap stride-1 conv over the two phases (2 x cin
                            // channels, re-arranged weights d_wh_ps); 2: a 1x1 stride-2 conv run at stride 1 on the even phase
  uint4 *d_wh_ps[2] = {nullptr, nullptr};   // ps_read == 1: weights for an even / odd input length (TF's SAME split differs)
  int64_t ps_half_items = 0;
  // a whole narrow residual block as one launch (jg_resblock.hip): conv1 is skipped, conv2's launch runs both
  int rb_first = -1;        // (conv2 of a fused block) index of the block's conv1 op
  int rb_second = -1;       // (conv1 of a fused block) index of the conv2 op whose launch computes this one too
  uint4 *d_rb_wfrag = nullptr;
  float *d_rb_epi = nullptr;
  bool f16_ok = false;      // (CONV ops) runs on the split-f16 kernel when the model is in split-f16 mode
  int n_cvt = 0;            // layout conversions queued in front of this op (any op kind): slot, direction
  int cvt_slot[3] = {-1, -1, -1};
  bool cvt_to_f32[3] = {false, false, false};
  unsigned ep = JG_EP_GENERIC;
  float alpha1 = 0.f, alpha2 = 0.f;
  int dytmask1 = 0, dytmask2 = 0;
};

struct JgSmallNet;
struct jg_model {
  jg_engine *e = nullptr;
  JgSmallNet *small = nullptr;    // fused small-window network (jg_small.hip) when the program matches that family
  std::vector<jg_op> ops;
  std::vector<ConvHPrep> hprep;   // parallel to ops
  int precision = 0;              // 0 = exact f32 MFMA, 1 = split-f16 (f16x3)
  bool f16_eligible = false;
  bool f16_mixed = false;         // some convs stay on the exact-f32 kernel inside the split-f16 program
  bool needs_cvt = false;         // the program holds F16S <-> f32 layout conversions (needs the scratch tensor)
  float *cvt_scratch = nullptr;
  int64_t cvt_cap = 0;
  std::string f16_reason;         // why the fast path is unavailable
  int *d_overflow = nullptr;
  float *d_w = nullptr;
  int64_t n_w = 0;
  int vocab = 0;
  // workspace (grown on demand)
  int64_t act_cap[JG_MAX_BUFS] = {}, msk_cap[JG_MAX_BUFS] = {}, nmd_cap[JG_MAX_BUFS] = {}, vec_cap[JG_MAX_VECS] = {};  // elements allocated
  float *act[JG_MAX_BUFS] = {};
  int64_t act_elems[JG_MAX_BUFS] = {}; // per window
  uint8_t *msk[JG_MAX_BUFS] = {};
  int64_t msk_elems[JG_MAX_BUFS] = {};
  float *vec[JG_MAX_VECS] = {};
  int vec_w[JG_MAX_VECS] = {};
  float *nmd_part[JG_MAX_BUFS] = {};
  int64_t nmd_part_elems[JG_MAX_BUFS] = {};
  uint8_t *d_ids = nullptr;
  int64_t d_ids_cap = 0;
  int32_t *d_counts = nullptr;
  int64_t d_counts_cap = 0;
  void *d_win = nullptr;
  int64_t d_win_cap = 0;
  void *d_bases_buf = nullptr;     // device copy of a host base buffer (whole-buffer path of jg_predict_windows): kept between
  int64_t d_bases_cap = 0;         // calls - the short-contig pass makes thousands of 96-window calls
  uint8_t *d_lut = nullptr;
  float *pool_part = nullptr;       // fused max-pool partial rows (split-f16 path)
  int64_t pool_part_cap = 0;
  std::vector<int> pool_fused_by;   // per op: conv op index that produces this OP_POOL's partials, or -1
  // nucleotide two-strand models (JG_OP_STRANDS): ids (W, strands, L), every strand runs through the program as a row of
  // its own with ONE frame; the strands' outputs are merged into the window's behind the last op
  int strands = 1;
  int id_frames = 6;                // frames per program row of the id tensor: 6 codon frames, or 1 (a strand)
  int id_bytes = 1;                 // bytes per id: 1, or 2 for a dicodon model (the program holds a JG_OP_EMBED)
  int merge_kind = 0;               // jg_merge_kind of the prediction
  float *merged[JG_MAX_VECS] = {};
  int64_t merged_cap[JG_MAX_VECS] = {};
  // "table net": an unmasked first conv straight on the ids, bias / activation, and the global pool behind it - the whole
  // representation learner of a strand branch - as ONE kernel (jg_kernels.hip: tab_conv_pool_kernel): the conv of a gathered
  // table row is a sum of k rows of the table (k, vocab, cout) = emb @ W[t]; nothing but ids in, pooled vectors out
  float *tab_table = nullptr;       // (k, vocab, cq) float4 quads, cq = ceil(cout / 4)
  float *tab_bias = nullptr;        // (cq) quads (zeros without a bias stage)
  int tab_conv = -1, tab_pool = -1; // ops replaced by the kernel (-1: the program does not match)
  uint16_t *tab_wfrag = nullptr;    // matrix-core form of the table (nucleotide one-hot input): jg_tabnet.hip
  float *tab_bias512 = nullptr;
  int tab_act = 0, tab_cq = 0, tab_vocab = 0, tab_zero = 0;   // table rows incl. an appended zero row when row 0 is not one
  int part_rows[JG_MAX_BUFS] = {};  // split-f16 path: partial rows per window the last conv wrote to each NMD slot
  int pool_rows = 0;                // same for the fused max pool
};

// cores this process may use: affinity mask, cgroup CPU quota, divided by LOCAL_WORLD_SIZE (jg_dust.hip)
int jg_usable_cores();

// ---- kernel launchers (defined in jg_kernels.hip) ---------------------------
int jg_launch_conv(jg_engine *e, const ConvArgs &a, hipStream_t s);
int jg_launch_embed(const uint16_t *ids, int64_t n_pos, const float *table, int vocab, int c, float *out, uint8_t *mask,
                    hipStream_t s);
int jg_launch_mask(const uint8_t *in, int rows, int L_in, int L_out, int k, int stride, int dil,
                   int pad_left, int mode, uint8_t *out, hipStream_t s);
int jg_launch_pool(const float *x, const uint8_t *mask, int n_win, int positions, int c, int kind,
                   float *out, int out_ld, hipStream_t s);
int jg_launch_dense(const float *in, int in_ld, const float *w, const float *b, int n_win, int cin,
                    int cout, int act, float *out, int out_ld, hipStream_t s);
int jg_launch_nmd_final(const float *part, int parts_per_win, const uint8_t *mask, int positions,
                        const float *moving_mean, float eps, int n_win, int c, float *out,
                        int out_ld, int out_off, hipStream_t s);
int jg_launch_eltwise(const EltArgs &a, hipStream_t s);
int jg_launch_layernorm(const EltArgs &a, int rows, int L, int tiles_m, hipStream_t s);
int jg_launch_encode(const uint8_t *bases, const int64_t *win_start, const int32_t *win_len,
                     int64_t n_win, int fsize, const uint8_t *lut, int flags, int l_pad,
                     uint8_t *ids, int32_t *counts, hipStream_t s);
int jg_launch_oodsig(const float *logits, int logits_ld, int n_cls, const float *nmd, int nmd_ld, int nmd_w,
                     int n_win, unsigned signal_bits, float eps, float *out, int out_ld, int out_off,
                     hipStream_t s);
int jg_launch_maxpool1d(const float *x, const uint8_t *mask_in, int rows, int L_in, int L_out, int c,
                        float *y, uint8_t *mask_out, hipStream_t s);
int jg_launch_pool_final(const float *part, int rows_per_win, int n_win, int c, float *out, int out_ld,
                         hipStream_t s);
int jg_launch_f32_to_f16s(const float *x, int64_t rows, int L, int c, uint4 *y, hipStream_t s, int *overflow);
int jg_launch_f16s_to_f32(const uint4 *x, int64_t rows, int L, int c, float *y, hipStream_t s);
int jg_launch_maxpool1d_f16s(const uint4 *x, int rows, int L_in, int L_out, int c, uint4 *y, hipStream_t s);
struct JgTabArgs {
  const uint8_t *ids;      // (rows, L)
  const float *table, *bias;
  float *out;              // (rows, out_ld)
  int out_ld, rows, L, L_out, pad_left, k, dil, vocab, cout, cq, act, pool_kind;
  int zero_id;             // table row that is all zeros in every tap (positions outside the sequence select it)
};
int64_t jg_tab_lds_bytes(int k, int vocab, int cq, int L, int dil);
// the same op on the matrix cores (jg_tabnet.hip): nucleotide one-hot rows only (vocab 5, table row 0 = zeros)
struct JgTabMArgs {
  const uint8_t *ids;      // (rows, L)
  const uint16_t *wfrag;   // f16 weight fragments, hi | lo planes (jg_tab_mfma_frag_halves)
  const float *bias;       // 512 floats (zero padded)
  float *out;              // (rows, out_ld)
  int out_ld, rows, L, L_out, pad_left, k, dil, cout, act, pool_kind;
};
bool jg_tab_mfma_supports(int k, int vocab, int cout, int dil);
int64_t jg_tab_mfma_frag_halves(int k);
int jg_launch_tab_mfma(jg_engine *e, const JgTabMArgs &a, hipStream_t s);
int jg_launch_tab_conv_pool(jg_engine *e, const JgTabArgs &a, hipStream_t s);
int jg_launch_strand_merge(const float *x, int x_ld, int n_win, int strands, int width, int kind, float *y, hipStream_t s);
int jg_launch_framesum(const float *x, int n_win, int frames, int64_t per_frame, float *y,
                       hipStream_t s);
int jg_launch_dust(uint8_t *d_bases, int64_t origin, int64_t span_len, const int64_t *d_rec_off, int64_t n_rec,
                   int window, int threshold, int64_t own0, int64_t own1, unsigned long long *d_masked, hipStream_t s);
int jg_conv_tile_m(int l_out);
int jg_conv_tile_m_for(int l_out, int k, int cin, int stride, int dil);
int jg_launch_conv_f16(jg_engine *e, const ConvHArgs &a, hipStream_t s);
int jg_conv_f16_lds_bytes(int k, int dil);
bool jg_conv_f16_supports(int k, int dil);
int jg_conv_f16_tile_m(void);
bool jg_conv_f16_has_pattern(unsigned ep, bool first_layer);
bool jg_conv_f16_has_flat_pattern(unsigned ep);
bool jg_conv_f16_has_narrow_pattern(unsigned ep);
int jg_conv_lut_lds_bytes(int k, int vocab);
bool jg_conv_lut_supports(int k, int dil, int vocab);
