/*
 * jaeger_hip.h - C-ABI of libjaeger_hip.so, the MI355X (gfx950) engine for the
 * Jaeger `predict` hot path: window table -> 6-frame codon ids -> conv forward.
 *
 * The reference has no FFI: its de-facto plugin boundary is the duck-typed
 * inference-engine class selected in src/jaeger/commands/predict.py:688-745
 * (InferModel / TFLiteInferModel / ONNXEngine / TensorRTEngine,
 * src/jaeger/nnlib/inference.py:300,486,626,925).  Each entry point below names
 * the reference code it replaces; jaeger_amd/engine.py wraps them into that
 * duck type (class_map, string_processor_config, predict()).
 *
 * Conventions: every function returns 0 on success and a negative jg_status on
 * failure (jg_last_error() then holds a message); nothing throws across the
 * ABI; the caller owns every buffer it passes in; the library owns its device
 * workspace.  A handle is bound to one GPU and is not thread-safe; distinct
 * handles are independent.  Pointers marked "dev/host" may be either: pass
 * JG_PTR_DEVICE or JG_PTR_HOST in the matching *_loc argument.  `stream` is a
 * hipStream_t passed as void* (NULL = the handle's own stream).
 */
#ifndef JAEGER_HIP_H
#define JAEGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JG_ABI_VERSION 1
#define JG_MAX_STAGES 8
#define JG_MAX_BUFS 6      /* activation / mask buffer slots */
#define JG_MAX_VECS 12     /* per-window vector slots */

typedef enum {
  JG_OK = 0,
  JG_ERR_INVALID = -1,      /* bad argument / malformed program */
  JG_ERR_HIP = -2,          /* a HIP runtime call failed */
  JG_ERR_UNSUPPORTED = -3,  /* op or geometry the kernels do not implement */
  JG_ERR_NOMEM = -4,
  JG_ERR_IO = -5            /* writing to a caller-supplied file descriptor failed (jg_table_write) */
} jg_status;

enum { JG_PTR_HOST = 0, JG_PTR_DEVICE = 1 };

/* ---- op program (the compiled layer plan) ------------------------------ */
typedef enum {
  JG_OP_CONV = 1,      /* MaskedConv1D (+ fused epilogue stages)  layers.py:1128-1332 */
  JG_OP_MASK = 2,      /* conv output-mask rule                    layers.py:1226-1255 */
  JG_OP_POOL = 3,      /* MaskedGlobalMax/AvgPooling               layers.py:455-538   */
  JG_OP_DENSE = 4,     /* Dense head layer                         builder.py:295      */
  JG_OP_ELTWISE = 5,   /* standalone norm / activation stages                          */
  JG_OP_NMD_FINAL = 6, /* NMDLayer mean - moving_mean              nmd.py:52-77        */
  JG_OP_OODSIG = 7,    /* OODSignalLayer                           layers.py:1632-1667 */
  JG_OP_MAXPOOL1D = 8, /* MaxPooling1D(2) of the legacy tower      v1/layers.py:154-207*/
  JG_OP_FRAMESUM = 9,  /* legacy frame Add                         v1/layers.py:399-423*/
  JG_OP_EMBED = 11,    /* Embedding lookup of WIDE ids (codon: DICODON - 4 096 codon pairs + the padding id, nnlib/inference.py:
                          430-451, builder.py:858-867): ids (rows, L) u16 -> out_buf f32 (rows, L, cout) = table[id] (b_off,
                          vocab x cout floats), out_mask = (id != 0).  Codon ids (one byte) are gathered inside the first
                          conv instead (in_buf = JG_BUF_IDS); a program with this op takes 16-bit id tensors when its vocabulary
                          exceeds 256.  w_off >= 0: + postab[position in the row] (k rows x cout floats: the rows of
                          SinusoidalPositionEmbedding, use_positional_embeddings, builder.py:886-892) - then also for one-byte ids */
  JG_OP_VECMAX = 12,   /* NMDMerge(mode="max") (nnlib/v2/nmd.py:150-152): out_vec[vec_off + c] = max over g < k of in_vec[g * cout + c] */
  JG_OP_STRANDS = 10   /* a branched (shared-weight) model over the k strands of a nucleotide input: every strand is a
                          program row of its own (ids (W, k, L), one frame per row); arg = how the strands' predictions
                          merge (jg_merge_kind); the embedding output is their average.  builder.py:1195-1266, :776-791 */
} jg_op_kind;

typedef enum {
  JG_ST_NONE = 0,
  JG_ST_BIAS = 1,    /* + bias[c]                         p0=bias                       */
  JG_ST_BN = 2,      /* g*((x-mu)*inv_std)+b              p0=mu p1=inv_std p2=g p3=b    */
  JG_ST_DYT = 3,     /* tanh(alpha*x)*g+b (*mask)         f0=alpha p2=g p3=b arg=use mask*/
  JG_ST_ADD = 4,     /* + other[buf arg] (residual add)   arg=activation buffer slot    */
  JG_ST_ACT = 5,     /* activation                        arg=jg_act                    */
  JG_ST_NMD = 6,     /* tap: partial masked channel sums  arg=partial-sum slot          */
  JG_ST_MASKMUL = 7, /* * out mask                                                      */
  JG_ST_LN = 8       /* MaskedLayerNormalization (eltwise op only) p2=g p3=b f0=eps     */
} jg_stage_kind;

typedef enum {
  JG_ACT_NONE = 0,
  JG_ACT_GELU_TANH = 1,  /* tf.nn.gelu(approximate=True) layers.py:29 */
  JG_ACT_GELU_ERF = 2,   /* legacy exact GELU v1/layers.py:72-79 */
  JG_ACT_RELU = 3,
  JG_ACT_TANH = 4,
  JG_ACT_SIGMOID = 5
} jg_act;

typedef enum { JG_MASK_ANY = 0, JG_MASK_MAJORITY = 1, JG_MASK_STRICT = 2 } jg_mask_mode;
typedef enum { JG_PAD_VALID = 0, JG_PAD_SAME = 1 } jg_padding;
typedef enum { JG_POOL_MAX = 0, JG_POOL_AVG = 1, JG_POOL_MAX_NOMASK = 2 } jg_pool_kind;
typedef enum { JG_MERGE_AVERAGE = 0, JG_MERGE_SUM = 1, JG_MERGE_MAX = 2, JG_MERGE_CONCAT = 3 } jg_merge_kind;   /* CONCAT (round 6): builder.py:1262-1265 */

/* buffer slot constants */
#define JG_BUF_NONE (-1)
#define JG_BUF_IDS (-2) /* conv input = embedding gather of the id tensor; mask = ids != 0 */

typedef struct {
  int32_t kind;   /* jg_stage_kind */
  int32_t arg;
  int64_t p0, p1, p2, p3; /* offsets (in floats) into the weight blob, -1 = unused */
  float f0;
  int32_t pad_;
} jg_stage;

typedef struct {
  int32_t kind;                 /* jg_op_kind */
  int32_t in_buf, out_buf;      /* activation slots (JG_BUF_IDS allowed for in_buf) */
  int32_t in_mask, out_mask;    /* mask slots, JG_BUF_NONE = no mask, JG_BUF_IDS = ids != 0 */
  int32_t k, cin, cout;         /* conv: taps, in/out channels; dense: cin->cout */
  int32_t stride, dilation;
  int32_t padding;              /* jg_padding */
  int32_t mask_mode;            /* jg_mask_mode */
  int32_t in_vec, out_vec;      /* vector slots (pool out, dense in/out, nmd out) */
  int32_t vec_off;              /* column offset inside out_vec (nmd concat) */
  int32_t arg;                  /* pool kind / dense activation / nmd partial slot */
  int64_t w_off;                /* conv/dense kernel offset (floats) */
  int64_t b_off;                /* dense bias / embedding table / nmd moving_mean offset */
  float f0;                     /* eps */
  int32_t n_stages;
  jg_stage stages[JG_MAX_STAGES];
} jg_op;

typedef struct jg_engine jg_engine; /* opaque, one per GPU */
typedef struct jg_model jg_model;   /* opaque, weights + op program on one engine */

/* ---- lifecycle ---------------------------------------------------------- */
/* Replaces the TF device / strategy setup of commands/predict.py:583-664. */
int jg_abi_version(void);
/* sizeof(jg_op) (which=0) / sizeof(jg_stage) (which=1): lets a binding check its struct layout */
int jg_sizeof(int which);
const char *jg_last_error(void);
int jg_engine_create(int device_id, jg_engine **out);
int jg_engine_destroy(jg_engine *e);
int jg_engine_sync(jg_engine *e);
/* Engine options.  JG_OPT_STREAM_BYTES: host-resident base buffers larger than this (default 32 MiB) are
 * streamed by jg_predict_windows - the start-sorted window list is cut into groups whose base span fits the
 * budget (and that hold a whole number of forward passes), each span goes host -> pinned staging buffer -> device
 * buffer on a copy stream (two of each) while the previous group is encoded and classified, outputs come back through
 * pinned staging, and no step of the loop synchronises the compute stream; the device never holds more than two
 * spans of bases.  This is the
 * "host-DRAM -> HBM streamed" ingest of BASELINE.json configs[4]; the reference streams Python strings through
 * tf.data instead (commands/predict.py:186-245).
 * JG_OPT_CONV_PC (default 0; values other than 0 only in the experiment build, JG_ERR_UNSUPPORTED in the shipped library):
 * which kernel runs the 128 -> 128 channel five-tap convs of the residual stacks
 * (layers.py:1882-1915): 0 = the two-workgroup kernel, 1 = the producer / consumer kernel (jg_conv_pc.hip), 2 = the
 * two-workgroup kernel with the producer / consumer experiment's pipelined main loop.  Same results bit for bit - the
 * switch exists for A/B timing and for the test that asserts exactly that.
 * JG_OPT_TERMINI_EXACT (default 0): 1 = jg_terminal_repeats runs every alignment through the kernel that carries length and
 * gap count through the dynamic programme, instead of only those the packed score-only pass leaves open (score > 100);
 * same table either way (tests/test_gpu_termini.py).
 * JG_OPT_TERMINI_REPORT_MIN (default 0; 2 .. 15): jg_terminal_repeats may report an alignment of FEWER than this many columns
 * as no alignment (score 0, length 0, ends -1) instead of scoring it.  The reference's decision rule only looks at alignments
 * longer than 12 columns (utils/termini.py:137-154), and under its scoring an alignment of L <= 50 columns is an exact run of
 * L matches: with the value 13 a record whose ends share no 13 matching bases in either orientation skips the dynamic
 * programme altogether (a hash probe of the 13-mers decides), and one whose only shared run is a single long one is settled
 * by the one-run check alone; every alignment of at least that many columns is reported exactly as with 0.
 * JG_OPT_DUST_ON_COPY_STREAM (default 1): the DUST pass of a streamed span runs on the copy stream behind the span's upload
 * (beside the previous group's convolutions) or, 0, on the compute stream in front of the span's encoder; same masks.
 * JG_OPT_TABLE_NET_LDS (default 0): a strand branch's conv + pool ("table net") runs on the matrix cores (0) or, 1, as the
 * exact-f32 LDS-table kernel - the form every shape the matrix-core kernel does not cover takes anyway.
 * JG_OPT_FUSE_RESBLOCK (default 1): narrow residual blocks (32 channels: three or five taps, any dilation with (k - 1) d <= 32;
 * 64 channels: dilation 1 or 2; stride 1, no bypass) run as ONE launch that keeps the block's intermediate tensor in LDS (1),
 * or conv by conv (0: A/B timing and the test that compares the two).
 * JG_OPT_RESET_PROGRESS (value ignored): sets JG_STAT_WINDOWS_DONE back to 0.  A thread that polls the mark of a
 * jg_predict_windows call ANOTHER thread is about to make calls this first (the call resets the mark itself, but only
 * once it has been entered - a poller that starts earlier would read the previous call's final count).
 * JG_OPT_STREAM_PRIORITY (default 0; round 6): 1 = the engine's stream is re-created at the device's HIGHEST stream priority (0: the
 *   default one again); set it on an idle engine, before its first launch.  run_core gives it to the side engine of the
 *   terminal-repeat scan: its short kernels then take the CUs the network's launches free between each other instead of
 *   queueing behind them, the repeat table exists long before the forward ends, and finished batches' rows are written beside it. */
enum { JG_OPT_STREAM_BYTES = 1, JG_OPT_CONV_PC = 2, JG_OPT_TERMINI_EXACT = 3, JG_OPT_DUST_ON_COPY_STREAM = 4,
       JG_OPT_TABLE_NET_LDS = 5, JG_OPT_RESET_PROGRESS = 6, JG_OPT_FUSE_RESBLOCK = 7, JG_OPT_TERMINI_REPORT_MIN = 8,
       JG_OPT_STREAM_PRIORITY = 9 };
int jg_engine_set_option(jg_engine *e, int key, int64_t value);
/* statistics of the engine's last jg_predict_windows call: number of streamed groups (0 = not streamed), bytes sent
 * through the staging buffers, peak bytes of bases resident on the device.  JG_STAT_WINDOWS_DONE may be read from
 * ANOTHER thread while a jg_predict_windows call with host outputs is running (reset it with JG_OPT_RESET_PROGRESS before
 * the call is handed to its thread): the output rows (and counts) of windows
 * [0, value) are final and may be consumed - the reference only sees its results when InferModel.predict returns
 * (nnlib/inference.py:341-373); here per-contig aggregation runs beside the forward of later windows. */
enum { JG_STAT_STREAM_GROUPS = 1, JG_STAT_STREAM_BYTES = 2, JG_STAT_PEAK_DEVICE_BASES = 3, JG_STAT_DUST_MASKED = 4,
       JG_STAT_WINDOWS_DONE = 5 };
int64_t jg_engine_get_stat(const jg_engine *e, int key);
/* DUST inside the fused path (replaces the per-contig pydustmasker call of seqops/io.py:104-108 without a host pass):
 * attach the record table (n_records + 1 offsets into the HOST base buffer the following jg_predict_windows /
 * jg_encode calls are given) and those calls soft-mask their uploaded copy of the bases on the device before encoding
 * it - whole-buffer uploads in one launch, streamed spans group by group with 64 bases of context - and encode with
 * the case respected (soft_mask bit 0).  The host buffer is not modified.  window <= 64.  n_records = 0 detaches.
 * JG_STAT_DUST_MASKED: bases lower-cased since the records were attached (overlapping streamed spans count twice). */
int jg_engine_set_dust(jg_engine *e, const int64_t *rec_off, int64_t n_records, int32_t window, int32_t threshold);

/* Replaces tf.saved_model.load + serving_default (nnlib/inference.py:307-325):
 * `ops` is the layer plan compiled by jaeger_amd/program.py, `weights` one f32
 * blob (host) the op offsets index into. */
int jg_model_create(jg_engine *e, const jg_op *ops, int n_ops, const float *weights,
                    int64_t n_weights, int32_t vocab, jg_model **out);
int jg_model_destroy(jg_model *m);
/* Arithmetic of the conv stack: 0 = exact-f32 MFMA, 1 = split-f16 ("f16x3": each f32 operand
 * as an f16 hi/lo pair, three f16 MFMAs per product, f32 accumulate; ~f32 accuracy).  A model
 * starts in mode 1 when at least one conv is eligible (k = 1 .. 5: 32, 64, 80..128 or a multiple of 128 output
 * channels, stride 1 or 2; k = 7 / 9: 128 channels, stride 1; a compiled epilogue pattern) or the program is the
 * 32-channel small-window family (one fused kernel), else 0; mode 1
 * falls back to 0 by itself if an activation ever leaves the f16 range.  Replaces the
 * --precision switch of commands/predict.py:604-613 (which trades accuracy; this one does not). */
int jg_model_set_precision(jg_model *m, int mode);
int jg_model_get_precision(const jg_model *m);
/* How the program was placed: JG_MSTAT_CONVS convolutions, JG_MSTAT_CONVS_F16X3 of them on the split-f16 kernels in
 * mode 1 (the others - 1x1 bypasses, widths or strides outside the kernel - run on the exact-f32 kernel with a layout
 * conversion either side: JG_MSTAT_LAYOUT_CONVERSIONS), JG_MSTAT_SMALL_FUSED = 1 when the whole conv stack runs as the
 * fused small-window kernel.  -1 for an unknown key. */
enum { JG_MSTAT_CONVS = 0, JG_MSTAT_CONVS_F16X3 = 1, JG_MSTAT_LAYOUT_CONVERSIONS = 2, JG_MSTAT_SMALL_FUSED = 3 };
int64_t jg_model_get_stat(const jg_model *m, int key);
/* ... and conv by conv, as text (one line each: geometry -> kernel; for a conv on the exact-f32 kernel the rule that kept
 * it there), NUL-terminated, truncated to cap. */
int jg_model_describe(const jg_model *m, char *buf, int64_t cap);

/* ---- hot path ----------------------------------------------------------- */
/* Replaces fragment_generator's per-window slice + 4x str.count
 * (seqops/io.py:119-133) and process_string_inference (seqops/encode.py:228-302)
 * for input_type="translated", ngram_width=3, seq_onehot=False:
 *   bases      concatenated contig bytes (ASCII), dev/host
 *   win_start  n_win byte offsets of the windows into `bases`
 *   win_len    n_win window lengths (<= fsize; shorter = whole-contig window)
 *   fsize      crop_size the frame offset is derived from (encode.py:232-236)
 *   lut65      65-byte table: entry 16*b0+4*b1+b2 (TCAG=0..3) -> codon_id+1; [64] unused
 *   soft_mask  0: upper-case before lookup/counting (masking=False, dustmask off); bit 0: bases are pre-cased (lower case
 *              = soft-masked, not counted); bit 1: ids are case sensitive (string_processor.masking = true);
 *              bit 2 (JG_ENC_NUCLEOTIDE): input_type="nucleotide" (encode.py:265-271, _map_nucleotide :36-41,
 *              _map_complement :28-33) - ids (n_win, 2, l_pad) u8, row 0 the window's first min(len, fsize) bases as
 *              A,G,C,T (either case) -> 1,2,3,4, row 1 the reverse complement of those bases, 0 = any other byte /
 *              padding (the all-zero one-hot row); l_pad counts bases; lut65 is not read
 *              bit 3 (JG_ENC_DICODON): codon = DICODON (ngram_width 6, encode.py:272-284 with the 4 096 pairs of
 *              seqops/maps.py:544-546) - ids (n_win, 6, l_pad) u16 (TWO bytes per id), entry i of frame j = the 6-gram at
 *              base j + 6 i of the strand -> 64 * (lut65[first codon] - 1) + (lut65[second codon] - 1) + 1 (lut65 = the
 *              plain codon table), 0 when either half is invalid; ceil((n - 8 + off) / 6) entries per frame
 *   l_pad      codons per frame row in the output (>= frame length of the longest window when the
 *              window table is on the host, >= frame length of fsize when it is on the device)
 * outputs (device or host per out_loc):
 *   ids        (n_win, 6, l_pad) u8, rows f1,f2,f3,r1,r2,r3, 0 = invalid / padding
 *   counts     (n_win, 4) i32 upper-case G,C,A,T counts of each window
 */
enum { JG_ENC_PRECASED = 1, JG_ENC_CASE_SENSITIVE = 2, JG_ENC_NUCLEOTIDE = 4, JG_ENC_DICODON = 8 };
int jg_encode(jg_engine *e, const uint8_t *bases, int64_t n_bases, int bases_loc,
              const int64_t *win_start, const int32_t *win_len, int win_loc, int64_t n_win,
              int32_t fsize, const uint8_t *lut65, int32_t soft_mask, int32_t l_pad,
              uint8_t *ids, int32_t *counts, int out_loc, void *stream);

/* Replaces InferModel.predict's per-batch serving_default call
 * (nnlib/inference.py:355-363): ids (n_win, 6, l) u8 - (n_win, 2, l) for a two-strand nucleotide model
 * (JG_OP_STRANDS), (n_win, 6, l) u16 for a dicodon model (a program with JG_OP_EMBED) - -> per-window outputs.
 * Any output pointer may be NULL.  Output widths are those of the program
 * (jg_model_vec_width).  `chunk` = windows per launch group (0 = default). */
int jg_forward(jg_model *m, const uint8_t *ids, int ids_loc, int64_t n_win, int32_t l,
               float *prediction, float *reliability, float *embedding, float *nmd,
               int out_loc, int32_t chunk, void *stream);

/* encode + forward on device-resident bases in one call (no id tensor round trip).  A two-strand nucleotide model
 * encodes with JG_ENC_NUCLEOTIDE whatever soft_mask says; l_pad then counts bases. */
int jg_predict_windows(jg_model *m, const uint8_t *bases, int64_t n_bases, int bases_loc,
                       const int64_t *win_start, const int32_t *win_len, int win_loc,
                       int64_t n_win, int32_t fsize, const uint8_t *lut65, int32_t soft_mask,
                       int32_t l_pad, float *prediction, float *reliability, float *embedding,
                       float *nmd, int32_t *counts, int out_loc, int32_t chunk, void *stream);

/* widths of the named outputs: which = 0 prediction, 1 reliability, 2 embedding, 3 nmd */
int jg_model_vec_width(const jg_model *m, int which);
/* algorithmic conv FLOPs of one window at l codons per frame */
double jg_model_flops_per_window(const jg_model *m, int32_t l);

/* ---- device memory helpers (so callers need no other GPU runtime) ------- */
int jg_dev_alloc(jg_engine *e, int64_t bytes, void **out);
int jg_dev_free(jg_engine *e, void *p);
int jg_memcpy_h2d(jg_engine *e, void *dst, const void *src, int64_t bytes);
int jg_memcpy_d2h(jg_engine *e, void *dst, const void *src, int64_t bytes);

/* ---- measurement (bench.py): HIP-event timing on the engine's stream ---- */
int jg_timer_start(jg_engine *e, void *stream);
int jg_timer_stop_ms(jg_engine *e, void *stream, float *ms);
/* accumulated HIP-event time and launch count of the dominant (conv) kernel since reset;
 * enabled by jg_profile_enable(e, 1), which inserts events around every conv launch */
int jg_profile_enable(jg_engine *e, int on);
int jg_profile_read(jg_engine *e, double *conv_ms, int64_t *conv_launches, double *conv_flops);
/* the same accumulators split by kernel family: split-f16 matrix-core convs, exact-f32 matrix-core convs, the
 * first layer's table-lookup kernel (no matrix cores: its "FLOPs" are the algorithmic ones of the conv it replaces),
 * and the fused small-window network kernel */
enum { JG_PROF_MFMA_F16X3 = 0, JG_PROF_MFMA_F32 = 1, JG_PROF_TABLE = 2, JG_PROF_FUSED_SMALL = 3 };
int jg_profile_read_class(jg_engine *e, int cls, double *ms, int64_t *launches, double *flops);

/* Box calibration (bench.py's `box` object; no counterpart in the reference): about `seconds` (0 < seconds <= 30) of
 * back-to-back launches of a bare v_mfma_f32_32x32x16_f16 loop on random register operands, two waves per SIMD on every
 * CU, no memory traffic.  *tflops = dense f16 matrix-core rate of the last launch, *clock_ghz = the shader clock the chip
 * held inside it (d s_memtime / d s_memrealtime x 100 MHz, median over workgroups).  info (optional, 5 doubles): launches,
 * ms of the last launch, mean TFLOP/s over all launches, lowest / highest per-workgroup clock.  MI355X devices differ by
 * up to 12 % on matrix-core-dense loops: a headline is comparable across boxes only relative to this figure. */
int jg_box_calibrate(jg_engine *e, double seconds, double *tflops, double *clock_ghz, double *info);

/* ---- FASTA ingest (host only; replaces the pyfastx iteration of seqops/io.py:98-103 and the
 * per-record Python strings of utils/fs.py:99-115) ------------------------------------------
 * jg_fasta_count : number of records (lines starting with '>') in a file image, and an upper bound of
 *                  the bytes their names take
 * jg_fasta_parse : bases (whitespace-stripped sequence lines joined; may alias `text` for in-place
 *                  compaction), record offsets (max_records + 1 entries, offsets[i+1]-offsets[i] =
 *                  length), names (header up to the first whitespace) back to back in `names` with
 *                  name_off (max_records + 1 entries) */
int jg_fasta_count(const uint8_t *text, int64_t n, int64_t *n_records, int64_t *name_bytes);
int jg_fasta_parse(const uint8_t *text, int64_t n, int64_t max_records, uint8_t *bases, int64_t *offsets,
                   uint8_t *names, int64_t *name_off, int64_t *n_records, int64_t *n_bases);

/* jg_fasta_index : the records of a file image without extracting them: byte offset of each record's header line
 *                  (rec_off, max_records + 1 entries, last = n), whitespace-stripped sequence length, names.  Under
 *                  torchrun rank 0 indexes, every rank parses only the byte ranges of the contigs it owns. */
int jg_fasta_index(const uint8_t *text, int64_t n, int64_t max_records, int64_t *rec_off, int64_t *seq_len,
                   uint8_t *names, int64_t *name_off, int64_t *n_records);

/* jg_fasta_scan / jg_fasta_fill : the same ingest on every core the process may use (n_threads <= 0: affinity mask,
 *                  cgroup CPU quota, divided by LOCAL_WORLD_SIZE).  scan cuts the image into one slice per thread,
 *                  finds the records and counts their bases, and returns the exact sizes the caller allocates; fill
 *                  copies bases / names to their final places in parallel and optionally returns the header-line byte
 *                  offsets jg_fasta_index gives (rec_off, n_records + 1 entries).  `text` must stay valid (and may be a
 *                  read-only mapping) until jg_fasta_scan_free.  Identical output to jg_fasta_parse. */
typedef struct jg_fasta_scan_t jg_fasta_scan_t;
int jg_fasta_scan(const uint8_t *text, int64_t n, int32_t n_threads, jg_fasta_scan_t **out, int64_t *n_records,
                  int64_t *n_bases, int64_t *name_bytes);
int jg_fasta_fill(const jg_fasta_scan_t *scan, uint8_t *bases, int64_t *offsets, uint8_t *names, int64_t *name_off,
                  int64_t *rec_off);
void jg_fasta_scan_free(jg_fasta_scan_t *scan);

/* ---- DUST soft-masking (host only; replaces pydustmasker.DustMasker(seq, window_size=64,
 * score_threshold=20).mask() of seqops/io.py:104-108) -------------------------------------------
 * Upper-cases every record of the base buffer, then lower-cases the symmetric-DUST intervals, in
 * place; n_threads <= 0 = all cores.  The buffer is then "pre-cased" for jg_encode (soft_mask bit 0). */
int jg_dust_mask(uint8_t *bases, const int64_t *offsets, int64_t n_records, int32_t window,
                 int32_t threshold, int32_t n_threads, int64_t *n_masked);
/* The same masks for DEVICE-resident bases, in place (window <= 64): symmetric DUST evaluated from its definition on the
 * GPU - every interval of up to window - 2 triplets by dynamic programme, one thread per interval start - bit-identical
 * to jg_dust_mask.  offsets (n_records + 1 entries) host or device per offsets_loc; n_masked (host, optional) makes the
 * call synchronous.  jg_predict_windows runs it by itself on the uploaded bases once records are attached to the engine
 * (jg_engine_set_dust). */
int jg_dust_mask_device(jg_engine *e, uint8_t *d_bases, int64_t n_bases, const int64_t *offsets, int offsets_loc,
                        int64_t n_records, int32_t window, int32_t threshold, int64_t *n_masked, void *stream);

/* ---- result table text (host only; replaces df.to_csv(path, sep="\t", index=False, float_format="%.3f") of
 * postprocess/collect.py:578-580 and 602-607 for the rows of `rows` (NULL: rows 0 .. n_rows - 1), without a header line) --
 * Column c is cols[c] read as kinds[c]: JG_COL_STRING = UTF-8 strings laid end to end with ONE separator byte behind each,
 * string r = bytes [starts[c][r], starts[c][r + 1] - 1) (copied as they are: the caller keeps strings the csv writer would
 * quote - tab, double quote, line break - away from this call); JG_COL_INT = int64; JG_COL_FLOAT = float64, printed as
 * CPython's "%.3f" % v (exact value, round-half-even) and as nothing for NaN; JG_COL_BOOL = uint8, "True" / "False".
 * *text (n_bytes bytes, no terminator) is released by jg_table_free.  n_threads <= 0: every usable core. */
enum { JG_COL_STRING = 0, JG_COL_INT = 1, JG_COL_FLOAT = 2, JG_COL_BOOL = 3, JG_COL_SPANS = 4 };
int jg_table_format(int32_t n_cols, const int32_t *kinds, const void *const *cols, const int64_t *const *starts,
                    const int64_t *rows, int64_t n_rows, int32_t n_threads, char **text, int64_t *n_bytes);
void jg_table_free(char *text);
/* (round 6) JG_COL_SPANS = strings as explicit spans of a byte buffer: string r = bytes [starts[c][2 r], starts[c][2 r + 1])
 * of cols[c] - record names straight out of the FASTA parser's name buffer (contig_id, collect.py:441), class labels and
 * repeat kinds as spans of a short label blob (collect.py:443, termini.py:137-154): no per-row string objects on the host.
 * jg_table_write renders the same rows and writes them to the open file descriptor fd (at its position, in row order)
 * instead of returning them. */
int jg_table_write(int32_t n_cols, const int32_t *kinds, const void *const *cols, const int64_t *const *starts,
                   const int64_t *rows, int64_t n_rows, int32_t n_threads, int32_t fd, int64_t *n_bytes);
/* *unique = 1 when the n strings buf[off[i] .. off[i + 1]) are pairwise different (host only).  Replaces the uniqueness test of
 * pandas' merge on contig_id (postprocess/collect.py:527-532): with unique record names the repeat table joins by record number. */
int jg_names_unique(const uint8_t *buf, const int64_t *off, int64_t n, int32_t n_threads, int32_t *unique);
/* window_summary strings (replaces get_window_summary of postprocess/helpers.py:73-108 over the run lengths of :8-40, one
 * Python call per contig in postprocess/collect.py:520-523): contig c owns calls[first[c] .. first[c] + count[c]); every run of
 * equal calls prints as its length followed by letters[class] (0 = no letter).  *text holds one NUL-terminated string per
 * contig back to back (n_bytes in all); release with jg_table_free. */
int jg_run_summaries(const int32_t *calls, int64_t n_calls, const int64_t *first, const int64_t *count, int64_t n_contigs,
                     const uint8_t *letters, int32_t n_letters, int32_t n_threads, char **text, int64_t *n_bytes);

/* ---- per-contig reductions (host only; replaces the contig-by-contig np.mean / np.var of postprocess/collect.py:332-356 and
 * the np.mean calls over every contig's entropy / energy / G+C / N% slice, :393-395, :319-327) -----------------------------
 * Segment s covers rows [first[s], first[s] + count[s]) (count >= 1).  Both restate numpy's own summation order, so that the
 * values - rounded to fp16 and printed with three decimals downstream - do not move by a bit:
 * jg_segment_mean_var: x (n_rows, n_cols) f32 row-major -> mean, var (n_seg, n_cols) f32 (var may be NULL; ddof 0):
 *                      rows added one after the other in f32 (n_cols = 1: pairwise, as numpy reduces a contiguous axis);
 * jg_segment_mean_1d:  v (n) f32 (is_f64 = 0) or f64 (1) -> out (n_seg) of the same type, numpy's pairwise summation.
 * n_threads <= 0: every usable core. */
int jg_segment_mean_var(const float *x, int64_t n_rows, int32_t n_cols, const int64_t *first, const int64_t *count,
                        int64_t n_seg, float *mean, float *var, int32_t n_threads);
int jg_segment_mean_1d(const void *v, int32_t is_f64, int64_t n, const int64_t *first, const int64_t *count, int64_t n_seg,
                       void *out, int32_t n_threads);

/* ---- CRF window decoding (host only; replaces the per-contig loop over postprocess/helpers.py:398-449
 * viterbi_decode that postprocess/collect.py:343-346 runs for `jaeger predict --crf`) ---------------
 * logits (n_windows, n_classes) f32 row-major; chain c covers windows [first[c], first[c+1]) (first has
 * n_chains + 1 entries); costs (n_classes, n_classes) f64, costs[a][b] = price of a -> b between adjacent
 * windows (build_transition_costs, helpers.py:347-395); path (n_windows) receives the MAP class of every
 * window.  f64 log-softmax emissions, ties to the lowest class index, chains of one window = argmax. */
int jg_viterbi_decode(const float *logits, int64_t n_windows, int32_t n_classes, const int64_t *first,
                      int64_t n_chains, const double *costs, int32_t *path);

/* ---- terminal-repeat scan (replaces utils/termini.py:88-189 scan_for_terminal_repeats: parasail
 * sw_trace_scan_16 of the first vs the last min(max(int(0.04 len), 400), 4000) bases, direct and
 * reverse-complemented; match 2 / mismatch -100 / gap 100 + 5(k-1)) -------------------------------
 * results: (n_records, 10) int32 on the host, per record DTR then ITR: score, alignment length,
 * gaps in the query row, end in the query, end in the reference; -1 for records shorter than min_len.
 * bases: host or device per bases_loc; offsets (n_records + 1) on the host.  Synchronous. */
int jg_terminal_repeats(jg_engine *e, const uint8_t *bases, int64_t n_bases, int bases_loc,
                        const int64_t *offsets, int64_t n_records, int32_t min_len, int32_t *results);

#ifdef __cplusplus
}
#endif
#endif /* JAEGER_HIP_H */
