// C-ABI of libjaeger_hip.so (see include/jaeger_hip.h): engine / model lifecycle,
// the op-program interpreter that sequences the gfx950 kernels, and the timing
// hooks bench.py uses.  Host logic only - kernels live in jg_kernels.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>

#include "jg_common.h"
#include "jg_small.h"
#include "jg_resblock64.h"
#include "jg_vecmax.h"

static thread_local char g_err[1024] = "";

void jg_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *jg_last_error(void) { return g_err; }
extern "C" int jg_abi_version(void) { return JG_ABI_VERSION; }
extern "C" int jg_sizeof(int which) {
  return which == 0 ? (int)sizeof(jg_op) : (which == 1 ? (int)sizeof(jg_stage) : -1);
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
extern "C" int jg_engine_create(int device_id, jg_engine **out) {
  JG_REQUIRE(out != nullptr, JG_ERR_INVALID, "jg_engine_create: out is NULL");
  int n_dev = 0;
  JG_HIP(hipGetDeviceCount(&n_dev));
  JG_REQUIRE(device_id >= 0 && device_id < n_dev, JG_ERR_INVALID,
             "jg_engine_create: device %d not present (%d visible)", device_id, n_dev);
  JG_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  JG_HIP(hipGetDeviceProperties(&prop, device_id));
  JG_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, JG_ERR_UNSUPPORTED,
             "jg_engine_create: device %d is %s; this library targets gfx950 (MI355X) only",
             device_id, prop.gcnArchName);
  jg_engine *e = new jg_engine();
  e->dev = device_id;
  e->n_cu = prop.multiProcessorCount;
  JG_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  JG_HIP(hipEventCreate(&e->t0));
  JG_HIP(hipEventCreate(&e->t1));
  *out = e;
  return JG_OK;
}

extern "C" int jg_engine_destroy(jg_engine *e) {
  if (e == nullptr) return JG_OK;
  (void)hipSetDevice(e->dev);
  (void)hipStreamSynchronize(e->stream);
  for (auto &p : e->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto ev : e->pool) (void)hipEventDestroy(ev);
  (void)hipEventDestroy(e->t0);
  (void)hipEventDestroy(e->t1);
  for (int i = 0; i < 2; ++i) {
    if (e->pin[i]) (void)hipHostFree(e->pin[i]);
    if (e->dbase[i]) (void)hipFree(e->dbase[i]);
    if (e->h2d_done[i]) (void)hipEventDestroy(e->h2d_done[i]);
    if (e->enc_done[i]) (void)hipEventDestroy(e->enc_done[i]);
    if (e->grp_done[i]) (void)hipEventDestroy(e->grp_done[i]);
    if (e->pin_io[i]) (void)hipHostFree(e->pin_io[i]);
  }
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->d_rec_off) (void)hipFree(e->d_rec_off);
  if (e->d_dust_cnt) (void)hipFree(e->d_dust_cnt);
  (void)hipStreamDestroy(e->stream);
  delete e;
  return JG_OK;
}

extern "C" int jg_engine_sync(jg_engine *e) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_engine_sync: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  JG_HIP(hipStreamSynchronize(e->stream));
  return JG_OK;
}

extern "C" int jg_engine_set_option(jg_engine *e, int key, int64_t value) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_engine_set_option: NULL engine");
  switch (key) {
    case JG_OPT_STREAM_BYTES:
      JG_REQUIRE(value >= 4096, JG_ERR_INVALID, "jg_engine_set_option: stream budget %lld < 4096 bytes", (long long)value);
      e->stream_bytes = value;
      return JG_OK;
    case JG_OPT_CONV_PC:
      JG_REQUIRE(value >= 0 && value <= 2, JG_ERR_INVALID, "jg_engine_set_option: JG_OPT_CONV_PC takes 0, 1 or 2, got %lld", (long long)value);
#ifndef JG_EXPERIMENT
      JG_REQUIRE(value == 0, JG_ERR_UNSUPPORTED, "jg_engine_set_option: JG_OPT_CONV_PC = %lld needs the experiment build (make -C jaeger_amd/csrc "
                 "exp; JAEGER_HIP_LIB=jaeger_amd/libjaeger_hip_exp.so): the producer / consumer kernels are not in the shipped library",
                 (long long)value);
#endif
      e->conv_pc = (int)value;
      return JG_OK;
    case JG_OPT_TERMINI_EXACT:
      e->termini_exact = value != 0;
      return JG_OK;
    case JG_OPT_TERMINI_REPORT_MIN:
      JG_REQUIRE(value == 0 || (value >= 2 && value <= 15), JG_ERR_INVALID,
                 "jg_engine_set_option: JG_OPT_TERMINI_REPORT_MIN = %lld (0, or 2 .. 15 columns)", (long long)value);
      e->termini_report_min = (int)value;
      return JG_OK;
    case JG_OPT_DUST_ON_COPY_STREAM:
      e->dust_on_copy = value != 0;
      return JG_OK;
    case JG_OPT_TABLE_NET_LDS:
      e->tab_lds_only = value != 0;
      return JG_OK;
    case JG_OPT_FUSE_RESBLOCK:
      e->fuse_resblock = value != 0;
      return JG_OK;
    case JG_OPT_RESET_PROGRESS:
      e->windows_done.store(0, std::memory_order_release);
      return JG_OK;
    case JG_OPT_STREAM_PRIORITY: {
      JG_REQUIRE(value == 0 || value == 1, JG_ERR_INVALID, "jg_engine_set_option: JG_OPT_STREAM_PRIORITY takes 0 or 1, got %lld",
                 (long long)value);
      JG_HIP(hipSetDevice(e->dev));
      JG_HIP(hipStreamSynchronize(e->stream));                   // (an idle engine: nothing is waited for)
      int least = 0, greatest = 0;                               // numerically lower = more urgent
      JG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      hipStream_t fresh = nullptr;
      JG_HIP(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, value ? greatest : least));
      (void)hipStreamDestroy(e->stream);
      e->stream = fresh;
      return JG_OK;
    }
    default:
      jg_set_error("jg_engine_set_option: unknown key %d", key);
      return JG_ERR_INVALID;
  }
}

extern "C" int64_t jg_engine_get_stat(const jg_engine *e, int key) {
  if (e == nullptr) return -1;
  switch (key) {
    case JG_STAT_STREAM_GROUPS: return e->streamed_groups;
    case JG_STAT_STREAM_BYTES: return e->streamed_bytes;
    case JG_STAT_PEAK_DEVICE_BASES: return e->peak_dev_bases;
    case JG_STAT_WINDOWS_DONE: return e->windows_done.load(std::memory_order_acquire);
    case JG_STAT_DUST_MASKED: {          // bases the device DUST lower-cased since the records were attached (syncs)
      if (e->d_dust_cnt == nullptr) return 0;
      unsigned long long h = 0;
      if (hipSetDevice(e->dev) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess ||
          hipMemcpy(&h, e->d_dust_cnt, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
      return (int64_t)h;
    }
    default: return -1;
  }
}

// Attach the record table of the host base buffer the following jg_predict_windows / jg_encode calls will be given:
// their uploaded copy of the bases is then soft-masked on the device (symmetric DUST, jg_dust.hip) before it is
// encoded, and the encoder respects the case.  n_records = 0 (or rec_off NULL) detaches.
extern "C" int jg_engine_set_dust(jg_engine *e, const int64_t *rec_off, int64_t n_records, int32_t window,
                                  int32_t threshold) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_engine_set_dust: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  if (rec_off == nullptr || n_records <= 0) {
    e->n_rec = 0;
    return JG_OK;
  }
  JG_REQUIRE(window >= 4 && window <= 64 && threshold > 0, JG_ERR_UNSUPPORTED,
             "jg_engine_set_dust: window %d outside 4..64 (mask on the host with jg_dust_mask)", window);
  for (int64_t r = 0; r < n_records; ++r)
    JG_REQUIRE(rec_off[r] >= 0 && rec_off[r + 1] >= rec_off[r], JG_ERR_INVALID, "jg_engine_set_dust: record %lld has a negative length",
               (long long)r);
  JG_HIP(hipStreamSynchronize(e->stream));            // (a previous call may still read the old table)
  if (n_records + 1 > e->rec_cap) {
    if (e->d_rec_off) JG_HIP(hipFree(e->d_rec_off));
    e->d_rec_off = nullptr;
    JG_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_rec_off), (size_t)(n_records + 1) * sizeof(int64_t)));
    e->rec_cap = n_records + 1;
  }
  if (e->d_dust_cnt == nullptr) JG_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_dust_cnt), sizeof(unsigned long long)));
  JG_HIP(hipMemcpy(e->d_rec_off, rec_off, (size_t)(n_records + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  JG_HIP(hipMemset(e->d_dust_cnt, 0, sizeof(unsigned long long)));
  e->n_rec = n_records;
  e->rec_end = rec_off[n_records];
  e->dust_window = window;
  e->dust_threshold = threshold;
  return JG_OK;
}

static hipStream_t pick_stream(jg_engine *e, void *stream) {
  return stream != nullptr ? reinterpret_cast<hipStream_t>(stream) : e->stream;
}

extern "C" int jg_dev_alloc(jg_engine *e, int64_t bytes, void **out) {
  JG_REQUIRE(e != nullptr && out != nullptr && bytes >= 0, JG_ERR_INVALID, "jg_dev_alloc: bad args");
  JG_HIP(hipSetDevice(e->dev));
  *out = nullptr;
  if (bytes == 0) return JG_OK;
  hipError_t err = hipMalloc(out, (size_t)bytes);
  if (err != hipSuccess) {
    jg_set_error("jg_dev_alloc: hipMalloc(%lld) -> %s", (long long)bytes, hipGetErrorString(err));
    return JG_ERR_NOMEM;
  }
  return JG_OK;
}

extern "C" int jg_dev_free(jg_engine *e, void *p) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_dev_free: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  if (p != nullptr) JG_HIP(hipFree(p));
  return JG_OK;
}

extern "C" int jg_memcpy_h2d(jg_engine *e, void *dst, const void *src, int64_t bytes) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_memcpy_h2d: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  if (bytes > 0) {
    JG_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
    JG_HIP(hipStreamSynchronize(e->stream));
  }
  return JG_OK;
}

extern "C" int jg_memcpy_d2h(jg_engine *e, void *dst, const void *src, int64_t bytes) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_memcpy_d2h: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  if (bytes > 0) {
    JG_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
    JG_HIP(hipStreamSynchronize(e->stream));
  }
  return JG_OK;
}

extern "C" int jg_timer_start(jg_engine *e, void *stream) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_timer_start: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  JG_HIP(hipEventRecord(e->t0, pick_stream(e, stream)));
  return JG_OK;
}

extern "C" int jg_timer_stop_ms(jg_engine *e, void *stream, float *ms) {
  JG_REQUIRE(e != nullptr && ms != nullptr, JG_ERR_INVALID, "jg_timer_stop_ms: bad args");
  JG_HIP(hipSetDevice(e->dev));
  JG_HIP(hipEventRecord(e->t1, pick_stream(e, stream)));
  JG_HIP(hipEventSynchronize(e->t1));
  JG_HIP(hipEventElapsedTime(ms, e->t0, e->t1));
  return JG_OK;
}

extern "C" int jg_profile_enable(jg_engine *e, int on) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_profile_enable: NULL engine");
  e->profile = on != 0;
  e->conv_ms = 0.0;
  e->conv_flops = 0.0;
  e->conv_launches = 0;
  for (int i = 0; i < 4; ++i) { e->cls_ms[i] = 0.0; e->cls_flops[i] = 0.0; e->cls_launches[i] = 0; }
  return JG_OK;
}

static int drain_profile(jg_engine *e) {
  for (auto &p : e->pending) {
    JG_HIP(hipEventSynchronize(p.b));
    float ms = 0.f;
    JG_HIP(hipEventElapsedTime(&ms, p.a, p.b));
    e->conv_ms += ms;
    e->conv_flops += p.flops;
    e->conv_launches += 1;
    e->cls_ms[p.cls & 3] += ms;
    e->cls_flops[p.cls & 3] += p.flops;
    e->cls_launches[p.cls & 3] += 1;
    e->pool.push_back(p.a);
    e->pool.push_back(p.b);
  }
  e->pending.clear();
  return JG_OK;
}

extern "C" int jg_profile_read(jg_engine *e, double *conv_ms, int64_t *conv_launches,
                               double *conv_flops) {
  JG_REQUIRE(e != nullptr, JG_ERR_INVALID, "jg_profile_read: NULL engine");
  JG_HIP(hipSetDevice(e->dev));
  int rc = drain_profile(e);
  if (rc != JG_OK) return rc;
  if (conv_ms) *conv_ms = e->conv_ms;
  if (conv_launches) *conv_launches = e->conv_launches;
  if (conv_flops) *conv_flops = e->conv_flops;
  return JG_OK;
}

extern "C" int jg_profile_read_class(jg_engine *e, int cls, double *ms, int64_t *launches, double *flops) {
  JG_REQUIRE(e != nullptr && cls >= 0 && cls < 4, JG_ERR_INVALID, "jg_profile_read_class: bad arguments");
  JG_HIP(hipSetDevice(e->dev));
  int rc = drain_profile(e);
  if (rc != JG_OK) return rc;
  if (ms) *ms = e->cls_ms[cls];
  if (launches) *launches = e->cls_launches[cls];
  if (flops) *flops = e->cls_flops[cls];
  return JG_OK;
}

static int prof_event(jg_engine *e, hipEvent_t *ev) {
  if (!e->pool.empty()) {
    *ev = e->pool.back();
    e->pool.pop_back();
    return JG_OK;
  }
  JG_HIP(hipEventCreate(ev));
  return JG_OK;
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
struct Shape {
  int frames = 0, L = 0, C = 0;
};

static void conv_geometry(int L_in, int k, int stride, int dil, int padding, int *L_out,
                          int *pad_left) {
  if (padding == JG_PAD_SAME) {
    // TF 'SAME': L_out = ceil(L/s); pad_left = pad_total // 2
    const int lo = (L_in + stride - 1) / stride;
    int total = (lo - 1) * stride + (k - 1) * dil + 1 - L_in;
    if (total < 0) total = 0;
    *L_out = lo;
    *pad_left = total / 2;
  } else {
    const int span = dil * (k - 1) + 1;
    *L_out = L_in >= span ? (L_in - span) / stride + 1 : 0;
    *pad_left = 0;
  }
}

static int validate_program(const jg_model *m) {
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    auto slot_ok = [](int s, bool allow_ids) {
      return (s >= 0 && s < JG_MAX_BUFS) || s == JG_BUF_NONE || (allow_ids && s == JG_BUF_IDS);
    };
    JG_REQUIRE(op.kind >= JG_OP_CONV && op.kind <= JG_OP_VECMAX, JG_ERR_INVALID,
               "op %zu: unknown kind %d", i, op.kind);
    if (op.kind == JG_OP_VECMAX)
      JG_REQUIRE(op.in_vec >= 0 && op.out_vec >= 0 && op.in_vec != op.out_vec && op.k >= 1 && op.cout >= 1 && op.vec_off >= 0,
                 JG_ERR_INVALID, "op %zu: vecmax takes k >= 1 groups of cout values from one vector into another", i);
    if (op.kind == JG_OP_STRANDS)
      JG_REQUIRE(i + 1 == m->ops.size() && op.k >= 2 && op.k <= 8 && op.arg >= JG_MERGE_AVERAGE && op.arg <= JG_MERGE_CONCAT,
                 JG_ERR_INVALID, "op %zu: a strands op closes the program, merges 2 - 8 strands by average / sum / max", i);
    JG_REQUIRE(slot_ok(op.in_buf, true) && slot_ok(op.out_buf, false) && slot_ok(op.in_mask, true) &&
                   slot_ok(op.out_mask, false),
               JG_ERR_INVALID, "op %zu: buffer slot out of range", i);
    JG_REQUIRE(op.in_vec >= -1 && op.in_vec < JG_MAX_VECS && op.out_vec >= -1 &&
                   op.out_vec < JG_MAX_VECS,
               JG_ERR_INVALID, "op %zu: vector slot out of range", i);
    JG_REQUIRE(op.n_stages >= 0 && op.n_stages <= JG_MAX_STAGES, JG_ERR_INVALID,
               "op %zu: %d stages", i, op.n_stages);
    auto off_ok = [&](int64_t off, int64_t n) { return off >= 0 && off + n <= m->n_w; };
    if (op.kind == JG_OP_CONV) {
      JG_REQUIRE(op.k >= 1 && op.cin >= 1 && op.cout >= 1 && op.stride >= 1 && op.dilation >= 1,
                 JG_ERR_INVALID, "op %zu: bad conv geometry", i);
      const int64_t cin_pad = (op.cin + 1) & ~1, cout_pad = (op.cout + 31) / 32 * 32;
      JG_REQUIRE(off_ok(op.w_off, (int64_t)op.k * cin_pad * cout_pad), JG_ERR_INVALID,
                 "op %zu: conv kernel outside the weight blob", i);
      if (op.in_buf == JG_BUF_IDS)
        JG_REQUIRE(off_ok(op.b_off, (int64_t)m->vocab * op.cin), JG_ERR_INVALID,
                   "op %zu: embedding table outside the weight blob", i);
    }
    if (op.kind == JG_OP_EMBED) {
      JG_REQUIRE(i == 0 && op.out_buf >= 0 && op.out_mask >= 0 && op.cout >= 4 && op.cout % 4 == 0 && m->vocab >= 2 &&
                     m->vocab <= 65536 && off_ok(op.b_off, (int64_t)m->vocab * op.cout),
                 JG_ERR_INVALID, "op %zu: an embedding op opens the program (vocabulary 2 .. 65536 - 16-bit ids above 256 -, table inside the weight blob)", i);
      // w_off >= 0: rows of a position table (k positions x cout floats) added to the looked-up rows
      JG_REQUIRE(op.w_off < 0 || (op.k >= 1 && off_ok(op.w_off, (int64_t)op.k * op.cout)), JG_ERR_INVALID,
                 "op %zu: position table outside the weight blob", i);
    } else if (!m->ops.empty() && m->ops[0].kind == JG_OP_EMBED) {
      JG_REQUIRE(op.in_buf != JG_BUF_IDS && op.in_mask != JG_BUF_IDS, JG_ERR_INVALID,
                 "op %zu: reads the id tensor directly in a program that opens with an embedding op (its buffer and mask take the tensor's place)", i);
    }
    if (op.kind == JG_OP_DENSE) {
      JG_REQUIRE(off_ok(op.w_off, (int64_t)op.cin * op.cout), JG_ERR_INVALID,
                 "op %zu: dense kernel outside the weight blob", i);
      JG_REQUIRE(op.b_off < 0 || off_ok(op.b_off, op.cout), JG_ERR_INVALID,
                 "op %zu: dense bias outside the weight blob", i);
    }
    {   // NMD taps per op: the conv kernels carry two accumulators, the element-wise / LayerNorm kernels one
      int n_nmd = 0;
      for (int s = 0; s < op.n_stages; ++s) n_nmd += op.stages[s].kind == JG_ST_NMD;
      JG_REQUIRE(n_nmd <= (op.kind == JG_OP_CONV ? 2 : 1), JG_ERR_UNSUPPORTED,
                 "op %zu: %d NMD taps in one stage list (at most %d)", i, n_nmd, op.kind == JG_OP_CONV ? 2 : 1);
    }
    for (int s = 0; s < op.n_stages; ++s) {
      const jg_stage &st = op.stages[s];
      const int64_t c = op.cout;
      switch (st.kind) {
        case JG_ST_BIAS:
          JG_REQUIRE(off_ok(st.p0, c), JG_ERR_INVALID, "op %zu stage %d: bias offset", i, s);
          break;
        case JG_ST_BN:
          JG_REQUIRE(off_ok(st.p0, c) && off_ok(st.p1, c) && off_ok(st.p2, c) && off_ok(st.p3, c),
                     JG_ERR_INVALID, "op %zu stage %d: batchnorm offsets", i, s);
          break;
        case JG_ST_DYT:
          JG_REQUIRE(off_ok(st.p2, c) && off_ok(st.p3, c), JG_ERR_INVALID,
                     "op %zu stage %d: dyt offsets", i, s);
          break;
        case JG_ST_ADD:
          JG_REQUIRE(st.arg >= 0 && st.arg < JG_MAX_BUFS, JG_ERR_INVALID,
                     "op %zu stage %d: add slot", i, s);
          break;
        case JG_ST_NMD:
          JG_REQUIRE(st.arg >= 0 && st.arg < JG_MAX_BUFS, JG_ERR_INVALID,
                     "op %zu stage %d: nmd partial slot", i, s);
          break;
        case JG_ST_LN:
          JG_REQUIRE(op.kind == JG_OP_ELTWISE && s == 0, JG_ERR_UNSUPPORTED,
                     "op %zu stage %d: a layer norm must lead an element-wise op", i, s);
          JG_REQUIRE(off_ok(st.p2, c) && off_ok(st.p3, c), JG_ERR_INVALID, "op %zu stage %d: layernorm offsets", i, s);
          break;
        case JG_ST_ACT:
        case JG_ST_MASKMUL:
          break;
        default:
          jg_set_error("op %zu stage %d: stage kind %d is not implemented", i, s, st.kind);
          return JG_ERR_UNSUPPORTED;
      }
    }
  }
  return JG_OK;
}

// Dry-run the program at `l` codons per frame: per-slot element counts (per
// window) and vector widths.  Also used to validate that shapes line up.
bool jg_tab_mfma_row_fits(int L_out, int k, int dil);       // jg_tabnet.hip: the matrix-core form's id image holds the row

// rows of l positions run on the table-net kernel (else - rows too long for the LDS image - layer by layer)
static bool tab_usable(const jg_model *m, int l) {
  if (m->tab_conv < 0) return false;
  const jg_op &c = m->ops[(size_t)m->tab_conv];
  int lo, pl;
  conv_geometry(l, c.k, 1, c.dilation, c.padding, &lo, &pl);
  return lo >= 1 && jg_tab_lds_bytes(c.k, m->tab_vocab, m->tab_cq, l, c.dilation) <= 160 * 1024;
}

static int plan_shapes(jg_model *m, int l, int64_t act_elems[JG_MAX_BUFS],
                       int64_t msk_elems[JG_MAX_BUFS], int64_t nmd_elems[JG_MAX_BUFS],
                       int vec_w[JG_MAX_VECS], double *flops) {
  Shape sh[JG_MAX_BUFS];
  int mlen[JG_MAX_BUFS] = {};   // positions per window of each mask slot
  for (int i = 0; i < JG_MAX_BUFS; ++i) act_elems[i] = msk_elems[i] = nmd_elems[i] = 0;
  for (int i = 0; i < JG_MAX_VECS; ++i) vec_w[i] = 0;
  double fl = 0.0;
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    switch (op.kind) {
      case JG_OP_CONV: {
        Shape in;
        if (op.in_buf == JG_BUF_IDS) { in.frames = m->id_frames; in.L = l; in.C = op.cin; }
        else in = sh[op.in_buf];
        JG_REQUIRE(in.C == op.cin, JG_ERR_INVALID, "op %zu: conv expects %d channels, input has %d",
                   i, op.cin, in.C);
        int lo, pl;
        conv_geometry(in.L, op.k, op.stride, op.dilation, op.padding, &lo, &pl);
        JG_REQUIRE(lo > 0, JG_ERR_INVALID,
                   "op %zu: conv output is empty at %d codons per frame (window too short)", i, in.L);
        JG_REQUIRE(op.out_buf >= 0, JG_ERR_INVALID, "op %zu: conv needs an output slot", i);
        sh[op.out_buf] = Shape{in.frames, lo, op.cout};
        fl += 2.0 * op.k * op.cin * op.cout * (double)in.frames * lo;
        if ((int)i == m->tab_conv && tab_usable(m, l)) break;      // table net: the activation never exists
        // (+ one position for an odd row: a phase-split tensor holds two phases of (lo + 1) / 2 positions)
        act_elems[op.out_buf] = std::max<int64_t>(act_elems[op.out_buf], (int64_t)in.frames * (lo + (lo & 1)) * op.cout);
        const int tiles = std::max((lo + 63) / 64, 8 * ((lo + 255) / 256));
        for (int s = 0; s < op.n_stages; ++s)
          if (op.stages[s].kind == JG_ST_NMD)
            nmd_elems[op.stages[s].arg] = std::max<int64_t>(nmd_elems[op.stages[s].arg],
                                                           (int64_t)in.frames * tiles * op.cout);
      } break;
      case JG_OP_EMBED: {
        JG_REQUIRE(op.out_buf >= 0 && op.out_mask >= 0 && op.cout > 0, JG_ERR_INVALID, "op %zu: bad embedding op", i);
        JG_REQUIRE(op.w_off < 0 || l <= op.k, JG_ERR_UNSUPPORTED,
                   "op %zu: rows of %d positions, the model's position table holds %d", i, l, op.k);
        sh[op.out_buf] = Shape{m->id_frames, l, op.cout};
        act_elems[op.out_buf] = std::max<int64_t>(act_elems[op.out_buf], (int64_t)m->id_frames * (l + (l & 1)) * op.cout);
        mlen[op.out_mask] = m->id_frames * l;
        msk_elems[op.out_mask] = std::max<int64_t>(msk_elems[op.out_mask], (int64_t)m->id_frames * l);
      } break;
      case JG_OP_MASK: {
        const int L_in = op.in_mask == JG_BUF_IDS ? l : (mlen[op.in_mask] / m->id_frames);
        int lo, pl;
        conv_geometry(L_in, op.k, op.stride, op.dilation, op.padding, &lo, &pl);
        JG_REQUIRE(op.out_mask >= 0 && lo > 0, JG_ERR_INVALID, "op %zu: bad mask op", i);
        mlen[op.out_mask] = m->id_frames * lo;
        msk_elems[op.out_mask] = std::max<int64_t>(msk_elems[op.out_mask], (int64_t)m->id_frames * lo);
      } break;
      case JG_OP_ELTWISE: {
        const Shape in = sh[op.in_buf];
        JG_REQUIRE(in.C == op.cout, JG_ERR_INVALID, "op %zu: eltwise channel mismatch", i);
        sh[op.out_buf] = in;
        act_elems[op.out_buf] = std::max<int64_t>(act_elems[op.out_buf], (int64_t)in.frames * in.L * in.C);
        for (int s = 0; s < op.n_stages; ++s)           // an NMD tap behind a LayerNorm: partial rows like a conv's
          if (op.stages[s].kind == JG_ST_NMD) {
            const int tiles = std::max((in.L + 63) / 64, 8 * ((in.L + 255) / 256));
            nmd_elems[op.stages[s].arg] = std::max<int64_t>(nmd_elems[op.stages[s].arg], (int64_t)in.frames * tiles * in.C);
          }
      } break;
      case JG_OP_MAXPOOL1D: {
        const Shape in = sh[op.in_buf];
        const int lo = in.L / 2;
        JG_REQUIRE(lo > 0, JG_ERR_INVALID, "op %zu: maxpool output empty", i);
        sh[op.out_buf] = Shape{in.frames, lo, in.C};
        act_elems[op.out_buf] = std::max<int64_t>(act_elems[op.out_buf], (int64_t)in.frames * lo * in.C);
      } break;
      case JG_OP_FRAMESUM: {
        const Shape in = sh[op.in_buf];
        sh[op.out_buf] = Shape{1, in.L, in.C};
        act_elems[op.out_buf] = std::max<int64_t>(act_elems[op.out_buf], (int64_t)in.L * in.C);
      } break;
      case JG_OP_POOL: {
        const Shape in = sh[op.in_buf];
        JG_REQUIRE(op.out_vec >= 0, JG_ERR_INVALID, "op %zu: pool needs an output vector", i);
        vec_w[op.out_vec] = std::max(vec_w[op.out_vec], op.vec_off + in.C);
      } break;
      case JG_OP_DENSE:
        JG_REQUIRE(op.in_vec >= 0 && op.out_vec >= 0, JG_ERR_INVALID, "op %zu: dense vectors", i);
        JG_REQUIRE(vec_w[op.in_vec] >= op.cin, JG_ERR_INVALID,
                   "op %zu: dense expects %d inputs, vector %d has %d", i, op.cin, op.in_vec,
                   vec_w[op.in_vec]);
        vec_w[op.out_vec] = std::max(vec_w[op.out_vec], op.vec_off + op.cout);
        break;
      case JG_OP_NMD_FINAL:
        JG_REQUIRE(op.out_vec >= 0, JG_ERR_INVALID, "op %zu: nmd needs an output vector", i);
        vec_w[op.out_vec] = std::max(vec_w[op.out_vec], op.vec_off + op.cout);
        break;
      case JG_OP_OODSIG:
        JG_REQUIRE(op.out_vec >= 0, JG_ERR_INVALID, "op %zu: oodsig needs an output vector", i);
        vec_w[op.out_vec] = std::max(vec_w[op.out_vec], op.vec_off + op.cout);
        break;
      case JG_OP_VECMAX:
        JG_REQUIRE(vec_w[op.in_vec] >= op.k * op.cout, JG_ERR_INVALID, "op %zu: vecmax expects %d x %d inputs, vector %d has %d", i,
                   op.k, op.cout, op.in_vec, vec_w[op.in_vec]);
        vec_w[op.out_vec] = std::max(vec_w[op.out_vec], op.vec_off + op.cout);
        break;
      default:
        break;
    }
  }
  for (int i = 0; i < JG_MAX_VECS; ++i) vec_w[i] = (vec_w[i] + 3) & ~3;  // float4-aligned rows
  if (flops) *flops = fl;
  return JG_OK;
}


// ---------------------------------------------------------------------------
// split-f16 operand preparation (host): see jg_conv_f16.hip for the layouts
// ---------------------------------------------------------------------------
static inline uint16_t f16_bits(float x) {
  _Float16 h = (_Float16)x;
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static inline float f16_value(float x) { return (float)(_Float16)x; }

// ---------------------------------------------------------------------------
// fused small-window network (jg_small.hip): does the op program match the family, and its operands
//   [MASK] CONV(ids, k0, E -> 32)  { [MASK] CONV(k 3, 32 -> 32, SAME) } x 2 | 4   POOL(avg | max)   ...heads
// every conv's stages being  [BIAS] [BN]  [ADD]  GELU(tanh)  [ [BN] GELU(tanh) ]
// ---------------------------------------------------------------------------
static void free_small(jg_model *m) {
  if (m->small == nullptr) return;
  JgSmallNet *sn = m->small;
  if (sn->d_lut) (void)hipFree(sn->d_lut);
  if (sn->d_epi) (void)hipFree(sn->d_epi);
  if (sn->d_part) (void)hipFree(sn->d_part);
  if (sn->d_wfrag) (void)hipFree(sn->d_wfrag);
  delete sn;
  m->small = nullptr;
}

static int prepare_small(jg_model *m, const float *weights) {
  std::vector<int> convs;
  int pool_op = -1;
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    if (op.kind == JG_OP_MASK || op.kind == JG_OP_NMD_FINAL) continue;     // (NMD finishes: matched to taps below)
    if (op.kind == JG_OP_CONV) { convs.push_back((int)i); continue; }
    if (op.kind == JG_OP_POOL) { pool_op = (int)i; break; }
    return JG_OK;                                   // anything else in front of the pool: not this family
  }
  const int nc = (int)convs.size() - 1;
  if (pool_op < 0 || nc < 1 || !jg_small_supports(nc, m->ops[convs[0]].k, m->vocab)) return JG_OK;
  const jg_op &c0 = m->ops[convs[0]];
  if (c0.in_buf != JG_BUF_IDS || c0.cout != 32 || c0.stride != 1 || c0.dilation != 1 || c0.mask_mode != JG_MASK_ANY ||
      !(c0.in_mask == JG_BUF_IDS || c0.in_mask == JG_BUF_NONE))
    return JG_OK;
  const bool use_mask = c0.in_mask == JG_BUF_IDS;
  for (int q = 1; q <= nc; ++q) {
    const jg_op &c = m->ops[convs[q]];
    if (c.in_buf != m->ops[convs[q - 1]].out_buf || c.k != 3 || c.cin != 32 || c.cout != 32 || c.stride != 1 ||
        c.dilation != 1 || c.padding != JG_PAD_SAME || c.mask_mode != JG_MASK_ANY || (c.in_mask >= 0) != use_mask)
      return JG_OK;
    if (use_mask && c.in_mask != m->ops[convs[q - 1]].out_mask) return JG_OK;
  }
  const jg_op &pl = m->ops[pool_op];
  if (pl.in_buf != m->ops[convs[nc]].out_buf || !(pl.arg == JG_POOL_AVG || pl.arg == JG_POOL_MAX) ||
      (pl.in_mask >= 0) != use_mask || (use_mask && pl.in_mask != m->ops[convs[nc]].out_mask))
    return JG_OK;
  // no later op may read an activation slot (the kernel never writes them)
  for (size_t i = (size_t)pool_op + 1; i < m->ops.size(); ++i) {
    const int k = m->ops[i].kind;
    if (k == JG_OP_CONV || k == JG_OP_MASK || k == JG_OP_POOL || k == JG_OP_ELTWISE || k == JG_OP_MAXPOOL1D ||
        k == JG_OP_FRAMESUM || k == JG_OP_NMD_FINAL)
      return JG_OK;
  }
  JgSmallNet *sn = new JgSmallNet();
  for (JgSmallLayer &ly : sn->layer) ly.add = ly.aff2 = ly.save = ly.tap = 0;
  std::vector<float> epi((size_t)(nc + 1) * 4 * 32, 0.f);
  bool ok = true;
  // split-f16 weight fragments of the k = 3 convs: built behind the fold below (the first affine's scale goes into them)
  std::vector<uint16_t> frag((size_t)nc * 12 * 64 * 8, 0);
  std::vector<double> scale1((size_t)(nc + 1) * 32, 1.0), shift1((size_t)(nc + 1) * 32, 0.0);
  // epilogue parameters: fold BIAS / BN chains (f64), match  affine [ADD] GELU [affine GELU]
  for (int q = 0; q <= nc && ok; ++q) {
    const jg_op &c = m->ops[convs[q]];
    std::vector<double> s1(32, 1.0), t1(32, 0.0), s2(32, 1.0), t2(32, 0.0);
    int st = 0;
    auto fold = [&](std::vector<double> &sc, std::vector<double> &sh) {
      bool any = false;
      for (; st < c.n_stages; ++st) {
        const jg_stage &g = c.stages[st];
        if (g.kind == JG_ST_BIAS) {
          for (int n = 0; n < 32; ++n) sh[n] += (double)weights[g.p0 + n];
        } else if (g.kind == JG_ST_BN) {
          for (int n = 0; n < 32; ++n) {
            const double mu = weights[g.p0 + n], is = weights[g.p1 + n], ga = weights[g.p2 + n], be = weights[g.p3 + n];
            sc[n] = sc[n] * is * ga;
            sh[n] = (sh[n] - mu) * is * ga + be;
          }
        } else break;
        any = true;
      }
      return any;
    };
    fold(s1, t1);
    JgSmallLayer &ly = sn->layer[q];
    if (st < c.n_stages && c.stages[st].kind == JG_ST_ADD) {
      // the shortcut must be the output of an earlier layer of this chain, and the only one alive
      int src = -1;
      for (int r = q - 1; r >= 0; --r)
        if (m->ops[convs[r]].out_buf == c.stages[st].arg) { src = r; break; }
      if (src < 0) { ok = false; break; }
      bool clobbered = false;
      for (int r = src + 1; r < q; ++r) clobbered |= m->ops[convs[r]].out_buf == c.stages[st].arg;
      if (clobbered) { ok = false; break; }
      sn->layer[src].save = 1;
      ly.add = 1;
      ++st;
    }
    if (!(st < c.n_stages && c.stages[st].kind == JG_ST_ACT && c.stages[st].arg == JG_ACT_GELU_TANH)) { ok = false; break; }
    ++st;
    if (st < c.n_stages && (c.stages[st].kind == JG_ST_BIAS || c.stages[st].kind == JG_ST_BN)) {
      fold(s2, t2);
      if (!(st < c.n_stages && c.stages[st].kind == JG_ST_ACT && c.stages[st].arg == JG_ACT_GELU_TANH)) { ok = false; break; }
      ++st;
      ly.aff2 = 1;
    }
    if (st < c.n_stages && c.stages[st].kind == JG_ST_NMD && st == c.n_stages - 1) {
      // a tap behind the layer's last stage: masked channel sums of the layer's output, finished by the NMD_FINAL op
      // that reads this partial slot (the program's slot number is kept to find it)
      ly.tap = ++sn->n_taps;
      sn->tap_part_slot[ly.tap] = c.stages[st].arg;
      sn->tap_conv_op[ly.tap] = convs[q];
      ++st;
    }
    if (st != c.n_stages) { ok = false; break; }
    for (int n = 0; n < 32; ++n) {
      // the first affine lives in the weights (scale) and in the accumulators' initial value (shift): jg_small.hip
      scale1[(size_t)q * 32 + n] = s1[n];
      shift1[(size_t)q * 32 + n] = t1[n];
      epi[((size_t)q * 4 + 0) * 32 + n] = 1.0f;
      epi[((size_t)q * 4 + 1) * 32 + n] = (float)t1[n];
      epi[((size_t)q * 4 + 2) * 32 + n] = (float)s2[n];
      epi[((size_t)q * 4 + 3) * 32 + n] = (float)t2[n];
    }
  }
  for (int q = 1; q <= nc && ok; ++q) {
    const jg_op &c = m->ops[convs[q]];
    const float *w = weights + c.w_off;               // (3, 32, 32) f32 (cin even, cout multiple of 32: no padding)
    // the folded weights w * scale1 go into f16 planes WITHOUT a power-of-two pre-scale: they must sit inside the f16
    // range (a large batch-norm scale would turn hi into inf and lo into -inf: NaN logits), and the layer's largest
    // weight must stay well above the subnormal quantum 2^-24 the lo plane resolves (hi + lo then still carries ~19 bits of
    // it); otherwise the model stays on the generic split-f16 / exact-f32 kernels, which pre-scale per conv
    double vmax = 0.0;
    for (int t = 0; t < 3; ++t)
      for (int ci = 0; ci < 32; ++ci)
        for (int co = 0; co < 32; ++co) {
          const double v = std::fabs((double)w[((size_t)t * 32 + ci) * 32 + co] * scale1[(size_t)q * 32 + co]);
          if (!(v <= 65000.0)) ok = false;            // (also catches NaN)
          vmax = std::max(vmax, v);
        }
    if (vmax != 0.0 && vmax < 0.015625) ok = false;
    if (!ok) break;
    for (int t = 0; t < 3; ++t)
      for (int cc = 0; cc < 2; ++cc)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int co = lane & 31, ci = cc * 16 + (lane >> 5) * 8 + j;
            const float v = (float)((double)w[((size_t)t * 32 + ci) * 32 + co] * scale1[(size_t)q * 32 + co]);
            const float hi = f16_value(v);
            const size_t base = ((((size_t)(q - 1) * 3 + t) * 2 + cc) * 2) * 64 * 8;
            frag[base + (size_t)lane * 8 + j] = f16_bits(hi);
            frag[base + 64 * 8 + (size_t)lane * 8 + j] = f16_bits(v - hi);      // (may be an f16 subnormal: the MFMA honours those)
          }
  }
  // a shortcut saved by layer r is read by exactly the next ADD: saves must not overlap
  if (ok) {
    int pending = -1;
    for (int q = 0; q <= nc; ++q) {
      if (sn->layer[q].add) pending = -1;
      if (sn->layer[q].save) {
        if (pending >= 0) ok = false;
        pending = q;
      }
    }
  }
  // every NMD_FINAL in front of the pool must finish one of the taps (the one most recently written to its slot)
  for (int i = 0; i < pool_op && ok; ++i) {
    if (m->ops[(size_t)i].kind != JG_OP_NMD_FINAL) continue;
    int tap = 0;
    for (int t = 1; t <= sn->n_taps; ++t)
      if (sn->tap_part_slot[t] == m->ops[(size_t)i].arg && sn->tap_conv_op[t] < i) tap = t;
    if (tap == 0 || m->ops[(size_t)i].cout != 32) ok = false;
  }
  if (sn->n_taps > JG_SMALL_MAX_LAYERS) ok = false;
  if (!ok) { delete sn; return JG_OK; }
  sn->n_slots = 1 + sn->n_taps;
  // first-layer table T_t[id] = E[id] . W_t (f64), row `vocab` = zeros (padding), row 0 = zeros when ids mask
  const int k0 = c0.k, vr = m->vocab + 1, cin_pad = (c0.cin + 1) & ~1;
  std::vector<float> lut((size_t)k0 * vr * 32, 0.f);
  const float *emb = weights + c0.b_off, *w0 = weights + c0.w_off;
  for (int t = 0; t < k0; ++t)
    for (int id = (use_mask ? 1 : 0); id < m->vocab; ++id)
      for (int n = 0; n < 32; ++n) {
        double acc = 0.0;
        for (int ci = 0; ci < c0.cin; ++ci)
          acc += (double)emb[(size_t)id * c0.cin + ci] * (double)w0[((size_t)t * cin_pad + ci) * 32 + n];
        lut[((size_t)t * vr + id) * 32 + n] = (float)(acc * scale1[(size_t)n]);
      }
  // every output position reads exactly one row of tap 0 (a codon's, the masked id 0's or the padding row `vocab`):
  // the first affine's shift rides on all of them
  for (int id = 0; id < vr; ++id)
    for (int n = 0; n < 32; ++n) lut[(size_t)id * 32 + n] = (float)((double)lut[(size_t)id * 32 + n] + shift1[(size_t)n]);
  JG_HIP(hipMalloc(reinterpret_cast<void **>(&sn->d_lut), lut.size() * sizeof(float)));
  JG_HIP(hipMemcpy(sn->d_lut, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
  JG_HIP(hipMalloc(reinterpret_cast<void **>(&sn->d_epi), epi.size() * sizeof(float)));
  JG_HIP(hipMemcpy(sn->d_epi, epi.data(), epi.size() * sizeof(float), hipMemcpyHostToDevice));
  JG_HIP(hipMalloc(reinterpret_cast<void **>(&sn->d_wfrag), frag.size() * 2));
  JG_HIP(hipMemcpy(sn->d_wfrag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));

  sn->valid = true;
  sn->n_conv = nc;
  sn->k0 = k0;
  sn->pad_same0 = c0.padding == JG_PAD_SAME;
  sn->use_mask = use_mask ? 1 : 0;
  sn->pool_kind = pl.arg;
  sn->first_op = 0;
  sn->pool_op = pool_op;
  sn->flops_per_pos0 = 2.0 * k0 * c0.cin * 32;
  sn->flops_per_pos = 2.0 * 3 * 32 * 32 * nc;
  m->small = sn;
  return JG_OK;
}

// exact-f32 conv operands: weights grouped by 8 input channels so that a lane fetches the four
// k-steps of a group with one 16-byte load (see conv_f32_kernel)
// ---- table net (jg_kernels.hip: tab_conv_pool_kernel) ------------------------------------------------------------
// The program matches when its first op is an UNMASKED stride-1 conv on the ids whose stages are [bias] [activation]
// and whose output goes to an unmasked global pool and nowhere else: the strand branch of the nucleotide model
// (conv1d -> relu -> max1d, train_config/nn_config_500bp_dvf.yaml).  Table entry (t, id) = embedding row id times W[t]
// (f64 sums, rounded once): for one-hot input W[t][id - 1] itself, the zero row for id 0.
static int prepare_tab(jg_model *m, const float *weights) {
  if (m->ops.size() < 2 || m->id_frames != 1) return JG_OK;      // (rows of one frame: the pool is per row)
  const jg_op &c = m->ops[0], &pl = m->ops[1];
  if (c.kind != JG_OP_CONV || c.in_buf != JG_BUF_IDS || c.in_mask >= 0 || c.in_mask == JG_BUF_IDS || c.out_mask >= 0 ||
      c.stride != 1 || c.n_stages > 2)
    return JG_OK;
  if (pl.kind != JG_OP_POOL || pl.in_buf != c.out_buf || pl.in_mask >= 0 || pl.in_mask == JG_BUF_IDS) return JG_OK;
  int bias_off = -1, act = JG_ACT_NONE, seen = 0;
  for (int q = 0; q < c.n_stages; ++q) {
    const jg_stage &st = c.stages[q];
    if (st.kind == JG_ST_BIAS && q == 0) { bias_off = (int)st.p0; ++seen; }
    else if (st.kind == JG_ST_ACT && q == c.n_stages - 1) { act = st.arg; ++seen; }
  }
  if (seen != c.n_stages) return JG_OK;
  for (size_t i = 2; i < m->ops.size(); ++i) {                    // the conv's output must have no other reader
    const jg_op &o = m->ops[i];
    if (o.in_buf == c.out_buf || o.out_buf == c.out_buf) return JG_OK;
    for (int q = 0; q < o.n_stages; ++q)
      if (o.stages[q].kind == JG_ST_ADD && o.stages[q].arg == c.out_buf) return JG_OK;
  }
  const int cq = (c.cout + 3) / 4;
  const float *w = weights + c.w_off, *emb = weights + c.b_off;
  // positions outside the sequence (SAME padding) add nothing: they select an all-zero table row - row 0 when the
  // embedding's row 0 is zero (one-hot input), else a row appended behind the vocabulary
  bool row0_zero = true;
  for (int ci = 0; ci < c.cin; ++ci) row0_zero &= emb[ci] == 0.f;
  const int V = m->vocab + (row0_zero ? 0 : 1);
  if (V > 255 || cq > 256 || jg_tab_lds_bytes(c.k, V, cq, 64, c.dilation) > 160 * 1024) return JG_OK;
  const int cin_pad = (c.cin + 1) & ~1, cout_pad = (c.cout + 31) / 32 * 32;
  std::vector<float> tab((size_t)c.k * V * cq * 4, 0.f), bias((size_t)cq * 4, 0.f);
  for (int t = 0; t < c.k; ++t)
    for (int id = 0; id < m->vocab; ++id)
      for (int n = 0; n < c.cout; ++n) {
        double acc = 0.0;
        for (int ci = 0; ci < c.cin; ++ci)
          acc += (double)emb[(size_t)id * c.cin + ci] * (double)w[((size_t)t * cin_pad + ci) * cout_pad + n];
        tab[((size_t)t * V + id) * cq * 4 + n] = (float)acc;
      }
  if (bias_off >= 0)
    for (int n = 0; n < c.cout; ++n) bias[(size_t)n] = weights[bias_off + n];
  JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->tab_table), tab.size() * sizeof(float)));
  JG_HIP(hipMemcpy(m->tab_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->tab_bias), bias.size() * sizeof(float)));
  JG_HIP(hipMemcpy(m->tab_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  bool f16_range = true;                                          // (a weight beyond the f16 range keeps the exact-f32 form)
  for (float v : tab) f16_range &= std::fabs(v) < 32768.f;
  if (row0_zero && f16_range && jg_tab_mfma_supports(c.k, m->vocab, c.cout, c.dilation)) {
    // the same table as MFMA A-operand fragments (jg_tabnet.hip): [32-channel tile][k-step of 4 taps][hi | lo][lane][8]
    const int ks = (c.k + 3) / 4;
    std::vector<uint16_t> frag((size_t)jg_tab_mfma_frag_halves(c.k), 0);
    for (int tile = 0; tile < 16; ++tile)
      for (int st = 0; st < ks; ++st)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int co = tile * 32 + (lane & 31), kk = (lane >> 5) * 8 + j, tap = 4 * st + kk / 4, nuc = kk % 4;
            if (co >= c.cout || tap >= c.k) continue;
            const float v = tab[((size_t)tap * V + (nuc + 1)) * cq * 4 + co];
            const float hi = f16_value(v);
            const size_t base = (((size_t)tile * ks + st) * 2) * 64 * 8;
            frag[base + (size_t)lane * 8 + j] = f16_bits(hi);
            frag[base + 64 * 8 + (size_t)lane * 8 + j] = f16_bits(v - hi);
          }
    std::vector<float> b512(512, 0.f);
    for (int n = 0; n < c.cout; ++n) b512[(size_t)n] = bias[(size_t)n];
    JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->tab_wfrag), frag.size() * sizeof(uint16_t)));
    JG_HIP(hipMemcpy(m->tab_wfrag, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->tab_bias512), b512.size() * sizeof(float)));
    JG_HIP(hipMemcpy(m->tab_bias512, b512.data(), b512.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  m->tab_conv = 0;
  m->tab_pool = 1;
  m->tab_act = act;
  m->tab_cq = cq;
  m->tab_vocab = V;
  m->tab_zero = row0_zero ? 0 : m->vocab;
  return JG_OK;
}

static int prepare_f32(jg_model *m, const float *weights) {
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    if (op.kind != JG_OP_CONV) continue;
    const int cin_pad2 = (op.cin + 1) & ~1, cout_pad = (op.cout + 31) / 32 * 32, cin8 = (op.cin + 7) / 8 * 8;
    const float *w = weights + op.w_off;
    std::vector<float> w8((size_t)op.k * (cin8 / 8) * cout_pad * 8, 0.f);
    for (int t = 0; t < op.k; ++t)
      for (int c = 0; c < op.cin; ++c)
        for (int n = 0; n < op.cout; ++n)
          w8[(((size_t)t * (cin8 / 8) + c / 8) * cout_pad + n) * 8 + c % 8] = w[((size_t)t * cin_pad2 + c) * cout_pad + n];
    JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->hprep[i].d_w8), w8.size() * sizeof(float)));
    JG_HIP(hipMemcpy(m->hprep[i].d_w8, w8.data(), w8.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return JG_OK;
}

static int prepare_f16(jg_model *m, const float *weights) {
  m->hprep.assign(m->ops.size(), ConvHPrep());
  m->pool_fused_by.assign(m->ops.size(), -1);
  m->f16_eligible = true;
  m->f16_reason.clear();
  // Pass A - every conv on its own: can it run on the split-f16 kernel (taps / dilation inside the tiling, a compiled
  // epilogue pattern; 32, 64 or a multiple of 128 output channels - narrow convs on 64- / 32-channel workgroup tiles,
  // wider ones as one launch per 128 channels; stride 2 as the stride-1 conv whose even outputs are kept)?  Ineligible
  // convs (1x1 bypass, other strides or widths) keep the exact-f32 kernel inside an otherwise split-f16 program; pass B
  // below places the layout conversions between them.
  std::string first_reason;
  size_t cur = 0;
  auto fail = [&](const char *why) {
    if (first_reason.empty()) first_reason = why;
    if (m->hprep[cur].why_f32.empty()) m->hprep[cur].why_f32 = why;
  };
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    if (op.kind != JG_OP_CONV) continue;
    cur = i;
    ConvHPrep &hp = m->hprep[i];
    hp.f16_ok = false;
    if ((int)i == m->tab_conv) { fail("runs as the table-net kernel (exact f32, ids to pooled vectors)"); continue; }
    // a 1x1 conv (the bypass of a strided / widening residual block) and a 3-tap conv (ResidualBlock's default kernel
    // size, layers.py:1787) ride the 5-tap kernel: weights in the middle taps, the matrix-core work of the others skipped
    hp.as_k5 = op.k >= 1 && op.k <= 4 && op.in_buf != JG_BUF_IDS;       // (2- and 4-tap convs the same way)
    const int kk = hp.as_k5 ? 5 : op.k, kdil = (hp.as_k5 && op.k == 1) ? 1 : op.dilation;
    if (op.stride != 1 && !(op.stride == 2 && (kk == 5 || kk == 7 || kk == 9) && op.in_buf != JG_BUF_IDS)) { fail("strided conv"); continue; }
    // (a first conv on ids runs as the table variant whatever its tap count, when the table fits LDS)
    static const bool no_lut = jg_exp_env("JG_NO_LUT") != nullptr;
    const bool lut_ok = !no_lut && op.in_buf == JG_BUF_IDS && (op.in_mask == JG_BUF_IDS || op.in_mask < 0) && op.cout <= 128 &&
                        op.stride == 1 && jg_conv_lut_supports(op.k, op.dilation, m->vocab);
    const bool mfma_ok = jg_conv_f16_supports(kk, kdil);
    if (!mfma_ok && !lut_ok) { fail("taps / dilation outside the split-f16 tiling"); continue; }
    const bool narrow = op.cout == 32 || op.cout == 64;
    if (op.cout % 16 != 0 || !(narrow || (op.cout > 64 && op.cout <= 128) || op.cout % 128 == 0)) {
      fail("conv width is not 32, 64, 80..128 or a multiple of 128 channels");
      continue;
    }
    // (the k = 5, 7 and 9 kernels are all built with run-time output geometry - other widths than 128, stride 2; a first
    // conv of up to 128 channels runs as the table variant, which has it too: no table -> 128 channels only, below)
    if (op.in_buf == JG_BUF_IDS && op.cout > 128) { fail("first conv wider than 128 channels"); continue; }
    if (op.in_buf != JG_BUF_IDS && op.cin % 16 != 0) { fail("conv input width is not a multiple of 16"); continue; }
    bool conv_ok = true;
    auto cfail = [&](const char *why) { conv_ok = false; fail(why); };
    const int cin16 = (op.cin + 15) / 16 * 16, cin_pad = (op.cin + 1) & ~1, cout_pad = 128;
    hp.cc_in = cin16 / 16;
    hp.n_half = (op.cout + 127) / 128;
    hp.cw = (narrow && op.in_buf != JG_BUF_IDS) ? op.cout : 128;    // (a first conv runs as the table variant: 128-wide)
    const int cwide = hp.n_half * 128;                                // channels incl. zero padding
    const float *w = weights + op.w_off;   // (k, cin_pad, cout_pad32) f32
    const int cout_pad32 = (op.cout + 31) / 32 * 32;
    float maxabs = 0.f;
    for (int64_t q = 0; q < (int64_t)op.k * cin_pad * cout_pad32; ++q) maxabs = std::max(maxabs, fabsf(w[q]));
    int sexp = 0;
    if (maxabs > 0.f) {
      int e2;
      frexpf(maxabs, &e2);            // maxabs = f * 2^e2, f in [0.5, 1)
      sexp = 3 - e2;                  // scaled max in [4, 8)
    }
    const float wscale = ldexpf(1.f, sexp);
    hp.acc_scale = ldexpf(1.f, -sexp);
    const int kc_total = cin16 / 8;
    const size_t half_items = (size_t)2 * kk * kc_total * cout_pad;       // one 128-channel half: [plane][tap][kc][128]
    const size_t n_items = half_items * hp.n_half;
    hp.wh_half_items = (int64_t)half_items;
    std::vector<uint16_t> wh(n_items * 8, 0);
    for (int t = 0; t < op.k; ++t)
      for (int c = 0; c < op.cin; ++c)
        for (int n = 0; n < op.cout; ++n) {
          const float v = w[((size_t)t * cin_pad + c) * cout_pad32 + n] * wscale;
          const float hi = f16_value(v);
          const size_t base = (size_t)(n / 128) * half_items;
          const int tk = hp.as_k5 ? std::max(1, (5 - op.k) / 2) + t : t;   // (a 1x1 / 3-tap conv's taps sit in the middle of five)
          const size_t item = base + (((size_t)0 * kk + tk) * kc_total + c / 8) * cout_pad + n % 128;
          const size_t item_lo = base + (((size_t)1 * kk + tk) * kc_total + c / 8) * cout_pad + n % 128;
          wh[item * 8 + c % 8] = f16_bits(hi);
          wh[item_lo * 8 + c % 8] = f16_bits(v - hi);
        }
    JG_HIP(hipMalloc(reinterpret_cast<void **>(&hp.d_wh), n_items * 16));
    JG_HIP(hipMemcpy(hp.d_wh, wh.data(), n_items * 16, hipMemcpyHostToDevice));
    if (op.in_buf == JG_BUF_IDS) {
      const float *emb = weights + op.b_off;   // (vocab, cin)
      const size_t e_items = (size_t)m->vocab * hp.cc_in * 4;
      std::vector<uint16_t> eh(e_items * 8, 0);
      for (int id = 0; id < m->vocab; ++id)
        for (int c = 0; c < op.cin; ++c) {
          const float v = emb[(size_t)id * op.cin + c];
          const float hi = f16_value(v);
          if (!(fabsf(v) <= 65000.f)) cfail("embedding value outside the f16 range");
          const int cc = c / 16, hh = (c % 16) / 8, j = c % 8;
          eh[(((size_t)id * hp.cc_in + cc) * 4 + 0 * 2 + hh) * 8 + j] = f16_bits(hi);
          eh[(((size_t)id * hp.cc_in + cc) * 4 + 1 * 2 + hh) * 8 + j] = f16_bits(v - hi);
        }
      JG_HIP(hipMalloc(reinterpret_cast<void **>(&hp.d_embh), e_items * 16));
      JG_HIP(hipMemcpy(hp.d_embh, eh.data(), e_items * 16, hipMemcpyHostToDevice));
    }
    // compact epilogue: fold acc un-scale, bias and batch-norm chains into per-channel affines
    {
      std::vector<float> tab;                       // [n_epi_rows][2][cwide]; uploaded as [half][n_epi_rows][2][128]
      std::vector<double> sc(cwide, (double)hp.acc_scale), sh(cwide, 0.0);
      bool pending = true;                          // an affine (the un-scale) is always pending first
      hp.n_hst = 0;
      hp.n_epi_rows = 0;
      auto flush = [&]() {
        if (!pending) return;
        HStageArg h{JG_HST_AFFINE, 0, 0.f, hp.n_epi_rows++};
        hp.hst[hp.n_hst++] = h;
        for (int n = 0; n < cwide; ++n) tab.push_back((float)sc[n]);
        for (int n = 0; n < cwide; ++n) tab.push_back((float)sh[n]);
        std::fill(sc.begin(), sc.end(), 1.0);
        std::fill(sh.begin(), sh.end(), 0.0);
        pending = false;
      };
      for (int q = 0; q < op.n_stages && conv_ok; ++q) {
        const jg_stage &st = op.stages[q];
        auto vecp = [&](int64_t off) { return weights + off; };
        if (st.kind == JG_ST_BIAS) {
          for (int n = 0; n < op.cout; ++n) sh[n] += (double)vecp(st.p0)[n];
          pending = true;
          continue;
        }
        if (st.kind == JG_ST_BN) {   // g*((x-mu)*is)+b on top of x = v*sc+sh
          for (int n = 0; n < op.cout; ++n) {
            const double mu = vecp(st.p0)[n], is = vecp(st.p1)[n], g = vecp(st.p2)[n], b = vecp(st.p3)[n];
            sc[n] = sc[n] * is * g;
            sh[n] = (sh[n] - mu) * is * g + b;
          }
          pending = true;
          continue;
        }
        flush();
        if (hp.n_hst >= JG_MAX_STAGES) { cfail("epilogue too long"); break; }
        HStageArg h{0, st.arg, st.f0, 0};
        switch (st.kind) {
          case JG_ST_DYT:
            h.kind = JG_HST_DYT;
            h.pad_ = hp.n_epi_rows++;
            for (int n = 0; n < cwide; ++n) tab.push_back(n < op.cout ? vecp(st.p2)[n] : 0.f);
            for (int n = 0; n < cwide; ++n) tab.push_back(n < op.cout ? vecp(st.p3)[n] : 0.f);
            break;
          case JG_ST_ADD: h.kind = JG_HST_ADD; hp.add_slot = st.arg; break;
          case JG_ST_ACT:
            h.kind = JG_HST_ACT;
            break;
          case JG_ST_NMD:
            h.kind = JG_HST_NMD;
            if (hp.nmd_slot < 0) hp.nmd_slot = st.arg;
            else if (hp.nmd_slot2 < 0) hp.nmd_slot2 = st.arg;
            else cfail("more than two NMD taps in one conv");
            break;
          case JG_ST_MASKMUL: h.kind = JG_HST_MASKMUL; break;
          default: cfail("epilogue stage not supported by the split-f16 kernel"); break;
        }
        hp.hst[hp.n_hst++] = h;
      }
      if (conv_ok) {
        if (pending && hp.n_hst >= JG_MAX_STAGES) cfail("epilogue too long");
        else flush();
      }
      if (hp.n_epi_rows > JG_EPI_ROWS) cfail("more norm stages than the split-f16 epilogue table holds");
      // match the stage list against the compiled pattern
      //   affine [nmd] [norm1] [add] [gelu] [nmd] [norm2] [gelu]
      {
        unsigned ep = 0;
        int q = 0;
        const int n = hp.n_hst;
        auto is = [&](int kind) { return q < n && hp.hst[q].kind == kind; };
        bool ok = is(JG_HST_AFFINE);
        if (ok) {
          ++q;
          if (is(JG_HST_NMD)) { ep |= JG_EP_NMD1; ++q; }
          if (is(JG_HST_AFFINE)) { ep |= JG_EP_NORM1_AFF; ++q; }
          else if (is(JG_HST_DYT)) { ep |= JG_EP_NORM1_DYT; hp.alpha1 = hp.hst[q].f0; hp.dytmask1 = hp.hst[q].arg; ++q; }
          if (is(JG_HST_ADD)) { ep |= JG_EP_ADD; ++q; }
          int gelu_kind = 0;   // all activation stages of a compiled pattern share one kind
          auto is_gelu = [&]() {
            if (!is(JG_HST_ACT)) return false;
            const int k = hp.hst[q].arg;
            if (k != JG_ACT_GELU_TANH && k != JG_ACT_GELU_ERF && k != JG_ACT_RELU) return false;
            if (gelu_kind != 0 && gelu_kind != k) return false;
            gelu_kind = k;
            return true;
          };
          if (is_gelu()) { ep |= JG_EP_ACT1; ++q; }
          if (is(JG_HST_NMD)) { ep |= JG_EP_NMD2; ++q; }
          if (is(JG_HST_AFFINE)) { ep |= JG_EP_NORM2_AFF; ++q; }
          else if (is(JG_HST_DYT)) { ep |= JG_EP_NORM2_DYT; hp.alpha2 = hp.hst[q].f0; hp.dytmask2 = hp.hst[q].arg; ++q; }
          if (is_gelu()) { ep |= JG_EP_ACT2; ++q; }
          ok = q == n;
          hp.act_kind = gelu_kind != 0 ? gelu_kind : JG_ACT_GELU_TANH;
        }
        if (ok && (ep & (JG_EP_NORM1_DYT | JG_EP_NORM2_DYT)) && (ep & (JG_EP_ACT1 | JG_EP_ACT2)) &&
            hp.act_kind != JG_ACT_GELU_TANH)
          ok = false;               // the DyT patterns are compiled for the tanh-GELU only
        hp.ep = ok ? ep : JG_EP_GENERIC;
        hp.ep_rt = 0;
        // a canonical stage list without an instantiation of its own (incl. two NMD taps in one conv): the run-time-flag
        // epilogue - tanh-GELU stage lists only (it carries every stage kind at once; the erf / ReLU forms beside them spill)
        const bool narrow_geo = op.in_buf != JG_BUF_IDS && (op.cout != 128 || op.stride != 1 || hp.as_k5);
        if (ok && hp.act_kind == JG_ACT_GELU_TANH && !(ep & JG_EP_ADD && op.in_buf == JG_BUF_IDS) &&
            (!jg_conv_f16_has_pattern(ep, op.in_buf == JG_BUF_IDS) || (narrow_geo && !jg_conv_f16_has_narrow_pattern(ep)) ||
             ((ep & JG_EP_NMD1) && (ep & JG_EP_NMD2)))) {
          hp.ep_rt = ep;
          hp.ep = JG_EP_RUNTIME;
        }
        if (hp.ep != JG_EP_RUNTIME && hp.nmd_slot2 >= 0) ok = false, hp.ep = JG_EP_GENERIC;   // two taps need the second accumulator
        // Only compiled stage patterns run on the split-f16 path: the interpreted epilogue was measured
        // 12x slower than the compiled ones (and 3x slower than the exact-f32 kernels), so anything else
        // stays on the exact-f32 path.
        if (conv_ok && !jg_conv_f16_has_pattern(hp.ep, op.in_buf == JG_BUF_IDS)) {
          cfail("a conv's stage list is not one of the compiled split-f16 epilogue patterns");
        }
        if (conv_ok && op.in_buf != JG_BUF_IDS && (op.cout != 128 || op.stride != 1 || hp.as_k5) && !jg_conv_f16_has_narrow_pattern(hp.ep))
          cfail("the stage list of a conv of other than 128 channels / stride 1 is not one of the patterns compiled for it");
        if (conv_ok && op.stride == 2 && ((hp.ep == JG_EP_RUNTIME ? hp.ep_rt : hp.ep) & (JG_EP_ADD | JG_EP_NMD1 | JG_EP_NMD2)))
          cfail("strided conv with a shortcut or an NMD tap in its epilogue");
      }
      if (conv_ok) {
        std::vector<float> th(tab.size());                 // [half][row][2][128]
        const int nr = hp.n_epi_rows;
        for (int hf = 0; hf < hp.n_half; ++hf)
          for (int r = 0; r < nr * 2; ++r)
            for (int n = 0; n < 128; ++n) th[((size_t)hf * nr * 2 + r) * 128 + n] = tab[(size_t)r * cwide + hf * 128 + n];
        JG_HIP(hipMalloc(reinterpret_cast<void **>(&hp.d_epi), th.size() * sizeof(float)));
        JG_HIP(hipMemcpy(hp.d_epi, th.data(), th.size() * sizeof(float), hipMemcpyHostToDevice));
      }
      // first layer on ids: the conv is a sum of k table rows T_t[id] = E[id] . W_t (f64 on the
      // host); the kernel's table variant then needs no matrix cores and no acc un-scale
      if (conv_ok && lut_ok) {
        const float *emb = weights + op.b_off;   // (vocab, cin)
        const int vr = m->vocab + 1;             // + the all-zero padding row
        std::vector<float> lut((size_t)2 * op.k * vr * 64, 0.f);
        for (int t = 0; t < op.k; ++t)
          for (int id = (op.in_mask == JG_BUF_IDS ? 1 : 0); id < m->vocab; ++id)   // id 0 is masked: zero row
            for (int n = 0; n < op.cout; ++n) {
              double acc = 0.0;
              for (int c = 0; c < op.cin; ++c)
                acc += (double)emb[(size_t)id * op.cin + c] * (double)w[((size_t)t * cin_pad + c) * cout_pad32 + n];
              lut[(((size_t)(n >> 6) * op.k + t) * vr + id) * 64 + (n & 63)] = (float)acc;
            }
        std::vector<float> tab_lut(tab);
        for (int n = 0; n < 128; ++n) tab_lut[n] = (float)((double)tab[n] / (double)hp.acc_scale);   // row 0 scale
        JG_HIP(hipMalloc(reinterpret_cast<void **>(&hp.d_lut), lut.size() * sizeof(float)));
        JG_HIP(hipMemcpy(hp.d_lut, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice));
        JG_HIP(hipMalloc(reinterpret_cast<void **>(&hp.d_epi_lut), tab_lut.size() * sizeof(float)));
        JG_HIP(hipMemcpy(hp.d_epi_lut, tab_lut.data(), tab_lut.size() * sizeof(float), hipMemcpyHostToDevice));
      }
      if (conv_ok && op.in_buf == JG_BUF_IDS && hp.d_lut == nullptr && (op.cout != 128 || !mfma_ok))
        cfail("first conv without the table variant is not a 128-channel 5- / 7- / 9-tap conv");
    }
    hp.f16_ok = conv_ok;
  }
  // Pass B - tensor formats.  Walk the program with the format of every activation slot (f32 rows or F16S items):
  // split-f16 convs read and write F16S (f32 when the next reader needs it, or no tensor at all when only a max pool
  // reads it), everything else works on f32; where a reader meets the other format, a layout conversion is queued in
  // front of it (run_chunk converts into a scratch tensor and swaps the slot's pointer).
  int n_ok = 0, n_conv = 0;
  for (size_t i = 0; i < m->ops.size(); ++i)
    if (m->ops[i].kind == JG_OP_CONV) { ++n_conv; n_ok += m->hprep[i].f16_ok ? 1 : 0; }
  if (n_ok == 0) {
    m->f16_eligible = false;
    m->f16_reason = first_reason.empty() ? "program has no convolution" : first_reason;
    return JG_OK;
  }
  m->f16_mixed = n_ok < n_conv;
  bool is_f32[JG_MAX_BUFS] = {};
  auto wants_f16s = [&](size_t j, int buf) {          // does op j read `buf` as an F16S tensor?
    const jg_op &o = m->ops[j];
    if (o.kind == JG_OP_MAXPOOL1D && o.in_buf == buf) return true;
    if (o.kind != JG_OP_CONV || !m->hprep[j].f16_ok) return false;
    if (o.in_buf == buf) return true;
    for (int q = 0; q < o.n_stages; ++q)
      if (o.stages[q].kind == JG_ST_ADD && o.stages[q].arg == buf) return true;
    return false;
  };
  auto reads = [&](size_t j, int buf) {
    const jg_op &o = m->ops[j];
    if ((o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE || o.kind == JG_OP_MAXPOOL1D || o.kind == JG_OP_FRAMESUM ||
         o.kind == JG_OP_POOL || o.kind == JG_OP_NMD_FINAL) && o.in_buf == buf)
      return o.kind != JG_OP_NMD_FINAL;               // (NMD_FINAL only takes the slot's shape)
    if (o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE)
      for (int q = 0; q < o.n_stages; ++q)
        if (o.stages[q].kind == JG_ST_ADD && o.stages[q].arg == buf) return true;
    return false;
  };
  auto writes = [&](size_t j, int buf) {
    const jg_op &o = m->ops[j];
    return (o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE || o.kind == JG_OP_MAXPOOL1D || o.kind == JG_OP_FRAMESUM ||
            o.kind == JG_OP_EMBED) &&
           o.out_buf == buf;
  };
  bool cvt_overflow = false;
  auto need = [&](size_t i, int buf, bool want_f32) {   // queue a conversion in front of op i if the slot is in the other format
    if (buf < 0 || is_f32[buf] == want_f32) return;
    ConvHPrep &hp = m->hprep[i];
    if (hp.n_cvt >= 3) {           // table full: the op would read a tensor in the wrong layout - give the fast path up instead
      cvt_overflow = true;
      return;
    }
    hp.cvt_slot[hp.n_cvt] = buf;
    hp.cvt_to_f32[hp.n_cvt] = want_f32;
    ++hp.n_cvt;
    is_f32[buf] = want_f32;
    m->needs_cvt = true;
  };
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    ConvHPrep &hp = m->hprep[i];
    switch (op.kind) {
      case JG_OP_CONV: {
        const bool f16 = hp.f16_ok;
        need(i, op.in_buf, !f16);
        for (int q = 0; q < op.n_stages; ++q)
          if (op.stages[q].kind == JG_ST_ADD) need(i, op.stages[q].arg, !f16);
        if (!f16) { is_f32[op.out_buf] = true; break; }
        // output format: what the first reader wants
        bool first_f16s = false, any_reader = false;
        for (size_t j = i + 1; j < m->ops.size(); ++j) {
          if (reads(j, op.out_buf)) { first_f16s = wants_f16s(j, op.out_buf); any_reader = true; break; }
          if (writes(j, op.out_buf)) break;
        }
        hp.out_f16s = any_reader && first_f16s;
        is_f32[op.out_buf] = !hp.out_f16s;
        if (!hp.out_f16s && jg_exp_env("JG_NO_POOL_FUSE") == nullptr) {
          // the only reader of the f32 output is a masked global max pool over the conv's own output mask:
          // reduce in the epilogue instead of storing 4 B per element and reading it back
          int readers = 0, pool_idx = -1;
          for (size_t j = i + 1; j < m->ops.size(); ++j) {
            const jg_op &o = m->ops[j];
            if (reads(j, op.out_buf)) {
              ++readers;
              if (o.kind == JG_OP_POOL && o.arg == JG_POOL_MAX && o.in_mask == op.out_mask) pool_idx = (int)j;
              else pool_idx = -2;
            }
            if (writes(j, op.out_buf)) break;
          }
          if (readers == 1 && pool_idx >= 0) {
            hp.pool_op = pool_idx;
            m->pool_fused_by[(size_t)pool_idx] = (int)i;
          }
        }
      } break;
      case JG_OP_EMBED:
        is_f32[op.out_buf] = true;                      // the lookup writes f32 rows: the first conv converts if it wants F16S
        break;
      case JG_OP_ELTWISE:
        need(i, op.in_buf, true);
        for (int q = 0; q < op.n_stages; ++q)
          if (op.stages[q].kind == JG_ST_ADD) need(i, op.stages[q].arg, true);
        is_f32[op.out_buf] = true;
        break;
      case JG_OP_MAXPOOL1D:
        hp.pool_f16s = op.in_buf >= 0 && !is_f32[op.in_buf];
        is_f32[op.out_buf] = !hp.pool_f16s;
        break;
      case JG_OP_FRAMESUM:
        need(i, op.in_buf, true);
        is_f32[op.out_buf] = true;
        break;
      case JG_OP_POOL:
        if (m->pool_fused_by[i] < 0) need(i, op.in_buf, true);
        break;
      default: break;
    }
  }
  if (cvt_overflow) {
    m->f16_eligible = false;
    m->f16_mixed = false;
    m->f16_reason = "an op needs more than 3 layout conversions in front of it (mixed split-f16 / f32 program)";
    for (ConvHPrep &hp : m->hprep) { hp.f16_ok = false; hp.n_cvt = 0; hp.pool_op = -1; }
    std::fill(m->pool_fused_by.begin(), m->pool_fused_by.end(), -1);
  }
  return JG_OK;
}

// Pass C - stride-2 convs without dropped work.  A strided residual block (layers.py:1882-1915: conv1 of 5 taps and the
// 1x1 bypass, both stride 2, both reading the block's input) is evaluated by the split-f16 kernel at stride 1 with every
// second output dropped.  When ALL readers of a tensor are such convs and a split-f16 conv writes it, the writer stores it
// phase-split instead (ConvHArgs::psplit: even positions in the first cin channels, odd ones in the next cin, (L + 1) / 2
// positions, mask-multiplied) and the readers run at stride 1: the 5-tap conv as a 3-tap conv over 2 x cin channels -
//   y[m] = sum_t w_t x[2m + t - pl]  =  sum over q = t - pl of  w_t . phase(q mod 2)[m + floor(q / 2)]
// (pl = TF's SAME left pad: 2 for an odd input length, 1 for an even one - two weight arrangements) - the 1x1 conv on the
// even phase alone.  Same sums in another order of the taps: results agree to f32 rounding, no output is computed twice.
static int plan_phase_split(jg_model *m, const float *weights) {
  if (!m->f16_eligible) return JG_OK;
  static const bool off = jg_exp_env("JG_NO_PSPLIT") != nullptr;
  if (off) return JG_OK;
  const size_t n = m->ops.size();
  auto reads_buf = [&](size_t j, int buf) {
    const jg_op &o = m->ops[j];
    if ((o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE || o.kind == JG_OP_MAXPOOL1D || o.kind == JG_OP_FRAMESUM ||
         o.kind == JG_OP_POOL || o.kind == JG_OP_NMD_FINAL) && o.in_buf == buf)
      return true;
    if (o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE)
      for (int q = 0; q < o.n_stages; ++q)
        if (o.stages[q].kind == JG_ST_ADD && o.stages[q].arg == buf) return true;
    return false;
  };
  auto writes_buf = [&](size_t j, int buf) {
    const jg_op &o = m->ops[j];
    return (o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE || o.kind == JG_OP_MAXPOOL1D || o.kind == JG_OP_FRAMESUM ||
            o.kind == JG_OP_EMBED) &&
           o.out_buf == buf;
  };
  for (size_t p = 0; p < n; ++p) {
    const jg_op &po = m->ops[p];
    ConvHPrep &pp = m->hprep[p];
    if (po.kind != JG_OP_CONV || !pp.f16_ok || !pp.out_f16s || pp.pool_op >= 0 || po.stride != 1 || po.out_buf < 0 ||
        po.cout % 16 != 0 || (int)p == m->tab_conv || po.in_buf == JG_BUF_IDS ||
        !jg_conv_f16_has_narrow_pattern(pp.ep))           // (the store is built into the run-time-geometry tiles)
      continue;
    std::vector<size_t> readers;
    bool ok = true, mask_rewritten = false;
    for (size_t j = p + 1; j < n && ok; ++j) {
      const jg_op &o = m->ops[j];
      if (reads_buf(j, po.out_buf)) {
        const ConvHPrep &hr = m->hprep[j];
        const bool conv_reader = o.kind == JG_OP_CONV && o.in_buf == po.out_buf && hr.f16_ok && o.stride == 2 &&
                                 o.padding == JG_PAD_SAME && o.cin == po.cout && o.in_mask == po.out_mask &&
                                 ((o.k == 5 && o.dilation == 1) || o.k == 1);
        bool adds_it = false;
        for (int q = 0; q < o.n_stages; ++q) adds_it |= o.stages[q].kind == JG_ST_ADD && o.stages[q].arg == po.out_buf;
        bool cvt_here = false;
        for (int q = 0; q < hr.n_cvt; ++q) cvt_here |= hr.cvt_slot[q] == po.out_buf;
        // (the mask the writer multiplies by must still be the reader's input mask when it runs)
        if (!conv_reader || adds_it || cvt_here || mask_rewritten) ok = false;
        else readers.push_back(j);
      }
      if (o.kind == JG_OP_MASK && po.out_mask >= 0 && o.out_mask == po.out_mask) mask_rewritten = true;
      if (writes_buf(j, po.out_buf)) break;
    }
    if (!ok || readers.empty()) continue;
    // weights of the 5-tap readers, re-arranged for both parities of the input length
    bool built = true;
    for (size_t j : readers) {
      const jg_op &o = m->ops[j];
      ConvHPrep &hr = m->hprep[j];
      if (o.k != 5) continue;
      const int cin = o.cin, cin2 = 2 * cin, cout_pad32 = (o.cout + 31) / 32 * 32, cin_pad = (cin + 1) & ~1;
      const float *w = weights + o.w_off;                  // (5, cin_pad, cout_pad32)
      const float wscale = 1.0f / hr.acc_scale;            // the conv's own power-of-two scale (the epilogue table undoes it)
      const int kk = 5, kc_total = cin2 / 8, cout_pad = 128;
      const size_t half_items = (size_t)2 * kk * kc_total * cout_pad, n_items = half_items * hr.n_half;
      hr.ps_half_items = (int64_t)half_items;
      for (int par = 0; par < 2; ++par) {                   // par = input length & 1
        const int pl = par ? 2 : 1;
        std::vector<uint16_t> wh(n_items * 8, 0);
        for (int t = 0; t < 5; ++t) {
          const int q = t - pl, ph = ((q % 2) + 2) % 2, off3 = (q - ph) / 2 + 1;     // phase, tap of the 3-tap conv (0 .. 2)
          const int tk = 1 + off3;                                                   // ... in the middle of the kernel's five
          for (int c = 0; c < cin; ++c)
            for (int nn = 0; nn < o.cout; ++nn) {
              const float v = w[((size_t)t * cin_pad + c) * cout_pad32 + nn] * wscale;
              const float hi = f16_value(v);
              const int c2 = ph * cin + c;
              const size_t base = (size_t)(nn / 128) * half_items;
              const size_t item = base + (((size_t)0 * kk + tk) * kc_total + c2 / 8) * cout_pad + nn % 128;
              const size_t item_lo = base + (((size_t)1 * kk + tk) * kc_total + c2 / 8) * cout_pad + nn % 128;
              wh[item * 8 + c2 % 8] = f16_bits(hi);
              wh[item_lo * 8 + c2 % 8] = f16_bits(v - hi);
            }
        }
        if (hipMalloc(reinterpret_cast<void **>(&hr.d_wh_ps[par]), n_items * 16) != hipSuccess ||
            hipMemcpy(hr.d_wh_ps[par], wh.data(), n_items * 16, hipMemcpyHostToDevice) != hipSuccess) {
          built = false;
          break;
        }
      }
      if (!built) break;
    }
    if (!built) {
      (void)hipGetLastError();
      for (size_t j : readers)
        for (int par = 0; par < 2; ++par)
          if (m->hprep[j].d_wh_ps[par]) { (void)hipFree(m->hprep[j].d_wh_ps[par]); m->hprep[j].d_wh_ps[par] = nullptr; }
      continue;
    }
    pp.ps_store = true;
    for (size_t j : readers) m->hprep[j].ps_read = m->ops[j].k == 5 ? 1 : 2;
  }
  return JG_OK;
}

// Pass D - whole narrow residual blocks as one launch (jg_resblock.hip).  conv1 [bias, norm, GELU] and conv2 [bias, norm,
// + block input, GELU] of a stride-1 block without bypass (layers.py:1882-1915), 32 channels, five taps, one dilation: the
// intermediate tensor lives in LDS only.  conv1's op is skipped at run time, conv2's launch computes both; the mask ops
// between them run as before (the kernel reads conv2's input mask from their output).
static int plan_resblocks(jg_model *m, const float *weights) {
  if (!m->f16_eligible) return JG_OK;
  static const bool off = jg_exp_env("JG_NO_RESBLOCK") != nullptr;
  if (off) return JG_OK;
  const size_t n = m->ops.size();
  auto reads_buf = [&](size_t j, int buf) {
    const jg_op &o = m->ops[j];
    if ((o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE || o.kind == JG_OP_MAXPOOL1D || o.kind == JG_OP_FRAMESUM ||
         o.kind == JG_OP_POOL || o.kind == JG_OP_NMD_FINAL) && o.in_buf == buf)
      return true;
    if (o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE)
      for (int q = 0; q < o.n_stages; ++q)
        if (o.stages[q].kind == JG_ST_ADD && o.stages[q].arg == buf) return true;
    return false;
  };
  auto writes_buf = [&](size_t j, int buf) {
    const jg_op &o = m->ops[j];
    return (o.kind == JG_OP_CONV || o.kind == JG_OP_ELTWISE || o.kind == JG_OP_MAXPOOL1D || o.kind == JG_OP_FRAMESUM ||
            o.kind == JG_OP_EMBED) &&
           o.out_buf == buf;
  };
  // bias / batch-norm stages in front of the first other stage, folded with the weights' un-scale (as prepare_f16 does)
  auto fold = [&](const jg_op &op, float acc_scale, float *sc, float *sh) {
    std::vector<double> s((size_t)op.cout, (double)acc_scale), t((size_t)op.cout, 0.0);
    for (int q = 0; q < op.n_stages; ++q) {
      const jg_stage &st = op.stages[q];
      if (st.kind == JG_ST_BIAS) {
        for (int c = 0; c < op.cout; ++c) t[(size_t)c] += (double)weights[st.p0 + c];
      } else if (st.kind == JG_ST_BN) {
        for (int c = 0; c < op.cout; ++c) {
          const double mu = weights[st.p0 + c], is = weights[st.p1 + c], g = weights[st.p2 + c], b = weights[st.p3 + c];
          s[(size_t)c] = s[(size_t)c] * is * g;
          t[(size_t)c] = (t[(size_t)c] - mu) * is * g + b;
        }
      } else {
        break;
      }
    }
    for (int c = 0; c < op.cout; ++c) { sc[c] = (float)s[(size_t)c]; sh[c] = (float)t[(size_t)c]; }
  };
  for (size_t ia = 0; ia < n; ++ia) {
    const jg_op &A = m->ops[ia];
    ConvHPrep &ha = m->hprep[ia];
    if (A.kind != JG_OP_CONV || !ha.f16_ok || ha.rb_second >= 0 || ha.rb_first >= 0 || A.in_buf < 0 || A.stride != 1 ||
        A.padding != JG_PAD_SAME || A.cin != A.cout || !(jg_resblock_supports(A.cout, A.k, A.dilation) || jg_resblock64_supports(A.cout, A.k, A.dilation)) || !ha.out_f16s ||
        ha.ep != JG_EP_ACT1 || ha.act_kind != JG_ACT_GELU_TANH || ha.pool_op >= 0 || ha.ps_store || ha.ps_read != 0 ||
        ha.n_cvt != 0 || (int)ia == m->tab_conv)
      continue;
    // the only reader of conv1's output: conv2, before anything overwrites it (mask ops may sit in between)
    size_t ib = n;
    bool ok = true;
    for (size_t j = ia + 1; j < n; ++j) {
      if (reads_buf(j, A.out_buf)) {
        if (ib == n && m->ops[j].kind == JG_OP_CONV && m->ops[j].in_buf == A.out_buf) ib = j;
        else ok = false;
      }
      if (writes_buf(j, A.out_buf) && j != ib) break;
      if (writes_buf(j, A.out_buf) && j == ib) { ok = false; break; }       // (in place: not a residual block)
    }
    if (!ok || ib == n) continue;
    for (size_t j = ia + 1; j < ib && ok; ++j)                               // nothing but mask ops between the two
      ok = m->ops[j].kind == JG_OP_MASK && m->ops[j].out_mask != A.in_mask && m->ops[j].out_mask != A.out_mask;
    const jg_op &B = m->ops[ib];
    ConvHPrep &hb = m->hprep[ib];
    if (!ok || !hb.f16_ok || B.stride != 1 || B.padding != JG_PAD_SAME || B.k != A.k || B.dilation != A.dilation ||
        B.cin != A.cout || B.cout != A.cout || B.in_mask != A.out_mask || !hb.out_f16s ||
        hb.ep != (JG_EP_ADD | JG_EP_ACT1) || hb.act_kind != JG_ACT_GELU_TANH || hb.add_slot != A.in_buf ||
        B.out_buf == A.in_buf || B.out_buf == A.out_buf || hb.pool_op >= 0 || hb.ps_read != 0 || hb.n_cvt != 0 ||
        hb.nmd_slot >= 0 || ha.nmd_slot >= 0)
      continue;
    // weight fragments [conv][tap][chunk][plane][32-channel output tile][lane][8 halfs]: lane = (cin group of 8) x (output
    // channel of the tile)
    const int C = A.cout, K = A.k, cc_n = C / 16, ct_n = C / 32;
    std::vector<uint16_t> frag((size_t)2 * K * cc_n * 2 * ct_n * 64 * 8, 0);
    std::vector<float> epi((size_t)4 * C, 0.f);
    bool range_ok = true;
    for (int c = 0; c < 2; ++c) {
      const jg_op &op = c == 0 ? A : B;
      const ConvHPrep &hp = c == 0 ? ha : hb;
      const float *w = weights + op.w_off;                 // (k, cin_pad, cout_pad32)
      const int cin_pad = (op.cin + 1) & ~1, cout_pad32 = (op.cout + 31) / 32 * 32;
      const float wscale = 1.0f / hp.acc_scale;
      for (int t = 0; t < K; ++t)
        for (int cc = 0; cc < cc_n; ++cc)
          for (int ct = 0; ct < ct_n; ++ct)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int co = ct * 32 + (lane & 31), ci = cc * 16 + (lane >> 5) * 8 + j;
                const float v = w[((size_t)t * cin_pad + ci) * cout_pad32 + co] * wscale;
                const float hi = f16_value(v);
                if (!(fabsf(v) <= 65000.f)) range_ok = false;
                const size_t base = ((((size_t)c * K + t) * cc_n + cc) * 2) * ct_n * 64 * 8;
                frag[base + ((size_t)ct * 64 + lane) * 8 + j] = f16_bits(hi);
                frag[base + ((size_t)(ct_n + ct) * 64 + lane) * 8 + j] = f16_bits(v - hi);
              }
      fold(op, hp.acc_scale, epi.data() + (size_t)c * 2 * C, epi.data() + (size_t)c * 2 * C + C);
    }
    if (!range_ok) continue;
    if (hipMalloc(reinterpret_cast<void **>(&hb.d_rb_wfrag), frag.size() * 2) != hipSuccess ||
        hipMemcpy(hb.d_rb_wfrag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&hb.d_rb_epi), epi.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(hb.d_rb_epi, epi.data(), epi.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      if (hb.d_rb_wfrag) { (void)hipFree(hb.d_rb_wfrag); hb.d_rb_wfrag = nullptr; }
      if (hb.d_rb_epi) { (void)hipFree(hb.d_rb_epi); hb.d_rb_epi = nullptr; }
      continue;
    }
    ha.rb_second = (int)ib;
    hb.rb_first = (int)ia;
  }
  return JG_OK;
}

extern "C" int jg_model_set_precision(jg_model *m, int mode) {
  JG_REQUIRE(m != nullptr && (mode == 0 || mode == 1), JG_ERR_INVALID, "jg_model_set_precision: bad args");
  if (mode == 1 && !m->f16_eligible) {
    jg_set_error("split-f16 path unavailable for this model: %s", m->f16_reason.c_str());
    return JG_ERR_UNSUPPORTED;
  }
  if (mode != m->precision) {
    JG_HIP(hipSetDevice(m->e->dev));
    JG_HIP(hipStreamSynchronize(m->e->stream));
    m->precision = mode;
  }
  return JG_OK;
}

extern "C" int jg_model_get_precision(const jg_model *m) { return m ? m->precision : -1; }

// One line per convolution: geometry, the kernel it runs on in mode 1 and, for the exact-f32 ones, why.
extern "C" int jg_model_describe(const jg_model *m, char *buf, int64_t cap) {
  JG_REQUIRE(m != nullptr && buf != nullptr && cap > 0, JG_ERR_INVALID, "jg_model_describe: bad arguments");
  std::string out;
  char line[512];
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    if (op.kind != JG_OP_CONV) continue;
    const ConvHPrep &hp = m->hprep[i];
    const char *where = m->small != nullptr ? "fused small-window kernel"
                        : (m->f16_eligible && hp.f16_ok)
                            ? (hp.d_lut != nullptr ? "split-f16 (table lookup)"
                               : hp.rb_second >= 0 ? "split-f16 (fused residual block: computed by the block's second conv)"
                               : hp.rb_first >= 0 ? (hp.ps_store ? "split-f16 (fused residual block, phase-split store)" : "split-f16 (fused residual block)")
                               : hp.ps_read == 1 ? "split-f16 (stride 2 as a 3-tap conv over the two phases of its phase-split input)"
                               : hp.ps_read == 2 ? "split-f16 (stride 2 on the even phase of its phase-split input)"
                               : hp.ps_store ? (hp.cw != 128 ? "split-f16 (narrow tile, phase-split store)" : "split-f16 (phase-split store)")
                               : hp.as_k5 ? "split-f16 (tap range of the 5-tap kernel)"
                               : hp.cw != 128 ? "split-f16 (narrow tile)" : "split-f16")
                            : "exact-f32";
    static const char *const st_name[] = {"?", "bias", "bn", "dyt", "add", "act", "nmd", "maskmul", "ln"};
    std::string stages;
    for (int q = 0; q < op.n_stages; ++q) {
      const int kd = op.stages[q].kind;
      stages += (q ? " " : "");
      stages += (kd >= 1 && kd <= 8) ? st_name[kd] : "?";
    }
    snprintf(line, sizeof(line), "op %zu: conv k=%d cin=%d cout=%d stride=%d dilation=%d [%s] -> %s%s%s\n", i, op.k, op.cin,
             op.cout, op.stride, op.dilation, stages.c_str(), where,
             (!hp.f16_ok && !hp.why_f32.empty() && m->small == nullptr) ? ": " : "",
             (!hp.f16_ok && m->small == nullptr) ? hp.why_f32.c_str() : "");
    out += line;
  }
  const size_t n = std::min(out.size(), (size_t)cap - 1);
  memcpy(buf, out.data(), n);
  buf[n] = 0;
  return JG_OK;
}

extern "C" int64_t jg_model_get_stat(const jg_model *m, int key) {
  if (m == nullptr) return -1;
  int64_t n_conv = 0, n_f16 = 0, n_cvt = 0;
  for (size_t i = 0; i < m->ops.size(); ++i) {
    n_cvt += m->hprep[i].n_cvt;
    if (m->ops[i].kind != JG_OP_CONV) continue;
    ++n_conv;
    n_f16 += (m->f16_eligible && m->hprep[i].f16_ok) ? 1 : 0;
  }
  switch (key) {
    case JG_MSTAT_CONVS: return n_conv;
    case JG_MSTAT_CONVS_F16X3: return n_f16;
    case JG_MSTAT_LAYOUT_CONVERSIONS: return m->f16_eligible ? n_cvt : 0;
    case JG_MSTAT_SMALL_FUSED: return m->small != nullptr ? 1 : 0;
    default: return -1;
  }
}

extern "C" int jg_model_destroy(jg_model *m);
extern "C" int jg_model_create(jg_engine *e, const jg_op *ops, int n_ops, const float *weights,
                               int64_t n_weights, int32_t vocab, jg_model **out) {
  JG_REQUIRE(e != nullptr && ops != nullptr && n_ops > 0 && weights != nullptr && n_weights > 0 &&
                 out != nullptr,
             JG_ERR_INVALID, "jg_model_create: bad arguments");
  JG_HIP(hipSetDevice(e->dev));
  jg_model *m = new jg_model();
  m->e = e;
  m->ops.assign(ops, ops + n_ops);
  m->n_w = n_weights;
  m->vocab = vocab;
  int rc = validate_program(m);
  if (rc != JG_OK) { delete m; return rc; }
  if (m->ops[0].kind == JG_OP_EMBED && m->vocab > 256) m->id_bytes = 2;
  if (m->ops.back().kind == JG_OP_STRANDS) {
    m->strands = m->ops.back().k;
    m->id_frames = 1;
    m->merge_kind = m->ops.back().arg;
  }
  hipError_t err = hipMalloc(&m->d_w, (size_t)n_weights * sizeof(float));
  if (err != hipSuccess) {
    jg_set_error("jg_model_create: weights hipMalloc -> %s", hipGetErrorString(err));
    delete m;
    return JG_ERR_NOMEM;
  }
  JG_HIP(hipMemcpy(m->d_w, weights, (size_t)n_weights * sizeof(float), hipMemcpyHostToDevice));
  JG_HIP(hipMalloc(&m->d_lut, 80));
  JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->d_overflow), sizeof(int)));
  JG_HIP(hipMemset(m->d_overflow, 0, sizeof(int)));
  rc = prepare_tab(m, weights);
  if (rc != JG_OK) { jg_model_destroy(m); return rc; }
  rc = prepare_f16(m, weights);
  if (rc != JG_OK) { jg_model_destroy(m); return rc; }
  rc = plan_phase_split(m, weights);
  if (rc != JG_OK) { jg_model_destroy(m); return rc; }
  rc = plan_resblocks(m, weights);
  if (rc != JG_OK) { jg_model_destroy(m); return rc; }
  rc = prepare_f32(m, weights);
  if (rc != JG_OK) { jg_model_destroy(m); return rc; }
  // the 32-channel small-window family has a fused kernel of its own (same split-f16 arithmetic): where the program
  // matches it, it takes precedence over the layer-by-layer placement above (whose narrow-conv kernels would run the
  // same model several times slower)
  rc = prepare_small(m, weights);
  if (rc != JG_OK) { jg_model_destroy(m); return rc; }
  if (m->small != nullptr) m->f16_eligible = true;
  m->precision = m->f16_eligible ? 1 : 0;
  *out = m;
  return JG_OK;
}

static void free_workspace(jg_model *m) {
  for (int i = 0; i < JG_MAX_BUFS; ++i) {
    if (m->act[i]) (void)hipFree(m->act[i]);
    if (m->msk[i]) (void)hipFree(m->msk[i]);
    if (m->nmd_part[i]) (void)hipFree(m->nmd_part[i]);
    m->act[i] = nullptr; m->msk[i] = nullptr; m->nmd_part[i] = nullptr;
    m->act_cap[i] = m->msk_cap[i] = m->nmd_cap[i] = 0;
  }
  for (int i = 0; i < JG_MAX_VECS; ++i) {
    if (m->vec[i]) (void)hipFree(m->vec[i]);
    m->vec[i] = nullptr;
    m->vec_cap[i] = 0;
  }
  if (m->cvt_scratch) (void)hipFree(m->cvt_scratch);
  m->cvt_scratch = nullptr;
  m->cvt_cap = 0;
}

extern "C" int jg_model_destroy(jg_model *m) {
  if (m == nullptr) return JG_OK;
  (void)hipSetDevice(m->e->dev);
  (void)hipStreamSynchronize(m->e->stream);
  free_workspace(m);
  free_small(m);
  for (int i = 0; i < JG_MAX_VECS; ++i)
    if (m->merged[i]) (void)hipFree(m->merged[i]);
  if (m->tab_wfrag) (void)hipFree(m->tab_wfrag);
  if (m->tab_bias512) (void)hipFree(m->tab_bias512);
  if (m->tab_table) (void)hipFree(m->tab_table);
  if (m->tab_bias) (void)hipFree(m->tab_bias);
  if (m->d_w) (void)hipFree(m->d_w);
  if (m->d_ids) (void)hipFree(m->d_ids);
  if (m->d_counts) (void)hipFree(m->d_counts);
  if (m->d_win) (void)hipFree(m->d_win);
  if (m->d_bases_buf) (void)hipFree(m->d_bases_buf);
  if (m->d_lut) (void)hipFree(m->d_lut);
  if (m->d_overflow) (void)hipFree(m->d_overflow);
  if (m->pool_part) (void)hipFree(m->pool_part);
  for (auto &hp : m->hprep) {
    if (hp.d_wh) (void)hipFree(hp.d_wh);
    for (int par = 0; par < 2; ++par) if (hp.d_wh_ps[par]) (void)hipFree(hp.d_wh_ps[par]);
    if (hp.d_rb_wfrag) (void)hipFree(hp.d_rb_wfrag);
    if (hp.d_rb_epi) (void)hipFree(hp.d_rb_epi);
    if (hp.d_embh) (void)hipFree(hp.d_embh);
    if (hp.d_epi) (void)hipFree(hp.d_epi);
    if (hp.d_w8) (void)hipFree(hp.d_w8);
    if (hp.d_lut) (void)hipFree(hp.d_lut);
    if (hp.d_epi_lut) (void)hipFree(hp.d_epi_lut);
  }
  delete m;
  return JG_OK;
}

// Workspace for `chunk` windows of `l` codons per frame.  Buffers are kept as long as they are large enough
// (the short-contig pass calls with a different l for every batch: commands/predict.py:236-245), and grow to the
// largest request seen.
static int ensure_workspace(jg_model *m, int64_t chunk, int l) {
  int64_t nmd_elems[JG_MAX_BUFS];
  int rc = plan_shapes(m, l, m->act_elems, m->msk_elems, nmd_elems, m->vec_w, nullptr);
  if (rc != JG_OK) return rc;
  bool fits = true;
  for (int i = 0; i < JG_MAX_BUFS; ++i) {
    m->nmd_part_elems[i] = nmd_elems[i];
    fits &= chunk * m->act_elems[i] <= m->act_cap[i] && chunk * m->msk_elems[i] <= m->msk_cap[i] &&
            chunk * nmd_elems[i] <= m->nmd_cap[i];
  }
  for (int i = 0; i < JG_MAX_VECS; ++i) fits &= chunk * m->vec_w[i] <= m->vec_cap[i];
  // programs with layout conversions swap a slot's tensor with the scratch tensor: every activation slot and the
  // scratch then need the LARGEST slot's size (a smaller buffer would otherwise wander into a larger slot - found by
  // the architecture fuzz on a net with two strided blocks)
  int64_t cvt_need = 0;
  if (m->needs_cvt) {
    for (int i = 0; i < JG_MAX_BUFS; ++i) cvt_need = std::max(cvt_need, chunk * m->act_elems[i]);
    for (int i = 0; i < JG_MAX_BUFS; ++i)
      if (m->act_elems[i] > 0) fits &= cvt_need <= m->act_cap[i];
  }
  fits &= cvt_need <= m->cvt_cap;
  if (fits) return JG_OK;
  JG_HIP(hipStreamSynchronize(m->e->stream));
  int64_t want_act[JG_MAX_BUFS], want_msk[JG_MAX_BUFS], want_nmd[JG_MAX_BUFS], want_vec[JG_MAX_VECS];
  for (int i = 0; i < JG_MAX_BUFS; ++i) {
    want_act[i] = std::max(m->act_cap[i], chunk * m->act_elems[i]);
    if (m->needs_cvt && m->act_elems[i] > 0) want_act[i] = std::max(want_act[i], cvt_need);
    want_msk[i] = std::max(m->msk_cap[i], chunk * m->msk_elems[i]);
    want_nmd[i] = std::max(m->nmd_cap[i], chunk * nmd_elems[i]);
  }
  for (int i = 0; i < JG_MAX_VECS; ++i) want_vec[i] = std::max(m->vec_cap[i], chunk * (int64_t)m->vec_w[i]);
  const int64_t want_cvt = std::max(m->cvt_cap, cvt_need);
  free_workspace(m);
  if (want_cvt > 0) {
    JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->cvt_scratch), (size_t)want_cvt * sizeof(float)));
    m->cvt_cap = want_cvt;
  }
  for (int i = 0; i < JG_MAX_BUFS; ++i) {
    if (want_act[i] > 0) JG_HIP(hipMalloc(&m->act[i], (size_t)want_act[i] * sizeof(float)));
    if (want_msk[i] > 0) JG_HIP(hipMalloc(&m->msk[i], (size_t)want_msk[i]));
    if (want_nmd[i] > 0) JG_HIP(hipMalloc(&m->nmd_part[i], (size_t)want_nmd[i] * sizeof(float)));
    m->act_cap[i] = want_act[i]; m->msk_cap[i] = want_msk[i]; m->nmd_cap[i] = want_nmd[i];
  }
  for (int i = 0; i < JG_MAX_VECS; ++i)
    if (want_vec[i] > 0) {
      JG_HIP(hipMalloc(&m->vec[i], (size_t)want_vec[i] * sizeof(float)));
      JG_HIP(hipMemsetAsync(m->vec[i], 0, (size_t)want_vec[i] * sizeof(float), m->e->stream));
      m->vec_cap[i] = want_vec[i];
    }
  return JG_OK;
}

static void resolve_stages(const jg_model *m, const jg_op &op, StageArg *dst, int *n) {
  *n = op.n_stages;
  for (int s = 0; s < op.n_stages; ++s) {
    const jg_stage &st = op.stages[s];
    StageArg &g = dst[s];
    g.kind = st.kind;
    g.arg = st.arg;
    g.f0 = st.f0;
    g.pad_ = 0;
    auto wp = [&](int64_t off) -> const float * { return off >= 0 ? m->d_w + off : nullptr; };
    g.p0 = wp(st.p0); g.p1 = wp(st.p1); g.p2 = wp(st.p2); g.p3 = wp(st.p3);
    if (st.kind == JG_ST_ADD) g.p0 = m->act[st.arg];
    if (st.kind == JG_ST_NMD) g.p0 = m->nmd_part[st.arg];
  }
}

template <typename T>
static int grow(T **p, int64_t *cap, int64_t need_bytes) {
  if (need_bytes <= *cap) return JG_OK;
  if (*p) JG_HIP(hipFree(*p));
  *p = nullptr;
  JG_HIP(hipMalloc(reinterpret_cast<void **>(p), (size_t)need_bytes));
  *cap = need_bytes;
  return JG_OK;
}

// Run the op program over `nw` windows whose ids (nw, 6, l) are on the device (a two-strand model: nw = strand rows,
// ids (nw, 1, l)).
static int run_chunk(jg_model *m, const uint8_t *d_ids, int nw, int l, hipStream_t s) {
  jg_engine *e = m->e;
  Shape sh[JG_MAX_BUFS];
  int mlen[JG_MAX_BUFS] = {};
  // the 32-channel family: one fused kernel from ids to pooled sums when the rows fit its 160 positions; longer
  // rows of such a model run layer by layer
  int small_L0 = 0, small_pad0 = 0;
  bool small = false;
  if (m->precision == 1 && m->small != nullptr) {
    const jg_op &c0 = m->ops[0].kind == JG_OP_CONV ? m->ops[0] : m->ops[1];
    conv_geometry(l, c0.k, 1, 1, c0.padding, &small_L0, &small_pad0);
    small = small_L0 >= 1 && small_L0 <= jg_small_max_positions() && l <= 192;
  }
  // arithmetic of the per-layer path (with `small` the convs in front of the pool are skipped; rows too long for the
  // fused kernel run layer by layer on the narrow split-f16 kernels - 3-tap convs as tap-masked 5-tap ones)
  const int prec = m->precision;
  const bool tab = tab_usable(m, l);
  if (small) {
    JgSmallNet *sn = m->small;
    const int64_t rows = (int64_t)nw * 6;
    int rc = grow(&sn->d_part, &sn->part_cap, rows * sn->n_slots * JG_SMALL_PARTW * (int64_t)sizeof(float));
    if (rc != JG_OK) return rc;
    JgSmallArgs a;
    memset(&a, 0, sizeof(a));
    a.ids = d_ids; a.lut = sn->d_lut; a.wfrag = sn->d_wfrag; a.epi = sn->d_epi; a.part = sn->d_part;
    a.overflow = m->d_overflow;
    a.rows = rows; a.L = l; a.L0 = small_L0; a.pad0 = small_pad0; a.vocab = m->vocab;
    a.use_mask = sn->use_mask; a.pool_kind = sn->pool_kind; a.n_slots = sn->n_slots;
    { const char *ev = jg_exp_env("JG_SMALL_DBG"); a.dbg = ev ? atoi(ev) : 0; }
    for (int q = 0; q < JG_SMALL_MAX_LAYERS; ++q) a.layer[q] = sn->layer[q];
    ProfEvent pe;
    if (e->profile) {
      if ((rc = prof_event(e, &pe.a)) != JG_OK || (rc = prof_event(e, &pe.b)) != JG_OK) return rc;
      pe.flops = (sn->flops_per_pos0 + sn->flops_per_pos) * (double)rows * small_L0;
      pe.cls = JG_PROF_FUSED_SMALL;
      JG_HIP(hipEventRecord(pe.a, s));
    }
    rc = jg_launch_small_net(e, a, sn->n_conv, sn->k0, s);
    if (rc != JG_OK) return rc;
    if (e->profile) {
      JG_HIP(hipEventRecord(pe.b, s));
      e->pending.push_back(pe);
    }
  }
  for (size_t i = 0; i < m->ops.size(); ++i) {
    const jg_op &op = m->ops[i];
    int rc = JG_OK;
    if (small && (int)i < m->small->pool_op) {
      if (op.kind == JG_OP_NMD_FINAL) {             // finish the tap the fused kernel accumulated for this slot
        const JgSmallNet *sn = m->small;
        int tap = 0;
        for (int t = 1; t <= sn->n_taps; ++t)
          if (sn->tap_part_slot[t] == op.arg && sn->tap_conv_op[t] < (int)i) tap = t;
        rc = jg_launch_small_pool_final(sn->d_part, 6, sn->n_slots, tap, nw, 2, m->d_w + op.b_off, op.f0,
                                        m->vec[op.out_vec] + op.vec_off, m->vec_w[op.out_vec], s);
        if (rc != JG_OK) return rc;
      }
      continue;
    }
    if (prec == 1) {
      // layout conversions queued by the format plan: into the scratch tensor, then the slot takes the scratch's
      // place (same bytes per element in both layouts)
      const ConvHPrep &hq = m->hprep[i];
      for (int q = 0; q < hq.n_cvt; ++q) {
        const int slot = hq.cvt_slot[q];
        const Shape &t = sh[slot];
        const int64_t rows = (int64_t)nw * t.frames;
        if (hq.cvt_to_f32[q]) rc = jg_launch_f16s_to_f32(reinterpret_cast<const uint4 *>(m->act[slot]), rows, t.L, t.C, m->cvt_scratch, s);
        else rc = jg_launch_f32_to_f16s(m->act[slot], rows, t.L, t.C, reinterpret_cast<uint4 *>(m->cvt_scratch), s, m->d_overflow);
        if (rc != JG_OK) return rc;
        std::swap(m->act[slot], m->cvt_scratch);
        std::swap(m->act_cap[slot], m->cvt_cap);
      }
    }
    if (tab && (int)i == m->tab_pool) continue;
    if (tab && (int)i == m->tab_conv) {
      const jg_op &po = m->ops[(size_t)m->tab_pool];
      int lo, pl;
      conv_geometry(l, op.k, 1, op.dilation, op.padding, &lo, &pl);
      JgTabArgs a;
      memset(&a, 0, sizeof(a));
      a.ids = d_ids; a.table = m->tab_table; a.bias = m->tab_bias;
      a.out = m->vec[po.out_vec] + po.vec_off; a.out_ld = m->vec_w[po.out_vec];
      a.rows = nw * m->id_frames; a.L = l; a.L_out = lo; a.pad_left = pl; a.k = op.k; a.dil = op.dilation;
      a.vocab = m->tab_vocab; a.zero_id = m->tab_zero; a.cout = op.cout; a.cq = m->tab_cq; a.act = m->tab_act; a.pool_kind = po.arg;
      JG_REQUIRE(m->id_frames == 1, JG_ERR_UNSUPPORTED, "table net: rows of one frame only");
      ProfEvent pe;
      if (e->profile) {
        if ((rc = prof_event(e, &pe.a)) != JG_OK || (rc = prof_event(e, &pe.b)) != JG_OK) return rc;
        pe.flops = 2.0 * op.k * op.cin * op.cout * (double)a.rows * lo;
        pe.cls = JG_PROF_TABLE;
        JG_HIP(hipEventRecord(pe.a, s));
      }
      if (m->tab_wfrag != nullptr && jg_tab_mfma_row_fits(lo, op.k, op.dilation) && !e->tab_lds_only) {
        JgTabMArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.ids = d_ids; ma.wfrag = m->tab_wfrag; ma.bias = m->tab_bias512; ma.out = a.out; ma.out_ld = a.out_ld;
        ma.rows = a.rows; ma.L = l; ma.L_out = lo; ma.pad_left = pl; ma.k = op.k; ma.dil = op.dilation; ma.cout = op.cout;
        ma.act = m->tab_act; ma.pool_kind = po.arg;
        rc = jg_launch_tab_mfma(e, ma, s);
      } else {
        rc = jg_launch_tab_conv_pool(e, a, s);
      }
      if (rc != JG_OK) return rc;
      if (e->profile) {
        JG_HIP(hipEventRecord(pe.b, s));
        e->pending.push_back(pe);
      }
      sh[op.out_buf] = Shape{m->id_frames, lo, op.cout};
      continue;
    }
    if (small && (int)i == m->small->pool_op) {
      rc = jg_launch_small_pool_final(m->small->d_part, 6, m->small->n_slots, 0, nw, m->small->pool_kind, nullptr, 0.f,
                                      m->vec[op.out_vec] + op.vec_off, m->vec_w[op.out_vec], s);
      if (rc != JG_OK) return rc;
      continue;
    }
    switch (op.kind) {
      case JG_OP_EMBED: {
        rc = jg_launch_embed_pos(d_ids, m->id_bytes, (int64_t)nw * m->id_frames * l, l, m->d_w + op.b_off, m->vocab, op.cout,
                                 op.w_off >= 0 ? m->d_w + op.w_off : nullptr, m->act[op.out_buf], m->msk[op.out_mask], s);
        sh[op.out_buf] = Shape{m->id_frames, l, op.cout};
        mlen[op.out_mask] = m->id_frames * l;
      } break;
      case JG_OP_CONV: {
        Shape in;
        if (op.in_buf == JG_BUF_IDS) { in.frames = m->id_frames; in.L = l; in.C = op.cin; }
        else in = sh[op.in_buf];
        int lo, pl;
        conv_geometry(in.L, op.k, op.stride, op.dilation, op.padding, &lo, &pl);
        ProfEvent pe;
        if (e->profile) {
          if ((rc = prof_event(e, &pe.a)) != JG_OK || (rc = prof_event(e, &pe.b)) != JG_OK) return rc;
          pe.flops = 2.0 * op.k * op.cin * op.cout * (double)nw * in.frames * lo;
          JG_HIP(hipEventRecord(pe.a, s));
        }
        const bool rb_on = prec == 1 && e->fuse_resblock != 0;
        if (rb_on && m->hprep[i].f16_ok && m->hprep[i].rb_second >= 0) {
          // first conv of a fused residual block: computed by the second conv's launch (jg_resblock.hip)
          sh[op.out_buf] = Shape{in.frames, lo, op.cout};
          if (e->profile) { e->pool.push_back(pe.a); e->pool.push_back(pe.b); }
          break;
        }
        if (rb_on && m->hprep[i].f16_ok && m->hprep[i].rb_first >= 0) {
          const ConvHPrep &hp = m->hprep[i];
          const jg_op &first = m->ops[(size_t)hp.rb_first];
          JgResBlockArgs ra;
          memset(&ra, 0, sizeof(ra));
          ra.xh = reinterpret_cast<const uint4 *>(m->act[first.in_buf]);
          ra.y = reinterpret_cast<uint4 *>(m->act[op.out_buf]);
          ra.m0 = first.in_mask >= 0 ? m->msk[first.in_mask] : nullptr;
          ra.m1 = op.in_mask >= 0 ? m->msk[op.in_mask] : nullptr;
          ra.m2 = op.out_mask >= 0 ? m->msk[op.out_mask] : nullptr;
          ra.wfrag = hp.d_rb_wfrag;
          ra.epi = hp.d_rb_epi;
          ra.overflow = m->d_overflow;
          ra.rows = nw * in.frames; ra.L = in.L; ra.k = op.k; ra.dil = op.dilation;
          const bool wide = op.cout == 64;                   // jg_resblock64.hip: the waves split into conv1 / conv2 roles
          if (wide) jg_resblock64_tiling(in.L, op.k, op.dilation, &ra.nb, &ra.tile_out, &ra.tiles_per_row);
          else jg_resblock_tiling(in.L, op.k, op.dilation, &ra.nb, &ra.tile_out, &ra.tiles_per_row);
          ra.psplit = hp.ps_store ? 1 : 0;
          if (e->profile) pe.flops *= 2.0;                   // both convs of the block
          pe.cls = JG_PROF_MFMA_F16X3;
          rc = wide ? jg_launch_resblock64(e, ra, s) : jg_launch_resblock(e, ra, s);
          if (e->profile && rc == JG_OK) {
            JG_HIP(hipEventRecord(pe.b, s));
            e->pending.push_back(pe);
          }
          sh[op.out_buf] = Shape{in.frames, lo, op.cout};
          break;
        }
        if (prec == 1 && m->hprep[i].f16_ok) {
          const ConvHPrep &hp = m->hprep[i];
          ConvHArgs a;
          memset(&a, 0, sizeof(a));
          a.xh = op.in_buf == JG_BUF_IDS ? nullptr : reinterpret_cast<const uint4 *>(m->act[op.in_buf]);
          a.ids = op.in_buf == JG_BUF_IDS ? d_ids : nullptr;
          a.embh = hp.d_embh;
          a.mask_from_ids = op.in_mask == JG_BUF_IDS;
          a.mask_in = op.in_mask >= 0 ? m->msk[op.in_mask] : nullptr;
          a.mask_out = op.out_mask >= 0 ? m->msk[op.out_mask] : nullptr;
          a.wh = hp.d_wh;
          a.y = m->act[op.out_buf];
          a.overflow = m->d_overflow;
          a.rows = nw * in.frames;
          a.L_in = in.L; a.L_out = lo;
          a.cc_in = hp.cc_in; a.cout = op.cout; a.cout_pad = op.cout;
          a.k = op.k; a.dil = op.dilation; a.pad_left = pl;
          a.tap_lo = 0; a.tap_hi = op.k - 1;
          if (hp.as_k5) {              // taps (5 - k) / 2 .. of five: the same input offsets when the left pad grows with them
            a.k = 5;
            a.dil = op.k == 1 ? 1 : op.dilation;
            a.tap_lo = std::max(1, (5 - op.k) / 2);      // >= 1: tap_lo != 0 is the kernel's "some taps are skipped" flag
            a.tap_hi = a.tap_lo + op.k - 1;
            a.pad_left = pl + a.tap_lo * a.dil;
          }
          a.cw = hp.cw;
          a.ostride = op.stride;
          a.cc_row = hp.cc_in;
          int eff_stride = op.stride, eff_lin = in.L;
          if (hp.ps_read != 0) {
            // the input was stored phase-split and masked by its writer (plan_phase_split): this stride-2 conv runs at
            // stride 1 over (L + 1) / 2 positions - five taps as three over the two phases, the 1x1 bypass on the even phase
            eff_stride = 1;
            eff_lin = (in.L + 1) / 2;                      // == lo (TF SAME at stride 2: ceil(L / 2))
            a.mask_in = nullptr;
            a.L_in = eff_lin;
            a.cc_row = 2 * hp.cc_in;
            a.k = 5; a.dil = 1; a.ostride = 1;
            if (hp.ps_read == 1) {
              a.cc_in = 2 * hp.cc_in;
              a.wh = hp.d_wh_ps[in.L & 1];
              a.tap_lo = 1; a.tap_hi = 3;
              a.pad_left = 1 + a.tap_lo * a.dil;           // a 3-tap SAME conv pads one position on the left
            } else {
              a.tap_lo = a.tap_hi = 2;
              a.pad_left = 0 + a.tap_lo * a.dil;
            }
          }
          a.psplit = hp.ps_store ? 1 : 0;
          a.L_res = eff_stride == 2 ? 2 * lo - 1 : lo;
          a.tiles_m = (a.L_res + jg_conv_f16_tile_m() - 1) / jg_conv_f16_tile_m();
          const int strips = hp.cw == 128 ? 2 : 4;               // wave strips per 256-position tile
          int strips_per_win = in.frames * a.tiles_m * strips;   // partial rows (128- / 64-position strips) per window
          {
            // window-packed tiling when the frames fill their own 256-position tiles badly (e.g. 665 codons)
            static const bool no_flat = jg_exp_env("JG_NO_FLAT") != nullptr;
            const int halo = (a.k - 1) * a.dil;
            const int gap = std::max(a.pad_left, halo - a.pad_left);
            const int fp = lo + gap;
            // a window's pitch: a multiple of the strip height (128) when the conv leaves per-strip partial rows (NMD taps, the
            // fused max pool: a strip must not straddle two windows), else only of 32 - at 83 positions and dilation 8
            // (six frames of 99) 608 instead of 640 positions per window
            const bool strip_rows = hp.nmd_slot >= 0 || hp.nmd_slot2 >= 0 || hp.pool_op >= 0;
            const int wp_unit = strip_rows ? 128 : 32;
            const int wp = (in.frames * fp + wp_unit - 1) / wp_unit * wp_unit;
            const int64_t flat_tiles = ((int64_t)nw * wp + 255) / 256;
            const int64_t row_tiles = (int64_t)a.rows * a.tiles_m;
            if (!no_flat && a.k == 5 && op.in_buf != JG_BUF_IDS && hp.d_lut == nullptr && eff_stride == 1 && eff_lin == lo &&
                ((op.cout == 128 && !hp.as_k5 && hp.ps_read == 0 && !hp.ps_store) ? jg_conv_f16_has_flat_pattern(hp.ep) : jg_conv_f16_has_narrow_pattern(hp.ep)) &&
                (int64_t)nw * wp < (1 << 24) && flat_tiles * 100 <= row_tiles * 95) {
              a.flat = 1;
              a.flat_p = fp;
              a.flat_wp = wp;
              a.flat_frames = in.frames;
              a.flat_tiles = (int)flat_tiles;
              a.flat_inv_p = 1.0f / (float)fp;
              a.flat_inv_wp = 1.0f / (float)wp;
              strips_per_win = wp / (256 / strips);
            }
          }
          if (hp.nmd_slot >= 0) m->part_rows[hp.nmd_slot] = strips_per_win;
          if (hp.nmd_slot2 >= 0) m->part_rows[hp.nmd_slot2] = strips_per_win;
          if (hp.pool_op >= 0) m->pool_rows = strips_per_win;
          a.out_f16s = hp.out_f16s ? 1 : 0;
          a.act_kind = hp.act_kind;
          a.n_hst = hp.n_hst;
          a.ep = hp.ep;
          a.alpha1 = hp.alpha1; a.alpha2 = hp.alpha2;
          a.dytmask1 = hp.dytmask1; a.dytmask2 = hp.dytmask2;
          a.n_epi_rows = hp.n_epi_rows;
          a.epi = hp.d_epi;
          for (int q = 0; q < hp.n_hst; ++q) a.hst[q] = hp.hst[q];
          if (hp.add_slot >= 0) a.addh = reinterpret_cast<const uint4 *>(m->act[hp.add_slot]);
          if (hp.nmd_slot >= 0) a.nmd_out = m->nmd_part[hp.nmd_slot];
          if (hp.nmd_slot2 >= 0) a.nmd_out2 = m->nmd_part[hp.nmd_slot2];
          a.ep_rt = hp.ep_rt;
          if (hp.pool_op >= 0) {
            const int64_t need = (int64_t)a.rows * a.tiles_m * strips * op.cout;
            if (need > m->pool_part_cap) {
              JG_HIP(hipStreamSynchronize(s));
              if (m->pool_part) (void)hipFree(m->pool_part);
              m->pool_part = nullptr;
              JG_HIP(hipMalloc(reinterpret_cast<void **>(&m->pool_part), (size_t)need * sizeof(float)));
              m->pool_part_cap = need;
            }
            a.pool_out = m->pool_part;
          }
          if (hp.d_lut != nullptr) {
            a.lut = hp.d_lut;
            a.lut_vocab = m->vocab;
            a.epi = hp.d_epi_lut;
            a.lut_one_half = op.cout <= 64 ? 1 : 0;
          }
          pe.cls = hp.d_lut != nullptr ? JG_PROF_TABLE : JG_PROF_MFMA_F16X3;
          rc = jg_launch_conv_f16(e, a, s);
          for (int hf = 1; hf < hp.n_half && rc == JG_OK; ++hf) {      // wider than 128 channels: one launch per 128
            a.ch0 = hf * 128;
            a.wh = hp.ps_read == 1 ? hp.d_wh_ps[in.L & 1] + (int64_t)hf * hp.ps_half_items : hp.d_wh + (int64_t)hf * hp.wh_half_items;
            a.epi = hp.d_epi + (int64_t)hf * hp.n_epi_rows * 2 * 128;
            rc = jg_launch_conv_f16(e, a, s);
          }
        } else {
          pe.cls = JG_PROF_MFMA_F32;
          ConvArgs a;
          memset(&a, 0, sizeof(a));
          a.x = op.in_buf == JG_BUF_IDS ? nullptr : m->act[op.in_buf];
          a.ids = op.in_buf == JG_BUF_IDS ? d_ids : nullptr;
          a.emb = op.in_buf == JG_BUF_IDS ? m->d_w + op.b_off : nullptr;
          a.mask_from_ids = op.in_mask == JG_BUF_IDS;
          a.mask_in = op.in_mask >= 0 ? m->msk[op.in_mask] : nullptr;
          a.mask_out = op.out_mask >= 0 ? m->msk[op.out_mask] : nullptr;
          a.w = m->d_w + op.w_off;
          a.w8 = m->hprep[i].d_w8;
          a.y = m->act[op.out_buf];
          a.rows = nw * in.frames;
          a.L_in = in.L; a.L_out = lo;
          a.cin = op.cin; a.cin_pad = (op.cin + 7) / 8 * 8;
          a.cout = op.cout; a.cout_pad = (op.cout + 31) / 32 * 32;
          a.k = op.k; a.stride = op.stride; a.dil = op.dilation; a.pad_left = pl;
          {
            const int tm_ = jg_conv_tile_m_for(lo, op.k, op.cin, op.stride, op.dilation);
            a.tiles_m = (lo + tm_ - 1) / tm_;
          }
          for (int q = 0; q < op.n_stages; ++q)
            if (op.stages[q].kind == JG_ST_NMD) m->part_rows[op.stages[q].arg] = in.frames * a.tiles_m;
          resolve_stages(m, op, a.st, &a.n_stages);
          rc = jg_launch_conv(e, a, s);
        }
        if (e->profile && rc == JG_OK) {
          JG_HIP(hipEventRecord(pe.b, s));
          e->pending.push_back(pe);
        }
        sh[op.out_buf] = Shape{in.frames, lo, op.cout};
      } break;
      case JG_OP_MASK: {
        const int L_in = op.in_mask == JG_BUF_IDS ? l : mlen[op.in_mask] / m->id_frames;
        int lo, pl;
        conv_geometry(L_in, op.k, op.stride, op.dilation, op.padding, &lo, &pl);
        const uint8_t *src = op.in_mask == JG_BUF_IDS ? d_ids : m->msk[op.in_mask];
        rc = jg_launch_mask(src, nw * m->id_frames, L_in, lo, op.k, op.stride, op.dilation, pl, op.mask_mode,
                            m->msk[op.out_mask], s);
        mlen[op.out_mask] = m->id_frames * lo;
      } break;
      case JG_OP_ELTWISE: {
        const Shape in = sh[op.in_buf];
        EltArgs a;
        memset(&a, 0, sizeof(a));
        a.x = m->act[op.in_buf];
        a.y = m->act[op.out_buf];
        a.mask = op.out_mask >= 0 ? m->msk[op.out_mask] : nullptr;
        a.n_pos = (int64_t)nw * in.frames * in.L;
        a.c = in.C;
        for (int q = 0; q < op.n_stages; ++q)
          if (op.stages[q].kind == JG_ST_NMD)
            m->part_rows[op.stages[q].arg] = in.frames * ((in.L + jg_conv_tile_m(in.L) - 1) / jg_conv_tile_m(in.L));
        resolve_stages(m, op, a.st, &a.n_stages);
        if (a.n_stages > 0 && a.st[0].kind == JG_ST_LN)
          rc = jg_launch_layernorm(a, nw * in.frames, in.L, (in.L + jg_conv_tile_m(in.L) - 1) / jg_conv_tile_m(in.L), s);
        else
          rc = jg_launch_eltwise(a, s);
        sh[op.out_buf] = in;
      } break;
      case JG_OP_MAXPOOL1D: {
        const Shape in = sh[op.in_buf];
        const int lo = in.L / 2;
        if (prec == 1 && m->hprep[i].pool_f16s)
          rc = jg_launch_maxpool1d_f16s(reinterpret_cast<const uint4 *>(m->act[op.in_buf]), nw * in.frames, in.L, lo,
                                        in.C, reinterpret_cast<uint4 *>(m->act[op.out_buf]), s);
        else
          rc = jg_launch_maxpool1d(m->act[op.in_buf], nullptr, nw * in.frames, in.L, lo, in.C,
                                   m->act[op.out_buf], nullptr, s);
        sh[op.out_buf] = Shape{in.frames, lo, in.C};
      } break;
      case JG_OP_FRAMESUM: {
        const Shape in = sh[op.in_buf];
        rc = jg_launch_framesum(m->act[op.in_buf], nw, in.frames, (int64_t)in.L * in.C,
                                m->act[op.out_buf], s);
        sh[op.out_buf] = Shape{1, in.L, in.C};
      } break;
      case JG_OP_POOL: {
        const Shape in = sh[op.in_buf];
        const uint8_t *mk = op.in_mask >= 0 ? m->msk[op.in_mask] : nullptr;
        if (prec == 1 && m->pool_fused_by[i] >= 0) {
          rc = jg_launch_pool_final(m->pool_part, m->pool_rows, nw, in.C,
                                    m->vec[op.out_vec] + op.vec_off, m->vec_w[op.out_vec], s);
          break;
        }
        rc = jg_launch_pool(m->act[op.in_buf], mk, nw, in.frames * in.L, in.C, op.arg,
                            m->vec[op.out_vec] + op.vec_off, m->vec_w[op.out_vec], s);
      } break;
      case JG_OP_DENSE:
        rc = jg_launch_dense(m->vec[op.in_vec], m->vec_w[op.in_vec], m->d_w + op.w_off,
                             op.b_off >= 0 ? m->d_w + op.b_off : nullptr, nw, op.cin, op.cout,
                             op.arg, m->vec[op.out_vec] + op.vec_off, m->vec_w[op.out_vec], s);
        break;
      case JG_OP_NMD_FINAL: {
        // op.arg = partial slot, in_mask = mask the tap used, cout = channels,
        // in_buf = activation slot whose shape gives the position count
        const Shape in = sh[op.in_buf];
        // partial rows per window as the tap that filled the slot laid them out (a conv records it; an element-wise
        // LayerNorm tap uses the f32 tiling of the slot)
        const int rows_f32 = in.frames * ((in.L + jg_conv_tile_m(in.L) - 1) / jg_conv_tile_m(in.L));
        const int rows_per_win = m->part_rows[op.arg] > 0 ? m->part_rows[op.arg] : rows_f32;
        const uint8_t *mk = op.in_mask >= 0 ? m->msk[op.in_mask] : nullptr;
        rc = jg_launch_nmd_final(m->nmd_part[op.arg], rows_per_win, mk, in.frames * in.L,
                                 m->d_w + op.b_off, op.f0, nw, op.cout, m->vec[op.out_vec],
                                 m->vec_w[op.out_vec], op.vec_off, s);
      } break;
      case JG_OP_OODSIG:
        // in_vec = logits (cin classes), op.k = nmd vector slot (width op.stride), arg = order
        rc = jg_launch_oodsig(m->vec[op.in_vec], m->vec_w[op.in_vec], op.cin, m->vec[op.k], m->vec_w[op.k], op.stride,
                              nw, (unsigned)op.arg, op.f0, m->vec[op.out_vec], m->vec_w[op.out_vec],
                              op.vec_off, s);
        break;
      case JG_OP_VECMAX:
        rc = jg_launch_vecmax(m->vec[op.in_vec], m->vec_w[op.in_vec], op.k, op.cout, nw, m->vec[op.out_vec], m->vec_w[op.out_vec],
                              op.vec_off, s);
        break;
      case JG_OP_STRANDS:      // the strands' rows are merged into the window's behind the program (forward_chunks)
        break;
      default:
        jg_set_error("op %zu: kind %d not implemented", i, op.kind);
        return JG_ERR_UNSUPPORTED;
    }
    if (rc != JG_OK) return rc;
  }
  return JG_OK;
}

extern "C" int jg_model_vec_width(const jg_model *m, int which) {
  if (m == nullptr || which < 0 || which > 3) return 0;
  // slot convention: 0 embedding, 1 nmd, 2 prediction, 3 reliability
  static const int slot_of[4] = {2, 3, 0, 1};
  int64_t a[JG_MAX_BUFS], b[JG_MAX_BUFS], c[JG_MAX_BUFS];
  int vw[JG_MAX_VECS];
  // unpadded widths: recompute from the ops
  int width = 0;
  const int slot = slot_of[which];
  for (const jg_op &op : m->ops) {
    if (op.out_vec != slot) continue;
    int wd = 0;
    if (op.kind == JG_OP_DENSE || op.kind == JG_OP_NMD_FINAL || op.kind == JG_OP_VECMAX)
      wd = op.vec_off + op.cout;
    else if (op.kind == JG_OP_POOL)
      wd = op.vec_off + op.cout;
    width = std::max(width, wd);
  }
  (void)a; (void)b; (void)c; (void)vw;
  // a branched model whose classifier's merge layer is Concatenate (builder.py:1262-1265): the window's prediction is the
  // strands' head outputs side by side
  if (which == 0 && m->strands > 1 && m->merge_kind == JG_MERGE_CONCAT) width *= m->strands;
  return width;
}

extern "C" double jg_model_flops_per_window(const jg_model *m, int32_t l) {
  if (m == nullptr) return 0.0;
  int64_t a[JG_MAX_BUFS], b[JG_MAX_BUFS], c[JG_MAX_BUFS];
  int vw[JG_MAX_VECS];
  double fl = 0.0;
  if (plan_shapes(const_cast<jg_model *>(m), l, a, b, c, vw, &fl) != JG_OK) return 0.0;
  return fl;
}

static int copy_out(jg_model *m, int slot, int width, float *dst, int64_t row0, int nw, int out_loc,
                    hipStream_t s) {
  if (dst == nullptr || width <= 0) return JG_OK;
  JG_REQUIRE(m->vec[slot] != nullptr, JG_ERR_INVALID,
             "output requested but the model does not produce vector slot %d", slot);
  const hipMemcpyKind kind = out_loc == JG_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const float *src = m->vec[slot];
  size_t src_ld = (size_t)m->vec_w[slot];
  if (m->strands > 1) {
    // branched model: the vector slot holds one row per strand; the window's row is their merge (the prediction by the
    // classifier's merge layer, every other output by Average - builder.py:776-791)
    const int rc = grow(&m->merged[slot], &m->merged_cap[slot], (int64_t)nw * width * (int64_t)sizeof(float));
    if (rc != JG_OK) return rc;
    const int kind = slot == 2 ? m->merge_kind : JG_MERGE_AVERAGE;
    // (`width` is the OUTPUT's: a concatenated prediction is `strands` head vectors wide)
    const int mrc = jg_launch_strand_merge(src, (int)src_ld, nw, m->strands, kind == JG_MERGE_CONCAT ? width / m->strands : width, kind,
                                           m->merged[slot], s);
    if (mrc != JG_OK) return mrc;
    src = m->merged[slot];
    src_ld = (size_t)width;
  }
  JG_HIP(hipMemcpy2DAsync(dst + row0 * width, (size_t)width * sizeof(float), src, src_ld * sizeof(float),
                          (size_t)width * sizeof(float), (size_t)nw, kind, s));
  return JG_OK;
}

// windows per launch group: amortises launch + pipeline fill.  2 048 windows of 498 codons per frame (+0.7 % over
// 1 024; 3.1 GB per activation slot), proportionally more for shorter frames (same positions: +6 % at 500 bp),
// and the 32-bit DMA offset cap below for longer ones
static int effective_chunk(const jg_model *m, int chunk, int l, int64_t n_win) {
  if (chunk <= 0) chunk = (int)std::min<int64_t>(65536, std::max<int64_t>(1024, (int64_t)2048 * 498 / std::max(l, 1) / 256 * 256));
  // split-f16 DMA offsets are 32-bit: keep one activation tensor (6 frames x l x 512 B) < 3.5 GB
  if (m->precision == 1) {
    const int64_t cap = (int64_t)(3.5e9 / (6.0 * l * 512.0));
    if (chunk > cap) chunk = (int)std::max<int64_t>(cap, 1);
  }
  if (chunk > n_win) chunk = (int)std::max<int64_t>(n_win, 1);
  return chunk;
}

// enqueue only: the chunk loop of one id tensor (workspace already sized for `chunk`); no range-guard readback
static int forward_chunks(jg_model *m, const uint8_t *d_ids, int64_t n_win, int l, float *prediction,
                          float *reliability, float *embedding, float *nmd, int out_loc, int chunk, hipStream_t s) {
  const int w_pred = jg_model_vec_width(m, 0), w_rel = jg_model_vec_width(m, 1);
  const int w_emb = jg_model_vec_width(m, 2), w_nmd = jg_model_vec_width(m, 3);
  for (int64_t w0 = 0; w0 < n_win; w0 += chunk) {
    const int nw = (int)std::min<int64_t>(chunk, n_win - w0);
    int rc = run_chunk(m, d_ids + w0 * m->strands * m->id_frames * (int64_t)l * m->id_bytes, nw * m->strands, l, s);
    if (rc != JG_OK) return rc;
    if ((rc = copy_out(m, 2, w_pred, prediction, w0, nw, out_loc, s)) != JG_OK) return rc;
    if ((rc = copy_out(m, 3, w_rel, reliability, w0, nw, out_loc, s)) != JG_OK) return rc;
    if ((rc = copy_out(m, 0, w_emb, embedding, w0, nw, out_loc, s)) != JG_OK) return rc;
    if ((rc = copy_out(m, 1, w_nmd, nmd, w0, nw, out_loc, s)) != JG_OK) return rc;
  }
  return JG_OK;
}

static int forward_device_ids(jg_model *m, const uint8_t *d_ids, int64_t n_win, int l,
                              float *prediction, float *reliability, float *embedding, float *nmd,
                              int out_loc, int chunk, hipStream_t s) {
  chunk = effective_chunk(m, chunk, l, n_win);
  JG_REQUIRE((int64_t)chunk * 6 <= 0x7fffffff / 8, JG_ERR_INVALID, "chunk too large");
  int rc = ensure_workspace(m, (int64_t)chunk * m->strands, l);
  if (rc != JG_OK) return rc;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if ((rc = forward_chunks(m, d_ids, n_win, l, prediction, reliability, embedding, nmd, out_loc, chunk, s)) != JG_OK)
      return rc;
    if (m->precision != 1) break;
    // split-f16 range guard: an activation beyond the f16 range poisons the fast path;
    // fall back to the exact-f32 kernels for this and every later call of the model.
    int flag = 0;
    JG_HIP(hipMemcpyAsync(&flag, m->d_overflow, sizeof(int), hipMemcpyDeviceToHost, s));
    JG_HIP(hipStreamSynchronize(s));
    if (flag == 0) break;
    JG_HIP(hipMemsetAsync(m->d_overflow, 0, sizeof(int), s));
    m->precision = 0;
    m->f16_reason = "an activation left the f16 range at run time";
  }
  return JG_OK;
}

extern "C" int jg_forward(jg_model *m, const uint8_t *ids, int ids_loc, int64_t n_win, int32_t l,
                          float *prediction, float *reliability, float *embedding, float *nmd,
                          int out_loc, int32_t chunk, void *stream) {
  JG_REQUIRE(m != nullptr && ids != nullptr && n_win >= 0 && l > 0, JG_ERR_INVALID,
             "jg_forward: bad arguments");
  if (n_win == 0) return JG_OK;
  jg_engine *e = m->e;
  JG_HIP(hipSetDevice(e->dev));
  hipStream_t s = pick_stream(e, stream);
  const uint8_t *d_ids = ids;
  if (ids_loc == JG_PTR_HOST) {
    const int64_t bytes = n_win * m->strands * m->id_frames * (int64_t)l * m->id_bytes;
    int rc = grow(&m->d_ids, &m->d_ids_cap, bytes);
    if (rc != JG_OK) return rc;
    JG_HIP(hipMemcpyAsync(m->d_ids, ids, (size_t)bytes, hipMemcpyHostToDevice, s));
    d_ids = m->d_ids;
  }
  int rc = forward_device_ids(m, d_ids, n_win, l, prediction, reliability, embedding, nmd, out_loc,
                              chunk, s);
  if (rc != JG_OK) return rc;
  if (out_loc == JG_PTR_HOST || ids_loc == JG_PTR_HOST) JG_HIP(hipStreamSynchronize(s));
  return JG_OK;
}

static int frame_len(int nt) {
  if (nt < 3) return 0;
  const int off = (nt % 3 == 0) ? -2 : ((nt % 3 == 1) ? -1 : 0);
  const int usable = nt - 5 + off;
  return usable > 0 ? (usable + 2) / 3 : 0;
}

// shared by jg_encode / jg_predict_windows: stage host-side window tables and
// run the encoder into a device id tensor
static int encode_common(jg_engine *e, jg_model *scratch_owner, const uint8_t *bases, int64_t n_bases,
                         int bases_loc, const int64_t *win_start, const int32_t *win_len, int win_loc,
                         int64_t n_win, int32_t fsize, const uint8_t *lut65, int32_t flags,
                         int32_t l_pad, uint8_t *d_ids, int32_t *d_counts, uint8_t *d_lut,
                         std::vector<void *> &to_free, hipStream_t s) {
  // scratch_owner: a model whose grow-only device buffers hold the uploaded bases / window table (no hipMalloc / hipFree
  // per call); without one (jg_encode) the copies are temporary
  // l_pad must hold the longest frame: known exactly for host-side window tables (the short-contig
  // pass pads to the longest window of a batch, commands/predict.py:236-245), fsize-derived otherwise
  const bool nt_ids = (flags & JG_ENC_NUCLEOTIDE) != 0;       // a row holds bases, not codons
  const bool di_ids = (flags & JG_ENC_DICODON) != 0;          // ... or codon pairs (six bases apart)
  JG_REQUIRE(!(nt_ids && di_ids), JG_ERR_INVALID, "encode: nucleotide and dicodon ids are different encodings");
  auto row_len = [&](int n) {                                  // entries per frame of a window cropped to n bases
    const int off3 = (fsize % 3 == 0) ? -2 : ((fsize % 3 == 1) ? -1 : 0);
    if (di_ids) { const int u = n - 8 + off3; return u > 0 ? (u + 5) / 6 : 0; }
    const int u = n - 5 + off3;
    return u > 0 ? (u + 2) / 3 : 0;
  };
  int need = nt_ids ? fsize : (di_ids ? row_len(fsize) : frame_len(fsize));
  if (win_loc == JG_PTR_HOST && fsize >= 3) {
    int longest = 0;
    for (int64_t i = 0; i < n_win; ++i) longest = std::max(longest, std::min(win_len[i], fsize));
    need = nt_ids ? longest : row_len(longest);
  }
  JG_REQUIRE(fsize >= 3 && l_pad >= need && l_pad >= 1, JG_ERR_INVALID,
             "encode: l_pad=%d is smaller than the %d %s the longest window yields (fsize %d)", l_pad,
             need, nt_ids ? "bases" : "codons", fsize);
  const uint8_t *d_bases = bases;
  if (bases_loc == JG_PTR_HOST) {
    void *p = nullptr;
    if (scratch_owner != nullptr) {
      const int rc = grow(&scratch_owner->d_bases_buf, &scratch_owner->d_bases_cap, std::max<int64_t>(n_bases, 1));
      if (rc != JG_OK) return rc;
      p = scratch_owner->d_bases_buf;
    } else {
      JG_HIP(hipMalloc(&p, (size_t)std::max<int64_t>(n_bases, 1)));
      to_free.push_back(p);
    }
    JG_HIP(hipMemcpyAsync(p, bases, (size_t)n_bases, hipMemcpyHostToDevice, s));
    d_bases = static_cast<const uint8_t *>(p);
    if (e->n_rec > 0) {                       // records attached: DUST on the uploaded copy, the encoder respects the case
      JG_REQUIRE(e->rec_end <= n_bases, JG_ERR_INVALID, "encode: the attached records end at %lld, beyond the %lld-byte base buffer",
                 (long long)e->rec_end, (long long)n_bases);
      const int rc = jg_launch_dust(static_cast<uint8_t *>(p), 0, n_bases, e->d_rec_off, e->n_rec, e->dust_window,
                                    e->dust_threshold, 0, n_bases, e->d_dust_cnt, s);
      if (rc != JG_OK) return rc;
      flags |= 1;
    }
  }
  const int64_t *d_start = win_start;
  const int32_t *d_len = win_len;
  if (win_loc == JG_PTR_HOST) {
    // validate on the host: every window must lie inside the base buffer
    for (int64_t i = 0; i < n_win; ++i)
      JG_REQUIRE(win_start[i] >= 0 && win_len[i] >= 0 && win_start[i] + win_len[i] <= n_bases,
                 JG_ERR_INVALID, "encode: window %lld [%lld, +%d) outside the %lld-byte base buffer",
                 (long long)i, (long long)win_start[i], win_len[i], (long long)n_bases);
    void *p = nullptr;
    if (scratch_owner != nullptr) {
      const int rc = grow(&scratch_owner->d_win, &scratch_owner->d_win_cap, n_win * 12);
      if (rc != JG_OK) return rc;
      p = scratch_owner->d_win;
    } else {
      JG_HIP(hipMalloc(&p, (size_t)n_win * 12));
      to_free.push_back(p);
    }
    JG_HIP(hipMemcpyAsync(p, win_start, (size_t)n_win * 8, hipMemcpyHostToDevice, s));
    JG_HIP(hipMemcpyAsync(static_cast<char *>(p) + n_win * 8, win_len, (size_t)n_win * 4,
                          hipMemcpyHostToDevice, s));
    d_start = static_cast<const int64_t *>(p);
    d_len = reinterpret_cast<const int32_t *>(static_cast<char *>(p) + n_win * 8);
  }
  JG_HIP(hipMemcpyAsync(d_lut, lut65, 65, hipMemcpyHostToDevice, s));
  return jg_launch_encode(d_bases, d_start, d_len, n_win, fsize, d_lut, flags, l_pad, d_ids,
                          d_counts, s);
}

extern "C" int jg_encode(jg_engine *e, const uint8_t *bases, int64_t n_bases, int bases_loc,
                         const int64_t *win_start, const int32_t *win_len, int win_loc,
                         int64_t n_win, int32_t fsize, const uint8_t *lut65, int32_t soft_mask,
                         int32_t l_pad, uint8_t *ids, int32_t *counts, int out_loc, void *stream) {
  JG_REQUIRE(e != nullptr && bases != nullptr && win_start != nullptr && win_len != nullptr &&
                 lut65 != nullptr && ids != nullptr && n_win >= 0,
             JG_ERR_INVALID, "jg_encode: bad arguments");
  if (n_win == 0) return JG_OK;
  JG_HIP(hipSetDevice(e->dev));
  hipStream_t s = pick_stream(e, stream);
  std::vector<void *> to_free;
  uint8_t *d_ids = ids;
  int32_t *d_counts = counts;
  const int64_t id_bytes = n_win * ((soft_mask & JG_ENC_NUCLEOTIDE) ? 2 : 6) * (int64_t)l_pad * ((soft_mask & JG_ENC_DICODON) ? 2 : 1);
  void *d_lut = nullptr;
  JG_HIP(hipMalloc(&d_lut, 80));
  to_free.push_back(d_lut);
  if (out_loc == JG_PTR_HOST) {
    void *p = nullptr;
    JG_HIP(hipMalloc(&p, (size_t)id_bytes));
    to_free.push_back(p);
    d_ids = static_cast<uint8_t *>(p);
    if (counts != nullptr) {
      JG_HIP(hipMalloc(&p, (size_t)n_win * 16));
      to_free.push_back(p);
      d_counts = static_cast<int32_t *>(p);
    }
  }
  int rc = encode_common(e, nullptr, bases, n_bases, bases_loc, win_start, win_len, win_loc, n_win,
                         fsize, lut65, soft_mask, l_pad, d_ids, d_counts,
                         static_cast<uint8_t *>(d_lut), to_free, s);
  if (rc == JG_OK && out_loc == JG_PTR_HOST) {
    hipError_t err = hipMemcpyAsync(ids, d_ids, (size_t)id_bytes, hipMemcpyDeviceToHost, s);
    if (err == hipSuccess && counts != nullptr)
      err = hipMemcpyAsync(counts, d_counts, (size_t)n_win * 16, hipMemcpyDeviceToHost, s);
    if (err != hipSuccess) {
      jg_set_error("jg_encode: D2H copy -> %s", hipGetErrorString(err));
      rc = JG_ERR_HIP;
    }
  }
  if (!to_free.empty() || out_loc == JG_PTR_HOST) (void)hipStreamSynchronize(s);
  for (void *p : to_free) (void)hipFree(p);
  return rc;
}

// ---- streamed ingest -------------------------------------------------------------------------
// Host-resident bases larger than the engine's stream budget never exist on the device as a whole: the
// (start-sorted) window list is cut into groups whose base span fits the budget and whose window count is a whole
// number of forward passes.  The groups run as a two-deep pipeline that never drains the compute stream:
//   helper thread   span of group g+1: host -> pinned staging -> device buffer (g+1)%2 on the copy stream
//   compute stream  group g: window table (pinned) -> [DUST] -> encode -> forward passes -> outputs D2H into pinned
//                   staging g%2 -> event
//   calling thread  after ENQUEUEING group g it waits for group g-1's event, copies that group's rows from the pinned
//                   staging into the caller's arrays and publishes the progress (JG_STAT_WINDOWS_DONE): the rows of
//                   windows below that mark are final while the call is still running.
// Nothing in the loop synchronises the whole stream; buffers are recycled on events (device span: the encode that
// read it; pinned span: its H2D copy; pinned outputs: the calling thread's own copy-out).
static int stream_setup(jg_engine *e, int64_t span_cap, int64_t io_cap) {
  if (e->copy_stream == nullptr) JG_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    if (e->h2d_done[i] == nullptr) JG_HIP(hipEventCreateWithFlags(&e->h2d_done[i], hipEventDisableTiming));
    if (e->enc_done[i] == nullptr) JG_HIP(hipEventCreateWithFlags(&e->enc_done[i], hipEventDisableTiming));
    if (e->grp_done[i] == nullptr)
      JG_HIP(hipEventCreateWithFlags(&e->grp_done[i], hipEventDisableTiming | hipEventBlockingSync));
  }
  if (span_cap > e->pin_cap) {
    for (int i = 0; i < 2; ++i) {
      if (e->pin[i]) JG_HIP(hipHostFree(e->pin[i]));
      e->pin[i] = nullptr;
      JG_HIP(hipHostMalloc(&e->pin[i], (size_t)span_cap, hipHostMallocDefault));
    }
    e->pin_cap = span_cap;
  }
  if (span_cap > e->dbase_cap) {
    for (int i = 0; i < 2; ++i) {
      if (e->dbase[i]) JG_HIP(hipFree(e->dbase[i]));
      e->dbase[i] = nullptr;
      JG_HIP(hipMalloc(&e->dbase[i], (size_t)span_cap));
    }
    e->dbase_cap = span_cap;
  }
  if (io_cap > e->pin_io_cap) {
    for (int i = 0; i < 2; ++i) {
      if (e->pin_io[i]) JG_HIP(hipHostFree(e->pin_io[i]));
      e->pin_io[i] = nullptr;
      JG_HIP(hipHostMalloc(&e->pin_io[i], (size_t)io_cap, hipHostMallocDefault));
    }
    e->pin_io_cap = io_cap;
  }
  return JG_OK;
}

struct StreamGroup {
  int64_t w0, w1;      // windows [w0, w1)
  int64_t b0, b1;      // base span [b0, b1) they touch
};

// Groups of a start-sorted window list: base span <= budget; a group that holds at least one whole forward pass is cut
// back to a multiple of `chunk` windows, so that only a call's last pass is ragged.
static void stream_groups(const int64_t *win_start, const int32_t *win_len, int64_t n_win, int64_t budget, int64_t chunk,
                          std::vector<StreamGroup> &groups) {
  int64_t i = 0;
  while (i < n_win) {
    // ramp: the first span is an eighth of the budget and the second a half, so that the compute stream has work after a
    // millisecond of staging instead of after a whole span (nothing hides the first span's copy, upload and DUST pass)
    const size_t gi = groups.size();
    const int64_t cap = gi == 0 ? std::max<int64_t>(budget / 8, 4096) : (gi == 1 ? std::max<int64_t>(budget / 2, 4096) : budget);
    StreamGroup g{i, i, win_start[i], win_start[i] + win_len[i]};
    int64_t j = i + 1;
    for (; j < n_win; ++j) {
      const int64_t b1 = std::max(g.b1, win_start[j] + win_len[j]);
      if (b1 - g.b0 > cap) break;
      g.b1 = b1;
    }
    if (j < n_win && j - i >= chunk && (j - i) % chunk != 0) {
      j = i + (j - i) / chunk * chunk;
      g.b1 = g.b0;
      for (int64_t q = i; q < j; ++q) g.b1 = std::max(g.b1, win_start[q] + win_len[q]);
    }
    g.w1 = j;
    groups.push_back(g);
    i = j;
  }
}

static int predict_streamed(jg_model *m, const uint8_t *bases, int64_t n_bases, const int64_t *win_start,
                            const int32_t *win_len, int64_t n_win, int32_t fsize, const uint8_t *lut65,
                            int32_t flags, int32_t l_pad, float *prediction, float *reliability,
                            float *embedding, float *nmd, int32_t *counts, int out_loc, int32_t chunk,
                            hipStream_t s) {
  jg_engine *e = m->e;
  const int64_t budget = e->stream_bytes;
  const int fchunk = effective_chunk(m, chunk, l_pad, n_win);
  JG_REQUIRE((int64_t)fchunk * 6 <= 0x7fffffff / 8, JG_ERR_INVALID, "chunk too large");
  std::vector<StreamGroup> groups;
  stream_groups(win_start, win_len, n_win, budget, fchunk, groups);
  int64_t span_cap = 0, win_cap = 0;
  for (const StreamGroup &q : groups) {
    span_cap = std::max(span_cap, q.b1 - q.b0);
    win_cap = std::max(win_cap, q.w1 - q.w0);
  }
  // with records attached (jg_engine_set_dust) every span is staged with 64 bases of context either side and
  // soft-masked on the device before it is encoded: an interval that touches a window starts or ends < 64 bases outside it
  const bool dust = e->n_rec > 0;
  const int64_t ctx = dust ? 64 : 0;
  if (dust) {
    JG_REQUIRE(e->rec_end <= n_bases, JG_ERR_INVALID, "encode: the attached records end at %lld, beyond the %lld-byte base buffer",
               (long long)e->rec_end, (long long)n_bases);
    flags |= 1;
  }
  span_cap = (span_cap + 2 * ctx + 4095) / 4096 * 4096;
  // pinned staging per parity: [window starts i64][window lengths i32][outputs f32 ...][counts i32 x 4][range-guard flag]
  const bool host_out = out_loc == JG_PTR_HOST;
  const int w_pred = jg_model_vec_width(m, 0), w_rel = jg_model_vec_width(m, 1);
  const int w_emb = jg_model_vec_width(m, 2), w_nmd = jg_model_vec_width(m, 3);
  float *user[4] = {prediction, reliability, embedding, nmd};
  const int width[4] = {w_pred, w_rel, w_emb, w_nmd};
  int64_t off_out[4] = {0, 0, 0, 0};
  int64_t io = (win_cap * 12 + 63) / 64 * 64;
  for (int k = 0; k < 4; ++k) {
    off_out[k] = io;
    if (host_out && user[k] != nullptr && width[k] > 0) io += (win_cap * width[k] * 4 + 63) / 64 * 64;
  }
  const int64_t off_counts = io;
  if (host_out && counts != nullptr) io += win_cap * 16;
  const int64_t off_flag = io;
  io += 64;
  int rc = stream_setup(e, std::max<int64_t>(span_cap, 4096), (io + 4095) / 4096 * 4096);
  if (rc != JG_OK) return rc;
  if ((rc = grow(&m->d_ids, &m->d_ids_cap, win_cap * 6 * (int64_t)l_pad * m->id_bytes)) != JG_OK) return rc;
  if ((rc = grow(&m->d_win, &m->d_win_cap, win_cap * 12)) != JG_OK) return rc;
  if (counts != nullptr && host_out)
    if ((rc = grow(&m->d_counts, &m->d_counts_cap, win_cap * 16)) != JG_OK) return rc;
  if ((rc = ensure_workspace(m, (int64_t)fchunk * m->strands, l_pad)) != JG_OK) return rc;
  JG_HIP(hipMemcpyAsync(m->d_lut, lut65, 65, hipMemcpyHostToDevice, s));
  JG_HIP(hipStreamSynchronize(s));               // (lut65 is the caller's pageable memory; nothing else waits in the loop)
  e->streamed_groups = (int64_t)groups.size();
  e->streamed_bytes = 0;
  e->peak_dev_bases = 2 * e->dbase_cap;

  bool span_used[2] = {false, false};            // parity b's device span / pinned span have been used in this pipeline run
  auto stage = [&](size_t gi) -> int {           // host span -> pinned -> device buffer gi % 2 (copy stream)
    const StreamGroup &g = groups[gi];
    const int b = (int)(gi & 1);
    JG_HIP(hipSetDevice(e->dev));
    const int64_t h0 = std::max<int64_t>(0, g.b0 - ctx), h1 = std::min(n_bases, g.b1 + ctx);
    if (span_used[b]) JG_HIP(hipEventSynchronize(e->h2d_done[b]));                 // the pinned span's last copy has left it
    memcpy(e->pin[b], bases + h0, (size_t)(h1 - h0));
    if (span_used[b]) JG_HIP(hipStreamWaitEvent(e->copy_stream, e->enc_done[b], 0));   // the device span's last reader is done
    JG_HIP(hipMemcpyAsync(e->dbase[b], e->pin[b], (size_t)(h1 - h0), hipMemcpyHostToDevice, e->copy_stream));
    if (dust && e->dust_on_copy) {
      // DUST on the copy stream, behind the span's upload: a vector / LDS kernel that shares the CUs with the matrix-core
      // convolutions of the previous group instead of standing in front of this group's encoder on the compute stream
      const int drc = jg_launch_dust(static_cast<uint8_t *>(e->dbase[b]), h0, h1 - h0, e->d_rec_off, e->n_rec, e->dust_window,
                                     e->dust_threshold, g.b0, g.b1, e->d_dust_cnt, e->copy_stream);
      if (drc != JG_OK) return drc;
    }
    JG_HIP(hipEventRecord(e->h2d_done[b], e->copy_stream));
    span_used[b] = true;
    e->streamed_bytes += g.b1 - g.b0;
    return JG_OK;
  };
  auto enqueue = [&](size_t gi) -> int {         // everything group gi needs of the compute stream
    const StreamGroup &g = groups[gi];
    const int b = (int)(gi & 1);
    const int64_t nw = g.w1 - g.w0;
    char *io_b = static_cast<char *>(e->pin_io[b]);
    int64_t *p_start = reinterpret_cast<int64_t *>(io_b);
    int32_t *p_len = reinterpret_cast<int32_t *>(io_b + win_cap * 8);
    const int64_t h0 = std::max<int64_t>(0, g.b0 - ctx);                                       // start of the staged span
    for (int64_t i = 0; i < nw; ++i) p_start[i] = win_start[g.w0 + i] - h0;
    memcpy(p_len, win_len + g.w0, (size_t)nw * 4);
    char *dw = static_cast<char *>(m->d_win);
    JG_HIP(hipMemcpyAsync(dw, io_b, (size_t)(win_cap * 8 + nw * 4), hipMemcpyHostToDevice, s));
    JG_HIP(hipStreamWaitEvent(s, e->h2d_done[b], 0));          // (uploaded and - JG_OPT_DUST_ON_COPY_STREAM - soft-masked)
    if (dust && !e->dust_on_copy) {
      const int64_t h1 = std::min(n_bases, g.b1 + ctx);
      const int drc = jg_launch_dust(static_cast<uint8_t *>(e->dbase[b]), h0, h1 - h0, e->d_rec_off, e->n_rec, e->dust_window,
                                     e->dust_threshold, g.b0, g.b1, e->d_dust_cnt, s);
      if (drc != JG_OK) return drc;
    }
    int32_t *d_counts = counts == nullptr ? nullptr : (host_out ? m->d_counts : counts + g.w0 * 4);
    int erc = jg_launch_encode(static_cast<const uint8_t *>(e->dbase[b]), reinterpret_cast<const int64_t *>(dw),
                               reinterpret_cast<const int32_t *>(dw + win_cap * 8), nw, fsize, m->d_lut, flags, l_pad,
                               m->d_ids, d_counts, s);
    if (erc != JG_OK) return erc;
    JG_HIP(hipEventRecord(e->enc_done[b], s));
    float *dst[4];
    for (int k = 0; k < 4; ++k)
      dst[k] = user[k] == nullptr ? nullptr
                                  : (host_out ? reinterpret_cast<float *>(io_b + off_out[k]) : user[k] + g.w0 * width[k]);
    erc = forward_chunks(m, m->d_ids, nw, l_pad, dst[0], dst[1], dst[2], dst[3], out_loc, fchunk, s);
    if (erc != JG_OK) return erc;
    if (counts != nullptr && host_out)
      JG_HIP(hipMemcpyAsync(io_b + off_counts, d_counts, (size_t)nw * 16, hipMemcpyDeviceToHost, s));
    if (m->precision == 1)
      JG_HIP(hipMemcpyAsync(io_b + off_flag, m->d_overflow, sizeof(int), hipMemcpyDeviceToHost, s));
    JG_HIP(hipEventRecord(e->grp_done[b], s));
    return JG_OK;
  };
  // wait for group gi, hand its rows to the caller, publish the progress; *overflow: the f16 range guard tripped in it
  auto finalize = [&](size_t gi, bool *overflow) -> int {
    const StreamGroup &g = groups[gi];
    const int b = (int)(gi & 1);
    const int64_t nw = g.w1 - g.w0;
    JG_HIP(hipEventSynchronize(e->grp_done[b]));
    const char *io_b = static_cast<const char *>(e->pin_io[b]);
    if (m->precision == 1 && *reinterpret_cast<const int *>(io_b + off_flag) != 0) {
      *overflow = true;
      return JG_OK;
    }
    if (host_out) {
      for (int k = 0; k < 4; ++k)
        if (user[k] != nullptr && width[k] > 0)
          memcpy(user[k] + g.w0 * width[k], io_b + off_out[k], (size_t)nw * width[k] * 4);
      if (counts != nullptr) memcpy(counts + g.w0 * 4, io_b + off_counts, (size_t)nw * 16);
    }
    e->windows_done.store(g.w1, std::memory_order_release);
    return JG_OK;
  };

  size_t first = 0;
  while (first < groups.size()) {
    span_used[0] = span_used[1] = false;
    if ((rc = stage(first)) != JG_OK) return rc;
    bool overflow = false;
    size_t redo = groups.size();
    for (size_t gi = first; gi < groups.size() && !overflow; ++gi) {
      // the next group's span is staged by a helper thread while this group is enqueued and the previous one handed over
      int stage_rc = JG_OK;
      std::string stage_err;
      std::thread stager;
      if (gi + 1 < groups.size())
        stager = std::thread([&, gi]() {
          stage_rc = stage(gi + 1);
          if (stage_rc != JG_OK) stage_err = jg_last_error();
        });
      struct Joiner {
        std::thread &t;
        ~Joiner() { if (t.joinable()) t.join(); }
      } joiner{stager};
      if ((rc = enqueue(gi)) != JG_OK) return rc;
      if (gi > first) {
        if ((rc = finalize(gi - 1, &overflow)) != JG_OK) return rc;
        if (overflow) redo = gi - 1;
      }
      if (stager.joinable()) stager.join();
      if (stage_rc != JG_OK) {
        jg_set_error("%s", stage_err.c_str());
        return stage_rc;
      }
    }
    if (!overflow) {
      if ((rc = finalize(groups.size() - 1, &overflow)) != JG_OK) return rc;
      if (overflow) redo = groups.size() - 1;
    }
    if (!overflow) break;
    // split-f16 range guard: an activation beyond the f16 range poisons the fast path - drain the pipeline, fall back to the
    // exact-f32 kernels for the group that tripped it, for every later group and for every later call of the model
    JG_HIP(hipStreamSynchronize(s));
    JG_HIP(hipStreamSynchronize(e->copy_stream));
    JG_HIP(hipMemsetAsync(m->d_overflow, 0, sizeof(int), s));
    m->precision = 0;
    m->f16_reason = "an activation left the f16 range at run time";
    if ((rc = ensure_workspace(m, (int64_t)fchunk * m->strands, l_pad)) != JG_OK) return rc;
    first = redo;
  }
  JG_HIP(hipStreamSynchronize(s));
  return JG_OK;
}

extern "C" int jg_predict_windows(jg_model *m, const uint8_t *bases, int64_t n_bases, int bases_loc,
                                  const int64_t *win_start, const int32_t *win_len, int win_loc,
                                  int64_t n_win, int32_t fsize, const uint8_t *lut65,
                                  int32_t soft_mask, int32_t l_pad, float *prediction,
                                  float *reliability, float *embedding, float *nmd, int32_t *counts,
                                  int out_loc, int32_t chunk, void *stream) {
  JG_REQUIRE(m != nullptr && bases != nullptr && win_start != nullptr && win_len != nullptr &&
                 lut65 != nullptr && n_win >= 0,
             JG_ERR_INVALID, "jg_predict_windows: bad arguments");
  jg_engine *e = m->e;
  e->windows_done.store(0, std::memory_order_release);       // (also for an empty call: a poller must not see the previous call's mark)
  if (n_win == 0) return JG_OK;
  JG_HIP(hipSetDevice(e->dev));
  hipStream_t s = pick_stream(e, stream);
  e->streamed_groups = 0;
  e->streamed_bytes = 0;
  e->peak_dev_bases = bases_loc == JG_PTR_HOST ? n_bases : 0;
  e->windows_done.store(0, std::memory_order_release);
  if (m->strands > 1) soft_mask |= JG_ENC_NUCLEOTIDE;       // a two-strand model reads nucleotide ids (n_win, 2, l_pad)
  if (m->id_bytes == 2) soft_mask |= JG_ENC_DICODON;        // a dicodon model reads 16-bit ids of codon pairs
  if (bases_loc == JG_PTR_HOST && win_loc == JG_PTR_HOST && n_bases > e->stream_bytes) {
    // streamed ingest needs a start-sorted window list (the fragmenter's FASTA order is) inside the buffer
    bool sorted = true;
    int longest = 0;
    for (int64_t i = 0; i < n_win; ++i) {
      JG_REQUIRE(win_start[i] >= 0 && win_len[i] >= 0 && win_start[i] + win_len[i] <= n_bases, JG_ERR_INVALID,
                 "encode: window %lld [%lld, +%d) outside the %lld-byte base buffer", (long long)i,
                 (long long)win_start[i], win_len[i], (long long)n_bases);
      sorted &= i == 0 || win_start[i] >= win_start[i - 1];
      longest = std::max(longest, std::min(win_len[i], fsize));
    }
    if (sorted) {
      const int off3 = (fsize % 3 == 0) ? -2 : ((fsize % 3 == 1) ? -1 : 0);
      const int usable = longest - 5 + off3, usable6 = longest - 8 + off3;
      const int need = m->strands > 1 ? longest : m->id_bytes == 2 ? (usable6 > 0 ? (usable6 + 5) / 6 : 0)
                                                                   : (usable > 0 ? (usable + 2) / 3 : 0);
      JG_REQUIRE(fsize >= 3 && l_pad >= need && l_pad >= 1, JG_ERR_INVALID,
                 "encode: l_pad=%d is smaller than the %d %s the longest window yields (fsize %d)", l_pad, need,
                 m->strands > 1 ? "bases" : "codons", fsize);
      return predict_streamed(m, bases, n_bases, win_start, win_len, n_win, fsize, lut65, soft_mask, l_pad,
                              prediction, reliability, embedding, nmd, counts, out_loc, chunk, s);
    }
  }
  std::vector<void *> to_free;
  int rc = grow(&m->d_ids, &m->d_ids_cap, n_win * 6 * (int64_t)l_pad * m->id_bytes);
  if (rc != JG_OK) return rc;
  int32_t *d_counts = counts;
  if (counts != nullptr && out_loc == JG_PTR_HOST) {
    rc = grow(&m->d_counts, &m->d_counts_cap, n_win * 16);
    if (rc != JG_OK) return rc;
    d_counts = m->d_counts;
  }
  rc = encode_common(e, m, bases, n_bases, bases_loc, win_start, win_len, win_loc, n_win, fsize,
                     lut65, soft_mask, l_pad, m->d_ids, d_counts, m->d_lut, to_free, s);
  if (rc == JG_OK)
    rc = forward_device_ids(m, m->d_ids, n_win, l_pad, prediction, reliability, embedding, nmd,
                            out_loc, chunk, s);
  if (rc == JG_OK && counts != nullptr && out_loc == JG_PTR_HOST) {
    hipError_t err = hipMemcpyAsync(counts, d_counts, (size_t)n_win * 16, hipMemcpyDeviceToHost, s);
    if (err != hipSuccess) {
      jg_set_error("jg_predict_windows: counts D2H -> %s", hipGetErrorString(err));
      rc = JG_ERR_HIP;
    }
  }
  if (!to_free.empty() || out_loc == JG_PTR_HOST) (void)hipStreamSynchronize(s);
  for (void *p : to_free) (void)hipFree(p);
  if (rc == JG_OK && out_loc == JG_PTR_HOST) e->windows_done.store(n_win, std::memory_order_release);
  return rc;
}
