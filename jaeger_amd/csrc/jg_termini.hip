// Terminal-repeat scan (utils/termini.py:88-189): for every contig the first `scan` bases are locally
// aligned (Smith-Waterman, match +2 / mismatch -100, gap open 100 / extend 5 - a gap of k costs
// 100 + 5(k-1)) against the last `scan` bases (direct terminal repeat, DTR) and against their reverse
// complement (inverted, ITR), scan = min(max(int(0.04 len), 400), 4000).  The reference runs parasail's
// sw_trace_scan_16 on CPU threads and then only uses, per alignment: the score, the alignment length
// (traceback columns) and the gap count in the query row.  Those three are carried through the DP here
// as auxiliary values of each state's best path, so no traceback matrix is stored.
//
// One workgroup per (contig, DTR|ITR) job.  Thread t owns the query rows [t*R, (t+1)*R) (R <= 16) and walks
// its strip column by column one step behind thread t-1 (a systolic wavefront): the only values that cross
// threads are the H / F states of each strip's last row, double-buffered in LDS, one barrier per step.
// Letters are compared case-insensitively and only A/C/G/T can match (parasail.matrix_create("ACGT", 2, -100)).
// Ties: H prefers the diagonal, then the gap in the query row (E), then F; E and F prefer extension; the best
// cell is the first maximum in (column, row) order.  parasail is not installable here: parity with its
// tie-breaking is unpinned (scores are unique; lengths can differ only between co-optimal alignments).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "jg_common.h"

namespace {

constexpr int TT = 256;          // threads per job
constexpr int RMAX = 16;         // rows per thread (scan <= 4096)
constexpr int NEG = -1000000;    // "minus infinity" that survives a few subtractions
constexpr int S_MATCH = 2, S_MISMATCH = -100, G_OPEN = 100, G_EXT = 5;

struct TermJob {
  int64_t q_off;   // query  = bases[q_off + i]
  int64_t r_off;   // DTR: ref[j] = bases[r_off + j]; ITR: ref[j] = complement(bases[r_off + n - 1 - j])
  int32_t n;
  int32_t itr;
};

struct TermOut {
  int32_t score, len, fgaps, end_q, end_r;
};

__device__ __forceinline__ int base_code(uint8_t c) {   // A,C,G,T -> 0..3 (any case), else 4
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
  }
}

__global__ __launch_bounds__(TT) void termini_kernel(const uint8_t *__restrict__ bases,
                                                     const TermJob *__restrict__ jobs,
                                                     TermOut *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const TermJob job = jobs[blockIdx.x];
  const int n = job.n, tid = threadIdx.x;
  uint8_t *ref = smem;                                           // n codes
  int4 *edge = reinterpret_cast<int4 *>(smem + ((n + 15) & ~15)); // [2][TT]: (H, Haux, F, Faux) of a strip's last row
  int *red = reinterpret_cast<int *>(edge + 2 * TT);              // reduction scratch
  for (int j = tid; j < n; j += TT) {
    int c = job.itr ? base_code(bases[job.r_off + n - 1 - j]) : base_code(bases[job.r_off + j]);
    if (job.itr && c < 4) c = 3 - c;                               // A<->T, C<->G on ACGT = 0..3
    ref[j] = (uint8_t)c;
  }
  const int R = (n + TT - 1) / TT;
  const int i0 = tid * R;
  int q[RMAX], hl[RMAX], el[RMAX], hla[RMAX], ela[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int i = i0 + r;
    q[r] = (r < R && i < n) ? base_code(bases[job.q_off + i]) : 5;   // 5: row does not exist
    hl[r] = 0; hla[r] = 0; el[r] = NEG; ela[r] = 0;
  }
  edge[tid] = make_int4(0, 0, NEG, 0);
  edge[TT + tid] = make_int4(0, 0, NEG, 0);
  int best = 0, best_aux = 0, best_i = -1, best_j = -1;
  int prev_up_h = 0, prev_up_a = 0;                                  // H[i0-1][j-1]
  __syncthreads();
  const int n_active = (n + R - 1) / R;                              // threads that own rows
  const int steps = n + n_active - 1;
  for (int step = 0; step < steps; ++step) {
    const int j = step - tid;
    int4 pub = make_int4(0, 0, NEG, 0);
    if (tid < n_active && j >= 0 && j < n) {
      int up_h = 0, up_a = 0, up_f = NEG, up_fa = 0;
      if (tid > 0) {
        const int4 e = edge[((step + 1) & 1) * TT + tid - 1];          // written at step - 1
        up_h = e.x; up_a = e.y; up_f = e.z; up_fa = e.w;
      }
      int dg_h = prev_up_h, dg_a = prev_up_a;
      prev_up_h = up_h; prev_up_a = up_a;
      const int rc = ref[j];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        if (r < R && q[r] != 5) {
          // E: gap in the query row (consumes a ref base): from the left neighbour
          int e_s, e_a;
          if (el[r] - G_EXT >= hl[r] - G_OPEN) { e_s = el[r] - G_EXT; e_a = ela[r]; }
          else { e_s = hl[r] - G_OPEN; e_a = hla[r]; }
          e_a += (1 << 16) | 1;                                      // one more column, one more query gap
          // F: gap in the ref row: from above
          int f_s, f_a;
          if (up_f - G_EXT >= up_h - G_OPEN) { f_s = up_f - G_EXT; f_a = up_fa; }
          else { f_s = up_h - G_OPEN; f_a = up_a; }
          f_a += 1 << 16;
          const int sub = (q[r] < 4 && q[r] == rc) ? S_MATCH : S_MISMATCH;
          const int d_s = dg_h + sub, d_a = dg_a + (1 << 16);
          int h = d_s, ha = d_a;
          if (e_s > h) { h = e_s; ha = e_a; }
          if (f_s > h) { h = f_s; ha = f_a; }
          if (h <= 0) { h = 0; ha = 0; }
          if (h > best) { best = h; best_aux = ha; best_i = i0 + r; best_j = j; }
          dg_h = hl[r]; dg_a = hla[r];                               // H[i][j-1] is the next row's diagonal
          hl[r] = h; hla[r] = ha; el[r] = e_s; ela[r] = e_a;
          up_h = h; up_a = ha; up_f = f_s; up_fa = f_a;
        }
      }
      pub = make_int4(up_h, up_a, up_f, up_fa);
    }
    edge[(step & 1) * TT + tid] = pub;
    __syncthreads();
  }
  // best cell of the job: max score, then smallest column, then smallest row
  long long key = best > 0 ? (((long long)best << 40) | ((long long)(0xFFFFF - best_j) << 20) | (long long)(0xFFFFF - best_i)) : 0;
  long long *redl = reinterpret_cast<long long *>(red);
  int *reda = red + 2 * TT;
  redl[tid] = key;
  reda[tid] = best_aux;
  __syncthreads();
  for (int s = TT / 2; s > 0; s >>= 1) {
    if (tid < s && redl[tid + s] > redl[tid]) { redl[tid] = redl[tid + s]; reda[tid] = reda[tid + s]; }
    __syncthreads();
  }
  if (tid == 0) {
    TermOut o;
    const long long k = redl[0];
    o.score = (int)(k >> 40);
    o.len = reda[0] >> 16;
    o.fgaps = reda[0] & 0xffff;
    o.end_r = k ? 0xFFFFF - (int)((k >> 20) & 0xFFFFF) : -1;
    o.end_q = k ? 0xFFFFF - (int)(k & 0xFFFFF) : -1;
    out[blockIdx.x] = o;
  }
}

}  // namespace

// results: (n_records, 10) int32 rows = DTR(score, len, fgaps, end_q, end_r), ITR(score, len, fgaps, end_q, end_r);
// records shorter than min_len get all -1.  `bases` host or device per bases_loc; offsets and results on the host.
extern "C" int jg_terminal_repeats(jg_engine *e, const uint8_t *bases, int64_t n_bases, int bases_loc,
                                   const int64_t *offsets, int64_t n_records, int32_t min_len,
                                   int32_t *results) {
  JG_REQUIRE(e != nullptr && bases != nullptr && offsets != nullptr && results != nullptr && n_records >= 0,
             JG_ERR_INVALID, "jg_terminal_repeats: bad arguments");
  JG_HIP(hipSetDevice(e->dev));
  hipStream_t s = e->stream;
  std::vector<TermJob> jobs;
  std::vector<int64_t> owner;
  std::vector<uint8_t> ends;          // host bases: only the two scanned ends of every record go to the device
  int max_n = 0;
  for (int64_t r = 0; r < n_records; ++r) {
    const int64_t len = offsets[r + 1] - offsets[r];
    JG_REQUIRE(len >= 0 && offsets[r + 1] <= n_bases, JG_ERR_INVALID, "jg_terminal_repeats: record %lld outside the base buffer",
               (long long)r);
    for (int q = 0; q < 10; ++q) results[r * 10 + q] = -1;
    if (len < min_len || len < 1) continue;
    int scan = (int)std::min<int64_t>(std::max<int64_t>((int64_t)((double)len * 0.04), 400), 4000);
    if (scan > len) scan = (int)len;                   // str slicing clamps (termini.py:121-133)
    max_n = std::max(max_n, scan);
    int64_t q_off = offsets[r], r_off = offsets[r + 1] - scan;
    if (bases_loc == JG_PTR_HOST) {
      q_off = (int64_t)ends.size();
      ends.insert(ends.end(), bases + offsets[r], bases + offsets[r] + scan);
      r_off = (int64_t)ends.size();
      ends.insert(ends.end(), bases + offsets[r + 1] - scan, bases + offsets[r + 1]);
    }
    for (int itr = 0; itr < 2; ++itr) {
      jobs.push_back(TermJob{q_off, r_off, scan, itr});
      owner.push_back(r * 2 + itr);
    }
  }
  if (jobs.empty()) return JG_OK;
  JG_REQUIRE(max_n <= TT * RMAX, JG_ERR_UNSUPPORTED, "jg_terminal_repeats: scan length %d", max_n);
  {
    // longest jobs first: a 4 000-base scan costs 100x a 400-base one, and workgroups are dispatched in index order -
    // in FASTA order the last long jobs run alone at the end of the launch
    std::vector<size_t> order(jobs.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return jobs[a].n > jobs[b].n; });
    std::vector<TermJob> sj(jobs.size());
    std::vector<int64_t> so(jobs.size());
    for (size_t k = 0; k < order.size(); ++k) { sj[k] = jobs[order[k]]; so[k] = owner[order[k]]; }
    jobs.swap(sj);
    owner.swap(so);
  }
  const uint8_t *d_bases = bases;
  void *tmp_bases = nullptr, *d_jobs = nullptr, *d_out = nullptr;
  if (bases_loc == JG_PTR_HOST) {
    JG_HIP(hipMalloc(&tmp_bases, std::max<size_t>(ends.size(), 1)));
    JG_HIP(hipMemcpyAsync(tmp_bases, ends.data(), ends.size(), hipMemcpyHostToDevice, s));
    d_bases = static_cast<const uint8_t *>(tmp_bases);
  }
  JG_HIP(hipMalloc(&d_jobs, jobs.size() * sizeof(TermJob)));
  JG_HIP(hipMalloc(&d_out, jobs.size() * sizeof(TermOut)));
  JG_HIP(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(TermJob), hipMemcpyHostToDevice, s));
  const size_t smem = (size_t)((max_n + 15) & ~15) + 2 * TT * sizeof(int4) + TT * 12;
  hipLaunchKernelGGL(termini_kernel, dim3((unsigned)jobs.size()), dim3(TT), smem, s, d_bases,
                     static_cast<const TermJob *>(d_jobs), static_cast<TermOut *>(d_out));
  JG_HIP(hipGetLastError());
  std::vector<TermOut> host(jobs.size());
  JG_HIP(hipMemcpyAsync(host.data(), d_out, jobs.size() * sizeof(TermOut), hipMemcpyDeviceToHost, s));
  JG_HIP(hipStreamSynchronize(s));
  for (size_t k = 0; k < jobs.size(); ++k) {
    int32_t *dst = results + (owner[k] / 2) * 10 + (owner[k] % 2) * 5;
    dst[0] = host[k].score; dst[1] = host[k].len; dst[2] = host[k].fgaps; dst[3] = host[k].end_q; dst[4] = host[k].end_r;
  }
  (void)hipFree(d_jobs);
  (void)hipFree(d_out);
  if (tmp_bases) (void)hipFree(tmp_bases);
  return JG_OK;
}
