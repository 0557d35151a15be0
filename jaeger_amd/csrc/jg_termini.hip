// Terminal-repeat scan (utils/termini.py:88-189): for every contig the first `scan` bases are locally
// aligned (Smith-Waterman, match +2 / mismatch -100, gap open 100 / extend 5 - a gap of k costs
// 100 + 5(k-1)) against the last `scan` bases (direct terminal repeat, DTR) and against their reverse
// complement (inverted, ITR), scan = min(max(int(0.04 len), 400), 4000).  The reference runs parasail's
// sw_trace_scan_16 on CPU threads and then only uses, per alignment: the score, the alignment length
// (traceback columns) and the gap count in the query row.  Those three are carried through the DP here
// as auxiliary values of each state's best path, so no traceback matrix is stored.
//
// Two kernels.  termini_fast_kernel (below the exact one) scores BOTH alignments of a record in one wave, packed as two
// unsigned 16-bit lanes per register, without the auxiliary values: 7 instructions per cell instead of 35.  A best score
// <= 100 fixes everything else: under match +2 / mismatch -100 / gap open 100 a path that never exceeds 100 cannot hold a
// mismatch or a gap (the score behind one would be <= 0), so the alignment is an exact-match run of score / 2 columns,
// no gaps, and its end cell is the first cell in (column, row) order that closes a run of that length - the wave reports
// the column and the 64-row strip, the host finds the row among the strip's rows.  Only alignments scoring above 100
// (real repeats of more than 50 bases) go through termini_kernel.  Same five numbers either way (tests/test_gpu_termini.py
// runs every case through both).
//
// termini_kernel: one workgroup per (contig, DTR|ITR) job.  Thread t owns the query rows [t*R, (t+1)*R) (R <= 16) and walks
// its strip column by column one step behind thread t-1 (a systolic wavefront): the only values that cross
// threads are the H / F states of each strip's last row, double-buffered in LDS, one barrier per step.
// Letters are compared case-insensitively and only A/C/G/T can match (parasail.matrix_create("ACGT", 2, -100)).
// Ties: H prefers the diagonal, then the gap in the query row (E), then F; E and F prefer extension; the best
// cell is the first maximum in (column, row) order.  parasail is not installable here: parity with its
// tie-breaking is unpinned (scores are unique; lengths can differ only between co-optimal alignments).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

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


// ---- fast path: score, end column and row strip of both alignments of a record, one wave per record -----------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

struct TermRec {
  int64_t q_off;   // query = bases[q_off + i]; DTR ref[j] = bases[r_off + j]; ITR ref[j] = complement(bases[r_off + n - 1 - j])
  int64_t r_off;
  int32_t n;
  int32_t pad_;
};

struct TermFast {
  int32_t score[2], col[2], lane[2];   // [0] DTR, [1] ITR; col / lane = -1 when the score is 0
};

__device__ __forceinline__ us2 sat_sub(us2 a, unsigned short b) { return __builtin_elementwise_sub_sat(a, (us2){b, b}); }
__device__ __forceinline__ us2 pk_max(us2 a, us2 b) { return __builtin_elementwise_max(a, b); }

// rows per lane of the fast kernel's instantiations; a record runs on the smallest one that covers its scan length
__host__ __device__ inline int fast_rows(int n) {
  const int need = (n + 63) >> 6;
  const int cls[8] = {7, 8, 12, 16, 24, 32, 48, 64};
  for (int k = 0; k < 8; ++k)
    if (cls[k] >= need) return cls[k];
  return 64;
}

// packed-u16 mismatch penalty: 0 where the two codes of a half agree, 102 (= match - mismatch) where they differ.  Written
// as two instructions by hand: the compiler turns min(x, 1) * 102 into a compare / select / permute chain of five
__device__ __forceinline__ unsigned pk_penalty(unsigned x, unsigned k102) {
  unsigned m, p;
  asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(m) : "v"(x));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(p) : "v"(m), "v"(k102));
  return p;
}

// Lane t owns query rows [t R, (t + 1) R) and walks its strip column by column one step behind lane t - 1 (the systolic
// wavefront of termini_kernel inside ONE wave: strip edges cross lanes by DPP wave_shr, no LDS traffic, no barrier).
// States are unsigned and floored at 0: H is >= 0 by definition, and an E / F below 0 can neither win a maximum against
// H >= 0 nor turn positive again by extension, so max(., 0) of them gives the same H everywhere.  Rows beyond the scan
// length and columns outside the matrix carry letters that match nothing: their cells only ever hold decayed copies of
// real scores, which cannot exceed the maximum they came from.
template <int R>
__global__ __launch_bounds__(256) void termini_fast_kernel(const uint8_t *__restrict__ bases, const TermRec *__restrict__ recs,
                                                           TermFast *__restrict__ out, int n_recs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rec = blockIdx.x * 4 + wave;
  unsigned *ref = reinterpret_cast<unsigned *>(smem) + wave * (64 * R);   // ref[j] = DTR code | ITR code << 16
  const bool live = rec < n_recs;
  const TermRec job = recs[live ? rec : 0];
  const int n = live ? job.n : 0;
  for (int j = lane; j < n; j += 64) {
    const int d = base_code(bases[job.r_off + j]);
    int c = base_code(bases[job.r_off + n - 1 - j]);
    if (c < 4) c = 3 - c;
    ref[j] = (unsigned)d | ((unsigned)c << 16);
  }
  __syncthreads();
  const int i0 = lane * R;
  unsigned q2[R];
  us2 hl[R], el[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = i0 + r;
    int c = 5;                                                     // 5: the row does not exist
    if (i < n) { c = base_code(bases[job.q_off + i]); if (c == 4) c = 6; }   // 6: a letter that matches nothing
    q2[r] = (unsigned)c * 0x10001u;
    hl[r] = (us2){0, 0};
    el[r] = (us2){0, 0};
  }
  const unsigned k102 = (unsigned)(S_MATCH - S_MISMATCH) * 0x10001u;
  unsigned best = 0, pub_h = 0, pub_f = 0, prev_up = 0;
  int col_lo = -1, col_hi = -1;
  const int steps = n > 0 ? n + 63 : 0;
  for (int step = 0; step < steps; ++step) {
    const int j = step - lane;
    const unsigned rc2 = (j >= 0 && j < n) ? ref[j] : 0x00040004u;   // outside the matrix: a letter that matches nothing
    const unsigned up_hu = __builtin_amdgcn_update_dpp(0u, pub_h, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 reads 0
    const unsigned up_fu = __builtin_amdgcn_update_dpp(0u, pub_f, 0x138, 0xf, 0xf, false);
    us2 dg = __builtin_bit_cast(us2, prev_up);
    prev_up = up_hu;
    us2 up_h = __builtin_bit_cast(us2, up_hu), up_f = __builtin_bit_cast(us2, up_fu);
    us2 cm = (us2){0, 0};
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const us2 pen = __builtin_bit_cast(us2, pk_penalty(q2[r] ^ rc2, k102));
      const us2 hd = __builtin_elementwise_sub_sat(dg + (us2){S_MATCH, S_MATCH}, pen);
      const us2 e = pk_max(sat_sub(el[r], G_EXT), sat_sub(hl[r], G_OPEN));
      const us2 f = pk_max(sat_sub(up_f, G_EXT), sat_sub(up_h, G_OPEN));
      const us2 h = pk_max(hd, pk_max(e, f));
      cm = pk_max(cm, h);
      dg = hl[r];
      hl[r] = h;
      el[r] = e;
      up_h = h;
      up_f = f;
    }
    pub_h = __builtin_bit_cast(unsigned, up_h);
    pub_f = __builtin_bit_cast(unsigned, up_f);
    const unsigned nb = __builtin_bit_cast(unsigned, pk_max(__builtin_bit_cast(us2, best), cm));
    if (nb != best) {                                             // a strictly higher score in either half: its first column
      if ((nb & 0xffffu) != (best & 0xffffu)) col_lo = j;
      if ((nb >> 16) != (best >> 16)) col_hi = j;
      best = nb;
    }
  }
  // best cell per half over the wave: max score, then smallest column, then smallest lane (= smallest rows)
  unsigned long long k0 = (best & 0xffffu) ? ((unsigned long long)(best & 0xffffu) << 32) | ((unsigned long long)(0xffff - col_lo) << 16) | (unsigned)(63 - lane) : 0ull;
  unsigned long long k1 = (best >> 16) ? ((unsigned long long)(best >> 16) << 32) | ((unsigned long long)(0xffff - col_hi) << 16) | (unsigned)(63 - lane) : 0ull;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o0 = __shfl_xor(k0, off, 64), o1 = __shfl_xor(k1, off, 64);
    k0 = o0 > k0 ? o0 : k0;
    k1 = o1 > k1 ? o1 : k1;
  }
  if (live && lane == 0) {
    TermFast o;
    o.score[0] = (int)(k0 >> 32); o.col[0] = k0 ? 0xffff - (int)((k0 >> 16) & 0xffff) : -1; o.lane[0] = k0 ? 63 - (int)(k0 & 0xffff) : -1;
    o.score[1] = (int)(k1 >> 32); o.col[1] = k1 ? 0xffff - (int)((k1 >> 16) & 0xffff) : -1; o.lane[1] = k1 ? 63 - (int)(k1 & 0xffff) : -1;
    out[rec] = o;
  }
}

// One thread per (record, DTR | ITR): turn the wave's (score, column, strip) into the five numbers of TermOut when the
// score is <= 100 - the alignment is an exact-match run of score / 2 columns ending in that column, and its row is the
// first row of the strip that closes such a run; len = -1 marks an alignment that needs termini_kernel (score > 100).
__global__ void termini_finish_kernel(const uint8_t *__restrict__ bases, const TermRec *__restrict__ recs,
                                      const TermFast *__restrict__ fast, TermOut *__restrict__ out, int n_recs) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * n_recs) return;
  const int rec = idx >> 1, itr = idx & 1;
  const TermRec job = recs[rec];
  const int n = job.n, S = fast[rec].score[itr], j = fast[rec].col[itr];
  TermOut o{S, 0, 0, -1, -1};
  if (S > 100) {
    o.len = -1;
  } else if (S > 0) {
    const int L = S / 2, R = fast_rows(n), i0 = fast[rec].lane[itr] * R;
    o.len = L;
    o.end_r = j;
    for (int i = i0; i < i0 + R && i < n && o.end_q < 0; ++i) {
      if (i < L - 1 || j < L - 1) continue;
      bool run = true;
      for (int k = 0; k < L && run; ++k) {
        const int a = base_code(bases[job.q_off + i - k]);
        int b = itr ? base_code(bases[job.r_off + n - 1 - (j - k)]) : base_code(bases[job.r_off + j - k]);
        if (itr && b < 4) b = 3 - b;
        run = a < 4 && a == b;
      }
      if (run) o.end_q = i;
    }
  }
  out[idx] = o;
}

// ---- alignments scoring above 100 that are ONE exact run ---------------------------------------------------------------
// Under match +2 / mismatch -100 / gap open 100 an alignment is a chain of exact-match runs joined by steps that cost at
// least 100 each, and a junction only pays when the scores on both sides of it exceed 100: every alignment that scores above
// 100 contains a run of more than 50 matches ("long run"), and attaching anything to it - shorter runs, a mismatch, a gap -
// gains at most nothing.  So when the matrix holds EXACTLY ONE long run, the best alignment is that run: score 2 L, L columns,
// no gap, ending in the run's last cell - which is also the first maximum in (column, row) order (a 50-match run behind a
// mismatch ties the score only in a later column).  That is the shape of every record shorter than twice its scan length
// (a 500-bp record scanned over 400: the two ends overlap in one 300-base run on one diagonal), which otherwise sends a
// million alignments through the length / gap carrying kernel.
// One wave per alignment.  Every long run covers >= 51 reference positions, hence a whole 32-mer starting at a multiple of
// 20: the wave compares those sampled 32-mers of the reference with every 32-mer of the query (first differing base ends a
// comparison: 1.3 bases on random sequence), extends each hit to its maximal run and keeps the smallest and the largest
// (diagonal, first row) it has seen - equal: one run.  Anything else (two runs, none, a run of <= 50) is left to termini_kernel.
constexpr int SR_NMAX = TT * RMAX;        // codes per end and wave in LDS (scan <= 4 096)

__global__ __launch_bounds__(256) void termini_single_kernel(const uint8_t *__restrict__ bases, const TermJob *__restrict__ jobs,
                                                             TermOut *__restrict__ out, int n_jobs) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = blockIdx.x * 4 + wave;
  const bool live = idx < n_jobs;
  const TermJob job = jobs[live ? idx : 0];
  const int n = live ? job.n : 0;
  uint8_t *qc = smem + (size_t)wave * 2 * SR_NMAX, *rc = qc + SR_NMAX;
  for (int j = lane; j < n; j += 64) {
    qc[j] = (uint8_t)base_code(bases[job.q_off + j]);
    int c = job.itr ? base_code(bases[job.r_off + n - 1 - j]) : base_code(bases[job.r_off + j]);
    if (job.itr && c < 4) c = 3 - c;
    rc[j] = (uint8_t)(c < 4 ? c : 7);                               // (a letter that matches nothing, not even another one)
  }
  __syncthreads();
  // (round 6) Hits are handled wave-wide, one at a time: a hit inside the run already known is that run again (nothing to
  // do - the form before extended EVERY hit base by base in its one lane: a 300-base overlap run was walked fifteen times
  // by single lanes, 122 ms per million records); a hit outside it is extended by the whole wave, 64 bases per step, and
  // either becomes the known run or proves a second one - the alignment then is not settled here and the wave leaves.
  bool known = false, second = false;
  int k_diag = 0, k_r0 = 0, k_r1 = -1, k_len = 0, k_end_r = -1;        // the known run: diagonal, first / last query row
  const int n_iter = (n - 32 + 64) / 64;                               // query 32-mers: i = 64 it + lane, i + 32 <= n
  for (int p = 0; p + 32 <= n && !second; p += 20) {
    for (int it = 0; it < n_iter && !second; ++it) {
      const int i = it * 64 + lane;
      bool hit = false;
      if (i + 32 <= n) {
        int t = 0;
        while (t < 32 && qc[i + t] == rc[p + t]) ++t;
        hit = t == 32;
      }
      unsigned long long hits = __ballot(hit);
      while (hits != 0ull) {
        const int hq = it * 64 + __builtin_ctzll(hits);                // query row of this hit (wave-uniform)
        hits &= hits - 1ull;
        if (known && hq - p + n == k_diag && hq >= k_r0 && hq <= k_r1) continue;
        // the maximal run through (hq .. hq + 31, p .. p + 31): left of it, then right of it, 64 cells per step
        int a = 0, bnd = min(hq, p);
        for (int base = 0; base < bnd; base += 64) {
          const int t = base + lane;
          const bool same = t < bnd && qc[hq - 1 - t] == rc[p - 1 - t];
          const unsigned long long diff = ~__ballot(same);
          if (diff != 0ull) { a = base + __builtin_ctzll(diff); break; }
          a = base + 64;
        }
        a = min(a, bnd);
        int bext = 0;
        bnd = n - 32 - max(hq, p);
        for (int base = 0; base < bnd; base += 64) {
          const int t = base + lane;
          const bool same = t < bnd && qc[hq + 32 + t] == rc[p + 32 + t];
          const unsigned long long diff = ~__ballot(same);
          if (diff != 0ull) { bext = base + __builtin_ctzll(diff); break; }
          bext = base + 64;
        }
        bext = min(bext, max(bnd, 0));
        if (known) { second = true; break; }                           // a run that is not the known one
        known = true;
        k_diag = hq - p + n;
        k_r0 = hq - a;
        k_r1 = hq + 31 + bext;
        k_len = a + 32 + bext;
        k_end_r = p + 31 + bext;
      }
    }
  }
  TermOut o{0, -1, 0, -1, -1};                                      // len = -1: not settled here
  if (known && !second && k_len > 50) o = TermOut{2 * k_len, k_len, 0, k_r1, k_end_r};
  if (live && lane == 0) out[idx] = o;
}

// ---- records whose ends share no K matching bases ------------------------------------------------------------------------
// JG_OPT_TERMINI_REPORT_MIN = K: an alignment of fewer than K columns may be reported as none.  Under this scoring an
// alignment of L <= 50 columns is an exact run of L matches (and a longer one contains such a run), so an alignment of at
// least K columns exists exactly when the two ends share K consecutive matching bases - directly (DTR) or with the
// reverse complement (ITR: q[i ..] matches rc(ref) exactly when the reverse complement of the query K-mer occurs in the
// reference as it lies in the record).  One workgroup per record: the reference's K-mers go into an LDS hash set (open
// addressing, 2-bit codes, K <= 15), every query K-mer and its reverse complement is looked up; bit 0 / bit 1 of the
// record's flag = a direct / an inverted alignment of >= K columns exists.  4 000-base scans: 8 000 inserts and probes instead
// of 16 million cells; random sequence shares 13 bases in a quarter of them, 400-base scans in 0.3 %.
constexpr unsigned SEED_EMPTY = 0xffffffffu;

__global__ __launch_bounds__(256) void termini_seed_kernel(const uint8_t *__restrict__ bases, const TermRec *__restrict__ recs,
                                                           int K, uint8_t *__restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const TermRec job = recs[blockIdx.x];
  const int n = job.n, tid = threadIdx.x;
  uint8_t *qc = smem, *rc = smem + SR_NMAX;
  unsigned *table = reinterpret_cast<unsigned *>(smem + 2 * SR_NMAX);
  __shared__ unsigned found;
  int bits = 4;                                                     // slots = the power of two >= 2 n (<= 8 192)
  while ((1 << bits) < 2 * n) ++bits;
  const unsigned mask = (1u << bits) - 1u;
  for (int j = tid; j < n; j += 256) {
    qc[j] = (uint8_t)base_code(bases[job.q_off + j]);
    rc[j] = (uint8_t)base_code(bases[job.r_off + j]);
  }
  for (unsigned q = tid; q <= mask; q += 256) table[q] = SEED_EMPTY;
  if (tid == 0) found = 0u;
  __syncthreads();
  for (int j = tid; j + K <= n; j += 256) {
    unsigned code = 0;
    bool ok = true;
    for (int t = 0; t < K; ++t) { const unsigned c = rc[j + t]; ok = ok && c < 4; code = (code << 2) | (c & 3u); }
    if (!ok) continue;
    unsigned h = (code * 2654435761u) >> (32 - bits);
    for (;;) {
      const unsigned old = atomicCAS(&table[h], SEED_EMPTY, code);
      if (old == SEED_EMPTY || old == code) break;
      h = (h + 1u) & mask;
    }
  }
  __syncthreads();
  unsigned mine = 0;
  auto present = [&](unsigned code) {
    unsigned h = (code * 2654435761u) >> (32 - bits);
    for (;;) {
      const unsigned v = table[h];
      if (v == code) return true;
      if (v == SEED_EMPTY) return false;
      h = (h + 1u) & mask;
    }
  };
  for (int i = tid; i + K <= n; i += 256) {
    unsigned fwd = 0, rev = 0;
    bool ok = true;
    for (int t = 0; t < K; ++t) {
      const unsigned c = qc[i + t];
      ok = ok && c < 4;
      fwd = (fwd << 2) | (c & 3u);
      rev |= (3u - (c & 3u)) << (2 * t);                            // complement, first base last
    }
    if (!ok) continue;
    if (!(mine & 1u) && present(fwd)) mine |= 1u;
    if (!(mine & 2u) && present(rev)) mine |= 2u;
  }
  if (mine) atomicOr(&found, mine);
  __syncthreads();
  if (tid == 0) flags[blockIdx.x] = (uint8_t)found;
}

template <int R>
static int launch_fast(const uint8_t *d_bases, const TermRec *d_recs, TermFast *d_out, int n_recs, hipStream_t s) {
  if (n_recs <= 0) return JG_OK;
  hipLaunchKernelGGL(termini_fast_kernel<R>, dim3((unsigned)((n_recs + 3) / 4)), dim3(256), (size_t)4 * 64 * R * 4, s, d_bases,
                     d_recs, d_out, n_recs);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// ---- the host side's four device stages: upload what the stage reads, launch, bring its results back ------------------
// Device allocation released with its scope - STREAM-ORDERED (hipMallocAsync / hipFreeAsync on the scan's stream): hipFree
// waits for every stream of the device, so the scan - which runs beside the network's forward on a stream of its own - used
// to return only when the forward's whole queue had drained (round 6: the repeat table of a million records arrived with the
// forward's last launch, 0.2 s after the scan's own kernels had ended, and no result row could be written beside the forward).
struct DevBuf {
  void *p = nullptr;
  hipStream_t st = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFreeAsync(p, st); }
  int alloc(size_t bytes, hipStream_t s) {
    st = s;
    JG_HIP(hipMallocAsync(&p, std::max<size_t>(bytes, 1), s));
    return JG_OK;
  }
  template <typename T>
  T *as() const { return static_cast<T *>(p); }
};

template <typename T>
static int upload(DevBuf &d, const std::vector<T> &v, hipStream_t s) {
  int rc = d.alloc(v.size() * sizeof(T), s);
  if (rc != JG_OK) return rc;
  JG_HIP(hipMemcpyAsync(d.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return JG_OK;
}

template <typename T>
static int download(std::vector<T> &v, const DevBuf &d, hipStream_t s) {
  JG_HIP(hipMemcpyAsync(v.data(), d.p, v.size() * sizeof(T), hipMemcpyDeviceToHost, s));
  JG_HIP(hipStreamSynchronize(s));
  return JG_OK;
}

// which alignments of the records have at least k columns at all: bit 0 direct, bit 1 inverted (termini_seed_kernel)
static int run_seeds(const uint8_t *d_bases, const TermRec *d_recs, int n_recs, int k, std::vector<uint8_t> &flags, hipStream_t s) {
  DevBuf d_flags;
  int rc = d_flags.alloc((size_t)n_recs, s);
  if (rc != JG_OK) return rc;
  hipLaunchKernelGGL(termini_seed_kernel, dim3((unsigned)n_recs), dim3(256), (size_t)2 * SR_NMAX + 8192 * sizeof(unsigned), s, d_bases,
                     d_recs, k, d_flags.as<uint8_t>());
  JG_HIP(hipGetLastError());
  flags.resize((size_t)n_recs);
  return download(flags, d_flags, s);
}

// alignments above 100 through the one-run check (exact = false: len = -1 where it settles nothing) or through the kernel
// that carries length and gaps (exact = true)
static int run_jobs(const uint8_t *d_bases, const std::vector<TermJob> &jobs, bool exact, std::vector<TermOut> &out, hipStream_t s) {
  out.resize(jobs.size());
  if (jobs.empty()) return JG_OK;
  DevBuf d_jobs, d_out;
  int rc = upload(d_jobs, jobs, s);
  if (rc == JG_OK) rc = d_out.alloc(jobs.size() * sizeof(TermOut), s);
  if (rc != JG_OK) return rc;
  if (exact) {
    int max_n = 0;
    for (const TermJob &j : jobs) max_n = std::max(max_n, j.n);
    const size_t smem = (size_t)((max_n + 15) & ~15) + 2 * TT * sizeof(int4) + TT * 12;
    hipLaunchKernelGGL(termini_kernel, dim3((unsigned)jobs.size()), dim3(TT), smem, s, d_bases, d_jobs.as<const TermJob>(),
                       d_out.as<TermOut>());
  } else {
    hipLaunchKernelGGL(termini_single_kernel, dim3((unsigned)((jobs.size() + 3) / 4)), dim3(256), (size_t)4 * 2 * SR_NMAX, s, d_bases,
                       d_jobs.as<const TermJob>(), d_out.as<TermOut>(), (int)jobs.size());
  }
  JG_HIP(hipGetLastError());
  return download(out, d_out, s);
}

// the packed pass + its finish kernel over records sorted by scan length (d_recs: their device copy): both alignments of
// every record, out[2 r + {0, 1}]; one launch per run of a strip height
static int run_packed(const uint8_t *d_bases, const std::vector<TermRec> &recs, const TermRec *d_recs, std::vector<TermOut> &out,
                      hipStream_t s) {
  const int n = (int)recs.size();
  out.resize((size_t)2 * n);
  if (n == 0) return JG_OK;
  DevBuf d_fast, d_out;
  int rc = d_fast.alloc((size_t)n * sizeof(TermFast), s);
  if (rc == JG_OK) rc = d_out.alloc((size_t)2 * n * sizeof(TermOut), s);
  TermFast *df = d_fast.as<TermFast>();
  int a = 0;
  while (a < n && rc == JG_OK) {
    const int rows = fast_rows(recs[(size_t)a].n);
    int b = a;
    while (b < n && fast_rows(recs[(size_t)b].n) == rows) ++b;
    switch (rows) {
      case 7: rc = launch_fast<7>(d_bases, d_recs + a, df + a, b - a, s); break;
      case 8: rc = launch_fast<8>(d_bases, d_recs + a, df + a, b - a, s); break;
      case 12: rc = launch_fast<12>(d_bases, d_recs + a, df + a, b - a, s); break;
      case 16: rc = launch_fast<16>(d_bases, d_recs + a, df + a, b - a, s); break;
      case 24: rc = launch_fast<24>(d_bases, d_recs + a, df + a, b - a, s); break;
      case 32: rc = launch_fast<32>(d_bases, d_recs + a, df + a, b - a, s); break;
      case 48: rc = launch_fast<48>(d_bases, d_recs + a, df + a, b - a, s); break;
      default: rc = launch_fast<64>(d_bases, d_recs + a, df + a, b - a, s); break;
    }
    a = b;
  }
  if (rc != JG_OK) return rc;
  hipLaunchKernelGGL(termini_finish_kernel, dim3((unsigned)((2 * n + 255) / 256)), dim3(256), 0, s, d_bases, d_recs, df,
                     d_out.as<TermOut>(), n);
  JG_HIP(hipGetLastError());
  return download(out, d_out, s);
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
  // one pass over the records: scan lengths, and whether the two scanned ends of every record (what a host buffer's upload is
  // cut down to: 32 MB for 10 000 contigs of 400 Mbp) are most of the buffer anyway (a million 500-bp records: 800 MB of
  // ends for 500 MB of bases) - then the buffer goes up as it is
  std::vector<TermRec> recs;
  std::vector<int64_t> owner;
  recs.reserve((size_t)n_records);
  owner.reserve((size_t)n_records);
  int max_n = 0;
  int64_t ends_bytes = 0;
  for (int64_t r = 0; r < n_records; ++r) {
    const int64_t len = offsets[r + 1] - offsets[r];
    JG_REQUIRE(len >= 0 && offsets[r + 1] <= n_bases, JG_ERR_INVALID, "jg_terminal_repeats: record %lld outside the base buffer",
               (long long)r);
    for (int q = 0; q < 10; ++q) results[r * 10 + q] = -1;
    if (len < min_len || len < 1) continue;
    int scan = (int)std::min<int64_t>(std::max<int64_t>((int64_t)((double)len * 0.04), 400), 4000);
    if (scan > len) scan = (int)len;                   // str slicing clamps (termini.py:121-133)
    max_n = std::max(max_n, scan);
    ends_bytes += 2 * (int64_t)scan;
    recs.push_back(TermRec{offsets[r], offsets[r + 1] - scan, scan, 0});
    owner.push_back(r);
  }
  if (recs.empty()) return JG_OK;
  JG_REQUIRE(max_n <= TT * RMAX, JG_ERR_UNSUPPORTED, "jg_terminal_repeats: scan length %d", max_n);
  {
    // one launch per strip height of the packed kernel: records bucketed by it, tallest first (workgroups are dispatched in
    // index order, and a 4 000-base scan costs a hundred 400-base ones); FASTA order inside a bucket
    const int cls[8] = {64, 48, 32, 24, 16, 12, 8, 7};
    std::vector<TermRec> sr;
    std::vector<int64_t> so;
    sr.reserve(recs.size());
    so.reserve(recs.size());
    for (int c = 0; c < 8; ++c)
      for (size_t k = 0; k < recs.size(); ++k)
        if (fast_rows(recs[k].n) == cls[c]) { sr.push_back(recs[k]); so.push_back(owner[k]); }
    recs.swap(sr);
    owner.swap(so);
  }
  std::vector<uint8_t> ends;
  const bool whole = bases_loc == JG_PTR_HOST && ends_bytes * 10 >= n_bases * 6;
  if (bases_loc == JG_PTR_HOST && !whole) {
    ends.resize((size_t)ends_bytes);
    int64_t at = 0;
    for (TermRec &rc : recs) {
      memcpy(ends.data() + at, bases + rc.q_off, (size_t)rc.n);
      memcpy(ends.data() + at + rc.n, bases + rc.r_off, (size_t)rc.n);
      rc.q_off = at;
      rc.r_off = at + rc.n;
      at += 2 * (int64_t)rc.n;
    }
  }
  const int n_recs = (int)recs.size();
  const uint8_t *d_bases = bases;
  DevBuf tmp_bases, d_recs;
  int rc = JG_OK;
  if (bases_loc == JG_PTR_HOST) {
    const uint8_t *src = whole ? bases : ends.data();
    const size_t nb = whole ? (size_t)n_bases : ends.size();
    if ((rc = tmp_bases.alloc(nb, s)) != JG_OK) return rc;
    JG_HIP(hipMemcpyAsync(tmp_bases.p, src, nb, hipMemcpyHostToDevice, s));
    d_bases = tmp_bases.as<const uint8_t>();
  }
  if ((rc = upload(d_recs, recs, s)) != JG_OK) return rc;
  auto job_of = [&](size_t k) { return TermJob{recs[k / 2].q_off, recs[k / 2].r_off, recs[k / 2].n, (int32_t)(k & 1)}; };
  std::vector<TermOut> host(recs.size() * 2);            // alignment k = 2 record + (0 direct | 1 inverted)
  std::vector<uint8_t> settled(host.size(), 0);          // 1: host[k] is final
  const int seed_k = e->termini_exact ? 0 : e->termini_report_min;
  if (seed_k >= 2) {
    // JG_OPT_TERMINI_REPORT_MIN: (1) which alignments have >= seed_k columns at all - the others are reported as none;
    // (2) of those, the ones that are one exact run are settled; the packed pass only sees records with an alignment
    // that is neither
    JG_REQUIRE(seed_k <= 15, JG_ERR_INVALID, "jg_terminal_repeats: JG_OPT_TERMINI_REPORT_MIN = %d (2 .. 15)", seed_k);
    std::vector<uint8_t> flags;
    if ((rc = run_seeds(d_bases, d_recs.as<const TermRec>(), n_recs, seed_k, flags, s)) != JG_OK) return rc;
    std::vector<TermJob> jobs;
    std::vector<size_t> slot;
    for (size_t k = 0; k < host.size(); ++k) {
      if ((flags[k / 2] >> (k & 1)) & 1) {
        jobs.push_back(job_of(k));
        slot.push_back(k);
      } else {
        host[k] = TermOut{0, 0, 0, -1, -1};
        settled[k] = 1;
      }
    }
    std::vector<TermOut> single;
    if ((rc = run_jobs(d_bases, jobs, false, single, s)) != JG_OK) return rc;
    for (size_t k = 0; k < jobs.size(); ++k)
      if (single[k].len > 0) {
        host[slot[k]] = single[k];
        settled[slot[k]] = 1;
      }
  }
  // pass 1: both scores of the records that still have an open alignment, one wave each
  {
    std::vector<int> todo;
    for (int r = 0; r < n_recs; ++r)
      if (!settled[(size_t)2 * r] || !settled[(size_t)2 * r + 1]) todo.push_back(r);
    std::vector<TermRec> sub;                  // (a subset keeps the order: still runs of one strip height)
    DevBuf d_sub;
    const bool all = (int)todo.size() == n_recs;
    if (!all) {
      sub.reserve(todo.size());
      for (int r : todo) sub.push_back(recs[(size_t)r]);
      if ((rc = upload(d_sub, sub, s)) != JG_OK) return rc;
    }
    std::vector<TermOut> part;
    if ((rc = run_packed(d_bases, all ? recs : sub, (all ? d_recs : d_sub).as<const TermRec>(), part, s)) != JG_OK) return rc;
    for (size_t q = 0; q < todo.size(); ++q)
      for (int h = 0; h < 2; ++h) {
        const size_t k = (size_t)2 * todo[q] + h;
        if (!settled[k]) host[k] = part[2 * q + h];
      }
  }
  // pass 2: the alignments that scored above 100 (real repeats).  First the check for "one exact run, nothing else" (a wave
  // each; with a seed length it has already run); what it does not settle - two long runs, mismatches or gaps inside the
  // repeat - goes through the kernel that carries length and gaps
  std::vector<TermJob> jobs;
  std::vector<size_t> slot;
  for (size_t k = 0; k < host.size(); ++k)
    if (!settled[k] && (host[k].len < 0 || e->termini_exact)) {
      jobs.push_back(job_of(k));
      slot.push_back(k);
    }
  if (!jobs.empty() && !e->termini_exact && seed_k < 2) {
    std::vector<TermOut> single;
    if ((rc = run_jobs(d_bases, jobs, false, single, s)) != JG_OK) return rc;
    size_t kept = 0;
    for (size_t k = 0; k < jobs.size(); ++k) {
      // (the score is the packed pass's: a run that is not the whole story would not match it)
      if (single[k].len > 0 && single[k].score == host[slot[k]].score) {
        host[slot[k]] = single[k];
      } else {
        jobs[kept] = jobs[k];
        slot[kept] = slot[k];
        ++kept;
      }
    }
    jobs.resize(kept);
    slot.resize(kept);
  }
  std::vector<TermOut> exact;
  if ((rc = run_jobs(d_bases, jobs, true, exact, s)) != JG_OK) return rc;
  for (size_t k = 0; k < jobs.size(); ++k) host[slot[k]] = exact[k];
  for (size_t k = 0; k < host.size(); ++k) {
    int32_t *dst = results + owner[k / 2] * 10 + (k % 2) * 5;
    dst[0] = host[k].score; dst[1] = host[k].len; dst[2] = host[k].fgaps; dst[3] = host[k].end_q; dst[4] = host[k].end_r;
  }
  return JG_OK;
}
