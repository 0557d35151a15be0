// Low-complexity soft-masking (symmetric DUST, Morgulis et al. 2006, J Comput Biol 13:1028) for the
// fragmenter: the reference calls pydustmasker.DustMasker(seq, window_size=64, score_threshold=20)
// .mask() on every upper-cased contig (seqops/io.py:104-108), which lower-cases the union of all
// "perfect intervals" (sub-intervals of a W-base window whose triplet score 10*sum c_t(c_t-1)/2 /
// (l-1) exceeds T and is maximal).  This is a host-side restatement of the published streaming
// algorithm (window deque of triplets, running triplet counts for the window and for its
// unsaturated suffix, list of perfect intervals of the current window); an ambiguous base ends the
// current stretch and the scan restarts behind it.  pydustmasker is not installable here: parity
// with it is UNPINNED (tests hold the algorithm's defining properties instead).
// jg_dust_mask: host code, one std::thread per slice of records.  jg_dust_mask_device (bottom of the file): the same
// masks from the DEFINITION of symmetric DUST, evaluated on the GPU for device-resident bases - all intervals of up to
// W - 2 triplets by dynamic programme, one thread per interval start (bit-identical to the host scan, tests/test_gpu_dust.py).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jg_common.h"

namespace {

constexpr int WLEN = 3, WTOT = 64, WMSK = 63;

struct Perf { int start, finish, r, l; };

struct Dust {
  int W, T;
  std::vector<int> ring;       // window of triplet codes, capacity W
  int front = 0, count = 0;
  std::vector<Perf> P;         // perfect intervals of the current window: descending start
  std::vector<std::pair<int, int>> res;
  int cv[WTOT], cw[WTOT], rv = 0, rw = 0, L = 0;

  Dust(int w, int t) : W(w), T(t), ring((size_t)w + 1) { reset(); }
  void reset() {
    front = count = 0;
    rv = rw = L = 0;
    memset(cv, 0, sizeof(cv));
    memset(cw, 0, sizeof(cw));
  }
  int at(int i) const { return ring[(size_t)((front + i) % (int)ring.size())]; }
  void push(int t) { ring[(size_t)((front + count) % (int)ring.size())] = t; ++count; }
  int shift() { const int s = ring[(size_t)front]; front = (front + 1) % (int)ring.size(); --count; return s; }

  void shift_window(int t) {
    if (count >= W - WLEN + 1) {
      const int s = shift();
      rw -= --cw[s];
      if (L > count) { --L; rv -= --cv[s]; }
    }
    push(t);
    ++L;
    rw += cw[t]++;
    rv += cv[t]++;
    if (cv[t] * 10 > T * 2) {
      int s;
      do {
        s = at(count - L);
        rv -= --cv[s];
        --L;
      } while (s != t);
    }
  }

  void save_masked(int start) {
    if (P.empty() || P.back().start >= start) return;
    const Perf &p = P.back();
    bool saved = false;
    if (!res.empty() && p.start <= res.back().second) {
      res.back().second = std::max(res.back().second, p.finish);
      saved = true;
    }
    if (!saved) res.emplace_back(p.start, p.finish);
    while (!P.empty() && P.back().start < start) P.pop_back();
  }

  void find_perfect(int start) {
    int c[WTOT];
    memcpy(c, cv, sizeof(c));
    int r = rv, max_r = 0, max_l = 0;
    for (int i = count - L - 1; i >= 0; --i) {
      const int t = at(i);
      r += c[t]++;
      const int new_r = r, new_l = count - i - 1;
      if (new_r * 10 > T * new_l) {
        size_t j = 0;
        for (; j < P.size() && P[j].start >= i + start; ++j) {
          const Perf &p = P[j];
          if (max_r == 0 || p.r * max_l > max_r * p.l) { max_r = p.r; max_l = p.l; }
        }
        if (max_r == 0 || new_r * max_l >= max_r * new_l) {
          max_r = new_r;
          max_l = new_l;
          P.insert(P.begin() + (long)j, Perf{i + start, count + (WLEN - 1) + start, new_r, new_l});
        }
      }
    }
  }

  // intervals [start, finish) of `seq` (ACGT any case; everything else breaks the stretch)
  void run(const uint8_t *seq, int64_t n) {
    res.clear();
    P.clear();
    reset();
    int l = 0;
    unsigned t = 0;
    for (int64_t i = 0; i <= n; ++i) {
      int b = 4;
      if (i < n) {
        switch (seq[i]) {
          case 'A': case 'a': b = 0; break;
          case 'C': case 'c': b = 1; break;
          case 'G': case 'g': b = 2; break;
          case 'T': case 't': b = 3; break;
          default: break;
        }
      }
      if (b < 4) {
        ++l;
        t = (t << 2 | (unsigned)b) & WMSK;
        if (l >= WLEN) {
          const int start = (int)((l - W > 0 ? l - W : 0) + (i + 1 - l));
          save_masked(start);
          shift_window((int)t);
          if (rw * 10 > L * T) find_perfect(start);
        }
      } else {
        int start = (int)((l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l));
        while (!P.empty()) save_masked(start++);
        l = 0;
        t = 0;
        reset();
      }
    }
  }
};

}  // namespace

// Upper-case every base, then lower-case the DUST intervals, record by record, in place.
// cores this process may actually use: the affinity mask, cut down to the cgroup CPU quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us) - a container can see 256 cores and own 16; threads beyond the quota only
// preempt each other.  Under torchrun every rank masks its own contigs, so the quota is also shared by the ranks.
int jg_usable_cores() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  long long q = -1, per = -1;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char qs[32] = "";
    if (fscanf(f, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0) q = atoll(qs);
    fclose(f);
  } else {
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
    if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = -1; fclose(g); }
  }
  if (q > 0 && per > 0) n = std::min<long long>(n, std::max<long long>(1, (q + per / 2) / per));
  // one process per GPU: the ranks of a node share these cores (torchrun / bench.py export LOCAL_WORLD_SIZE)
  if (const char *lw = getenv("LOCAL_WORLD_SIZE")) {
    const int ranks = atoi(lw);
    if (ranks > 1) n = n / ranks;
  }
  return std::max(1, n);
}

extern "C" int jg_dust_mask(uint8_t *bases, const int64_t *offsets, int64_t n_records, int32_t window,
                            int32_t threshold, int32_t n_threads, int64_t *n_masked) {
  JG_REQUIRE(bases != nullptr && offsets != nullptr && n_records >= 0 && window >= 4 && window <= 4096 &&
                 threshold > 0,
             JG_ERR_INVALID, "jg_dust_mask: bad arguments");
  for (int64_t r = 0; r < n_records; ++r)
    JG_REQUIRE(offsets[r + 1] >= offsets[r] && offsets[r + 1] - offsets[r] < (int64_t)2000000000,
               JG_ERR_INVALID, "jg_dust_mask: record %lld length out of range", (long long)r);
  int nt = n_threads > 0 ? n_threads : jg_usable_cores();
  nt = std::max(1, std::min(nt, 256));
  if ((int64_t)nt > n_records) nt = (int)std::max<int64_t>(1, n_records);
  std::vector<int64_t> masked((size_t)nt, 0);
  // contiguous slices of records with about equal base counts
  const int64_t total = n_records > 0 ? offsets[n_records] - offsets[0] : 0;
  auto work = [&](int tix) {
    const int64_t lo_b = offsets[0] + total * tix / nt, hi_b = offsets[0] + total * (tix + 1) / nt;
    int64_t r0 = std::lower_bound(offsets, offsets + n_records, lo_b) - offsets;
    int64_t r1 = tix == nt - 1 ? n_records : std::lower_bound(offsets, offsets + n_records, hi_b) - offsets;
    Dust d(window, threshold);
    for (int64_t r = r0; r < r1; ++r) {
      uint8_t *s = bases + offsets[r];
      const int64_t n = offsets[r + 1] - offsets[r];
      for (int64_t i = 0; i < n; ++i)
        if (s[i] >= 'a' && s[i] <= 'z') s[i] -= 32;
      d.run(s, n);
      for (const auto &iv : d.res) {
        for (int q = iv.first; q < iv.second; ++q) s[q] |= 0x20;
        masked[(size_t)tix] += iv.second - iv.first;
      }
    }
  };
  std::vector<std::thread> pool;
  for (int tix = 1; tix < nt; ++tix) pool.emplace_back(work, tix);
  work(0);
  for (auto &th : pool) th.join();
  if (n_masked != nullptr) {
    int64_t sum = 0;
    for (int64_t v : masked) sum += v;
    *n_masked = sum;
  }
  return JG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric DUST on the GPU, by the definition (Morgulis et al. 2006): an interval of l triplets (2 <= l <= W - 2) with
// repeat count r = sum_t c_t (c_t - 1) / 2 scores r / (l - 1); it is PERFECT when its score exceeds T / 10 and no
// sub-interval scores higher; the mask is the union of the perfect intervals.  One thread per interval start s walks
// l = 1 .. W - 2: r grows by the number of earlier occurrences of the new triplet (a 64-entry byte table per thread,
// LDS, [triplet][thread]); M(s, l) = max(score(s, l), M(s, l - 1), M(s + 1, l - 1)) is the best score of any
// sub-interval, exchanged with the neighbour through LDS once per step.  Scores are compared as integer fractions
// (r <= 1 891, l <= 62: exact).  A workgroup owns 640 base positions and computes 64 starts either side of them:
// an interval that covers an owned position starts at most 63 bases before it, and a start needs its <= 61 right
// neighbours.  Triplets that hold a non-ACGT byte or straddle a record boundary exist in no interval.  HBM: one byte
// read (+ a 1.2x halo from L2) and at most one written per base - 0.8 GB for 400 Mbp, 0.1 ms of HBM time; the kernel
// is bound by the 62 dependent DP steps (~30 vector / LDS instructions and a barrier each per start: ~30 ms of vector
// issue for 400 Mbp on 256 CUs).  Measured 39 ms = 10.3 Gbp/s (workgroups of 768 = three waves per SIMD, two per CU;
// 384 threads: 64 ms, 512: 40, 640: 55, 1 024: 53), the host scan on 16 threads 1.5 Gbp/s.  Round 5: a barrier-free
// filter in front of the programme (does ANY interval of the tile score above the threshold?) lets the tiles of
// ordinary sequence leave after the repeat counts alone.
namespace {

constexpr int DT = 768, DOWN = 640, DHALO = 64;
constexpr int DPITCH = DT + 4;      // bytes per triplet row of the count table: 97 dwords, so that the rows of different
                                    // triplets start in different LDS banks

__device__ __forceinline__ bool frac_ge(int an, int ad, int bn, int bd) { return an * bd >= bn * ad; }   // an/ad >= bn/bd

__global__ __launch_bounds__(DT) void dust_kernel(uint8_t *bases, int64_t origin, int64_t span_len,
                                                  const int64_t *rec_off, int64_t n_rec, int lmax, int T,
                                                  int64_t own0, int64_t own1, unsigned long long *n_masked) {
  __shared__ uint32_t cnt32[64 * DPITCH / 4];      // byte table [triplet][thread]
  __shared__ uint8_t code[DT + 68];                // bits 0-2: 0-3 = A C G T, 4 = other; bit 3: last base of its record
  __shared__ uint8_t tri[DT + 64];                 // triplet code, 0xff = no triplet starts here
  __shared__ short2 Mbuf[2][DT + 1];              // (repeat count <= 1 891, triplets - 1 <= 61)
  __shared__ int endp[DT];
  uint8_t *cnt = reinterpret_cast<uint8_t *>(cnt32);
  const int j = threadIdx.x;
  const int64_t tile0 = own0 + (int64_t)blockIdx.x * DOWN - DHALO;     // global position of thread 0's start
  // record of the tile's first position inside the records: one (workgroup-uniform) binary search; every position
  // then walks forward from it - records are mostly far longer than a tile
  const int64_t first = rec_off[0], last = rec_off[n_rec];
  int64_t r0 = 0;
  {
    const int64_t p0 = tile0 < first ? first : tile0;
    int64_t lo = 0, hi = n_rec;                     // rec_off[lo] <= p0 < rec_off[hi] (when p0 < last)
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (rec_off[mid] <= p0) lo = mid; else hi = mid;
    }
    r0 = lo;
  }
  for (int q = j; q < DT + 66; q += DT) {
    const int64_t pos = tile0 + q, loc = pos - origin;
    int c = 4;
    if (loc >= 0 && loc < span_len && pos >= first && pos < last) {
      switch (bases[loc]) {
        case 'A': case 'a': c = 0; break;
        case 'C': case 'c': c = 1; break;
        case 'G': case 'g': c = 2; break;
        case 'T': case 't': c = 3; break;
        default: break;
      }
      // record of pos: the last r with rec_off[r] <= pos; pos is its last base when pos + 1 == rec_off[r + 1]
      int64_t r = r0;
      while (r + 1 < n_rec && rec_off[r + 1] <= pos) ++r;
      if (pos + 1 == rec_off[r + 1]) c |= 8;
    }
    code[q] = (uint8_t)c;
  }
  for (int q = j; q < 64 * DPITCH / 4; q += DT) cnt32[q] = 0u;
  Mbuf[0][j] = Mbuf[1][j] = make_short2(-1, 1);
  if (j == 0) Mbuf[0][DT] = Mbuf[1][DT] = make_short2(-1, 1);
  __syncthreads();
  for (int q = j; q < DT + 64; q += DT) {
    const int a = code[q], b = code[q + 1], c = code[q + 2];
    const bool ok = (a & 7) < 4 && (b & 7) < 4 && (c & 7) < 4 && !(a & 8) && !(b & 8);
    tri[q] = ok ? (uint8_t)((a & 3) * 16 + (b & 3) * 4 + (c & 3)) : (uint8_t)0xff;
  }
  __syncthreads();
  // Filter: a perfect interval scores above T / 10, so a tile in which NO interval (any start, any length) does has
  // nothing to mask - and the repeat counts alone need neither the neighbour exchange nor a barrier per step.  Most
  // tiles of ordinary sequence stop here; the others clear the count table and run the dynamic programme.
  {
    // the thread's counter of a triplet is a byte of the dword it shares with three neighbours: one returning LDS add
    // per step, nothing in the loop waits for the step before it (a dead start - its triplet chain broken - adds zero)
    bool hot = false, live = true;
    int rf = 0;
    const int sh = 8 * (j & 3);
    uint32_t *col = cnt32 + (j >> 2);
#pragma unroll 4
    for (int l = 1; l <= lmax; ++l) {
      const unsigned t = tri[j + l - 1];
      live = live && t != 0xffu;
      const uint32_t old = atomicAdd(col + (t & 63u) * (DPITCH / 4), live ? 1u << sh : 0u);
      rf += live ? (int)((old >> sh) & 0xffu) : 0;
      hot = hot || 10 * rf > T * (l - 1);          // (l = 1: r = 0; a dead start's count no longer grows)
    }
    if (__syncthreads_or(hot ? 1 : 0) == 0) {
      if (j >= DHALO && j < DHALO + DOWN) {        // nothing masked: the owned bases are upper-cased, that is all
        const int64_t pos = tile0 + j, loc = pos - origin;
        if (pos >= own0 && pos < own1 && loc >= 0 && loc < span_len) {
          const uint8_t b = bases[loc];
          if (b >= 'a' && b <= 'z') bases[loc] = (uint8_t)(b - 32);
        }
      }
      return;
    }
    for (int q = j; q < 64 * DPITCH / 4; q += DT) cnt32[q] = 0u;
    __syncthreads();
  }
  bool alive = true;
  int r = 0, reach = 0;
  for (int l = 1; l <= lmax; ++l) {
    const unsigned t = tri[j + l - 1];
    alive = alive && t != 0xffu;
    if (alive) {
      const unsigned c = cnt[t * DPITCH + j];
      r += (int)c;
      cnt[t * DPITCH + j] = (uint8_t)(c + 1);
    }
    if (l >= 2) {
      const short2 a = Mbuf[(l - 1) & 1][j], b = Mbuf[(l - 1) & 1][j + 1];
      short2 m = frac_ge(a.x, a.y, b.x, b.y) ? a : b;         // best score of any proper sub-interval
      if (alive && frac_ge(r, l - 1, m.x, m.y)) {
        if (10 * r > T * (l - 1)) reach = l + 2;               // perfect: masks bases [s, s + l + 2)
        m = make_short2((short)r, (short)(l - 1));
      }
      Mbuf[l & 1][j] = m;
    }
    __syncthreads();
  }
  endp[j] = reach > 0 ? j + reach : 0;
  __syncthreads();
  bool masked = false;
  if (j >= DHALO && j < DHALO + DOWN) {
    for (int k = 0; k < DHALO; ++k) masked = masked || endp[j - k] > j;
    const int64_t pos = tile0 + j, loc = pos - origin;
    if (pos >= own0 && pos < own1 && loc >= 0 && loc < span_len) {
      const uint8_t b = bases[loc];
      uint8_t u = (b >= 'a' && b <= 'z') ? (uint8_t)(b - 32) : b;
      if (masked) u |= 0x20;
      if (u != b) bases[loc] = u;
    } else {
      masked = false;
    }
  }
  const unsigned long long bal = __ballot(masked);
  if ((j & 63) == 0 && bal != 0ull && n_masked != nullptr) atomicAdd(n_masked, (unsigned long long)__popcll(bal));
}

}  // namespace

// device-side launch shared with jg_predict_windows (jg_api.hip): rec_off is a DEVICE array of n_rec + 1 global offsets
int jg_launch_dust(uint8_t *d_bases, int64_t origin, int64_t span_len, const int64_t *d_rec_off, int64_t n_rec,
                   int window, int threshold, int64_t own0, int64_t own1, unsigned long long *d_masked, hipStream_t s) {
  JG_REQUIRE(window >= 4 && window <= 64 && threshold > 0 && threshold <= 1000000, JG_ERR_UNSUPPORTED,
             "dust (device): window %d outside 4..64 or threshold %d out of range (use the host scan)", window, threshold);
  if (own1 <= own0 || n_rec <= 0) return JG_OK;
  const int64_t blocks = (own1 - own0 + DOWN - 1) / DOWN;
  JG_REQUIRE(blocks < (int64_t)2147483647, JG_ERR_UNSUPPORTED, "dust (device): %lld bases in one launch", (long long)(own1 - own0));
  hipLaunchKernelGGL(dust_kernel, dim3((unsigned)blocks), dim3(DT), 0, s, d_bases, origin, span_len, d_rec_off, n_rec,
                     window - 2, threshold, own0, own1, d_masked);
  JG_HIP(hipGetLastError());
  return JG_OK;
}

// Soft-mask device-resident bases in place (all records of the buffer): upper-case everything, lower-case the DUST
// intervals.  offsets: n_records + 1 entries, host or device (offsets_loc).  n_masked (host, optional) forces a sync.
extern "C" int jg_dust_mask_device(jg_engine *e, uint8_t *d_bases, int64_t n_bases, const int64_t *offsets, int offsets_loc,
                                   int64_t n_records, int32_t window, int32_t threshold, int64_t *n_masked, void *stream) {
  JG_REQUIRE(e != nullptr && d_bases != nullptr && offsets != nullptr && n_records >= 0 && n_bases >= 0, JG_ERR_INVALID,
             "jg_dust_mask_device: bad arguments");
  JG_HIP(hipSetDevice(e->dev));
  hipStream_t s = stream != nullptr ? static_cast<hipStream_t>(stream) : e->stream;
  if (n_masked != nullptr) *n_masked = 0;
  if (n_records == 0 || n_bases == 0) return JG_OK;
  void *tmp = nullptr;
  const int64_t *d_off = offsets;
  if (offsets_loc == JG_PTR_HOST) {
    JG_REQUIRE(offsets[0] >= 0 && offsets[n_records] <= n_bases, JG_ERR_INVALID, "jg_dust_mask_device: records outside the base buffer");
    JG_HIP(hipMalloc(&tmp, (size_t)(n_records + 1) * sizeof(int64_t) + 8));
    JG_HIP(hipMemcpyAsync(tmp, offsets, (size_t)(n_records + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    d_off = static_cast<const int64_t *>(tmp);
  }
  unsigned long long *d_cnt = nullptr;
  int rc = JG_OK;
  if (n_masked != nullptr) {
    hipError_t er = hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long));
    if (er == hipSuccess) er = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s);
    if (er != hipSuccess) { jg_set_error("jg_dust_mask_device: %s", hipGetErrorString(er)); rc = JG_ERR_NOMEM; }
  }
  if (rc == JG_OK) rc = jg_launch_dust(d_bases, 0, n_bases, d_off, n_records, window, threshold, 0, n_bases, d_cnt, s);
  if (rc == JG_OK && n_masked != nullptr) {
    unsigned long long h = 0;
    hipError_t er = hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, s);
    if (er == hipSuccess) er = hipStreamSynchronize(s);
    if (er != hipSuccess) { jg_set_error("jg_dust_mask_device: %s", hipGetErrorString(er)); rc = JG_ERR_HIP; }
    *n_masked = (int64_t)h;
  } else if (tmp != nullptr) {
    (void)hipStreamSynchronize(s);                 // the offsets copy is freed below
  }
  if (d_cnt) (void)hipFree(d_cnt);
  if (tmp) (void)hipFree(tmp);
  return rc;
}
