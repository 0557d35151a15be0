// Low-complexity soft-masking (symmetric DUST, Morgulis et al. 2006, J Comput Biol 13:1028) for the
// fragmenter: the reference calls pydustmasker.DustMasker(seq, window_size=64, score_threshold=20)
// .mask() on every upper-cased contig (seqops/io.py:104-108), which lower-cases the union of all
// "perfect intervals" (sub-intervals of a W-base window whose triplet score 10*sum c_t(c_t-1)/2 /
// (l-1) exceeds T and is maximal).  This is a host-side restatement of the published streaming
// algorithm (window deque of triplets, running triplet counts for the window and for its
// unsaturated suffix, list of perfect intervals of the current window); an ambiguous base ends the
// current stretch and the scan restarts behind it.  pydustmasker is not installable here: parity
// with it is UNPINNED (tests hold the algorithm's defining properties instead).
// Host code only, one std::thread per slice of records.
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
static int usable_cores() {
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
  int nt = n_threads > 0 ? n_threads : usable_cores();
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
