// Per-contig reductions of per-window arrays (host only): the statistics pred_to_dict takes contig by contig
// (postprocess/collect.py:332-356,393-395 - np.mean / np.var of every contig's (T, C) logit block, np.mean of its 1-D
// entropy / energy / G+C / N% slices) for ALL contigs of a batch in one call, on every usable core.
//
// The table bytes depend on the summation ORDER (results are rounded to fp16 and printed with three decimals), so both
// reductions restate numpy's own:
//   * along the window axis of a (T, C) block numpy adds row after row into the output row - sequential in f32;
//     np.var subtracts the (already divided) mean, squares, and sums the same way;
//   * along a contiguous 1-D slice numpy sums PAIRWISE (umath/loops_utils.h.src: fewer than 8 items sequentially; up to
//     128 items with eight accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and a sequential tail; more by
//     halving at a multiple of eight), in chunks of the iterator's buffer size (8 192 items) added up in order.
// No fused multiply-add may replace a rounded product here: contraction is switched off for this file's arithmetic.
// tests/test_postprocess.py compares with numpy on every length from 1 to 1 100 and on random segmentations.
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

#include "jg_common.h"

namespace {

template <typename T>
T pairwise_sum(const T *a, int64_t n) {
#pragma clang fp contract(off)
  if (n < 8) {
    T res = 0;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// numpy's reduction loop hands the inner loop at most `bufsize` (8 192) items at a time: a longer slice is summed chunk by
// chunk, every chunk pairwise, the chunk sums added up in order
template <typename T>
T numpy_sum(const T *a, int64_t n) {
#pragma clang fp contract(off)
  constexpr int64_t CHUNK = 8192;
  T res = pairwise_sum(a, std::min(n, CHUNK));
  for (int64_t off = CHUNK; off < n; off += CHUNK) res += pairwise_sum(a + off, std::min(CHUNK, n - off));
  return res;
}

template <typename F>
void parallel_segments(int64_t n_seg, int64_t work_items, int32_t n_threads, F &&body) {
  int nt = n_threads > 0 ? n_threads : jg_usable_cores();
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 64), work_items / 65536 + 1));
  if (nt == 1) {
    body(0, n_seg);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { body(n_seg * t / nt, n_seg * (t + 1) / nt); });
  for (auto &x : th) x.join();
}

}  // namespace

extern "C" int jg_segment_mean_var(const float *x, int64_t n_rows, int32_t n_cols, const int64_t *first, const int64_t *count,
                                   int64_t n_seg, float *mean, float *var, int32_t n_threads) {
  JG_REQUIRE(x != nullptr && first != nullptr && count != nullptr && mean != nullptr && n_cols > 0 && n_cols <= 4096 &&
                 n_seg >= 0 && n_rows >= 0,
             JG_ERR_INVALID, "jg_segment_mean_var: bad arguments");
  for (int64_t s = 0; s < n_seg; ++s)
    JG_REQUIRE(first[s] >= 0 && count[s] >= 1 && first[s] + count[s] <= n_rows, JG_ERR_INVALID,
               "jg_segment_mean_var: segment %lld [%lld, +%lld) outside the %lld rows", (long long)s, (long long)first[s],
               (long long)count[s], (long long)n_rows);
  parallel_segments(n_seg, n_rows * n_cols, n_threads, [&](int64_t s0, int64_t s1) {
#pragma clang fp contract(off)
    std::vector<float> acc((size_t)n_cols), sq;
    for (int64_t s = s0; s < s1; ++s) {
      const float *b = x + first[s] * n_cols;
      const int64_t t_n = count[s];
      const float div = (float)t_n;
      float *m = mean + s * n_cols;
      if (n_cols == 1) {
        // a single column: numpy drops the length-one axis, the window axis becomes the contiguous inner loop - pairwise
        m[0] = numpy_sum(b, t_n) / div;
        if (var != nullptr) {
          sq.resize((size_t)t_n);
          for (int64_t t = 0; t < t_n; ++t) { const float d = b[t] - m[0]; sq[(size_t)t] = d * d; }
          var[s] = numpy_sum(sq.data(), t_n) / div;
        }
        continue;
      }
      for (int c = 0; c < n_cols; ++c) acc[(size_t)c] = b[c];                    // the first row, then row after row
      for (int64_t t = 1; t < t_n; ++t)
        for (int c = 0; c < n_cols; ++c) acc[(size_t)c] += b[t * n_cols + c];
      for (int c = 0; c < n_cols; ++c) m[c] = acc[(size_t)c] / div;
      if (var != nullptr) {
        float *v = var + s * n_cols;
        for (int c = 0; c < n_cols; ++c) { const float d = b[c] - m[c]; acc[(size_t)c] = d * d; }
        for (int64_t t = 1; t < t_n; ++t)
          for (int c = 0; c < n_cols; ++c) {
            const float d = b[t * n_cols + c] - m[c];
            const float sq = d * d;
            acc[(size_t)c] += sq;
          }
        for (int c = 0; c < n_cols; ++c) v[c] = acc[(size_t)c] / div;
      }
    }
  });
  return JG_OK;
}

extern "C" int jg_segment_mean_1d(const void *v, int32_t is_f64, int64_t n, const int64_t *first, const int64_t *count,
                                  int64_t n_seg, void *out, int32_t n_threads) {
  JG_REQUIRE(v != nullptr && first != nullptr && count != nullptr && out != nullptr && n_seg >= 0 && n >= 0, JG_ERR_INVALID,
             "jg_segment_mean_1d: bad arguments");
  for (int64_t s = 0; s < n_seg; ++s)
    JG_REQUIRE(first[s] >= 0 && count[s] >= 1 && first[s] + count[s] <= n, JG_ERR_INVALID,
               "jg_segment_mean_1d: segment %lld [%lld, +%lld) outside the %lld items", (long long)s, (long long)first[s],
               (long long)count[s], (long long)n);
  parallel_segments(n_seg, n, n_threads, [&](int64_t s0, int64_t s1) {
#pragma clang fp contract(off)
    if (is_f64) {
      const double *a = static_cast<const double *>(v);
      double *o = static_cast<double *>(out);
      for (int64_t s = s0; s < s1; ++s) o[s] = numpy_sum(a + first[s], count[s]) / (double)count[s];
    } else {
      const float *a = static_cast<const float *>(v);
      float *o = static_cast<float *>(out);
      for (int64_t s = s0; s < s1; ++s) o[s] = numpy_sum(a + first[s], count[s]) / (float)count[s];
    }
  });
  return JG_OK;
}
