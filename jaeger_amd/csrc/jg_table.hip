// Result-table text (host only): the bytes pandas writes for
//   df.to_csv(path, sep="\t", index=False, float_format="%.3f")          (postprocess/collect.py:578-580, 602-607)
// from typed columns, on every core the process may use.  A run over an assembly of short contigs spends more time in
// pandas' per-value Python formatter than in the forward (one million 500-bp records: 5.0 s of 7.3 s), so the rows are
// rendered here: floats by an exact "%.3f" (below), integers, booleans ("True" / "False"), strings copied as they are.
// What the csv writer would QUOTE (a tab, a quote, a line break inside a string) is the caller's to detect - it then
// formats that batch with pandas (jaeger_amd/postprocess.py: _tsv_bytes).
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "jg_common.h"

namespace {

inline char *put_u64(char *p, uint64_t v) {
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}

// "%.3f" % v as CPython / glibc print it: the decimal expansion of the EXACT binary value, rounded half-even at the
// third place.  a * 1000 = x + e exactly (e from the fused multiply-add), and for x < 2^52 both x - floor(x) and 0.5 are
// multiples of ulp(x) while |e| <= ulp(x) / 2, so the comparison with one half is decided by (x - floor(x)) - 0.5 alone
// unless that is zero, and then by the sign of e; an exact tie goes to the even neighbour.
inline char *put_f3(char *p, double v) {
  if (std::isnan(v)) return p;                                       // na_rep = ""
  const double a = std::fabs(v);
  if (!(a < 4.0e12)) return p + snprintf(p, 400, "%.3f", v);         // inf and the range where ulp(x) > 1/2
  if (std::signbit(v)) *p++ = '-';                                   // "-0.000" keeps its sign, as printf does
  uint64_t n = 0;
  if (a >= 4.0e-4) {
    const double x = a * 1000.0, e = std::fma(a, 1000.0, -x), q = std::floor(x), t = (x - q) - 0.5;
    n = (uint64_t)q;
    if (t > 0.0 || (t == 0.0 && (e > 0.0 || (e == 0.0 && (n & 1))))) ++n;
  }
  p = put_u64(p, n / 1000);
  const unsigned f = (unsigned)(n % 1000);
  *p++ = '.';
  *p++ = (char)('0' + f / 100);
  *p++ = (char)('0' + f / 10 % 10);
  *p++ = (char)('0' + f % 10);
  return p;
}

}  // namespace

static int table_render(int32_t n_cols, const int32_t *kinds, const void *const *cols, const int64_t *const *starts,
                        const int64_t *rows, int64_t n_rows, int32_t n_threads, char **text, int fd, int64_t *n_bytes) {
  JG_REQUIRE(n_cols > 0 && kinds != nullptr && cols != nullptr && starts != nullptr && n_rows >= 0 &&
                 (text != nullptr || fd >= 0) && n_bytes != nullptr,
             JG_ERR_INVALID, "jg_table_format: bad arguments");
  for (int c = 0; c < n_cols; ++c)
    JG_REQUIRE(kinds[c] >= JG_COL_STRING && kinds[c] <= JG_COL_SPANS && cols[c] != nullptr &&
                   ((kinds[c] != JG_COL_STRING && kinds[c] != JG_COL_SPANS) || starts[c] != nullptr),
               JG_ERR_INVALID, "jg_table_format: column %d: kind %d / missing data", c, kinds[c]);
  int nt = n_threads > 0 ? n_threads : jg_usable_cores();
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 256), n_rows / 2048 + 1));
  std::vector<std::string> part((size_t)nt);
  std::vector<char> failed((size_t)nt, 0);                             // an exception must not leave a worker thread
  auto work = [&](int tix) {
   try {
    const int64_t r0 = n_rows * tix / nt, r1 = n_rows * (tix + 1) / nt;
    std::string &out = part[(size_t)tix];
    // rows are short (a few hundred bytes); strings may be long (a window summary), so size each row before it is written
    size_t fixed = 0;
    for (int c = 0; c < n_cols; ++c)
      fixed += (kinds[c] == JG_COL_STRING || kinds[c] == JG_COL_SPANS) ? 1 : (kinds[c] == JG_COL_FLOAT ? 400 : 24);
    size_t used = 0;
    for (int64_t i = r0; i < r1; ++i) {
      const int64_t r = rows ? rows[i] : i;
      size_t need = fixed;
      for (int c = 0; c < n_cols; ++c)
        if (kinds[c] == JG_COL_STRING) need += (size_t)(starts[c][r + 1] - 1 - starts[c][r]);
        else if (kinds[c] == JG_COL_SPANS) need += (size_t)(starts[c][2 * r + 1] - starts[c][2 * r]);
      if (out.size() < used + need) out.resize(std::max(out.size() * 2, used + need + (size_t)(r1 - i) * 64));
      char *p = &out[used];
      for (int c = 0; c < n_cols; ++c) {
        if (c) *p++ = '\t';
        switch (kinds[c]) {
          case JG_COL_STRING: {
            const int64_t a = starts[c][r], b = starts[c][r + 1] - 1;        // one separator byte behind every string
            memcpy(p, (const char *)cols[c] + a, (size_t)(b - a));
            p += b - a;
            break;
          }
          case JG_COL_SPANS: {
            const int64_t a = starts[c][2 * r], b = starts[c][2 * r + 1];    // an explicit [begin, end) per row
            memcpy(p, (const char *)cols[c] + a, (size_t)(b - a));
            p += b - a;
            break;
          }
          case JG_COL_INT: {
            const int64_t v = ((const int64_t *)cols[c])[r];
            if (v < 0) *p++ = '-';
            p = put_u64(p, v < 0 ? 0 - (uint64_t)v : (uint64_t)v);
            break;
          }
          case JG_COL_FLOAT: p = put_f3(p, ((const double *)cols[c])[r]); break;
          default: {
            const bool v = ((const uint8_t *)cols[c])[r] != 0;
            memcpy(p, v ? "True" : "False", v ? 4 : 5);
            p += v ? 4 : 5;
          }
        }
      }
      *p++ = '\n';
      used = (size_t)(p - out.data());
    }
    out.resize(used);
   } catch (const std::exception &) {
     failed[(size_t)tix] = 1;
   }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto &t : th) t.join();
  }
  for (char f : failed) JG_REQUIRE(!f, JG_ERR_NOMEM, "jg_table_format: out of memory while rendering %lld rows", (long long)n_rows);
  size_t total = 0;
  for (const auto &s : part) total += s.size();
  if (text == nullptr) {                          // straight to the file: the threads' pieces in row order
    for (const auto &s : part) {
      size_t done = 0;
      while (done < s.size()) {
        const ssize_t w = write(fd, s.data() + done, s.size() - done);
        JG_REQUIRE(w > 0 || (w < 0 && errno == EINTR), JG_ERR_IO, "jg_table_write: write failed (%s)", strerror(errno));
        if (w > 0) done += (size_t)w;
      }
    }
    *n_bytes = (int64_t)total;
    return JG_OK;
  }
  char *buf = (char *)malloc(total ? total : 1);
  JG_REQUIRE(buf != nullptr, JG_ERR_NOMEM, "jg_table_format: out of memory (%zu bytes)", total);
  size_t at = 0;
  for (const auto &s : part) { memcpy(buf + at, s.data(), s.size()); at += s.size(); }
  *text = buf;
  *n_bytes = (int64_t)total;
  return JG_OK;
}

extern "C" int jg_table_format(int32_t n_cols, const int32_t *kinds, const void *const *cols, const int64_t *const *starts,
                               const int64_t *rows, int64_t n_rows, int32_t n_threads, char **text, int64_t *n_bytes) {
  JG_REQUIRE(text != nullptr, JG_ERR_INVALID, "jg_table_format: bad arguments");
  return table_render(n_cols, kinds, cols, starts, rows, n_rows, n_threads, text, -1, n_bytes);
}

extern "C" int jg_table_write(int32_t n_cols, const int32_t *kinds, const void *const *cols, const int64_t *const *starts,
                              const int64_t *rows, int64_t n_rows, int32_t n_threads, int32_t fd, int64_t *n_bytes) {
  JG_REQUIRE(fd >= 0, JG_ERR_INVALID, "jg_table_write: bad file descriptor");
  return table_render(n_cols, kinds, cols, starts, rows, n_rows, n_threads, nullptr, fd, n_bytes);
}

extern "C" void jg_table_free(char *text) { free(text); }

// Are the n strings buf[off[i] .. off[i + 1]) pairwise different?  (The repeat table is joined to the result rows by record
// NUMBER, which is the reference's merge on the record NAME exactly when no name occurs twice - postprocess/collect.py:527-532;
// a hash set of a million Python strings costs more than the forward of a million short records.)  64-bit hashes on every
// core, then one open-addressing pass; equal hashes are settled by comparing the bytes.
extern "C" int jg_names_unique(const uint8_t *buf, const int64_t *off, int64_t n, int32_t n_threads, int32_t *unique) {
  JG_REQUIRE(off != nullptr && unique != nullptr && n >= 0 && (buf != nullptr || n == 0 || off[n] == off[0]), JG_ERR_INVALID,
             "jg_names_unique: bad arguments");
  *unique = 1;
  if (n < 2) return JG_OK;
  JG_REQUIRE(n < ((int64_t)1 << 31), JG_ERR_UNSUPPORTED, "jg_names_unique: %lld names", (long long)n);
  try {
    std::vector<uint64_t> hash((size_t)n);
    int nt = n_threads > 0 ? n_threads : jg_usable_cores();
    nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 64), n / 65536 + 1));
    auto work = [&](int tix) {
      for (int64_t i = n * tix / nt, e = n * (tix + 1) / nt; i < e; ++i) {
        uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)(off[i + 1] - off[i]);
        for (int64_t j = off[i]; j < off[i + 1]; ++j) h = (h ^ buf[j]) * 0x100000001b3ull;
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        hash[(size_t)i] = h;
      }
    };
    if (nt == 1) work(0);
    else {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
      for (auto &t : th) t.join();
    }
    size_t cap = 1;
    while (cap < (size_t)n * 2) cap <<= 1;
    std::vector<uint32_t> slot(cap, 0u);
    for (int64_t i = 0; i < n; ++i) {
      size_t at = (size_t)hash[(size_t)i] & (cap - 1);
      while (slot[at]) {
        const int64_t j = (int64_t)slot[at] - 1;
        if (hash[(size_t)j] == hash[(size_t)i] && off[j + 1] - off[j] == off[i + 1] - off[i] &&
            memcmp(buf + off[j], buf + off[i], (size_t)(off[i + 1] - off[i])) == 0) {
          *unique = 0;
          return JG_OK;
        }
        at = (at + 1) & (cap - 1);
      }
      slot[at] = (uint32_t)(i + 1);
    }
  } catch (const std::exception &) {
    JG_REQUIRE(false, JG_ERR_NOMEM, "jg_names_unique: out of memory (%lld names)", (long long)n);
  }
  return JG_OK;
}

// window_summary of every contig (postprocess/helpers.py:8-40 run lengths, :73-108 letters): the per-window calls of contig c
// are calls[first[c] .. first[c] + count[c]); each run of equal calls prints as its length followed by the class's letter
// (letters[class] = 0: no letter), e.g. "12V3b".  One NUL-terminated string per contig, back to back.
extern "C" int jg_run_summaries(const int32_t *calls, int64_t n_calls, const int64_t *first, const int64_t *count,
                                int64_t n_contigs, const uint8_t *letters, int32_t n_letters, int32_t n_threads, char **text,
                                int64_t *n_bytes) {
  JG_REQUIRE(n_contigs >= 0 && n_calls >= 0 && (n_contigs == 0 || (calls != nullptr && first != nullptr && count != nullptr)) &&
                 (n_letters == 0 || letters != nullptr) && text != nullptr && n_bytes != nullptr,
             JG_ERR_INVALID, "jg_run_summaries: bad arguments");
  for (int64_t c = 0; c < n_contigs; ++c)
    JG_REQUIRE(first[c] >= 0 && count[c] >= 0 && first[c] + count[c] <= n_calls, JG_ERR_INVALID,
               "jg_run_summaries: contig %lld covers calls [%lld, +%lld) of %lld", (long long)c, (long long)first[c],
               (long long)count[c], (long long)n_calls);
  int nt = n_threads > 0 ? n_threads : jg_usable_cores();
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 256), n_contigs / 1024 + 1));
  std::vector<std::string> part((size_t)nt);
  std::vector<char> failed((size_t)nt, 0);
  auto work = [&](int tix) {
   try {
    std::string &out = part[(size_t)tix];
    char tmp[24];
    for (int64_t c = n_contigs * tix / nt; c < n_contigs * (tix + 1) / nt; ++c) {
      const int32_t *v = calls + first[c];
      const int64_t n = count[c];
      for (int64_t i = 0; i < n;) {
        int64_t j = i + 1;
        while (j < n && v[j] == v[i]) ++j;
        char *e = put_u64(tmp, (uint64_t)(j - i));
        if (v[i] >= 0 && v[i] < n_letters && letters[v[i]] != 0) *e++ = (char)letters[v[i]];
        out.append(tmp, (size_t)(e - tmp));
        i = j;
      }
      out.push_back('\0');
    }
   } catch (const std::exception &) {
     failed[(size_t)tix] = 1;
   }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto &t : th) t.join();
  }
  for (char f : failed) JG_REQUIRE(!f, JG_ERR_NOMEM, "jg_run_summaries: out of memory");
  size_t total = 0;
  for (const auto &s : part) total += s.size();
  char *buf = (char *)malloc(total ? total : 1);
  JG_REQUIRE(buf != nullptr, JG_ERR_NOMEM, "jg_run_summaries: out of memory (%zu bytes)", total);
  size_t at = 0;
  for (const auto &s : part) { memcpy(buf + at, s.data(), s.size()); at += s.size(); }
  *text = buf;
  *n_bytes = (int64_t)total;
  return JG_OK;
}
