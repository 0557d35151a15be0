// FASTA ingest for the predict path (host code only): one pass over the file image into the
// layout the encoder scans - a contiguous base buffer plus per-record offsets - instead of a Python
// string per record.  Record rules follow what the reference gets from pyfastx.Fasta(build_index=
// False) in seqops/io.py:98-103: a record starts at a line beginning with '>', its name is the header
// up to the first whitespace, sequence lines are joined with surrounding whitespace removed;
// anything before the first header is ignored.  Case and non-ACGT letters are kept (the encoder
// deals with them).
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "jg_common.h"

namespace {
inline bool is_space(uint8_t c) { return c == ' ' || (c >= '\t' && c <= '\r'); }
}  // namespace

extern "C" int jg_fasta_count(const uint8_t *text, int64_t n, int64_t *n_records, int64_t *name_bytes) {
  JG_REQUIRE((text != nullptr || n == 0) && n >= 0 && n_records != nullptr, JG_ERR_INVALID,
             "jg_fasta_count: bad arguments");
  int64_t count = 0, nbytes = 0;
  const uint8_t *p = text, *end = text + n;
  while (p < end) {
    const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(end - p)));
    if (*p == '>') {
      ++count;
      nbytes += (nl ? nl : end) - p;      // upper bound of the name's length
    }
    if (nl == nullptr) break;
    p = nl + 1;
  }
  *n_records = count;
  if (name_bytes != nullptr) *name_bytes = nbytes;
  return JG_OK;
}

// `bases` may alias `text` (in-place compaction: the write position never passes the read position),
// which saves a second file-sized buffer; names are therefore copied out into `names`.
extern "C" int jg_fasta_parse(const uint8_t *text, int64_t n, int64_t max_records, uint8_t *bases,
                              int64_t *offsets, uint8_t *names, int64_t *name_off, int64_t *n_records,
                              int64_t *n_bases) {
  JG_REQUIRE((text != nullptr || n == 0) && n >= 0 && bases != nullptr && offsets != nullptr &&
                 names != nullptr && name_off != nullptr && n_records != nullptr && n_bases != nullptr &&
                 max_records >= 0,
             JG_ERR_INVALID, "jg_fasta_parse: bad arguments");
  int64_t rec = 0, nb = 0, nn = 0;
  const uint8_t *p = text, *end = text + n;
  while (p < end) {
    const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(end - p)));
    const uint8_t *le = nl ? nl : end;     // line = [p, le)
    const uint8_t *next = nl ? nl + 1 : end;
    while (le > p && (le[-1] == '\r' || le[-1] == '\n')) --le;
    if (p < le && *p == '>') {
      JG_REQUIRE(rec < max_records, JG_ERR_INVALID, "jg_fasta_parse: more than %lld records",
                 (long long)max_records);
      const uint8_t *q = p + 1;
      while (q < le && is_space(*q)) ++q;
      const uint8_t *qe = q;
      while (qe < le && !is_space(*qe)) ++qe;
      name_off[rec] = nn;
      memcpy(names + nn, q, (size_t)(qe - q));
      nn += qe - q;
      offsets[rec] = nb;
      ++rec;
    } else if (rec > 0) {
      const uint8_t *a = p, *b = le;
      while (a < b && is_space(*a)) ++a;
      while (b > a && is_space(b[-1])) --b;
      memmove(bases + nb, a, (size_t)(b - a));
      nb += b - a;
    }
    p = next;
  }
  offsets[rec] = nb;
  name_off[rec] = nn;
  *n_records = rec;
  *n_bases = nb;
  return JG_OK;
}

// Record index of a file image without extracting the sequences: byte offset of every record's '>' line
// (rec_off[n_records] = n), whitespace-stripped sequence length, and the names.  The torchrun path indexes once
// (rank 0) and every rank then parses only the byte ranges of the contigs it owns.
extern "C" int jg_fasta_index(const uint8_t *text, int64_t n, int64_t max_records, int64_t *rec_off,
                              int64_t *seq_len, uint8_t *names, int64_t *name_off, int64_t *n_records) {
  JG_REQUIRE((text != nullptr || n == 0) && n >= 0 && rec_off != nullptr && seq_len != nullptr &&
                 names != nullptr && name_off != nullptr && n_records != nullptr && max_records >= 0,
             JG_ERR_INVALID, "jg_fasta_index: bad arguments");
  int64_t rec = 0, nn = 0;
  const uint8_t *p = text, *end = text + n;
  while (p < end) {
    const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(end - p)));
    const uint8_t *le = nl ? nl : end;
    const uint8_t *next = nl ? nl + 1 : end;
    while (le > p && (le[-1] == '\r' || le[-1] == '\n')) --le;
    if (p < le && *p == '>') {
      JG_REQUIRE(rec < max_records, JG_ERR_INVALID, "jg_fasta_index: more than %lld records", (long long)max_records);
      const uint8_t *q = p + 1;
      while (q < le && is_space(*q)) ++q;
      const uint8_t *qe = q;
      while (qe < le && !is_space(*qe)) ++qe;
      name_off[rec] = nn;
      memcpy(names + nn, q, (size_t)(qe - q));
      nn += qe - q;
      rec_off[rec] = p - text;
      seq_len[rec] = 0;
      ++rec;
    } else if (rec > 0) {
      const uint8_t *a = p, *b = le;
      while (a < b && is_space(*a)) ++a;
      while (b > a && is_space(b[-1])) --b;
      seq_len[rec - 1] += b - a;
    }
    p = next;
  }
  rec_off[rec] = n;
  name_off[rec] = nn;
  *n_records = rec;
  return JG_OK;
}

// ---- parallel ingest ------------------------------------------------------------------------------------------
// jg_fasta_scan cuts the file image into one slice per thread (slices start at line starts), finds the records of
// every slice and counts their bases; the serial merge turns that into record offsets / lengths / name offsets.
// jg_fasta_fill then copies bases and names to their final places, slice by slice in parallel: bases of consecutive
// records lie back to back, so a slice's write position is the number of record bases in front of it.  Same record
// rules as jg_fasta_parse (tests/test_fasta_native.py compares the two on every fixture).
struct jg_fasta_scan_t {
  const uint8_t *text = nullptr;
  int64_t n = 0;
  struct Slice {
    int64_t a = 0, b = 0;                 // byte range [a, b), a at a line start
    std::vector<int64_t> hdr;             // byte offset of every '>' line in the slice
    std::vector<int64_t> name_a, name_b;  // the names' byte ranges
    std::vector<int64_t> seg;             // hdr.size() + 1 entries: bases in front of the first header, then behind each
    int64_t base0 = 0;                    // write position of the slice's first record base
    int64_t rec0 = 0;                     // global index of the slice's first record
    int64_t name0 = 0;                    // write position of its first name
    bool lead_dropped = false;            // the bases in front of its first header belong to no record
  };
  std::vector<Slice> slices;
  int64_t n_records = 0, n_bases = 0, name_bytes = 0;
};

namespace {
template <typename OnHeader, typename OnSeq>
inline void walk_lines(const uint8_t *text, int64_t a, int64_t b, OnHeader on_header, OnSeq on_seq) {
  const uint8_t *p = text + a, *end = text + b;
  while (p < end) {
    const uint8_t *nl = static_cast<const uint8_t *>(memchr(p, '\n', (size_t)(end - p)));
    const uint8_t *le = nl ? nl : end;
    const uint8_t *next = nl ? nl + 1 : end;
    while (le > p && (le[-1] == '\r' || le[-1] == '\n')) --le;
    if (p < le && *p == '>') {
      const uint8_t *q = p + 1;
      while (q < le && is_space(*q)) ++q;
      const uint8_t *qe = q;
      while (qe < le && !is_space(*qe)) ++qe;
      on_header(p, q, qe);
    } else {
      const uint8_t *x = p, *y = le;
      while (x < y && is_space(*x)) ++x;
      while (y > x && is_space(y[-1])) --y;
      if (y > x) on_seq(x, y);
    }
    p = next;
  }
}
}  // namespace

extern "C" int jg_fasta_scan(const uint8_t *text, int64_t n, int32_t n_threads, jg_fasta_scan_t **out,
                             int64_t *n_records, int64_t *n_bases, int64_t *name_bytes) {
  JG_REQUIRE((text != nullptr || n == 0) && n >= 0 && out != nullptr && n_records != nullptr && n_bases != nullptr &&
                 name_bytes != nullptr, JG_ERR_INVALID, "jg_fasta_scan: bad arguments");
  // automatic: one slice per usable core, at least 1 MiB each; an explicit count is taken as given (tests cut tiny files)
  int nt = n_threads > 0 ? (int)std::min<int64_t>(n_threads, n / 8 + 1)
                         : (int)std::min<int64_t>(jg_usable_cores(), n / (1 << 20) + 1);
  nt = std::max(1, std::min(nt, 64));
  jg_fasta_scan_t *h = new jg_fasta_scan_t();
  h->text = text;
  h->n = n;
  h->slices.resize((size_t)nt);
  for (int t = 0; t < nt; ++t) {                          // slice starts: the first line start at or behind n * t / nt
    int64_t a = n * t / nt;
    if (t > 0 && a > 0 && text[a - 1] != '\n') {
      const uint8_t *nl = static_cast<const uint8_t *>(memchr(text + a, '\n', (size_t)(n - a)));
      a = nl ? (nl - text) + 1 : n;
    }
    h->slices[(size_t)t].a = a;
  }
  for (int t = 0; t < nt; ++t) h->slices[(size_t)t].b = t + 1 < nt ? h->slices[(size_t)t + 1].a : n;
  auto work = [&](int t) {
    jg_fasta_scan_t::Slice &sl = h->slices[(size_t)t];
    sl.seg.push_back(0);
    walk_lines(text, sl.a, sl.b,
               [&](const uint8_t *p, const uint8_t *q, const uint8_t *qe) {
                 sl.hdr.push_back(p - text);
                 sl.name_a.push_back(q - text);
                 sl.name_b.push_back(qe - text);
                 sl.seg.push_back(0);
               },
               [&](const uint8_t *x, const uint8_t *y) { sl.seg.back() += y - x; });
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  int64_t rec = 0, nb = 0, nn = 0;
  for (auto &sl : h->slices) {
    sl.rec0 = rec;
    sl.name0 = nn;
    sl.lead_dropped = rec == 0;                           // anything in front of the first header is ignored
    sl.base0 = nb;
    if (!sl.lead_dropped) nb += sl.seg[0];
    for (size_t k = 0; k < sl.hdr.size(); ++k) {
      nb += sl.seg[k + 1];
      nn += sl.name_b[k] - sl.name_a[k];
    }
    rec += (int64_t)sl.hdr.size();
  }
  h->n_records = rec;
  h->n_bases = nb;
  h->name_bytes = nn;
  *out = h;
  *n_records = rec;
  *n_bases = nb;
  *name_bytes = nn;
  return JG_OK;
}

// bases: n_bases bytes (NULL: index only); offsets / name_off: n_records + 1 entries; names: name_bytes bytes (or NULL); rec_off (optional):
// n_records + 1 byte offsets of the header lines (last = n), as jg_fasta_index returns them
extern "C" int jg_fasta_fill(const jg_fasta_scan_t *h, uint8_t *bases, int64_t *offsets, uint8_t *names,
                             int64_t *name_off, int64_t *rec_off) {
  JG_REQUIRE(h != nullptr && offsets != nullptr && name_off != nullptr, JG_ERR_INVALID, "jg_fasta_fill: bad arguments");
  const uint8_t *text = h->text;
  auto work = [&](size_t t) {
    const jg_fasta_scan_t::Slice &sl = h->slices[t];
    int64_t nb = sl.base0, nn = sl.name0, rec = sl.rec0;
    bool in_record = !sl.lead_dropped;
    walk_lines(text, sl.a, sl.b,
               [&](const uint8_t *p, const uint8_t *q, const uint8_t *qe) {
                 offsets[rec] = nb;
                 name_off[rec] = nn;
                 if (rec_off != nullptr) rec_off[rec] = p - text;
                 if (names != nullptr) memcpy(names + nn, q, (size_t)(qe - q));
                 nn += qe - q;
                 ++rec;
                 in_record = true;
               },
               [&](const uint8_t *x, const uint8_t *y) {
                 if (!in_record) return;
                 if (bases != nullptr) memcpy(bases + nb, x, (size_t)(y - x));
                 nb += y - x;
               });
  };
  std::vector<std::thread> pool;
  for (size_t t = 1; t < h->slices.size(); ++t) pool.emplace_back(work, t);
  work(0);
  for (auto &th : pool) th.join();
  offsets[h->n_records] = h->n_bases;
  name_off[h->n_records] = h->name_bytes;
  if (rec_off != nullptr) rec_off[h->n_records] = h->n;
  return JG_OK;
}

extern "C" void jg_fasta_scan_free(jg_fasta_scan_t *h) { delete h; }
