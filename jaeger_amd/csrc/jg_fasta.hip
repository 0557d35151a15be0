// FASTA ingest for the predict path (host code only): one pass over the file image into the
// layout the encoder scans - a contiguous base buffer plus per-record offsets - instead of a Python
// string per record.  Record rules follow what the reference gets from pyfastx.Fasta(build_index=
// False) in seqops/io.py:98-103: a record starts at a line beginning with '>', its name is the header
// up to the first whitespace, sequence lines are joined with surrounding whitespace removed;
// anything before the first header is ignored.  Case and non-ACGT letters are kept (the encoder
// deals with them).
#include <stdint.h>
#include <string.h>

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
