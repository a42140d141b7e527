// nb_ingest.cpp — FASTQ(.gz) record reader that packs straight into the device layout (host, product code).
//
// Replaces the FASTQ decode the Rust binary performs on each "--input" file (/root/reference/nimble/__main__.py:177-179)
// and, for 10x data, the barcode/UMI split of R1 that the reference does in Python
// (/root/reference/nimble/fastq_barcode_processor.py:131-141: CB = r1[:cb_len], UMI = r1[cb_len:cb_len+umi_len]).
// zlib's gzopen reads plain and gzip files alike. One thread inflates/reads and indexes record boundaries (memchr);
// `threads` workers (the reference's -c/--num_cores, __main__.py:398) then pack the sequences in parallel.
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "nb_internal.h"

struct nb_fastq {
  gzFile f = nullptr;
  std::vector<char> buf;      // raw text; [pos, end) is unconsumed
  size_t pos = 0, end = 0;
  bool eof = false;
  int64_t record = 0;
  int threads = 1;
  std::vector<size_t> seq_b, seq_e, qual_b, qual_e;   // per record of the current batch: offsets into buf
};

namespace {

bool fill(nb_fastq* q) {   // append more bytes; the buffer may grow (callers keep OFFSETS, never pointers)
  if (q->eof) return false;
  if (q->end == q->buf.size()) q->buf.resize(q->buf.size() * 2);
  const int got = gzread(q->f, q->buf.data() + q->end, (unsigned)std::min<size_t>(q->buf.size() - q->end, (size_t)1 << 30));
  if (got <= 0) { q->eof = true; return false; }
  q->end += (size_t)got; return true;
}

// next line [*b, *e) as OFFSETS into buf (without \n / \r), scanning from *cursor; advances *cursor. 0 at end of data.
int next_line(nb_fastq* q, size_t* cursor, size_t* b, size_t* e) {
  for (;;) {
    const char* base = q->buf.data();
    const char* nl = (const char*)std::memchr(base + *cursor, '\n', q->end - *cursor);
    if (nl) {
      *b = *cursor; *e = (size_t)(nl - base);
      *cursor = *e + 1;
      if (*e > *b && base[*e - 1] == '\r') --*e;
      return 1;
    }
    if (!fill(q)) {
      if (*cursor < q->end) {  // last line without newline
        *b = *cursor; *e = q->end; *cursor = q->end;
        if (*e > *b && q->buf[*e - 1] == '\r') --*e;
        return 1;
      }
      return 0;
    }
  }
}

inline void pack_one(const char* b, int64_t L, int nwb, int64_t stride, uint64_t* w) {
  std::memset(w, 0, (size_t)stride * 8);
  for (int64_t j = 0; j < L; ++j) {
    uint64_t c;
    switch (b[j]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break;
      default: c = 0; w[nwb + (j >> 6)] |= 1ull << (j & 63); }
    w[j >> 5] |= c << ((j & 31) * 2);
  }
}

}  // namespace

extern "C" {

int32_t nb_fastq_open(const char* path, nb_fastq** out) {
  if (!path || !out) { nb::set_error("nb_fastq_open: null argument"); return NB_ERR_INVALID; }
  gzFile f = gzopen(path, "rb");
  if (!f) { nb::set_error(std::string("nb_fastq_open: cannot open ") + path); return NB_ERR_INVALID; }
  gzbuffer(f, 1 << 20);
  nb_fastq* q = new nb_fastq(); q->f = f; q->buf.resize((size_t)1 << 24);
  *out = q; return NB_OK;
}

int32_t nb_fastq_close(nb_fastq* q) { if (q) { if (q->f) gzclose(q->f); delete q; } return NB_OK; }

int32_t nb_fastq_set_threads(nb_fastq* q, int32_t threads) {
  if (!q) { nb::set_error("nb_fastq_set_threads: null handle"); return NB_ERR_INVALID; }
  q->threads = threads < 1 ? 1 : (threads > 256 ? 256 : threads); return NB_OK;
}

// Reads up to max_reads records; *n_out = records read (0 = end of file).
//   out_words != NULL: the whole sequence line is packed at stride nb_packed_stride_words(max_len), length to out_lens.
//   cb_out / umi_out != NULL: bases [0, cb_len) and [cb_len, cb_len + umi_len) of the sequence line are copied, fixed
//   width, to cb_out[i * cb_len] / umi_out[i * umi_len] (the 10x R1 layout); cbqual_out != NULL additionally receives
//   the quality characters of the CB bases (for the lowest-quality tie-break of the whitelist correction).
int32_t nb_fastq_next_batch(nb_fastq* q, int64_t max_reads, int32_t max_len, uint64_t* out_words, uint16_t* out_lens,
                            int64_t* n_out, char* cb_out, int32_t cb_len, char* umi_out, int32_t umi_len, char* cbqual_out) {
  if (!q || !n_out || max_reads < 0) { nb::set_error("nb_fastq_next_batch: bad arguments"); return NB_ERR_INVALID; }
  // compact: drop consumed bytes so the offsets recorded below stay valid while the buffer grows
  if (q->pos > 0) { std::memmove(q->buf.data(), q->buf.data() + q->pos, q->end - q->pos); q->end -= q->pos; q->pos = 0; }
  q->seq_b.clear(); q->seq_e.clear(); q->qual_b.clear(); q->qual_e.clear();
  size_t cur = 0; int64_t n = 0;
  // phase 1 (serial): read / inflate and index record boundaries
  while (n < max_reads) {
    size_t b, e;
    if (!next_line(q, &cur, &b, &e)) break;
    if (b == e) continue;  // blank line between records
    if (q->buf[b] != '@') { nb::set_error("FASTQ parse error: record " + std::to_string(q->record + n) + " does not start with '@'"); return NB_ERR_INVALID; }
    size_t sb, se, pb, pe, qb, qe;
    if (!next_line(q, &cur, &sb, &se)) { nb::set_error("FASTQ parse error: truncated record"); return NB_ERR_INVALID; }
    if (!next_line(q, &cur, &pb, &pe) || pb == pe || q->buf[pb] != '+') { nb::set_error("FASTQ parse error: missing '+' line in record " + std::to_string(q->record + n)); return NB_ERR_INVALID; }
    if (!next_line(q, &cur, &qb, &qe)) { nb::set_error("FASTQ parse error: missing quality line"); return NB_ERR_INVALID; }
    q->seq_b.push_back(sb); q->seq_e.push_back(se); q->qual_b.push_back(qb); q->qual_e.push_back(qe);
    ++n;
  }
  q->pos = cur;
  // phase 2 (parallel): validate + pack
  const int nwb = (max_len + 31) / 32; const int64_t stride = nb_packed_stride_words(max_len);
  const char* base = q->buf.data();
  const int T = (int)std::min<int64_t>(q->threads, std::max<int64_t>(1, n / 4096));
  std::vector<int64_t> bad((size_t)T, -1); std::vector<int> bad_kind((size_t)T, 0);
  auto work = [&](int t) {
    const int64_t lo = n * t / T, hi = n * (t + 1) / T;
    for (int64_t i = lo; i < hi; ++i) {
      const char* b = base + q->seq_b[(size_t)i]; const int64_t L = (int64_t)(q->seq_e[(size_t)i] - q->seq_b[(size_t)i]);
      if (cb_out || umi_out || cbqual_out) {
        if (L < cb_len + umi_len) { if (bad[(size_t)t] < 0) { bad[(size_t)t] = i; bad_kind[(size_t)t] = 1; } continue; }
        if (cb_out) std::memcpy(cb_out + i * cb_len, b, (size_t)cb_len);
        if (umi_out) std::memcpy(umi_out + i * umi_len, b + cb_len, (size_t)umi_len);
        if (cbqual_out) {
          if ((int64_t)(q->qual_e[(size_t)i] - q->qual_b[(size_t)i]) < cb_len) { if (bad[(size_t)t] < 0) { bad[(size_t)t] = i; bad_kind[(size_t)t] = 2; } continue; }
          std::memcpy(cbqual_out + i * cb_len, base + q->qual_b[(size_t)i], (size_t)cb_len);
        }
      }
      if (out_words) {
        if (L > max_len) { if (bad[(size_t)t] < 0) { bad[(size_t)t] = i; bad_kind[(size_t)t] = 3; } continue; }
        pack_one(b, L, nwb, stride, out_words + i * stride);
        if (out_lens) out_lens[i] = (uint16_t)L;
      }
    }
  };
  if (T <= 1) work(0);
  else { std::vector<std::thread> th; for (int t = 1; t < T; ++t) th.emplace_back(work, t); work(0); for (auto& x : th) x.join(); }
  for (int t = 0; t < T; ++t) if (bad[(size_t)t] >= 0) {
      const std::string rec = std::to_string(q->record + bad[(size_t)t]);
      if (bad_kind[(size_t)t] == 1) { nb::set_error("FASTQ record " + rec + ": R1 shorter than cb_len + umi_len"); return NB_ERR_INVALID; }
      if (bad_kind[(size_t)t] == 2) { nb::set_error("FASTQ record " + rec + ": quality line shorter than cb_len"); return NB_ERR_INVALID; }
      nb::set_error("FASTQ record " + rec + ": read longer than max_len " + std::to_string(max_len)); return NB_ERR_UNSUPPORTED;
    }
  q->record += n;
  *n_out = n; return NB_OK;
}

}  // extern "C"
