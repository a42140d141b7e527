// nb_ingest.cpp — FASTQ(.gz) record reader that packs straight into the device layout (host, product code).
//
// Replaces the FASTQ decode the Rust binary performs on each "--input" file (/root/reference/nimble/__main__.py:177-179)
// and, for 10x data, the barcode/UMI split of R1 that the reference does in Python
// (/root/reference/nimble/fastq_barcode_processor.py:131-141: CB = r1[:cb_len], UMI = r1[cb_len:cb_len+umi_len]).
// zlib's gzopen reads plain and gzip files alike.
#include <zlib.h>

#include <cstring>
#include <string>
#include <vector>

#include "nb_internal.h"

struct nb_fastq {
  gzFile f = nullptr;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  int64_t record = 0;
};

namespace {

bool fill(nb_fastq* q) {
  if (q->eof) return false;
  if (q->pos > 0 && q->pos < q->end) std::memmove(q->buf.data(), q->buf.data() + q->pos, q->end - q->pos);
  q->end -= q->pos; q->pos = 0;
  if (q->end == q->buf.size()) q->buf.resize(q->buf.size() * 2);
  const int got = gzread(q->f, q->buf.data() + q->end, (unsigned)(q->buf.size() - q->end));
  if (got <= 0) { q->eof = true; return false; }
  q->end += (size_t)got; return true;
}

// returns 1 and [*b, *e) of the next line (without \n / \r), 0 at end of file
int next_line(nb_fastq* q, const char** b, const char** e) {
  for (;;) {
    const char* s = q->buf.data() + q->pos;
    const char* nl = (const char*)std::memchr(s, '\n', q->end - q->pos);
    if (nl) {
      *b = s; *e = nl; if (*e > *b && (*e)[-1] == '\r') --*e;
      q->pos = (size_t)(nl - q->buf.data()) + 1; return 1;
    }
    if (!fill(q)) {
      if (q->pos < q->end) {  // last line without newline
        *b = q->buf.data() + q->pos; *e = q->buf.data() + q->end; if (*e > *b && (*e)[-1] == '\r') --*e;
        q->pos = q->end; return 1;
      }
      return 0;
    }
  }
}

}  // namespace

extern "C" {

int32_t nb_fastq_open(const char* path, nb_fastq** out) {
  if (!path || !out) { nb::set_error("nb_fastq_open: null argument"); return NB_ERR_INVALID; }
  gzFile f = gzopen(path, "rb");
  if (!f) { nb::set_error(std::string("nb_fastq_open: cannot open ") + path); return NB_ERR_INVALID; }
  gzbuffer(f, 1 << 20);
  nb_fastq* q = new nb_fastq(); q->f = f; q->buf.resize(1 << 22);
  *out = q; return NB_OK;
}

int32_t nb_fastq_close(nb_fastq* q) { if (q) { if (q->f) gzclose(q->f); delete q; } return NB_OK; }

// Reads up to max_reads records; *n_out = records read (0 = end of file).
//   out_words != NULL: the whole sequence line is packed at stride nb_packed_stride_words(max_len), length to out_lens.
//   cb_out / umi_out != NULL: bases [0, cb_len) and [cb_len, cb_len + umi_len) of the sequence line are copied, fixed
//   width, to cb_out[i * cb_len] / umi_out[i * umi_len] (the 10x R1 layout); cbqual_out != NULL additionally receives
//   the quality characters of the CB bases (for the lowest-quality tie-break of the whitelist correction).
int32_t nb_fastq_next_batch(nb_fastq* q, int64_t max_reads, int32_t max_len, uint64_t* out_words, uint16_t* out_lens,
                            int64_t* n_out, char* cb_out, int32_t cb_len, char* umi_out, int32_t umi_len, char* cbqual_out) {
  if (!q || !n_out || max_reads < 0) { nb::set_error("nb_fastq_next_batch: bad arguments"); return NB_ERR_INVALID; }
  const int nwb = (max_len + 31) / 32; const int64_t stride = nb_packed_stride_words(max_len);
  int64_t n = 0;
  while (n < max_reads) {
    const char *b, *e;
    if (!next_line(q, &b, &e)) break;
    if (b == e) continue;  // blank line between records
    if (*b != '@') { nb::set_error("FASTQ parse error: record " + std::to_string(q->record) + " does not start with '@'"); return NB_ERR_INVALID; }
    if (!next_line(q, &b, &e)) { nb::set_error("FASTQ parse error: truncated record"); return NB_ERR_INVALID; }
    const int64_t L = e - b;
    if (cb_out || umi_out) {
      if (L < cb_len + umi_len) { nb::set_error("FASTQ record " + std::to_string(q->record) + ": R1 shorter than cb_len + umi_len"); return NB_ERR_INVALID; }
      if (cb_out) std::memcpy(cb_out + n * cb_len, b, (size_t)cb_len);
      if (umi_out) std::memcpy(umi_out + n * umi_len, b + cb_len, (size_t)umi_len);
    }
    if (out_words) {
      if (L > max_len) { nb::set_error("FASTQ record " + std::to_string(q->record) + ": read longer than max_len " + std::to_string(max_len)); return NB_ERR_UNSUPPORTED; }
      uint64_t* w = out_words + n * stride; std::memset(w, 0, (size_t)stride * 8);
      for (int64_t j = 0; j < L; ++j) {
        uint64_t c;
        switch (b[j]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break;
          default: c = 0; w[nwb + (j >> 6)] |= 1ull << (j & 63); }
        w[j >> 5] |= c << ((j & 31) * 2);
      }
      if (out_lens) out_lens[n] = (uint16_t)L;
    }
    // '+' line and quality line
    const char *b2, *e2;
    if (!next_line(q, &b2, &e2) || b2 == e2 || *b2 != '+') { nb::set_error("FASTQ parse error: missing '+' line in record " + std::to_string(q->record)); return NB_ERR_INVALID; }
    if (!next_line(q, &b2, &e2)) { nb::set_error("FASTQ parse error: missing quality line"); return NB_ERR_INVALID; }
    if (cbqual_out) {
      if (e2 - b2 < cb_len) { nb::set_error("FASTQ record " + std::to_string(q->record) + ": quality line shorter than cb_len"); return NB_ERR_INVALID; }
      std::memcpy(cbqual_out + n * cb_len, b2, (size_t)cb_len);
    }
    ++n; ++q->record;
  }
  *n_out = n; return NB_OK;
}

}  // extern "C"
