// nb_ingest.cpp — FASTQ(.gz) record reader that packs straight into the device layout (host, product code).
//
// Replaces the FASTQ decode the Rust binary performs on each "--input" file (/root/reference/nimble/__main__.py:177-179)
// and, for 10x data, the barcode/UMI split of R1 that the reference does in Python
// (/root/reference/nimble/fastq_barcode_processor.py:131-141: CB = r1[:cb_len], UMI = r1[cb_len:cb_len+umi_len]).
//
// Three stages, sized so that the host keeps up with the GPU for as long as the storage format allows:
//   1. a background thread per file reads (plain: read(2); gzip: the table-driven inflate of nb_inflate.cpp, CRC checked —
//      inherently serial per stream) into chunks, ahead of the consumer, so R1 and R2 inflate concurrently and overlap
//      packing, the GPU and the writer;
//   2. record boundaries: `threads` workers (the reference's -c/--num_cores, __main__.py:398) find the newlines of the
//      buffered text in parallel; strict 4-line framing is verified per record ('@' and '+' lines) and any anomaly
//      (blank lines between records, truncation) falls back to the serial line walker, which also words the errors;
//   3. the same workers pack the sequence lines 32 bases per word through a lookup table.
#include <fcntl.h>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "nb_inflate.h"
#include "nb_internal.h"

namespace {

constexpr size_t CHUNK_BYTES = (size_t)8 << 20;
constexpr size_t MAX_QUEUED_CHUNKS = 16;
// a chunk buffer = [history space the gzip decoder writes its 32 KiB window into][payload][overshoot slack of its match copies]
constexpr size_t CHUNK_PREFIX = nb::GzipStream::HISTORY, CHUNK_ALLOC = CHUNK_PREFIX + CHUNK_BYTES + nb::GzipStream::SLACK + 64;

struct Chunk { char* p = nullptr; size_t n = 0; };   // p = allocation base; the n payload bytes start at p + CHUNK_PREFIX

struct CodeTable {
  uint8_t t[256];
  CodeTable() {
    std::memset(t, 4, sizeof t);   // 4 = invalid (N): base bits 0, mask bit 1
    t[(int)'A'] = t[(int)'a'] = 0; t[(int)'C'] = t[(int)'c'] = 1; t[(int)'G'] = t[(int)'g'] = 2; t[(int)'T'] = t[(int)'t'] = 3;
  }
};
const CodeTable CODES;

// device layout of one read (nb_packed_stride_words): nwb words of 2-bit bases, then ceil(max_len/64) words of N mask
inline void pack_one(const char* b, int64_t L, int nwb, int64_t stride, uint64_t* w) {
  const uint8_t* s = (const uint8_t*)b;
  int64_t j = 0; int word = 0;
  for (; word < nwb && j < L; ++word) {
    const int64_t m = std::min<int64_t>(32, L - j);
    uint64_t bits = 0, nm = 0;
    for (int64_t u = 0; u < m; ++u) {
      const uint64_t c = CODES.t[s[j + u]];
      bits |= (c & 3) << (2 * u);
      nm |= (c >> 2) << u;
    }
    w[word] = bits;
    if (word & 1) w[nwb + (word >> 1)] |= nm << 32; else w[nwb + (word >> 1)] = nm;
    j += m;
  }
  for (int z = word; z < nwb; ++z) w[z] = 0;
  for (int64_t z = nwb + ((word + 1) >> 1); z < stride; ++z) w[z] = 0;
}

template <class F>
void parallel_for(int T, F&& f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> th;
  th.reserve((size_t)T - 1);
  for (int t = 1; t < T; ++t) th.emplace_back([&f, t] { f(t); });
  f(0);
  for (auto& x : th) x.join();
}

}  // namespace

struct nb_fastq {
  // source + read-ahead thread
  int fd = -1;
  nb::GzipStream* gz = nullptr;
  nb::BgzfReader* bgzf = nullptr;   // a gzip file made of BGZF blocks (bgzip): inflated block-parallel, like a BAM
  FILE* bgzf_fp = nullptr;
  std::thread reader;
  std::mutex mu;
  std::condition_variable cv_data, cv_space;
  std::deque<Chunk> chunks;
  std::vector<char*> spare;      // recycled chunk buffers (first touch of fresh pages is the expensive part of a chunk)
  bool src_done = false, stop = false;
  std::string src_error;
  int64_t range_left = -1;       // plain files opened with nb_fastq_open_range: bytes still to read (-1 = to the end of the file)
  // text of the current batch; [pos, end) is unconsumed. malloc-managed (no zero fill on growth)
  char* buf = nullptr;
  size_t cap = 0, pos = 0, end = 0, next_pos = 0;   // next_pos: where the indexed batch ends (pos moves there when it is packed)
  bool eof = false;
  int64_t record = 0;
  std::atomic<int> threads{1};   // written by nb_fastq_set_threads after open, read by the read-ahead thread (BGZF inflate)
  // index of the current batch: offsets into buf
  std::vector<size_t> nl;
  std::vector<size_t> seq_b, seq_e, qual_b, qual_e;
  std::vector<size_t> name_b, name_e;                 // the read id: header line behind '@' up to the first blank
  int64_t n_indexed = 0;
  int32_t longest = 0;
  bool indexed = false;

  void read_loop() {
    for (;;) {
      Chunk c;
      {
        std::lock_guard<std::mutex> lk(mu);
        if (!spare.empty()) { c.p = spare.back(); spare.pop_back(); }
      }
      if (!c.p) c.p = (char*)std::malloc(CHUNK_ALLOC);
      std::string err;
      if (!c.p) err = "out of memory";
      if (c.p && gz) {
        const long got = gz->read((uint8_t*)c.p + CHUNK_PREFIX, CHUNK_BYTES);   // (may exceed CHUNK_BYTES by one match: slack)
        if (got < 0) err = std::string("gzip stream error: ") + (gz->error ? gz->error : "?");
        else c.n = (size_t)got;
      }
      while (c.p && bgzf && c.n + 65536 <= CHUNK_BYTES) {
        bgzf->threads = threads.load(std::memory_order_relaxed);
        const long got = bgzf->read((uint8_t*)c.p + CHUNK_PREFIX + c.n, CHUNK_BYTES - c.n);
        if (got < 0) { err = std::string("gzip stream error: ") + (bgzf->error ? bgzf->error : "?"); break; }
        if (got == 0) break;
        c.n += (size_t)got;
      }
      while (c.p && !gz && !bgzf && c.n < CHUNK_BYTES) {
        size_t want = CHUNK_BYTES - c.n;
        if (range_left >= 0 && (int64_t)want > range_left) want = (size_t)range_left;
        if (want == 0) break;
        const long got = (long)::read(fd, c.p + CHUNK_PREFIX + c.n, want);
        if (got < 0) { err = "read error"; break; }
        if (got == 0) break;
        c.n += (size_t)got;
        if (range_left >= 0) range_left -= got;
      }
      std::unique_lock<std::mutex> lk(mu);
      if (!err.empty()) { src_error = err; src_done = true; std::free(c.p); cv_data.notify_all(); return; }
      const bool last = bgzf ? c.n + 65536 <= CHUNK_BYTES : c.n < CHUNK_BYTES;   // (a BGZF chunk stops when the next block might not fit)
      if (c.n) chunks.push_back(c); else std::free(c.p);
      if (last) src_done = true;
      cv_data.notify_all();
      if (last) return;
      cv_space.wait(lk, [&] { return stop || chunks.size() < MAX_QUEUED_CHUNKS; });
      if (stop) return;
    }
  }

  // append the next chunk to buf; false at end of data (or on a source error: see src_error)
  bool fill() {
    if (eof) return false;
    Chunk c;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_data.wait(lk, [&] { return !chunks.empty() || src_done; });
      if (chunks.empty()) { eof = true; return false; }
      c = chunks.front(); chunks.pop_front();
      cv_space.notify_all();
    }
    if (end + c.n > cap) {
      size_t ncap = std::max<size_t>(cap * 2, end + c.n);
      char* nb_ = (char*)std::realloc(buf, ncap);
      if (!nb_) { std::free(c.p); src_error = "out of memory"; eof = true; return false; }
      buf = nb_; cap = ncap;
    }
    std::memcpy(buf + end, c.p + CHUNK_PREFIX, c.n);
    end += c.n;
    { std::lock_guard<std::mutex> lk(mu); spare.push_back(c.p); }
    return true;
  }
};

namespace {

// next line [*b, *e) as OFFSETS into buf (without \n / \r), scanning from *cursor; advances *cursor. 0 at end of data.
int next_line(nb_fastq* q, size_t* cursor, size_t* b, size_t* e) {
  for (;;) {
    const char* base = q->buf;
    const char* nl = *cursor < q->end ? (const char*)std::memchr(base + *cursor, '\n', q->end - *cursor) : nullptr;
    if (nl) {
      *b = *cursor; *e = (size_t)(nl - base);
      *cursor = *e + 1;
      if (*e > *b && base[*e - 1] == '\r') --*e;
      return 1;
    }
    if (!q->fill()) {
      if (*cursor < q->end) {  // last line without newline
        *b = *cursor; *e = q->end; *cursor = q->end;
        if (*e > *b && q->buf[*e - 1] == '\r') --*e;
        return 1;
      }
      return 0;
    }
  }
}

// serial indexer: tolerates blank lines between records and words the parse errors
int32_t index_serial(nb_fastq* q, int64_t max_reads) {
  q->seq_b.clear(); q->seq_e.clear(); q->qual_b.clear(); q->qual_e.clear(); q->name_b.clear(); q->name_e.clear();
  size_t cur = 0; int64_t n = 0;
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
    { size_t ne = b + 1; while (ne < e && q->buf[ne] != ' ' && q->buf[ne] != '\t') ++ne; q->name_b.push_back(b + 1); q->name_e.push_back(ne); }
    ++n;
  }
  q->next_pos = cur; q->n_indexed = n;
  return NB_OK;
}

#if defined(__x86_64__)
bool have_avx2() { static const bool ok = __builtin_cpu_supports("avx2"); return ok; }
__attribute__((target("avx2"))) void newlines_avx2(const char* base, size_t lo, size_t hi, std::vector<size_t>& v) {
  const __m256i nl = _mm256_set1_epi8('\n');
  size_t i = lo;
  for (; i + 32 <= hi; i += 32) {
    uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(base + i)), nl));
    while (m) { v.push_back(i + (size_t)__builtin_ctz(m)); m &= m - 1; }
  }
  for (; i < hi; ++i) if (base[i] == '\n') v.push_back(i);
}
#endif

// parallel newline scan of buf[from, to): appends the offsets of '\n' to q->nl in order
void scan_newlines(nb_fastq* q, size_t from, size_t to) {
  if (to <= from) return;
  const int T = (int)std::min<size_t>((size_t)q->threads.load(), std::max<size_t>(1, (to - from) >> 20));
  std::vector<std::vector<size_t>> part((size_t)T);
  const char* base = q->buf;
  parallel_for(T, [&](int t) {
    const size_t lo = from + (to - from) * (size_t)t / (size_t)T, hi = from + (to - from) * (size_t)(t + 1) / (size_t)T;
    auto& v = part[(size_t)t];
    v.reserve((hi - lo) / 64 + 16);
#if defined(__x86_64__)
    if (have_avx2()) { newlines_avx2(base, lo, hi, v); return; }   // (short lines: one memchr call per line costs more than the scan)
#endif
    const char* p = base + lo; const char* e = base + hi;
    while (p < e) {
      const char* x = (const char*)std::memchr(p, '\n', (size_t)(e - p));
      if (!x) break;
      v.push_back((size_t)(x - base));
      p = x + 1;
    }
  });
  size_t total = q->nl.size();
  for (auto& v : part) total += v.size();
  q->nl.reserve(total);
  for (auto& v : part) q->nl.insert(q->nl.end(), v.begin(), v.end());
}

// fast indexer: strict 4-line records. Returns 1 when it indexed the batch, 0 when the serial walker must take over.
int index_fast(nb_fastq* q, int64_t max_reads) {
  const size_t want = (size_t)max_reads * 4;
  q->nl.clear();
  size_t scanned = 0;
  bool virtual_nl = false;
  for (;;) {
    scan_newlines(q, scanned, q->end);
    scanned = q->end;
    if (q->nl.size() >= want) break;
    if (!q->fill()) {
      if (q->end > 0 && q->buf[q->end - 1] != '\n') { q->nl.push_back(q->end); virtual_nl = true; }   // last line without newline
      break;
    }
  }
  const int64_t n = (int64_t)std::min<size_t>((size_t)max_reads, q->nl.size() / 4);
  if (n == 0) return q->end == 0 ? 1 : 0;   // nothing left at all, or a tail the serial walker must judge
  q->seq_b.resize((size_t)n); q->seq_e.resize((size_t)n); q->qual_b.resize((size_t)n); q->qual_e.resize((size_t)n);
  q->name_b.resize((size_t)n); q->name_e.resize((size_t)n);
  const int T = (int)std::min<int64_t>(q->threads.load(), std::max<int64_t>(1, n / 4096));
  std::vector<int> bad((size_t)T, 0);
  const char* base = q->buf; const size_t* nl = q->nl.data();
  parallel_for(T, [&](int t) {
    const int64_t lo = n * t / T, hi = n * (t + 1) / T;
    for (int64_t i = lo; i < hi; ++i) {
      const size_t h = i ? nl[4 * i - 1] + 1 : 0;
      const size_t sb = nl[4 * i] + 1, pb = nl[4 * i + 1] + 1, qb = nl[4 * i + 2] + 1;
      size_t se = nl[4 * i + 1], qe = nl[4 * i + 3];
      if (base[h] != '@' || nl[4 * i] == h || base[pb] != '+') { bad[(size_t)t] = 1; return; }
      if (se > sb && base[se - 1] == '\r') --se;
      if (qe > qb && base[qe - 1] == '\r') --qe;
      q->seq_b[(size_t)i] = sb; q->seq_e[(size_t)i] = se; q->qual_b[(size_t)i] = qb; q->qual_e[(size_t)i] = qe;
      { size_t he = nl[4 * i]; if (he > h && base[he - 1] == '\r') --he; size_t ne = h + 1; while (ne < he && base[ne] != ' ' && base[ne] != '\t') ++ne; q->name_b[(size_t)i] = h + 1; q->name_e[(size_t)i] = ne; }
    }
  });
  for (int b : bad) if (b) return 0;
  q->next_pos = std::min(q->end, nl[4 * n - 1] + 1);
  (void)virtual_nl;
  q->n_indexed = n;
  return 1;
}

}  // namespace

extern "C" {

int32_t nb_fastq_open(const char* path, nb_fastq** out) {
  if (!path || !out) { nb::set_error("nb_fastq_open: null argument"); return NB_ERR_INVALID; }
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) { nb::set_error(std::string("nb_fastq_open: cannot open ") + path); return NB_ERR_INVALID; }
  unsigned char magic[2] = {0, 0};
  const long got = (long)::read(fd, magic, 2);
  ::lseek(fd, 0, SEEK_SET);
  nb_fastq* q = new nb_fastq();
  unsigned char first[18]; const long got18 = (long)::pread(fd, first, 18, 0);
  if (got18 == 18 && nb::BgzfReader::block_size(first, 18) > 0) {
    q->bgzf_fp = ::fdopen(::dup(fd), "rb");
    if (q->bgzf_fp) { q->bgzf = new nb::BgzfReader(); q->bgzf->fp = q->bgzf_fp; }
  }
  if (!q->bgzf && got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
    q->gz = new nb::GzipStream(); q->gz->fd = fd;
  }
  q->fd = fd;
  q->cap = (size_t)1 << 24;
  q->buf = (char*)std::malloc(q->cap);
  if (!q->buf) { nb_fastq_close(q); nb::set_error("nb_fastq_open: out of memory"); return NB_ERR_INVALID; }
  q->reader = std::thread([q] { q->read_loop(); });
  *out = q; return NB_OK;
}

// A byte range [begin, end) of a PLAIN (uncompressed) FASTQ file that starts at a record boundary: what one rank of a multi-GPU job
// reads (nimble_amd/align.py finds the boundaries with nb_file_count_newlines / nb_file_nth_newline). Compressed files cannot be
// entered in the middle: NB_ERR_UNSUPPORTED (the caller deals batches round-robin instead).
int32_t nb_fastq_open_range(const char* path, int64_t begin, int64_t end, nb_fastq** out) {
  if (!path || !out || begin < 0 || (end >= 0 && end < begin)) { nb::set_error("nb_fastq_open_range: bad arguments"); return NB_ERR_INVALID; }
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) { nb::set_error(std::string("nb_fastq_open_range: cannot open ") + path); return NB_ERR_INVALID; }
  unsigned char magic[2] = {0, 0};
  if ((long)::pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
    ::close(fd); nb::set_error("nb_fastq_open_range: compressed input cannot be read from the middle"); return NB_ERR_UNSUPPORTED;
  }
  if (::lseek(fd, (off_t)begin, SEEK_SET) < 0) { ::close(fd); nb::set_error("nb_fastq_open_range: seek failed"); return NB_ERR_INVALID; }
  nb_fastq* q = new nb_fastq();
  q->fd = fd;
  q->range_left = end < 0 ? -1 : end - begin;
  q->cap = (size_t)1 << 24;
  q->buf = (char*)std::malloc(q->cap);
  if (!q->buf) { nb_fastq_close(q); nb::set_error("nb_fastq_open_range: out of memory"); return NB_ERR_INVALID; }
  q->reader = std::thread([q] { q->read_loop(); });
  *out = q; return NB_OK;
}

// newlines in [begin, end) of a file (end < 0: to the end); *last_byte_out (may be NULL) = the last byte of that range, or -1 if it is empty
int32_t nb_file_count_newlines(const char* path, int64_t begin, int64_t end, int64_t* count_out, int32_t* last_byte_out) {
  if (!path || !count_out || begin < 0) { nb::set_error("nb_file_count_newlines: bad arguments"); return NB_ERR_INVALID; }
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) { nb::set_error(std::string("nb_file_count_newlines: cannot open ") + path); return NB_ERR_INVALID; }
  std::vector<char> buf((size_t)1 << 22);
  int64_t pos = begin, n = 0; int last = -1;
  for (;;) {
    size_t want = buf.size();
    if (end >= 0) { if (pos >= end) break; if ((int64_t)want > end - pos) want = (size_t)(end - pos); }
    const long got = (long)::pread(fd, buf.data(), want, (off_t)pos);
    if (got < 0) { ::close(fd); nb::set_error("nb_file_count_newlines: read error"); return NB_ERR_INVALID; }
    if (got == 0) break;
    const char* p = buf.data(); const char* e = p + got;
    while ((p = (const char*)std::memchr(p, '\n', (size_t)(e - p))) != nullptr) { ++n; ++p; }
    last = (unsigned char)buf[(size_t)got - 1];
    pos += got;
  }
  ::close(fd);
  *count_out = n; if (last_byte_out) *last_byte_out = last;
  return NB_OK;
}

// offset of the byte after the n-th newline (n >= 1) at or after `begin`; -1 in *offset_out if the file has fewer
int32_t nb_file_nth_newline(const char* path, int64_t begin, int64_t n, int64_t* offset_out) {
  if (!path || !offset_out || begin < 0 || n < 1) { nb::set_error("nb_file_nth_newline: bad arguments"); return NB_ERR_INVALID; }
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) { nb::set_error(std::string("nb_file_nth_newline: cannot open ") + path); return NB_ERR_INVALID; }
  std::vector<char> buf((size_t)1 << 22);
  int64_t pos = begin, seen = 0; *offset_out = -1;
  for (;;) {
    const long got = (long)::pread(fd, buf.data(), buf.size(), (off_t)pos);
    if (got <= 0) break;
    const char* p = buf.data(); const char* e = p + got;
    while ((p = (const char*)std::memchr(p, '\n', (size_t)(e - p))) != nullptr) {
      ++p;
      if (++seen == n) { *offset_out = pos + (p - buf.data()); ::close(fd); return NB_OK; }
    }
    pos += got;
  }
  ::close(fd);
  return NB_OK;
}

int32_t nb_fastq_close(nb_fastq* q) {
  if (!q) return NB_OK;
  if (q->reader.joinable()) {
    { std::lock_guard<std::mutex> lk(q->mu); q->stop = true; }
    q->cv_space.notify_all();
    q->reader.join();
  }
  for (auto& c : q->chunks) std::free(c.p);
  for (char* p : q->spare) std::free(p);
  delete q->gz;
  delete q->bgzf;
  if (q->bgzf_fp) std::fclose(q->bgzf_fp);
  if (q->fd >= 0) ::close(q->fd);
  std::free(q->buf);
  delete q;
  return NB_OK;
}

int32_t nb_fastq_set_threads(nb_fastq* q, int32_t threads) {
  if (!q) { nb::set_error("nb_fastq_set_threads: null handle"); return NB_ERR_INVALID; }
  q->threads = threads < 1 ? 1 : (threads > 256 ? 256 : threads); return NB_OK;
}

// Decodes a whole gzip file (any number of members) into `out` with the engine's own inflater (nb_inflate.cpp); *n_out =
// decoded bytes. NB_ERR_OVERFLOW if cap is too small. The tests hold the inflater to zlib through this entry point.
int32_t nb_gunzip_file(const char* path, char* out, int64_t cap, int64_t* n_out) {
  if (!path || !out || !n_out || cap < 0) { nb::set_error("nb_gunzip_file: bad arguments"); return NB_ERR_INVALID; }
  const int fd = ::open(path, O_RDONLY);
  if (fd < 0) { nb::set_error(std::string("nb_gunzip_file: cannot open ") + path); return NB_ERR_INVALID; }
  nb::GzipStream gz; gz.fd = fd;
  constexpr size_t PIECE = (size_t)1 << 20;
  std::vector<uint8_t> buf(nb::GzipStream::HISTORY + PIECE + nb::GzipStream::SLACK + 64);
  int64_t total = 0; int32_t rc = NB_OK;
  for (;;) {
    const long got = gz.read(buf.data() + nb::GzipStream::HISTORY, PIECE);
    if (got < 0) { nb::set_error(std::string("nb_gunzip_file: ") + (gz.error ? gz.error : "?")); rc = NB_ERR_INVALID; break; }
    if (got == 0) break;
    if (total + got > cap) { nb::set_error("nb_gunzip_file: output buffer too small"); rc = NB_ERR_OVERFLOW; break; }
    std::memcpy(out + total, buf.data() + nb::GzipStream::HISTORY, (size_t)got); total += got;
  }
  ::close(fd);
  *n_out = total; return rc;
}

// Stage 2: finds the next up-to-max_reads records. *n_out = records found (0 = end of file), *max_seq_len_out = the
// longest sequence line among them (so that the caller can size the packed layout to the batch).
int32_t nb_fastq_index_batch(nb_fastq* q, int64_t max_reads, int64_t* n_out, int32_t* max_seq_len_out) {
  if (!q || !n_out || max_reads < 0) { nb::set_error("nb_fastq_index_batch: bad arguments"); return NB_ERR_INVALID; }
  // compact: drop consumed bytes so the offsets recorded below stay valid while the buffer grows
  if (q->pos > 0) { std::memmove(q->buf, q->buf + q->pos, q->end - q->pos); q->end -= q->pos; q->pos = 0; }
  q->n_indexed = 0; q->indexed = false; q->next_pos = 0;
  if (max_reads > 0 && !index_fast(q, max_reads)) {
    const int32_t rc = index_serial(q, max_reads);
    if (rc != NB_OK) return rc;
  }
  if (!q->src_error.empty()) { nb::set_error("FASTQ read error: " + q->src_error); return NB_ERR_INVALID; }
  const int64_t n = q->n_indexed;
  int32_t mx = 0;
  for (int64_t i = 0; i < n; ++i) mx = std::max<int32_t>(mx, (int32_t)std::min<size_t>(q->seq_e[(size_t)i] - q->seq_b[(size_t)i], (size_t)INT32_MAX));
  if (max_seq_len_out) *max_seq_len_out = mx;
  q->longest = mx;
  q->indexed = true;
  *n_out = n; return NB_OK;
}

// Stage 3: packs the records found by the last nb_fastq_index_batch (each output may be NULL; see nb_fastq_next_batch).
int32_t nb_fastq_pack_batch(nb_fastq* q, int32_t max_len, uint64_t* out_words, uint16_t* out_lens, char* cb_out, int32_t cb_len,
                            char* umi_out, int32_t umi_len, char* cbqual_out) {
  if (!q || !q->indexed) { nb::set_error("nb_fastq_pack_batch: no indexed batch"); return NB_ERR_INVALID; }
  if (out_words && (max_len < 1 || max_len > 65535)) { nb::set_error("nb_fastq_pack_batch: max_len out of range"); return NB_ERR_INVALID; }
  const int64_t n = q->n_indexed;
  const int nwb = (max_len + 31) / 32; const int64_t stride = out_words ? nb_packed_stride_words(max_len) : 0;
  const char* base = q->buf;
  const int T = (int)std::min<int64_t>(q->threads.load(), std::max<int64_t>(1, n / 4096));
  std::vector<int64_t> bad((size_t)T, -1); std::vector<int> bad_kind((size_t)T, 0);
  parallel_for(T, [&](int t) {
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
  });
  for (int t = 0; t < T; ++t) if (bad[(size_t)t] >= 0) {
      const std::string rec = std::to_string(q->record + bad[(size_t)t]);
      if (bad_kind[(size_t)t] == 1) { nb::set_error("FASTQ record " + rec + ": R1 shorter than cb_len + umi_len"); return NB_ERR_INVALID; }
      if (bad_kind[(size_t)t] == 2) { nb::set_error("FASTQ record " + rec + ": quality line shorter than cb_len"); return NB_ERR_INVALID; }
      nb::set_error("FASTQ record " + rec + ": read longer than max_len " + std::to_string(max_len)); return NB_ERR_UNSUPPORTED;
    }
  q->record += n;
  q->pos = q->next_pos;
  q->indexed = false;
  return NB_OK;
}

// The streaming entry point of align.py: up to max_reads records, indexed and packed in cache-sized sub-batches (the text
// of a sub-batch is packed while it is still in the host caches), all at the layout of ONE max_len = the longest read of
// the first sub-batch, reported in *max_len_out. The batch ends early, before a sub-batch that holds a longer read (that
// sub-batch stays indexed and opens the next call), so out_words needs max_reads * nb_packed_stride_words(cap_len) words
// where cap_len is the longest read the caller accepts. Other outputs as for nb_fastq_next_batch.
int32_t nb_fastq_next_batch_tight(nb_fastq* q, int64_t max_reads, int32_t cap_len, uint64_t* out_words, uint16_t* out_lens,
                                  int64_t* n_out, int32_t* max_len_out, char* cb_out, int32_t cb_len, char* umi_out, int32_t umi_len,
                                  char* cbqual_out) {
  if (!q || !n_out || !max_len_out || max_reads < 0) { nb::set_error("nb_fastq_next_batch_tight: bad arguments"); return NB_ERR_INVALID; }
  constexpr int64_t SUB = 32768;
  int64_t n = 0; int32_t batch_len = 0;
  while (n < max_reads) {
    int64_t m = q->n_indexed; int32_t longest = q->longest;
    if (q->indexed && m > max_reads - n) {
      if (n > 0) break;      // a pending sub-batch larger than the room left: it opens the next call
      q->indexed = false;    // (nothing is consumed until a batch is packed: index again, smaller)
    }
    if (!q->indexed) {
      const int32_t rc = nb_fastq_index_batch(q, std::min<int64_t>(SUB, max_reads - n), &m, &longest);
      if (rc != NB_OK) return rc;
      if (m == 0) { q->indexed = false; break; }
    }
    if (out_words) {
      if (longest > cap_len) { nb::set_error("FASTQ read of " + std::to_string(longest) + " bases: longer than the supported " + std::to_string(cap_len)); return NB_ERR_UNSUPPORTED; }
      if (n == 0) batch_len = std::max<int32_t>(longest, 1);
      else if (longest > batch_len) break;
    }
    const int64_t stride = out_words ? nb_packed_stride_words(batch_len) : 0;
    const int32_t rc = nb_fastq_pack_batch(q, out_words ? batch_len : 1, out_words ? out_words + n * stride : nullptr, out_lens ? out_lens + n : nullptr,
                                           cb_out ? cb_out + n * cb_len : nullptr, cb_len, umi_out ? umi_out + n * umi_len : nullptr, umi_len,
                                           cbqual_out ? cbqual_out + n * cb_len : nullptr);
    if (rc != NB_OK) return rc;
    n += m;
  }
  *n_out = n; *max_len_out = batch_len;
  return NB_OK;
}

// `nimble fastq-to-bam` (/root/reference/nimble/fastq_barcode_processor.py:131-320): 10x R1/R2 FASTQ(.gz) -> unaligned BAM whose
// records carry the whitelist-corrected cell barcode (CB:Z) and the raw UMI (UB:Z). Per pair, in file order (process_pair,
// :145-210): ids (first word of the header, without a trailing /1 resp. /2) must agree; R1 = CB + UMI + a non-empty rest; the
// CB is corrected against the whitelist (nb_whitelist_correct) or the pair is dropped; r1 = the rest of R1 with flag 77, r2 =
// all of R2 with flag 141, both unplaced (refID / pos -1, bin 4680), qualities = phred, tags CB then UB. Header as the
// reference writes it (:237-240). stats: total_pairs, written_pairs, cb_perfect_match, cb_corrected, cb_no_correction,
// name_mismatch, too_short, no_remaining_seq (the counters the reference prints, :290-317).
int32_t nb_fastq_to_bam(const char* r1_path, const char* r2_path, nb_whitelist* wl, const char* out_bam, int32_t threads, int32_t cb_len,
                        int32_t umi_len, int64_t* stats) {
  if (!r1_path || !r2_path || !wl || !out_bam || !stats || cb_len < 1 || umi_len < 0) { nb::set_error("nb_fastq_to_bam: bad arguments"); return NB_ERR_INVALID; }
  for (int i = 0; i < 8; ++i) stats[i] = 0;
  nb_fastq *q1 = nullptr, *q2 = nullptr;
  int32_t rc = nb_fastq_open(r1_path, &q1); if (rc != NB_OK) return rc;
  rc = nb_fastq_open(r2_path, &q2); if (rc != NB_OK) { nb_fastq_close(q1); return rc; }
  nb_fastq_set_threads(q1, threads); nb_fastq_set_threads(q2, threads);
  FILE* out = std::fopen(out_bam, "wb");
  if (!out) { nb_fastq_close(q1); nb_fastq_close(q2); nb::set_error(std::string("nb_fastq_to_bam: cannot write ") + out_bam); return NB_ERR_INVALID; }
  std::vector<char> raw, comp;
  auto put32 = [](std::vector<char>& v, int32_t x) { for (int j = 0; j < 4; ++j) v.push_back((char)((uint32_t)x >> (8 * j))); };
  {
    const std::string text = "@HD\tVN:1.6\tSO:queryname\n@PG\tID:nimble-fastq-to-bam\tPN:nimble\tVN:1.2\tCL:whitelist-based CB correction\n";
    raw.insert(raw.end(), {'B', 'A', 'M', 1}); put32(raw, (int32_t)text.size()); raw.insert(raw.end(), text.begin(), text.end()); put32(raw, 0);
  }
  auto flush = [&](bool all) -> int32_t {
    if (raw.empty() || (!all && raw.size() < ((size_t)32 << 20))) return NB_OK;
    const int64_t n = (int64_t)raw.size(), cap = n + (n >> 9) + 64 * ((n >> 20) + 1) + 64;
    comp.resize((size_t)cap); int64_t got = 0;
    const int32_t r = nb_gzip_members(raw.data(), n, 6, threads, comp.data(), cap, &got);
    if (r != NB_OK) return r;
    if (std::fwrite(comp.data(), 1, (size_t)got, out) != (size_t)got) { nb::set_error("nb_fastq_to_bam: write error"); return NB_ERR_INVALID; }
    raw.clear(); return NB_OK;
  };
  static const uint8_t NT16[128] = {   // htslib seq_nt16_table restricted to ASCII: "=ACMGRSVTWYHKDBN", everything else N
      15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15,
      15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 1, 2, 4, 8, 15, 15, 15, 15, 15, 15, 15, 15, 15, 0, 15, 15,
      15, 1, 14, 2, 13, 15, 15, 4, 11, 15, 15, 12, 15, 3, 15, 15, 15, 15, 5, 6, 8, 15, 7, 9, 15, 10, 15, 15, 15, 15, 15, 15,
      15, 1, 14, 2, 13, 15, 15, 4, 11, 15, 15, 12, 15, 3, 15, 15, 15, 15, 5, 6, 8, 15, 7, 9, 15, 10, 15, 15, 15, 15, 15, 15};
  constexpr int64_t SUB = 65536;
  std::vector<char> cb_raw, cb_qual, cb_fix; std::vector<uint8_t> cb_ok; std::vector<int64_t> cand;
  for (;;) {
    int64_t n1 = 0, n2 = 0;
    rc = nb_fastq_index_batch(q1, SUB, &n1, nullptr); if (rc != NB_OK) break;
    rc = nb_fastq_index_batch(q2, SUB, &n2, nullptr); if (rc != NB_OK) break;
    const int64_t n = std::min(n1, n2);                 // zip(): the shorter file ends the run
    if (n == 0) break;
    stats[0] += n;
    const char* b1 = q1->buf; const char* b2 = q2->buf;
    // pass 1 (serial, cheap): the filters in the reference's order, and the barcodes that go to the corrector
    std::vector<uint8_t> keep((size_t)n, 0);
    cand.clear(); cb_raw.clear(); cb_qual.clear();
    for (int64_t i = 0; i < n; ++i) {
      size_t a0 = q1->name_b[(size_t)i], a1 = q1->name_e[(size_t)i], c0 = q2->name_b[(size_t)i], c1 = q2->name_e[(size_t)i];
      if (a1 - a0 >= 2 && b1[a1 - 2] == '/' && b1[a1 - 1] == '1') a1 -= 2;
      if (c1 - c0 >= 2 && b2[c1 - 2] == '/' && b2[c1 - 1] == '2') c1 -= 2;
      if (a1 - a0 != c1 - c0 || std::memcmp(b1 + a0, b2 + c0, a1 - a0) != 0) { ++stats[5]; continue; }
      const int64_t L1 = (int64_t)(q1->seq_e[(size_t)i] - q1->seq_b[(size_t)i]);
      if (L1 < cb_len + umi_len) { ++stats[6]; continue; }
      if (L1 == cb_len + umi_len) { ++stats[7]; continue; }
      if ((int64_t)(q1->qual_e[(size_t)i] - q1->qual_b[(size_t)i]) < L1 || (q2->qual_e[(size_t)i] - q2->qual_b[(size_t)i]) < (q2->seq_e[(size_t)i] - q2->seq_b[(size_t)i])) {
        nb::set_error("nb_fastq_to_bam: quality line shorter than the sequence in record " + std::to_string(q1->record + i)); rc = NB_ERR_INVALID; break; }
      cand.push_back(i);
      cb_raw.insert(cb_raw.end(), b1 + q1->seq_b[(size_t)i], b1 + q1->seq_b[(size_t)i] + cb_len);
      cb_qual.insert(cb_qual.end(), b1 + q1->qual_b[(size_t)i], b1 + q1->qual_b[(size_t)i] + cb_len);
    }
    if (rc != NB_OK) break;
    cb_fix.resize(cb_raw.size()); cb_ok.assign(cand.size(), 0);
    int64_t wst[3] = {0, 0, 0};
    if (!cand.empty()) { rc = nb_whitelist_correct(wl, cb_raw.data(), cb_qual.data(), (int64_t)cand.size(), cb_len, cb_fix.data(), cb_ok.data(), wst); if (rc != NB_OK) break; }
    stats[2] = wst[0]; stats[3] = wst[1]; stats[4] = wst[2];   // (cumulative inside the corrector)
    // pass 2 (parallel): the BAM records of the surviving pairs, in file order
    const int64_t m = (int64_t)cand.size();
    const int T = (int)std::min<int64_t>(threads < 1 ? 1 : threads, std::max<int64_t>(1, m / 4096));
    std::vector<std::vector<char>> part((size_t)T);
    std::vector<int64_t> wrote((size_t)T, 0);
    parallel_for(T, [&](int t) {
      std::vector<char>& v = part[(size_t)t];
      auto p32 = [&](int32_t x) { for (int j = 0; j < 4; ++j) v.push_back((char)((uint32_t)x >> (8 * j))); };
      auto record = [&](const char* name, size_t ln, const char* seq, const char* qual, size_t L, int flag, const char* cb, const char* umi) {
        const size_t size = 32 + ln + 1 + (L + 1) / 2 + L + (3 + (size_t)cb_len + 1) + (3 + (size_t)umi_len + 1);
        p32((int32_t)size); p32(-1); p32(-1);
        v.push_back((char)(ln + 1)); v.push_back(0); v.push_back((char)(4680 & 0xff)); v.push_back((char)(4680 >> 8));   // l_read_name, mapq, bin
        v.push_back(0); v.push_back(0); v.push_back((char)(flag & 0xff)); v.push_back((char)(flag >> 8));                 // n_cigar_op, flag
        p32((int32_t)L); p32(-1); p32(-1); p32(0);
        v.insert(v.end(), name, name + ln); v.push_back(0);
        for (size_t j = 0; j < L; j += 2) { const uint8_t hi = NT16[(uint8_t)seq[j] & 127], lo = j + 1 < L ? NT16[(uint8_t)seq[j + 1] & 127] : 0; v.push_back((char)((hi << 4) | lo)); }
        for (size_t j = 0; j < L; ++j) v.push_back((char)((uint8_t)qual[j] - 33));
        v.push_back('C'); v.push_back('B'); v.push_back('Z'); v.insert(v.end(), cb, cb + cb_len); v.push_back(0);
        v.push_back('U'); v.push_back('B'); v.push_back('Z'); v.insert(v.end(), umi, umi + umi_len); v.push_back(0);
      };
      for (int64_t k = m * t / T, e = m * (t + 1) / T; k < e; ++k) {
        if (!cb_ok[(size_t)k]) continue;
        const int64_t i = cand[(size_t)k];
        size_t a0 = q1->name_b[(size_t)i], a1 = q1->name_e[(size_t)i];
        if (a1 - a0 >= 2 && b1[a1 - 2] == '/' && b1[a1 - 1] == '1') a1 -= 2;
        if (a1 - a0 > 254) a1 = a0 + 254;                // (BAM keeps read names below 255 bytes)
        const char* s1 = b1 + q1->seq_b[(size_t)i]; const char* ql1 = b1 + q1->qual_b[(size_t)i];
        const size_t L1 = q1->seq_e[(size_t)i] - q1->seq_b[(size_t)i], skip = (size_t)(cb_len + umi_len);
        const size_t L2 = q2->seq_e[(size_t)i] - q2->seq_b[(size_t)i];
        const char* cb = cb_fix.data() + k * cb_len; const char* umi = s1 + cb_len;
        record(b1 + a0, a1 - a0, s1 + skip, ql1 + skip, L1 - skip, 77, cb, umi);
        record(b1 + a0, a1 - a0, b2 + q2->seq_b[(size_t)i], b2 + q2->qual_b[(size_t)i], L2, 141, cb, umi);
        ++wrote[(size_t)t];
      }
    });
    for (int t = 0; t < T; ++t) { raw.insert(raw.end(), part[(size_t)t].begin(), part[(size_t)t].end()); stats[1] += wrote[(size_t)t]; }
    // consume the indexed records (a batch is consumed by packing it; nothing to pack here)
    rc = nb_fastq_pack_batch(q1, 1, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr); if (rc != NB_OK) break;
    rc = nb_fastq_pack_batch(q2, 1, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr); if (rc != NB_OK) break;
    rc = flush(false); if (rc != NB_OK) break;
    if (n1 != n2) break;
  }
  if (rc == NB_OK) rc = flush(true);
  if (rc == NB_OK) {
    static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (std::fwrite(eof_block, 1, 28, out) != 28) { nb::set_error("nb_fastq_to_bam: write error"); rc = NB_ERR_INVALID; }
  }
  if (std::fclose(out) != 0 && rc == NB_OK) { nb::set_error("nb_fastq_to_bam: write error"); rc = NB_ERR_INVALID; }
  nb_fastq_close(q1); nb_fastq_close(q2);
  return rc;
}

// Reads up to max_reads records; *n_out = records read (0 = end of file).
//   out_words != NULL: the whole sequence line is packed at stride nb_packed_stride_words(max_len), length to out_lens.
//   cb_out / umi_out != NULL: bases [0, cb_len) and [cb_len, cb_len + umi_len) of the sequence line are copied, fixed
//   width, to cb_out[i * cb_len] / umi_out[i * umi_len] (the 10x R1 layout); cbqual_out != NULL additionally receives
//   the quality characters of the CB bases (for the lowest-quality tie-break of the whitelist correction).
int32_t nb_fastq_next_batch(nb_fastq* q, int64_t max_reads, int32_t max_len, uint64_t* out_words, uint16_t* out_lens,
                            int64_t* n_out, char* cb_out, int32_t cb_len, char* umi_out, int32_t umi_len, char* cbqual_out) {
  if (!q || !n_out || max_reads < 0) { nb::set_error("nb_fastq_next_batch: bad arguments"); return NB_ERR_INVALID; }
  int32_t rc = nb_fastq_index_batch(q, max_reads, n_out, nullptr);
  if (rc != NB_OK) return rc;
  return nb_fastq_pack_batch(q, max_len, out_words, out_lens, cb_out, cb_len, umi_out, umi_len, cbqual_out);
}

}  // extern "C"
