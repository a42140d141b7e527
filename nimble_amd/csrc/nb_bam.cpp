// nb_bam.cpp — minimal BAM reader for `nimble align --input x.bam` (host, product code; zlib only, no htslib).
//
// The reference hands a BAM to the Rust aligner after `samtools sort -t UR -n` (/root/reference/nimble/__main__.py:171-175,
// 312-370): the sort only exists so that the aligner sees one UMI's records together. This engine emits one row per
// template and lets `report` group by (cell, UMI) afterwards, so no sort is needed; the records are consumed in file
// order. What is read from each record (SAM/BAM spec v1.6 §4.2): flag, l_seq, the 4-bit sequence, the qualities are
// skipped, and the Z-typed tags CB and UB (the columns r1_CB / r1_UB consumed at __main__.py:237-245; they are the
// tags the reference's own fastq-to-bam writes, fastq_barcode_processor.py:189-190,203-204).
//   * secondary / supplementary records (0x100, 0x800) are skipped;
//   * a record stored reverse-complemented (0x10) is turned back into the read as sequenced;
//   * paired records (0x1): adjacent mates (the order samtools sort -n / collate and the reference's fastq-to-bam produce)
//     are paired on the spot; in any other order (coordinate-sorted files) a record waits, keyed by read name, until its
//     mate shows up — what the reference obtains from its samtools sort -n staging; a read whose mate never comes is
//     handed on with an empty partner.
// BGZF is a series of gzip members of <= 64 KiB, each carrying its compressed size in a 'BC' extra field (SAM spec §4.1),
// so members inflate independently: a refill reads a run of whole blocks and inflates them on `threads` workers, each into
// its own slice of the output (sizes come from the ISIZE trailers) with the decoder of nb_inflate.cpp, CRC of every block checked. A gzip file
// that is not BGZF (one long member) falls back to zlib's serial gzread. Records are then gathered serially (a walk
// over the block_size chain plus the mate-adjacency rule) and packed in parallel.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "nb_inflate.h"
#include "nb_internal.h"

namespace {

// offsets and sizes of the record bodies of one template; an offset of SIZE_MAX = that mate is absent (single-end input, or an
// orphan of a paired file); side bit 0 / 1 = record a / b lives in the side arena (a mate that waited for its partner)
// instead of the inflated window
struct Tmpl { size_t a, b; uint32_t a_size, b_size; uint32_t side; };

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

struct nb_bam {
  FILE* fp = nullptr;          // BGZF path
  nb::BgzfReader bgzf;         // (block-parallel inflate, nb_inflate.cpp)
  gzFile gz = nullptr;         // fallback: plain gzip
  int threads = 1;
  std::vector<uint8_t> ubuf;   // inflated bytes: [upos, uend) unconsumed
  size_t upos = 0, uend = 0;
  bool eof = false;
  std::string error;
  int64_t record = 0;
  int paired = -1;             // -1 unknown, 0 single-end, 1 adjacent mates (fixed by the first primary record)
  // a paired file that is not collated by name (e.g. coordinate-sorted): a record whose mate is not the next primary
  // record waits here, keyed by read name, until the mate shows up (the reference gets the same effect from its
  // samtools sort -n staging, nimble/__main__.py:344-347); what is still waiting at the end of the file is an orphan
  std::unordered_map<std::string, std::vector<uint8_t>> waiting;
  std::vector<uint8_t> side;   // records of pending templates that came out of `waiting`
  // templates gathered but not yet packed (a batch may end before them: see nb_bam_next_batch_tight)
  std::vector<Tmpl> pend;
  size_t pend_next_upos = 0;
  int pend_l1 = 0, pend_l2 = 0;
  bool have_pend = false;
};

namespace {

int32_t le32(const uint8_t* p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// inflate more data behind uend; false at end of file or on error (b->error set)
bool refill(nb_bam* b) {
  if (b->eof) return false;
  const size_t want = (size_t)32 << 20;
  if (b->ubuf.size() < b->uend + want) b->ubuf.resize(std::max(b->ubuf.size() * 2, b->uend + want));
  if (b->gz) {
    const int r = gzread(b->gz, b->ubuf.data() + b->uend, (unsigned)want);
    if (r < 0) { int e = 0; const char* m = gzerror(b->gz, &e); b->error = std::string("BAM: gzip stream error: ") + (m ? m : "?"); b->eof = true; return false; }
    if (r == 0) { b->eof = true; return false; }
    b->uend += (size_t)r; return true;
  }
  b->bgzf.threads = b->threads;
  const long got = b->bgzf.read(b->ubuf.data() + b->uend, want);
  if (got < 0) { b->error = std::string("BAM: ") + (b->bgzf.error ? b->bgzf.error : "BGZF error"); b->eof = true; return false; }
  if (got == 0) { b->eof = true; return false; }
  b->uend += (size_t)got; return true;
}

// makes [off, off + n) of ubuf valid; false when the file ends first
bool ensure(nb_bam* b, size_t off, size_t n) {
  while (b->uend < off + n) if (!refill(b)) return false;
  return true;
}

// 1 = record body at [*off, *off + *size) (cursor advanced past it), 0 = clean end of file, <0 = error
int next_record(nb_bam* b, size_t* cursor, size_t* off, uint32_t* size) {
  if (!ensure(b, *cursor, 4)) {
    if (!b->error.empty()) { nb::set_error(b->error); return NB_ERR_INVALID; }
    if (b->uend > *cursor) { nb::set_error("BAM: truncated record header"); return NB_ERR_INVALID; }
    return 0;
  }
  const int32_t sz = le32(b->ubuf.data() + *cursor);
  if (sz < 32 || sz > (1 << 28)) { nb::set_error("BAM: implausible record size"); return NB_ERR_INVALID; }
  if (!ensure(b, *cursor + 4, (size_t)sz)) { nb::set_error(b->error.empty() ? "BAM: truncated record" : b->error); return NB_ERR_INVALID; }
  *off = *cursor + 4; *size = (uint32_t)sz; *cursor += 4 + (size_t)sz;
  return 1;
}

struct Fields { uint16_t flag; int32_t l_seq; const uint8_t* name; int l_name; const uint8_t* seq; const uint8_t* tags; const uint8_t* end; };

bool parse(const uint8_t* p, size_t size, Fields& f) {
  const int l_read_name = p[8]; const int n_cigar = le16(p + 12);
  f.flag = le16(p + 14); f.l_seq = le32(p + 16);
  size_t off = 32;
  f.name = p + off; f.l_name = l_read_name; off += (size_t)l_read_name;
  off += (size_t)n_cigar * 4;
  if (f.l_seq < 0) return false;
  f.seq = p + off; off += (size_t)(f.l_seq + 1) / 2;
  off += (size_t)f.l_seq;   // qualities
  if (off > size) return false;
  f.tags = p + off; f.end = p + size;
  return true;
}

// finds Z-typed tag `t`; returns pointer/length of the string or nullptr
const char* find_z_tag(const Fields& f, const char* t, size_t* len) {
  const uint8_t* p = f.tags;
  while (p + 3 <= f.end) {
    const char a = (char)p[0], b = (char)p[1], ty = (char)p[2]; p += 3;
    size_t skip = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': skip = 1; break;
      case 's': case 'S': skip = 2; break;
      case 'i': case 'I': case 'f': skip = 4; break;
      case 'Z': case 'H': { const uint8_t* e = (const uint8_t*)std::memchr(p, 0, (size_t)(f.end - p)); if (!e) return nullptr;
        if (ty == 'Z' && a == t[0] && b == t[1]) { *len = (size_t)(e - p); return (const char*)p; }
        skip = (size_t)(e - p) + 1; break; }
      case 'B': { if (p + 5 > f.end) return nullptr; const char sub = (char)p[0]; const uint32_t cnt = (uint32_t)le32(p + 1);
        const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4); skip = 5 + es * cnt; break; }
      default: return nullptr;
    }
    if (skip > (size_t)(f.end - p)) return nullptr;
    p += skip;
  }
  return nullptr;
}

// decode the 4-bit sequence into the packed device layout (restoring the sequenced orientation when 0x10 is set)
void pack_seq(const Fields& f, int nwb, int64_t stride, uint64_t* w) {
  static const uint8_t code[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};   // "=ACMGRSVTWYHKDBN" -> A C G T else N
  const bool rev = f.flag & 0x10;
  const int L = f.l_seq;
  int j = 0, word = 0;
  for (; word < nwb && j < L; ++word) {
    const int m = std::min(32, L - j);
    uint64_t bits = 0, nm = 0;
    for (int u = 0; u < m; ++u) {
      const int src = rev ? L - 1 - (j + u) : j + u;
      uint64_t c = code[(f.seq[src >> 1] >> ((src & 1) ? 0 : 4)) & 15];
      if (rev && c != 4) c = 3 - c;
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

// gathers up to max_t templates starting at upos into b->pend (offsets only; the window is not consumed until they are packed)
int32_t gather(nb_bam* b, int64_t max_t) {
  b->pend.clear(); b->side.clear(); b->pend_l1 = b->pend_l2 = 0; b->have_pend = false;
  size_t cursor = b->upos;
  int& paired = b->paired;
  auto to_side = [&](const std::vector<uint8_t>& rec) { const size_t o = b->side.size(); b->side.insert(b->side.end(), rec.begin(), rec.end()); return o; };
  while ((int64_t)b->pend.size() < max_t) {
    size_t off; uint32_t size;
    const int r = next_record(b, &cursor, &off, &size);
    if (r < 0) return r;
    if (r == 0) {                                       // end of file: what still waits for a mate is an orphan
      if (b->waiting.empty()) break;
      auto it = b->waiting.begin();
      Fields o; parse(it->second.data(), it->second.size(), o);
      const size_t so = to_side(it->second); const uint32_t sz = (uint32_t)it->second.size();
      if (o.flag & 0x80) { b->pend.push_back(Tmpl{SIZE_MAX, so, 0, sz, 2u}); b->pend_l2 = std::max(b->pend_l2, o.l_seq); }
      else { b->pend.push_back(Tmpl{so, SIZE_MAX, sz, 0, 1u}); b->pend_l1 = std::max(b->pend_l1, o.l_seq); }
      b->waiting.erase(it);
      continue;
    }
    Fields f;
    if (!parse(b->ubuf.data() + off, size, f)) { nb::set_error("BAM: malformed record " + std::to_string(b->record)); return NB_ERR_INVALID; }
    ++b->record;
    if (f.flag & 0x900) continue;                       // secondary / supplementary
    const bool is_paired = f.flag & 0x1;
    if (paired < 0) paired = is_paired ? 1 : 0;
    if ((paired == 1) != is_paired) { nb::set_error("BAM: mixed single-end and paired records"); return NB_ERR_UNSUPPORTED; }
    if (!is_paired) {
      b->pend.push_back(Tmpl{off, SIZE_MAX, size, 0, 0u});
      b->pend_l1 = std::max(b->pend_l1, f.l_seq);
      continue;
    }
    // fast path: the mate is the next primary record (name-collated files, the reference's fastq-to-bam output)
    Fields g; bool got = false; size_t off2 = 0; uint32_t size2 = 0;
    const size_t cursor_after_f = cursor; const int64_t record_after_f = b->record;
    for (;;) {
      const int r2 = next_record(b, &cursor, &off2, &size2);
      if (r2 < 0) return r2;
      if (r2 == 0) break;
      if (!parse(b->ubuf.data() + off2, size2, g)) { nb::set_error("BAM: malformed record " + std::to_string(b->record)); return NB_ERR_INVALID; }
      ++b->record;
      if (g.flag & 0x900) continue;
      got = true; break;
    }
    parse(b->ubuf.data() + off, size, f);               // (the buffer may have moved while the next record was read)
    const bool mates = got && (g.flag & 0x1) && f.l_name == g.l_name && std::memcmp(f.name, g.name, (size_t)f.l_name) == 0 && ((f.flag & 0xC0) != (g.flag & 0xC0));
    if (mates) {
      const bool f_first = (f.flag & 0x40) || !(g.flag & 0x40);
      if (f_first) { b->pend.push_back(Tmpl{off, off2, size, size2, 0u}); b->pend_l1 = std::max(b->pend_l1, f.l_seq); b->pend_l2 = std::max(b->pend_l2, g.l_seq); }
      else { b->pend.push_back(Tmpl{off2, off, size2, size, 0u}); b->pend_l1 = std::max(b->pend_l1, g.l_seq); b->pend_l2 = std::max(b->pend_l2, f.l_seq); }
      continue;
    }
    // not collated by name: f pairs with a record seen earlier, or waits; the record after it is looked at again
    cursor = cursor_after_f; b->record = record_after_f;
    std::string key((const char*)f.name, (size_t)f.l_name);
    auto it = b->waiting.find(key);
    if (it == b->waiting.end()) {
      if (b->waiting.size() >= ((size_t)1 << 26)) { nb::set_error("BAM: more than 2^26 reads wait for their mates (collate the file by read name)"); return NB_ERR_UNSUPPORTED; }
      b->waiting.emplace(std::move(key), std::vector<uint8_t>(b->ubuf.data() + off, b->ubuf.data() + off + size));
      continue;
    }
    Fields o; parse(it->second.data(), it->second.size(), o);
    if ((o.flag & 0xC0) == (f.flag & 0xC0)) { nb::set_error("BAM: two records of read '" + key.substr(0, key.find('\0')) + "' claim the same end of the pair"); return NB_ERR_INVALID; }
    const size_t so = to_side(it->second); const uint32_t sz = (uint32_t)it->second.size();
    const bool f_first = (f.flag & 0x40) || !(o.flag & 0x40);
    if (f_first) { b->pend.push_back(Tmpl{off, so, size, sz, 2u}); b->pend_l1 = std::max(b->pend_l1, f.l_seq); b->pend_l2 = std::max(b->pend_l2, o.l_seq); }
    else { b->pend.push_back(Tmpl{so, off, sz, size, 1u}); b->pend_l1 = std::max(b->pend_l1, o.l_seq); b->pend_l2 = std::max(b->pend_l2, f.l_seq); }
    b->waiting.erase(it);
  }
  b->pend_next_upos = cursor;
  b->have_pend = true;
  return NB_OK;
}

// packs the first `count` gathered templates at layouts (len1, len2); the window is consumed once all of them are packed
// (a gather is never redone: it moves records in and out of `waiting`)
void pack_pending(nb_bam* b, int64_t count, int32_t len1, int32_t len2, uint64_t* r1_words, uint16_t* r1_lens, uint64_t* r2_words, uint16_t* r2_lens,
                  char* cb_out, char* ub_out, int32_t tag_width) {
  const int64_t n = std::min<int64_t>(count, (int64_t)b->pend.size());
  const int nwb1 = (len1 + 31) / 32, nwb2 = (len2 + 31) / 32;
  const int64_t s1 = nb_packed_stride_words(len1), s2 = b->paired == 1 ? nb_packed_stride_words(len2) : 0;
  const uint8_t* base = b->ubuf.data(); const uint8_t* side = b->side.data();
  const int T = (int)std::min<int64_t>(b->threads, std::max<int64_t>(1, n / 2048));
  auto put_tag = [&](const Fields& f, const char* t, char* dst) {
    std::memset(dst, 0, (size_t)tag_width);
    size_t len = 0; const char* v = find_z_tag(f, t, &len);
    if (v) std::memcpy(dst, v, std::min(len, (size_t)tag_width));
  };
  parallel_for(T, [&](int t) {
    for (int64_t i = n * t / T, e = n * (t + 1) / T; i < e; ++i) {
      const Tmpl& m = b->pend[(size_t)i];
      Fields f, g; const bool has_a = m.a != SIZE_MAX, has_b = m.b != SIZE_MAX;
      if (has_a) { parse(((m.side & 1u) ? side : base) + m.a, m.a_size, f); pack_seq(f, nwb1, s1, r1_words + i * s1); r1_lens[i] = (uint16_t)f.l_seq; }
      else { std::memset(r1_words + i * s1, 0, (size_t)s1 * 8); r1_lens[i] = 0; }
      if (b->paired == 1) {
        if (has_b) { parse(((m.side & 2u) ? side : base) + m.b, m.b_size, g); pack_seq(g, nwb2, s2, r2_words + i * s2); r2_lens[i] = (uint16_t)g.l_seq; }
        else { std::memset(r2_words + i * s2, 0, (size_t)s2 * 8); r2_lens[i] = 0; }
      }
      const Fields& tagsrc = has_a ? f : g;              // the r1 version of the tags (__main__.py:236-237); an orphan r2 speaks for itself
      put_tag(tagsrc, "CB", cb_out + i * tag_width); put_tag(tagsrc, "UB", ub_out + i * tag_width);
    }
  });
  b->pend.erase(b->pend.begin(), b->pend.begin() + (std::ptrdiff_t)n);
  if (b->pend.empty()) { b->upos = b->pend_next_upos; b->side.clear(); b->have_pend = false; }
}

void compact(nb_bam* b) {
  if (b->have_pend || b->upos == 0) return;   // offsets of gathered templates must stay valid
  if (b->upos < ((size_t)16 << 20) && b->upos < b->uend / 2) return;   // (amortised: a refill leaves ~32 MB behind uend; moving that per sub-batch would copy every byte five times)
  std::memmove(b->ubuf.data(), b->ubuf.data() + b->upos, b->uend - b->upos);
  b->uend -= b->upos; b->upos = 0;
}

}  // namespace

extern "C" {

int32_t nb_bam_open(const char* path, nb_bam** out) {
  if (!path || !out) { nb::set_error("nb_bam_open: null argument"); return NB_ERR_INVALID; }
  FILE* fp = std::fopen(path, "rb");
  if (!fp) { nb::set_error(std::string("nb_bam_open: cannot open ") + path); return NB_ERR_INVALID; }
  uint8_t first[18]; const size_t got = std::fread(first, 1, 18, fp);
  nb_bam* b = new nb_bam();
  auto fail = [&](const std::string& m) { nb::set_error(m); nb_bam_close(b); return NB_ERR_INVALID; };
  if (nb::BgzfReader::block_size(first, got) > 0) {
    std::rewind(fp); b->fp = fp; b->bgzf.fp = fp;
  } else {
    std::fclose(fp);
    b->gz = gzopen(path, "rb");
    if (!b->gz) return fail(std::string("nb_bam_open: cannot open ") + path);
    gzbuffer(b->gz, 1 << 20);
  }
  auto need = [&](size_t off, size_t n) { return ensure(b, off, n); };
  if (!need(0, 8) || std::memcmp(b->ubuf.data(), "BAM\1", 4) != 0) return fail(b->error.empty() ? "nb_bam_open: not a BAM file" : b->error);
  const int32_t l_text = le32(b->ubuf.data() + 4);
  if (l_text < 0 || l_text > (1 << 30)) return fail("nb_bam_open: implausible header length");
  size_t cur = 8 + (size_t)l_text;
  if (!need(cur, 4)) return fail("nb_bam_open: truncated header");
  const int32_t n_ref = le32(b->ubuf.data() + cur); cur += 4;
  if (n_ref < 0) return fail("nb_bam_open: bad reference count");
  for (int32_t i = 0; i < n_ref; ++i) {
    if (!need(cur, 4)) return fail("nb_bam_open: truncated reference list");
    const int32_t l_name = le32(b->ubuf.data() + cur);
    if (l_name < 0 || l_name > (1 << 20)) return fail("nb_bam_open: implausible reference name length");
    if (!need(cur + 4, (size_t)l_name + 4)) return fail("nb_bam_open: truncated reference list");
    cur += 8 + (size_t)l_name;
  }
  b->upos = cur;
  *out = b; return NB_OK;
}

int32_t nb_bam_close(nb_bam* b) { if (b) { if (b->fp) std::fclose(b->fp); if (b->gz) gzclose(b->gz); delete b; } return NB_OK; }

int32_t nb_bam_set_threads(nb_bam* b, int32_t threads) {
  if (!b) { nb::set_error("nb_bam_set_threads: null handle"); return NB_ERR_INVALID; }
  b->threads = threads < 1 ? 1 : (threads > 256 ? 256 : threads); return NB_OK;
}

// Reads up to max_templates templates. r1 (and, when the records are paired, r2) go to the packed layout at
// nb_packed_stride_words(max_len); *paired_out = 1 if this batch holds mates (all templates of a file are alike).
// cb_out / ub_out: tag_width bytes per template, NUL padded (missing tag = empty). *n_out = templates read (0 = end).
int32_t nb_bam_next_batch(nb_bam* b, int64_t max_templates, int32_t max_len, uint64_t* r1_words, uint16_t* r1_lens, uint64_t* r2_words,
                          uint16_t* r2_lens, char* cb_out, char* ub_out, int32_t tag_width, int64_t* n_out, int32_t* paired_out) {
  if (!b || !r1_words || !r1_lens || !cb_out || !ub_out || !n_out || !paired_out || tag_width < 1 || max_len < 1 || max_templates < 0) { nb::set_error("nb_bam_next_batch: bad arguments"); return NB_ERR_INVALID; }
  if (!b->have_pend) {
    compact(b);
    const int32_t rc = gather(b, max_templates);
    if (rc != NB_OK) return rc;
  }
  if (std::max(b->pend_l1, b->pend_l2) > max_len) { nb::set_error("BAM: read of " + std::to_string(std::max(b->pend_l1, b->pend_l2)) + " bases: longer than max_len " + std::to_string(max_len)); return NB_ERR_UNSUPPORTED; }
  if (b->paired == 1 && !b->pend.empty() && (!r2_words || !r2_lens)) { nb::set_error("nb_bam_next_batch: paired BAM but no mate buffers"); return NB_ERR_INVALID; }
  const int64_t m = std::min<int64_t>(max_templates, (int64_t)b->pend.size());
  *n_out = m; *paired_out = b->paired == 1 ? 1 : 0;
  if (b->pend.empty()) b->have_pend = false;
  else pack_pending(b, m, max_len, max_len, r1_words, r1_lens, r2_words, r2_lens, cb_out, ub_out, tag_width);
  return NB_OK;
}

// Streaming form used by align.py, as nb_fastq_next_batch_tight: cache-sized sub-batches, r1 packed at the layout of
// *len1_out and r2 at *len2_out (the longest r1 / r2 of the first sub-batch); the batch ends early before a sub-batch that
// holds a longer read. The word buffers hold max_templates * nb_packed_stride_words(cap_len) words.
int32_t nb_bam_next_batch_tight(nb_bam* b, int64_t max_templates, int32_t cap_len, uint64_t* r1_words, uint16_t* r1_lens, uint64_t* r2_words,
                                uint16_t* r2_lens, char* cb_out, char* ub_out, int32_t tag_width, int64_t* n_out, int32_t* len1_out,
                                int32_t* len2_out, int32_t* paired_out) {
  if (!b || !r1_words || !r1_lens || !cb_out || !ub_out || !n_out || !len1_out || !len2_out || !paired_out || tag_width < 1) { nb::set_error("nb_bam_next_batch_tight: bad arguments"); return NB_ERR_INVALID; }
  constexpr int64_t SUB = 32768;
  int64_t n = 0; int32_t len1 = 1, len2 = 1;
  while (n < max_templates) {
    if (!b->have_pend) {
      compact(b);
      const int32_t rc = gather(b, std::min<int64_t>(SUB, max_templates - n));
      if (rc != NB_OK) return rc;
    }
    if (b->pend.empty()) { b->have_pend = false; break; }
    if (std::max(b->pend_l1, b->pend_l2) > cap_len) { nb::set_error("BAM: read of " + std::to_string(std::max(b->pend_l1, b->pend_l2)) + " bases: longer than the supported " + std::to_string(cap_len)); return NB_ERR_UNSUPPORTED; }
    if (b->paired == 1 && (!r2_words || !r2_lens)) { nb::set_error("nb_bam_next_batch_tight: paired BAM but no mate buffers"); return NB_ERR_INVALID; }
    if (n == 0) { len1 = std::max(b->pend_l1, 1); len2 = std::max(b->pend_l2, 1); }
    else if (b->pend_l1 > len1 || b->pend_l2 > len2) break;
    const int64_t m = std::min<int64_t>((int64_t)b->pend.size(), max_templates - n);   // (what does not fit stays gathered)
    const int64_t s1 = nb_packed_stride_words(len1), s2 = nb_packed_stride_words(len2);
    pack_pending(b, m, len1, len2, r1_words + n * s1, r1_lens + n, r2_words ? r2_words + n * s2 : nullptr, r2_lens ? r2_lens + n : nullptr,
                 cb_out + n * tag_width, ub_out + n * tag_width, tag_width);
    n += m;
  }
  *n_out = n; *len1_out = len1; *len2_out = len2; *paired_out = b->paired == 1 ? 1 : 0;
  return NB_OK;
}

}  // extern "C"
