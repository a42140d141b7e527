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
//   * paired records (0x1) must be adjacent mates (first 0x40 then last 0x80, same name): the order samtools sort -n /
//     collate and the reference's fastq-to-bam produce; anything else is an error, not a silent mis-pairing.
// BGZF is a series of gzip members; zlib's gzread walks across members transparently.
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "nb_internal.h"

struct nb_bam {
  gzFile f = nullptr;
  int64_t record = 0;
  bool have_held = false;
  std::vector<uint8_t> held;   // a record read ahead (mate lookup)
  int paired = -1;             // -1 unknown, 0 single-end, 1 adjacent mates (fixed by the first primary record)
};

namespace {

bool read_exact(nb_bam* b, void* dst, size_t n) {
  size_t got = 0;
  while (got < n) {
    const int r = gzread(b->f, (char*)dst + got, (unsigned)std::min<size_t>(n - got, (size_t)1 << 30));
    if (r <= 0) return false;
    got += (size_t)r;
  }
  return true;
}

int32_t le32(const uint8_t* p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// 1 = record read into rec (without the block_size word), 0 = clean end of file, <0 = error
int next_record(nb_bam* b, std::vector<uint8_t>& rec) {
  if (b->have_held) { rec.swap(b->held); b->have_held = false; return 1; }
  uint8_t szb[4];
  const int r = gzread(b->f, szb, 4);
  if (r == 0) return 0;
  if (r != 4) { nb::set_error("BAM: truncated record header"); return NB_ERR_INVALID; }
  const int32_t sz = le32(szb);
  if (sz < 32 || sz > (1 << 28)) { nb::set_error("BAM: implausible record size"); return NB_ERR_INVALID; }
  rec.resize((size_t)sz);
  if (!read_exact(b, rec.data(), (size_t)sz)) { nb::set_error("BAM: truncated record"); return NB_ERR_INVALID; }
  return 1;
}

struct Fields { uint16_t flag; int32_t l_seq; const uint8_t* name; int l_name; const uint8_t* seq; const uint8_t* tags; const uint8_t* end; };

bool parse(const std::vector<uint8_t>& rec, Fields& f) {
  const uint8_t* p = rec.data();
  const int l_read_name = p[8]; const int n_cigar = le16(p + 12);
  f.flag = le16(p + 14); f.l_seq = le32(p + 16);
  size_t off = 32;
  f.name = p + off; f.l_name = l_read_name; off += (size_t)l_read_name;
  off += (size_t)n_cigar * 4;
  f.seq = p + off; off += (size_t)(f.l_seq + 1) / 2;
  off += (size_t)f.l_seq;   // qualities
  if (f.l_seq < 0 || off > rec.size()) return false;
  f.tags = p + off; f.end = p + rec.size();
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
    p += skip;
  }
  return nullptr;
}

// decode the 4-bit sequence into the packed device layout (restoring the sequenced orientation when 0x10 is set)
void pack_seq(const Fields& f, int nwb, int64_t stride, uint64_t* w) {
  static const uint8_t code[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};   // "=ACMGRSVTWYHKDBN" -> A C G T else N
  std::memset(w, 0, (size_t)stride * 8);
  const bool rev = f.flag & 0x10;
  const int L = f.l_seq;
  for (int j = 0; j < L; ++j) {
    const int src = rev ? L - 1 - j : j;
    uint8_t c = code[(f.seq[src >> 1] >> ((src & 1) ? 0 : 4)) & 15];
    if (c != 4 && rev) c = (uint8_t)(3 - c);
    if (c == 4) w[nwb + (j >> 6)] |= 1ull << (j & 63);
    else w[j >> 5] |= (uint64_t)c << ((j & 31) * 2);
  }
}

}  // namespace

extern "C" {

int32_t nb_bam_open(const char* path, nb_bam** out) {
  if (!path || !out) { nb::set_error("nb_bam_open: null argument"); return NB_ERR_INVALID; }
  gzFile f = gzopen(path, "rb");
  if (!f) { nb::set_error(std::string("nb_bam_open: cannot open ") + path); return NB_ERR_INVALID; }
  gzbuffer(f, 1 << 20);
  nb_bam* b = new nb_bam(); b->f = f;
  uint8_t hdr[8];
  auto fail = [&](const char* m) { nb::set_error(m); gzclose(f); delete b; return NB_ERR_INVALID; };
  if (!read_exact(b, hdr, 8) || std::memcmp(hdr, "BAM\1", 4) != 0) return fail("nb_bam_open: not a BAM file");
  const int32_t l_text = le32(hdr + 4);
  if (l_text < 0 || l_text > (1 << 30)) return fail("nb_bam_open: implausible header length");
  std::vector<uint8_t> text((size_t)l_text);
  if (l_text && !read_exact(b, text.data(), (size_t)l_text)) return fail("nb_bam_open: truncated header text");
  uint8_t nb4[4];
  if (!read_exact(b, nb4, 4)) return fail("nb_bam_open: truncated header");
  const int32_t n_ref = le32(nb4);
  if (n_ref < 0) return fail("nb_bam_open: bad reference count");
  for (int32_t i = 0; i < n_ref; ++i) {
    uint8_t l4[4];
    if (!read_exact(b, l4, 4)) return fail("nb_bam_open: truncated reference list");
    const int32_t l_name = le32(l4);
    if (l_name < 0 || l_name > (1 << 20)) return fail("nb_bam_open: implausible reference name length");
    std::vector<uint8_t> nm((size_t)l_name + 4);
    if (!read_exact(b, nm.data(), (size_t)l_name + 4)) return fail("nb_bam_open: truncated reference list");
  }
  *out = b; return NB_OK;
}

int32_t nb_bam_close(nb_bam* b) { if (b) { if (b->f) gzclose(b->f); delete b; } return NB_OK; }

// Reads up to max_templates templates. r1 (and, when the records are paired, r2) go to the packed layout at
// nb_packed_stride_words(max_len); *paired_out = 1 if this batch holds mates (all templates of a file are alike).
// cb_out / ub_out: tag_width bytes per template, NUL padded (missing tag = empty). *n_out = templates read (0 = end).
int32_t nb_bam_next_batch(nb_bam* b, int64_t max_templates, int32_t max_len, uint64_t* r1_words, uint16_t* r1_lens, uint64_t* r2_words,
                          uint16_t* r2_lens, char* cb_out, char* ub_out, int32_t tag_width, int64_t* n_out, int32_t* paired_out) {
  if (!b || !r1_words || !r1_lens || !cb_out || !ub_out || !n_out || !paired_out || tag_width < 1) { nb::set_error("nb_bam_next_batch: bad arguments"); return NB_ERR_INVALID; }
  const int nwb = (max_len + 31) / 32; const int64_t stride = nb_packed_stride_words(max_len);
  std::vector<uint8_t> rec, rec2;
  int64_t n = 0; int& paired = b->paired;
  auto put_tag = [&](const Fields& f, const char* t, char* dst) {
    std::memset(dst, 0, (size_t)tag_width);
    size_t len = 0; const char* v = find_z_tag(f, t, &len);
    if (v) std::memcpy(dst, v, std::min(len, (size_t)tag_width));
  };
  while (n < max_templates) {
    const int r = next_record(b, rec);
    if (r == 0) break;
    if (r < 0) return r;
    Fields f;
    if (!parse(rec, f)) { nb::set_error("BAM: malformed record " + std::to_string(b->record)); return NB_ERR_INVALID; }
    ++b->record;
    if (f.flag & 0x900) continue;                       // secondary / supplementary
    if (f.l_seq > max_len) { nb::set_error("BAM record " + std::to_string(b->record - 1) + ": read longer than max_len " + std::to_string(max_len)); return NB_ERR_UNSUPPORTED; }
    const bool is_paired = f.flag & 0x1;
    if (paired < 0) paired = is_paired ? 1 : 0;
    if ((paired == 1) != is_paired) { nb::set_error("BAM: mixed single-end and paired records"); return NB_ERR_UNSUPPORTED; }
    if (!is_paired) {
      pack_seq(f, nwb, stride, r1_words + n * stride); r1_lens[n] = (uint16_t)f.l_seq;
      put_tag(f, "CB", cb_out + n * tag_width); put_tag(f, "UB", ub_out + n * tag_width);
      ++n; continue;
    }
    if (!r2_words || !r2_lens) { nb::set_error("nb_bam_next_batch: paired BAM but no mate buffers"); return NB_ERR_INVALID; }
    // the mate must be the next primary record
    Fields g; bool got = false;
    for (;;) {
      const int r2 = next_record(b, rec2);
      if (r2 < 0) return r2;
      if (r2 == 0) break;
      if (!parse(rec2, g)) { nb::set_error("BAM: malformed record " + std::to_string(b->record)); return NB_ERR_INVALID; }
      ++b->record;
      if (g.flag & 0x900) continue;
      got = true; break;
    }
    const bool names_match = got && f.l_name == g.l_name && std::memcmp(f.name, g.name, (size_t)f.l_name) == 0;
    if (!got || !names_match || !(g.flag & 0x1) || ((f.flag & 0xC0) == (g.flag & 0xC0))) {
      nb::set_error("BAM: paired records are not adjacent mates near record " + std::to_string(b->record) +
                    " (collate the file by read name: the reference itself runs samtools sort -n, nimble/__main__.py:344-347)");
      return NB_ERR_UNSUPPORTED;
    }
    if (g.l_seq > max_len) { nb::set_error("BAM record " + std::to_string(b->record - 1) + ": read longer than max_len " + std::to_string(max_len)); return NB_ERR_UNSUPPORTED; }
    const bool f_first = (f.flag & 0x40) || !(g.flag & 0x40);
    const Fields& a1 = f_first ? f : g; const Fields& a2 = f_first ? g : f;
    pack_seq(a1, nwb, stride, r1_words + n * stride); r1_lens[n] = (uint16_t)a1.l_seq;
    pack_seq(a2, nwb, stride, r2_words + n * stride); r2_lens[n] = (uint16_t)a2.l_seq;
    put_tag(a1, "CB", cb_out + n * tag_width); put_tag(a1, "UB", ub_out + n * tag_width);   // the r1 version of the tags (__main__.py:236-237)
    ++n;
  }
  *n_out = n; *paired_out = paired == 1 ? 1 : 0; return NB_OK;
}

}  // extern "C"
