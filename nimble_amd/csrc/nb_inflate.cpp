// nb_inflate.cpp — see nb_inflate.h. RFC 1951 restated; table layout and loop structure are this file's own.
#include "nb_inflate.h"

#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <thread>

namespace nb {

namespace {

// table entry: bits [0,8) = bits to consume, [8,12) = kind, [12,16) = extra-bit count (or sub-table width), [16,32) = payload
enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
constexpr uint32_t entry(uint32_t kind, uint32_t bits, uint32_t extra, uint32_t payload) { return bits | (kind << 8) | (extra << 12) | (payload << 16); }
constexpr uint32_t BAD_ENTRY = entry(K_BAD, 0, 0, 0);

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t reverse_bits(uint32_t code, int len) {
  uint32_t r = 0;
  for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
  return r;
}

inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }
inline void store64(uint8_t* p, uint64_t v) { std::memcpy(p, &v, 8); }

// Canonical Huffman code -> lookup table with `primary` index bits and second-level tables for longer codes.
// sym_entry(sym, len) gives the entry of a symbol (with `bits` = len, adjusted here for second-level entries).
template <class SymEntry>
bool build_table(uint32_t* table, int table_cap, int primary, const uint8_t* lens, int n, SymEntry sym_entry) {
  int count[16] = {0};
  for (int i = 0; i < n; ++i) ++count[lens[i]];
  count[0] = 0;
  uint32_t next[16]; uint32_t code = 0; long left = 1;
  for (int l = 1; l <= 15; ++l) {
    left = (left << 1) - count[l];
    if (left < 0) return false;                     // over-subscribed
    code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code;
  }
  const int psize = 1 << primary;
  for (int i = 0; i < psize; ++i) table[i] = BAD_ENTRY;
  // widths of the second-level tables: for every primary prefix, the longest code behind it
  uint8_t sub_bits[1 << 11];
  std::memset(sub_bits, 0, (size_t)psize);
  {
    uint32_t nx[16]; std::memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
      const int l = lens[s];
      if (l <= primary) { if (l) ++nx[l]; continue; }
      const uint32_t rev = reverse_bits(nx[l]++, l), prefix = rev & (uint32_t)(psize - 1);
      if (l - primary > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - primary);
    }
  }
  int used = psize;
  for (int p = 0; p < psize; ++p) if (sub_bits[p]) {
      const int sz = 1 << sub_bits[p];
      if (used + sz > table_cap) return false;
      table[p] = entry(K_SUB, (uint32_t)primary, sub_bits[p], (uint32_t)used);
      for (int i = 0; i < sz; ++i) table[used + i] = BAD_ENTRY;
      used += sz;
    }
  for (int s = 0; s < n; ++s) {
    const int l = lens[s];
    if (!l) continue;
    const uint32_t rev = reverse_bits(next[l]++, l);
    if (l <= primary) {
      const uint32_t e = sym_entry(s, (uint32_t)l);
      for (uint32_t i = rev; i < (uint32_t)psize; i += 1u << l) table[i] = e;
    } else {
      const uint32_t prefix = rev & (uint32_t)(psize - 1), sub = table[prefix];
      const int sb = (int)((sub >> 12) & 15), base = (int)(sub >> 16), rem = l - primary;
      const uint32_t e = sym_entry(s, (uint32_t)rem);
      for (uint32_t i = rev >> primary; i < (1u << sb); i += 1u << rem) table[base + (int)i] = e;
    }
  }
  return true;
}

uint32_t lit_entry(int s, uint32_t bits) {
  if (s < 256) return entry(K_LIT, bits, 0, (uint32_t)s);
  if (s == 256) return entry(K_EOB, bits, 0, 0);
  if (s <= 285) return entry(K_LEN, bits, LEN_EXTRA[s - 257], LEN_BASE[s - 257]);
  return entry(K_BAD, bits, 0, 0);
}
uint32_t dist_entry(int s, uint32_t bits) {
  if (s < 30) return entry(K_LEN, bits, DIST_EXTRA[s], DIST_BASE[s]);
  return entry(K_BAD, bits, 0, 0);
}

}  // namespace

bool Inflater::build_tables(const uint8_t* lit_lens, int n_lit, const uint8_t* dist_lens, int n_dist) {
  return build_table(lit, LIT_TABLE, LIT_PRIMARY, lit_lens, n_lit, lit_entry) && build_table(dist, DIST_TABLE, DIST_PRIMARY, dist_lens, n_dist, dist_entry);
}

Inflater::Status Inflater::run(const uint8_t** in, const uint8_t* in_end, bool in_final, uint8_t** out, uint8_t* out_limit, const uint8_t* window_begin) {
  const uint8_t* ip = *in; uint8_t* op = *out;
  uint64_t bb = bitbuf; int bc = bitcnt;
  // on every exit whole bytes still in the bit buffer go back to the input position (the caller may move its buffer)
  auto leave = [&](Status s) { ip -= bc >> 3; bc &= 7; bb &= (1ull << bc) - 1; bitbuf = bb; bitcnt = bc; *in = ip; *out = op; return s; };
  auto fail = [&](const char* msg) { error = msg; return leave(ERROR); };
  // careful bit reader for headers: never reads past in_end
  auto need = [&](int n) -> bool { while (bc < n) { if (ip >= in_end) return false; bb |= (uint64_t)*ip++ << bc; bc += 8; } return true; };
  auto take = [&](int n) -> uint32_t { const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1)); bb >>= n; bc -= n; return v; };

  for (;;) {
    if (phase == 3) return leave(STREAM_END);
    if (phase == 0) {
      if (!in_final && (size_t)(in_end - ip) + (size_t)(bc >> 3) < HEADER_BYTES) return leave(NEED_INPUT);
      if (!need(3)) return fail("deflate stream truncated (block header)");
      last_block = take(1);
      const uint32_t type = take(2);
      if (type == 0) {
        take(bc & 7);                                   // to the byte boundary
        if (!need(32)) return fail("deflate stream truncated (stored block)");
        const uint32_t len = take(16), nlen = take(16);
        if ((len ^ nlen) != 0xFFFFu) return fail("deflate stored block length check failed");
        stored_left = len; phase = 1;
      } else if (type == 1) {
        uint8_t ll[288], dl[32];
        for (int i = 0; i < 144; ++i) ll[i] = 8;
        for (int i = 144; i < 256; ++i) ll[i] = 9;
        for (int i = 256; i < 280; ++i) ll[i] = 7;
        for (int i = 280; i < 288; ++i) ll[i] = 8;
        for (int i = 0; i < 32; ++i) dl[i] = 5;
        if (!build_tables(ll, 288, dl, 32)) return fail("deflate fixed tables");
        phase = 2;
      } else if (type == 2) {
        if (!need(14)) return fail("deflate stream truncated (dynamic header)");
        const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
        if (hlit > 286 || hdist > 30) return fail("deflate dynamic header: too many codes");
        uint8_t cl[19] = {0};
        for (int i = 0; i < hclen; ++i) { if (!need(3)) return fail("deflate stream truncated (dynamic header)"); cl[CL_ORDER[i]] = (uint8_t)take(3); }
        uint32_t clt[128 + 8];
        if (!build_table(clt, 128 + 8, 7, cl, 19, [](int s, uint32_t bits) { return entry(K_LIT, bits, 0, (uint32_t)s); })) return fail("deflate dynamic header: bad code-length code");
        uint8_t lens[286 + 30 + 138]; int n = 0;
        while (n < hlit + hdist) {
          if (!need(7) && bc == 0) return fail("deflate stream truncated (dynamic header)");
          const uint32_t e = clt[bb & 127];
          if (((e >> 8) & 15) != K_LIT || (int)(e & 0xFF) > bc) return fail("deflate dynamic header: bad code-length symbol");
          take((int)(e & 0xFF));
          const uint32_t sym = e >> 16;
          if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
          int rep; uint8_t val = 0;
          if (sym == 16) { if (n == 0) return fail("deflate dynamic header: repeat without a previous length"); if (!need(2)) return fail("deflate stream truncated (dynamic header)"); rep = 3 + (int)take(2); val = lens[n - 1]; }
          else if (sym == 17) { if (!need(3)) return fail("deflate stream truncated (dynamic header)"); rep = 3 + (int)take(3); }
          else { if (!need(7)) return fail("deflate stream truncated (dynamic header)"); rep = 11 + (int)take(7); }
          if (n + rep > hlit + hdist) return fail("deflate dynamic header: repeat runs past the code counts");
          for (int i = 0; i < rep; ++i) lens[n++] = val;
        }
        if (lens[256] == 0) return fail("deflate dynamic header: no end-of-block code");
        if (!build_tables(lens, hlit, lens + hlit, hdist)) return fail("deflate dynamic header: over-subscribed code");
        phase = 2;
      } else return fail("deflate block type 3");
    }
    if (phase == 1) {
      while (stored_left && bc >= 8) { if (op >= out_limit) return leave(NEED_OUTPUT); *op++ = (uint8_t)take(8); --stored_left; }
      if (bc == 0) bb = 0;   // (the 8-byte refills leave stream bits above bc in the buffer; the copy below jumps past them)
      while (stored_left) {
        if (op >= out_limit) return leave(NEED_OUTPUT);
        size_t m = stored_left;
        if ((size_t)(in_end - ip) < m) m = (size_t)(in_end - ip);
        if ((size_t)(out_limit - op) < m) m = (size_t)(out_limit - op);
        if (m == 0) { if (in_final) return fail("deflate stream truncated (stored data)"); return leave(NEED_INPUT); }
        std::memcpy(op, ip, m); op += m; ip += m; stored_left -= (uint32_t)m;
      }
      phase = last_block ? 3 : 0;
      continue;
    }
    // phase 2: Huffman-coded block
    const uint32_t lmask = (1u << LIT_PRIMARY) - 1, dmask = (1u << DIST_PRIMARY) - 1;
    for (;;) {
      if (op >= out_limit) return leave(NEED_OUTPUT);
      if (!in_final) { if (ip + IN_MARGIN > in_end) return leave(NEED_INPUT); }
      else if (ip > in_end + 8) return fail("deflate stream truncated");
      bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56;
      uint32_t e = lit[bb & lmask];
      if (((e >> 8) & 15) == K_SUB) { bb >>= LIT_PRIMARY; bc -= LIT_PRIMARY; e = lit[(e >> 16) + (uint32_t)(bb & ((1u << ((e >> 12) & 15)) - 1))]; }
      bb >>= (e & 0xFF); bc -= (int)(e & 0xFF);
      uint32_t kind = (e >> 8) & 15;
      if (kind == K_LIT) {
        *op++ = (uint8_t)(e >> 16);
        // up to two more literals from the same refill (a code is at most 15 bits; >= 41 bits are left)
        e = lit[bb & lmask];
        if (((e >> 8) & 15) == K_LIT) {
          bb >>= (e & 0xFF); bc -= (int)(e & 0xFF); *op++ = (uint8_t)(e >> 16);
          e = lit[bb & lmask];
          if (((e >> 8) & 15) == K_LIT) { bb >>= (e & 0xFF); bc -= (int)(e & 0xFF); *op++ = (uint8_t)(e >> 16); }
        }
        continue;
      }
      if (kind == K_LEN) {
        const uint32_t xl = (e >> 12) & 15;
        const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1)); bb >>= xl; bc -= (int)xl;
        uint32_t d = dist[bb & dmask];
        if (((d >> 8) & 15) == K_SUB) { bb >>= DIST_PRIMARY; bc -= DIST_PRIMARY; d = dist[(d >> 16) + (uint32_t)(bb & ((1u << ((d >> 12) & 15)) - 1))]; }
        if (((d >> 8) & 15) != K_LEN) return fail("deflate: invalid distance code");
        bb >>= (d & 0xFF); bc -= (int)(d & 0xFF);
        const uint32_t xd = (d >> 12) & 15;
        const uint32_t distance = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1)); bb >>= xd; bc -= (int)xd;
        if ((size_t)(op - window_begin) < distance) return fail("deflate: distance reaches before the start of the data");
        const uint8_t* src = op - distance; uint8_t* end = op + len;
        if (distance >= 8) { do { store64(op, load64(src)); op += 8; src += 8; } while (op < end); }
        else if (distance == 1) { std::memset(op, *src, len); }
        else { while (op < end) *op++ = *src++; }
        op = end;
        continue;
      }
      if (kind == K_EOB) break;
      return fail("deflate: invalid literal/length code");
    }
    if (in_final && ip - (bc >> 3) > in_end) return fail("deflate stream truncated");
    phase = last_block ? 3 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// CRC-32 (the gzip polynomial) by carry-less multiplication: zlib 1.2.11's table CRC runs at ~1 GB/s, which would cost the
// reader thread a third of its time. The data is folded 64 bytes at a time down to 16 bytes (fold constants x^(512+64),
// x^512, x^(128+64), x^128 mod P, bit-reflected, as tabulated in Intel's "Fast CRC Computation Using PCLMULQDQ"); the
// last 16 bytes and the tail go through zlib. Checked against zlib at first use; any disagreement disables it for good.
namespace {

constexpr size_t IBUF = (size_t)4 << 20;

#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t crc, const uint8_t* p, size_t n) {
  const __m128i k64 = _mm_set_epi64x(0x00000001c6e41596ll, 0x0000000154442bd4ll);
  const __m128i k16 = _mm_set_epi64x(0x00000000ccaa009ell, 0x00000001751997d0ll);
  auto fold = [](__m128i x, __m128i k) __attribute__((target("pclmul,sse4.1"))) { return _mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)); };
  __m128i x0 = _mm_loadu_si128((const __m128i*)p), x1 = _mm_loadu_si128((const __m128i*)(p + 16));
  __m128i x2 = _mm_loadu_si128((const __m128i*)(p + 32)), x3 = _mm_loadu_si128((const __m128i*)(p + 48));
  x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)~crc));
  p += 64; n -= 64;
  while (n >= 64) {
    x0 = _mm_xor_si128(fold(x0, k64), _mm_loadu_si128((const __m128i*)p));
    x1 = _mm_xor_si128(fold(x1, k64), _mm_loadu_si128((const __m128i*)(p + 16)));
    x2 = _mm_xor_si128(fold(x2, k64), _mm_loadu_si128((const __m128i*)(p + 32)));
    x3 = _mm_xor_si128(fold(x3, k64), _mm_loadu_si128((const __m128i*)(p + 48)));
    p += 64; n -= 64;
  }
  __m128i x = _mm_xor_si128(fold(x0, k16), x1);
  x = _mm_xor_si128(fold(x, k16), x2);
  x = _mm_xor_si128(fold(x, k16), x3);
  while (n >= 16) { x = _mm_xor_si128(fold(x, k16), _mm_loadu_si128((const __m128i*)p)); p += 16; n -= 16; }
  uint8_t t[16]; _mm_storeu_si128((__m128i*)t, x);
  const uint32_t c = (uint32_t)crc32(0xFFFFFFFFul, t, 16);   // register 0 in, 16 folded bytes through: zlib's pre/post inversions cancel
  return (uint32_t)crc32(c, p, (uInt)n);
}

bool clmul_usable() {
  static const bool ok = [] {
    if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
    uint8_t buf[1500]; uint32_t s = 12345;
    for (auto& b : buf) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    for (size_t n : {(size_t)80, (size_t)81, (size_t)127, (size_t)128, (size_t)777, (size_t)1500})
      for (uint32_t c0 : {0u, 0xDEADBEEFu})
        if (crc32_clmul(c0, buf, n) != (uint32_t)crc32(c0, buf, (uInt)n)) return false;
    return true;
  }();
  return ok;
}
#else
bool clmul_usable() { return false; }
uint32_t crc32_clmul(uint32_t crc, const uint8_t* p, size_t n) { return (uint32_t)crc32(crc, p, (uInt)n); }
#endif

}  // namespace

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  while (n) {
    const size_t m = n < ((size_t)1 << 30) ? n : ((size_t)1 << 30);
    crc = (m >= 80 && clmul_usable()) ? crc32_clmul(crc, p, m) : (uint32_t)crc32(crc, p, (uInt)m);
    p += m; n -= m;
  }
  return crc;
}

GzipStream::GzipStream() { ibuf = (uint8_t*)std::malloc(IBUF + 64); hist = (uint8_t*)std::malloc(HISTORY); }
GzipStream::~GzipStream() { std::free(ibuf); std::free(hist); }

// moves the unread tail to the front and reads more; false once the file has ended (the window is then zero padded)
bool GzipStream::more_input() {
  if (file_eof) return false;
  if (ipos > 0) { std::memmove(ibuf, ibuf + ipos, iend - ipos); iend -= ipos; ipos = 0; }
  bool got_any = false;
  while (iend < IBUF) {
    const long got = (long)::read(fd, ibuf + iend, IBUF - iend);
    if (got < 0) { error = "read error"; file_eof = true; break; }
    if (got == 0) { file_eof = true; break; }
    iend += (size_t)got; got_any = true;
  }
  if (file_eof) std::memset(ibuf + iend, 0, 64);
  return got_any;
}

long GzipStream::read(uint8_t* payload, size_t cap) {
  if (!ibuf || !hist) { error = "out of memory"; return -1; }
  uint8_t* op = payload; uint8_t* const out_limit = payload + cap;
  // the history of a member that continues from the previous call sits directly in front of the payload
  std::memcpy(payload - hist_len, hist, hist_len);
  const uint8_t* window_begin = payload - hist_len;
  while (op < out_limit && stage != 3) {
    if (stage == 0) {                                   // member header (RFC 1952 §2.3)
      // a header is parsed in one go: make sure it is all there (FEXTRA <= 64 KiB, names are short in practice)
      if (!file_eof && iend - ipos < (size_t)1 << 17) more_input();
      if (error) return -1;
      const uint8_t* p = ibuf + ipos; const size_t n = iend - ipos;
      if (n == 0) { stage = 3; break; }
      bool zeros = true;
      for (size_t i = 0; i < n && zeros; ++i) zeros = p[i] == 0;
      if (zeros && file_eof) { stage = 3; break; }     // zero padding after the last member (tolerated, as gzip does)
      if (n < 18 || p[0] != 0x1f || p[1] != 0x8b) { error = "not a gzip member (bad magic)"; return -1; }
      if (p[2] != 8 || (p[3] & 0xE0)) { error = "unsupported gzip method or flags"; return -1; }
      const uint8_t flg = p[3]; size_t o = 10;
      if (flg & 4) { if (o + 2 > n) { error = "gzip header truncated"; return -1; } o += 2 + ((size_t)p[o] | ((size_t)p[o + 1] << 8)); }
      for (int name = 0; name < 2; ++name) if (flg & (name ? 16 : 8)) { while (o < n && p[o]) ++o; ++o; }
      if (flg & 2) o += 2;
      if (o > n) { error = "gzip header truncated"; return -1; }
      ipos += o;
      inf.reset(); crc = (uint32_t)crc32(0L, Z_NULL, 0); isize = 0; hist_len = 0; window_begin = op; stage = 1;
    }
    if (stage == 1) {
      const uint8_t* ip = ibuf + ipos; uint8_t* const before = op;
      const Inflater::Status st = inf.run(&ip, ibuf + iend, file_eof, &op, out_limit, window_begin);
      ipos = (size_t)(ip - ibuf);
      if (op > before) { crc = crc32_update(crc, before, (size_t)(op - before)); isize += (uint64_t)(op - before); }
      if (st == Inflater::ERROR) { error = inf.error; return -1; }
      if (st == Inflater::NEED_INPUT) { more_input(); if (error) return -1; continue; }   // (at end of file the next call runs with in_final)
      if (st == Inflater::STREAM_END) stage = 2;
    }
    if (stage == 2) {                                   // CRC-32 and ISIZE
      if (iend - ipos < 8) more_input();
      if (error) return -1;
      if (iend - ipos < 8) { error = "gzip member truncated (no trailer)"; return -1; }
      const uint8_t* t = ibuf + ipos;
      const uint32_t want_crc = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      const uint32_t want_size = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
      if (want_crc != crc) { error = "gzip member fails its CRC-32"; return -1; }
      if (want_size != (uint32_t)isize) { error = "gzip member fails its length check"; return -1; }
      ipos += 8; stage = 0; hist_len = 0; window_begin = op;
    }
  }
  // keep the history for a member that continues in the next call
  if (stage == 1) {
    const size_t have = (size_t)(op - window_begin), keep = have < HISTORY ? have : HISTORY;
    std::memmove(hist, op - keep, keep); hist_len = keep;
  } else hist_len = 0;
  return (long)(op - payload);
}

// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr size_t REFILL_COMPRESSED = (size_t)8 << 20;
inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
}  // namespace

long BgzfReader::block_size(const uint8_t* p, size_t n) {
  if (n < 18) return 0;
  if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
  const size_t xlen = rd16(p + 10);
  if (n < 12 + xlen) return 0;
  for (size_t o = 12; o + 4 <= 12 + xlen;) {
    const size_t slen = rd16(p + o + 2);
    if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2) return (long)rd16(p + o + 4) + 1;
    o += 4 + slen;
  }
  return -1;
}

long BgzfReader::read(uint8_t* dst, size_t cap) {
  if (cap < 65536) { error = "BGZF: output window smaller than a block"; return -1; }
  struct Blk { size_t coff, csize, uoff, usize; };
  std::vector<Blk> blks;
  for (;;) {
    if (!file_eof && cend - cpos < REFILL_COMPRESSED) {   // top up the compressed window
      if (cpos > 0) { std::memmove(cbuf.data(), cbuf.data() + cpos, cend - cpos); cend -= cpos; cpos = 0; }
      if (cbuf.size() < cend + REFILL_COMPRESSED + 64) cbuf.resize(cend + REFILL_COMPRESSED + 64);   // (+64: the decoder may look 16 bytes past a block)
      const size_t got = std::fread(cbuf.data() + cend, 1, REFILL_COMPRESSED, fp);
      if (got < REFILL_COMPRESSED) { if (std::ferror(fp)) { error = "BGZF: read error"; return -1; } file_eof = true; }
      cend += got;
    }
    size_t o = cpos, utotal = 0;
    blks.clear();
    while (o < cend) {
      const long bs = block_size(cbuf.data() + o, cend - o);
      if (bs < 0) { error = "BGZF: corrupt block header"; return -1; }
      if (bs == 0 || o + (size_t)bs > cend) break;
      if (bs < 26) { error = "BGZF: corrupt block size"; return -1; }
      const size_t usize = rd32(cbuf.data() + o + (size_t)bs - 4);
      if (usize > 65536) { error = "BGZF: corrupt block (ISIZE > 64 KiB)"; return -1; }
      if (utotal + usize > cap) break;
      blks.push_back(Blk{o, (size_t)bs, utotal, usize});
      utotal += usize; o += (size_t)bs;
    }
    if (blks.empty()) {
      if (o == cend && file_eof) return 0;                          // clean end of file
      if (file_eof) { error = "BGZF: truncated block"; return -1; }
      continue;                                                     // not even one whole block yet: read more
    }
    const int T = (int)std::min<size_t>((size_t)(threads < 1 ? 1 : threads), std::max<size_t>(1, blks.size() / 8));
    std::vector<int> bad((size_t)T, 0);
    const uint8_t* cb = cbuf.data();
    auto work = [&](int t) {
      // each block is decoded into scratch and copied to its slice: the decoder's match copies overshoot by a few bytes,
      // which must not touch the neighbouring slice another worker is filling
      std::vector<uint8_t> scratch(65536 + Inflater::OUT_MARGIN + 64);
      Inflater inf;
      for (size_t i = blks.size() * (size_t)t / (size_t)T, e = blks.size() * (size_t)(t + 1) / (size_t)T; i < e; ++i) {
        const Blk& k = blks[i];
        if (k.usize == 0) continue;   // (the empty end-of-file marker block)
        const size_t xlen = rd16(cb + k.coff + 10), hdr = 12 + xlen;
        if (k.csize < hdr + 8) { bad[(size_t)t] = 1; break; }
        inf.reset();
        const uint8_t* ip = cb + k.coff + hdr; uint8_t* op = scratch.data();
        const Inflater::Status st = inf.run(&ip, cb + k.coff + k.csize - 8, true, &op, scratch.data() + k.usize + 1, scratch.data());
        if (st != Inflater::STREAM_END || (size_t)(op - scratch.data()) != k.usize) { bad[(size_t)t] = 1; break; }
        if (crc32_update((uint32_t)crc32(0L, Z_NULL, 0), scratch.data(), k.usize) != rd32(cb + k.coff + k.csize - 8)) { bad[(size_t)t] = 1; break; }
        std::memcpy(dst + k.uoff, scratch.data(), k.usize);
      }
    };
    if (T <= 1) work(0);
    else { std::vector<std::thread> th; for (int t = 1; t < T; ++t) th.emplace_back(work, t); work(0); for (auto& x : th) x.join(); }
    for (int x : bad) if (x) { error = "BGZF: corrupt block (inflate / CRC failure)"; return -1; }
    cpos = o;
    if (utotal == 0) continue;   // only empty (end-of-file marker) blocks
    return (long)utotal;
  }
}

}  // namespace nb
