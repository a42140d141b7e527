// nb_inflate.h — streaming DEFLATE (RFC 1951) decoder for the gzip / BGZF inputs of `nimble align` (host, product code).
//
// zlib's inflate runs at 0.15-0.3 GB/s, which makes a gzip FASTQ the slowest stage of the whole job by an order of
// magnitude (end to end). This decoder follows the same specification with the usual fast-path structure:
// a 64-bit bit buffer refilled eight bytes at a time, one table lookup per symbol (11-bit primary table for
// literal/length codes, 8-bit for distances, second-level tables for longer codes), 8-byte match copies. It is
// resumable between symbols, so one long gzip member can be decoded piece by piece with a 32 KiB history in front of
// the output cursor. The gzip container (RFC 1952: header fields, CRC-32, ISIZE, concatenated members) is handled by the
// caller (nb_ingest.cpp), which also verifies the CRC of everything decoded here.
#ifndef NB_INFLATE_H
#define NB_INFLATE_H
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <vector>

namespace nb {

struct Inflater {
  enum Status { NEED_INPUT = 0, NEED_OUTPUT = 1, STREAM_END = 2, ERROR = -1 };
  static constexpr int LIT_PRIMARY = 11, DIST_PRIMARY = 8;
  static constexpr int LIT_TABLE = 2048 + 2048, DIST_TABLE = 256 + 1024;
  // input / output margins run() relies on (see run())
  static constexpr size_t IN_MARGIN = 16, OUT_MARGIN = 320, HEADER_BYTES = 1024;

  uint32_t lit[LIT_TABLE];
  uint32_t dist[DIST_TABLE];
  uint64_t bitbuf = 0;
  int bitcnt = 0;
  int phase = 0;              // 0 = expecting a block header, 1 = stored block body, 2 = Huffman block body, 3 = done
  bool last_block = false;
  uint32_t stored_left = 0;
  const char* error = nullptr;

  void reset() { bitbuf = 0; bitcnt = 0; phase = 0; last_block = false; stored_left = 0; error = nullptr; }

  // Decodes from [*in, in_end) to [*out, out_limit). History: bytes [window_begin, *out) must be the previously decoded
  // output (at least the last 32 KiB of it, or all of it).
  //   * `in_final` = no input exists beyond in_end. Unless in_final, run() returns NEED_INPUT when fewer than
  //     HEADER_BYTES remain in front of a block header or IN_MARGIN inside a block body; the caller moves the unread tail
  //     to the front of its buffer, appends more input and calls again. With in_final the caller must have IN_MARGIN
  //     readable zero bytes after in_end (over-reads are detected, not performed blindly: consumption past in_end is an
  //     error).
  //   * out_limit must leave OUT_MARGIN writable bytes behind it (match copies overshoot); NEED_OUTPUT is returned once
  //     *out >= out_limit.
  //   * STREAM_END: the final block ended; *in points behind the last byte the DEFLATE stream used (the bit buffer is
  //     drained back into the input position).
  Status run(const uint8_t** in, const uint8_t* in_end, bool in_final, uint8_t** out, uint8_t* out_limit, const uint8_t* window_begin);

 private:
  bool build_tables(const uint8_t* lit_lens, int n_lit, const uint8_t* dist_lens, int n_dist);
  bool read_block_header(const uint8_t** in, const uint8_t* in_end);
};

// CRC-32 of gzip / BGZF (carry-less multiplication where the CPU has it, checked against zlib at first use; else zlib)
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n);

// gzip container (RFC 1952) over a file descriptor: header fields, any number of concatenated members, CRC-32 and ISIZE of
// every member verified. read() fills `payload` with up to `cap` decoded bytes (a few more may follow: see below) and
// returns the count, 0 at the end of the file, -1 on error (`error` says what).
struct GzipStream {
  static constexpr size_t HISTORY = 32768;             // writable space the caller keeps IN FRONT of payload
  static constexpr size_t SLACK = Inflater::OUT_MARGIN; // writable space the caller keeps BEHIND payload + cap
  int fd = -1;
  const char* error = nullptr;
  long read(uint8_t* payload, size_t cap);
  GzipStream();
  ~GzipStream();
  GzipStream(const GzipStream&) = delete;
  GzipStream& operator=(const GzipStream&) = delete;

 private:
  Inflater inf;
  uint8_t* ibuf;                 // compressed window [ipos, iend), zero padded behind iend at end of file
  size_t ipos = 0, iend = 0;
  bool file_eof = false;
  int stage = 0;                 // 0 member header, 1 deflate body, 2 trailer, 3 end of file
  uint32_t crc = 0;
  uint64_t isize = 0;
  uint8_t* hist;                 // last <= 32 KiB of the current member's output
  size_t hist_len = 0;
  bool more_input();
};

// BGZF (SAM spec §4.1): gzip members of <= 64 KiB that carry their compressed size in a 'BC' extra field, so a run of whole
// blocks can be cut out of the file and inflated independently on worker threads, each into its own slice of the output
// (sizes come from the ISIZE trailers), CRC of every block checked. BAM files and the .tsv.gz tables this engine writes
// (nb_gzip_members) have this form.
struct BgzfReader {
  FILE* fp = nullptr;            // owned by the caller
  int threads = 1;
  const char* error = nullptr;
  // size of the BGZF block starting at p (n bytes available): 0 if the header is incomplete, -1 if it is not a BGZF block
  static long block_size(const uint8_t* p, size_t n);
  // decodes whole blocks into dst (cap >= 65536): returns the bytes produced, 0 at the end of the file, -1 on error
  long read(uint8_t* dst, size_t cap);

 private:
  std::vector<uint8_t> cbuf;     // compressed bytes not yet inflated: [cpos, cend)
  size_t cpos = 0, cend = 0;
  bool file_eof = false;
};

}  // namespace nb
#endif
