// nb_internal.h — layouts shared by the host index builder and the HIP kernels (not part of the C-ABI).
#pragma once
#include <cstdint>
#include <string>
#include <algorithm>
#include <thread>
#include <vector>

#include <memory>
#include <utility>

#include "nimble_amd.h"

// a vector whose resize() leaves new elements uninitialised: for tables of GB size whose every word is written by worker threads right
// after (value-initialising them first means one thread touching every page: 4.9 of the 12 s of a 10 000-sequence index build)
template <class T> struct NoInitAlloc : std::allocator<T> {
  template <class U> struct rebind { using other = NoInitAlloc<U>; };
  template <class U, class... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U; else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
  }
};
using WordsNoInit = std::vector<uint64_t, NoInitAlloc<uint64_t>>;

namespace nb {

// ---- k-mer encoding (product side): base j of the k-mer sits at bits [2j, 2j+2): FIRST base LEAST significant,
// so that a k-mer is a plain bit-field of the packed read (nimble_amd.h, nb_pack_reads). ----

// The canonical-k-mer table is an array of 32-byte BUCKETS of two 16-byte slots {key, pay[2]}; a lookup reads its home
// bucket with two dwordx4 loads issued together (one L2 request: both halves share a 64-byte line) and finds key AND
// payload there — no probe loop and no dependent second load. The builder keeps the load at <= 0.10 (0.2 keys per
// bucket on average), so a key overflows its home bucket with probability ~1e-3; when one does it goes to the next
// bucket and every bucket on the way gets the OVERFLOW flag, which sends the (rare) lookup one bucket further.
// Why not denser: one read per wave means 131 lookups in lock-step, so any per-lookup retry probability above ~1e-3
// makes nearly every wave pay a second (and third) iteration — per-slot linear probing at load 0.3 cost ~340 VALU
// instructions per read in its probe loop. Why not two-choice cuckoo: it doubles the L2-miss requests and hits the
// fabric request rate (measured 36 ms vs 22.5 ms per 10M reads). Why not 64-byte/4-key buckets with split key/payload
// halves: the payload becomes a dependent second load and the 4 key words cost registers (107 VGPRs, 24.8 ms).
//   key word  bits [0,62) canonical k-mer = min(x, rc(x)); bit 62 set = empty slot; bit 63 (slot 0 only) = overflow flag
//   pay[0]    payload of the k-mer equal to the key as it occurs in the reference (stranded), or PAY_ABSENT
//   pay[1]    payload of rc(key) as it occurs in the reference, or PAY_ABSENT (palindromes: both set)
// payload bits: [0,4) rext mask  [4,8) lext mask  8 r_internal  9 l_internal  [10,32) colour id
struct Slot {
  uint64_t key;
  uint32_t pay[2];
};
struct Bucket { Slot slot[2]; };
static_assert(sizeof(Bucket) == 32, "bucket must be 32 bytes");
constexpr int SLOTS_PER_BUCKET = 2;
constexpr double TABLE_MAX_LOAD = 0.10;

constexpr uint64_t KEY_EMPTY = 1ull << 62;
constexpr uint64_t KEY_OVERFLOW = 1ull << 63;
constexpr uint64_t KEY_MASK = ~KEY_OVERFLOW;
constexpr uint32_t PAY_ABSENT = 0xFFFFFFFFu;
constexpr uint32_t MAX_COLOURS = (1u << 22) - 1;  // colour ids are 22 bits; all-ones reserved by PAY_ABSENT
constexpr uint64_t HASH_MULT = 0x9E3779B97F4A7C15ull;   // class-table hash (content hash -> slot)

// bucket of a canonical k-mer: fold to 32 bits, one multiplicative mix, multiply-high range reduction
// (3 x v_mul_lo_u32 + 1 x v_mul_hi_u32 on the GPU); n_buckets need not be a power of two.
#if defined(__HIPCC__)   // (nb_index.cpp may be built by a plain host compiler in the CPU sanitizer tests)
#define NB_HD __host__ __device__
#else
#define NB_HD
#endif
struct KmerHash { uint32_t mf, m1, n_buckets; };
NB_HD inline uint32_t kmer_bucket(uint64_t canon, const KmerHash& h) {
  // two multiplicative hashes (the high bits of a product by an odd constant depend on every bit of the factor), xor-ed, then
  // a multiply-high range reduction that reads exactly those high bits: 2 mul_lo + 1 mul_hi + 1 xor per k-mer on the device
  const uint32_t m = ((uint32_t)canon * h.m1) ^ ((uint32_t)(canon >> 32) * h.mf);
  return (uint32_t)(((uint64_t)m * (uint64_t)h.n_buckets) >> 32);
}

// ---- COMPACT table (k <= 23): the form the kernels use whenever the k-mer is short enough ----------------------------------
// 16-byte buckets of TWO 8-byte entries {key word, payload}, (2,2) partial-key cuckoo hashing at load ~0.85: 9.4 bytes per stranded
// k-mer instead of 160 (the 1 000-sequence library of BASELINE configs[1]: 2.9 MB instead of 49 MB, i.e. resident in every XCD's
// 4 MiB L2; 50 sequences: 0.16 MB, which the kernels stage in LDS). A lookup reads BOTH candidate buckets with one dwordx4 each,
// issued together: no probe loop, no overflow chain, no dependent load. Exactness without storing the k-mer:
//   (hi, lo) = canonical k-mer split at bit 32           hb = max(2k - 32, 0) bits of hi
//   m   = lo * a  ^  hi * c                               a odd: a bijection of lo for every hi
//   hi2 = hi ^ ((m >> 7) & mask(hb))                      a bijection of hi for every m
//   rem = m & (2^30 - 1)                                  stored in the entry
//   g   = top 32 bits of the (hb + 32)-bit value hi2:m ;  b1 = mulhi(g, nb)        home bucket
//   b2  = (b1 + 1 + mulhi(rem bit-reversed and left-aligned, nb - 1)) mod nb         alternate bucket, a function of (b1, rem) only
// Two distinct k-mers with equal rem differ in the top hb + 2 bits of g, hence by >= 2^(30 - hb) in g, hence in b1 as soon as
// nb >= 2^(hb + 2) (enforced): (b1, rem) identifies the k-mer. An entry records WHICH of its two buckets it sits in, and b1 =
// b2 - d(rem) recovers the home bucket from the alternate one, so (bucket, rem, which) identifies it as well: a lookup that
// compares rem and `which` in the two buckets can never match another k-mer — absent k-mers included.
//   key word  bits [2,32) rem   bit 1 which (0 = home bucket, 1 = alternate)   bit 0 strand (0: the payload describes the k-mer
//             equal to the canonical key as it occurs in the reference, 1: its reverse complement as it occurs)
//   payload   as below (make_payload); an empty entry is {0xFFFFFFFF, PAY_ABSENT}; the builder re-seeds (a, c) if a real key
//             word would equal 0xFFFFFFFF
struct CEntry { uint32_t key, pay; };
struct CBucket { CEntry e[2]; };
static_assert(sizeof(CBucket) == 16, "compact bucket must be 16 bytes");
constexpr int COMPACT_MAX_K = 23;                       // hb <= 14: nb >= 2^16 buckets (1 MiB) at the most
constexpr double COMPACT_LOAD = 0.85;                   // (2,2) cuckoo threshold is 0.897
constexpr int64_t COMPACT_MAX_BYTES = 8ll << 20;        // larger tables are not L2-resident: the one-line-per-lookup wide table wins there (nb_index.cpp)
constexpr uint32_t CKEY_EMPTY = 0xFFFFFFFFu;
struct CHash { uint32_t a, c, nb, nbm1, hb, himask; };
struct CProbe { uint32_t want, b1, b2; };               // want = rem << 2 (which = 0, strand = 0)
NB_HD inline uint32_t nb_mulhi32(uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x * (uint64_t)y) >> 32); }   // v_mul_hi_u32 on the device
NB_HD inline uint32_t nb_brev32(uint32_t x) {
#if defined(__clang__)
  return __builtin_bitreverse32(x);                     // v_bfrev_b32 on the device
#else
  x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1); x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
  x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4); x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
  return (x >> 16) | (x << 16);
#endif
}
// in [1, nb - 1]; want = rem << 2. The fold brings every bit of rem into the 17 low bits that the reversal moves to the top: the reversal
// alone looked at the LOW bits of rem only, which k-mers that differ in one high base (allelic variants: the multiply of compact_probe
// carries upward only) share — and with them home AND alternate bucket (a library of 150 sequences at k = 23 could not be placed).
NB_HD inline uint32_t compact_alt_delta(uint32_t want, const CHash& h) { return 1u + nb_mulhi32(nb_brev32(want ^ (want >> 15)), h.nbm1); }
NB_HD inline CProbe compact_probe(uint64_t canon, const CHash& h) {
  const uint32_t lo = (uint32_t)canon, hi = (uint32_t)(canon >> 32);
  const uint32_t m = (lo * h.a) ^ (hi * h.c);
  const uint32_t hi2 = hi ^ ((m >> 7) & h.himask);
  const uint32_t g = h.hb ? ((hi2 << (32u - h.hb)) | (m >> h.hb)) : m;
  CProbe p; p.want = m << 2; p.b1 = nb_mulhi32(g, h.nb);
  const uint32_t b2 = p.b1 + compact_alt_delta(p.want, h);
  p.b2 = b2 >= h.nb ? b2 - h.nb : b2;
  return p;
}

// ---- PERFECT-HASH table of the lane-per-read kernels (nb_lane.hip): ONE 16-byte entry per lookup, first level in LDS -------------------
// Measured on MI355X (profiles/r02_gather_microbench.jsonl, r02_tuning_notes.md): the chip serves ~266 G random requests/s from an
// L2-resident table whatever the request width (8 / 16 B alike) — the lane-per-read kernel with the cuckoo table (2 buckets + 1 colour
// descriptor per position = 353 requests per 150-bp read) sits at 94 % of that. Fewer requests per position is the only lever: this table
// answers a lookup with a single dwordx4 that carries the payload AND the class word, i.e. 131 requests per read.
//   (t, rem) = the same bijective split of the canonical k-mer as the compact table (rem = 30 bits, t = the other hb + 2 bits);
//   group = mulhi(rem << 2, n_groups): a function of rem alone; pilot[group] (one byte, the table's first level: staged in LDS by every
//   workgroup); slot = (mulhi((rem ^ pilot * PM) * GM2, n_slots) + t * stride) mod n_slots, stride = n_slots >> (hb + 2). "Hash and
//   displace": the builder picks each group's pilot so that all its k-mers land on free slots. Exactness without stored k-mers: an
//   entry with the queried rem belongs to the queried group and was hashed with the same pilot; two k-mers with equal rem differ in
//   t, i.e. by a non-zero multiple of stride below n_slots: same (slot, rem) = same k-mer — absent k-mers can never match.
//   entry {key, pay, m}: key = rem << 2 | both << 1 | strand (strand as in the compact table; both = the k-mer occurs on both strands:
//   rare, the read is handed to the wave-per-read kernel); pay = rext:4 lext:4 r_int l_int | wide << 10 | (wide ? colour : w0) << 11;
//   m = the class word (internal numbering) when !wide. Empty slot: key = 0xFFFFFFFF.
struct PHEntry { uint32_t key, pay; uint64_t m; };
static_assert(sizeof(PHEntry) == 16, "perfect-hash entry must be 16 bytes");
struct PHash { uint32_t a, c, n_slots, n_groups, hb, himask, stride; };
constexpr uint32_t PH_PM = 0x9E3779u, PH_GM2 = 0xC2B2AE35u;   // PH_PM < 2^24: pilot * PH_PM is a full-rate 24-bit multiply
constexpr int64_t PH_MAX_PILOT_BYTES = 84 << 10;        // first level must fit LDS next to the record staging of 16 waves of 150-bp reads (nb_anchor.hip: 4 waves per SIMD)
constexpr int64_t PH_MAX_TABLE_BYTES = 16ll << 20;      // beyond that the second level leaves the L2s: the wave-per-read kernels take over
struct PHProbe { uint32_t want, toff, group; };         // toff = t * stride
NB_HD inline PHProbe ph_probe(uint64_t canon, const PHash& h) {
  const uint32_t lo = (uint32_t)canon, hi = (uint32_t)(canon >> 32);
  const uint32_t m = (lo * h.a) ^ ((hi & 0xFFFFu) * (h.c & 0xFFFFFFu));          // (24-bit operands: v_mul_u32_u24)
  const uint32_t hi2 = hi ^ ((m >> 7) & h.himask);
  PHProbe p; p.want = m << 2;
  p.toff = (((hi2 << 2) | (m >> 30)) & 0xFFFFu) * (h.stride & 0xFFFFFFu);          // t < 2^(hb + 2) <= 2^16, stride < 2^24
  p.group = nb_mulhi32(p.want, h.n_groups);                                         // the top bits of rem: mixed by the multiply above
  return p;
}
NB_HD inline uint32_t ph_slot(const PHProbe& p, uint32_t pilot, const PHash& h) {
  const uint32_t s = nb_mulhi32(((p.want >> 2) ^ ((pilot & 0xFFu) * PH_PM)) * PH_GM2, h.n_slots) + p.toff;     // both terms < n_slots
  return s >= h.n_slots ? s - h.n_slots : s;
}

// ---- ANCHOR structures of the anchor-extend kernel (nb_anchor.hip): walk a read along the library TEXT instead of looking every k-mer up ----
// Once a k-mer of the read is known to sit at text position t (a table hit: its entry names ONE occurrence, chosen by the index builder),
// the next k-mers of the walk of docs/SPEC.md §3 need no lookup while the read's next bases equal the text's: the (K+1)-mer read[i..i+K]
// then occurs in the library by construction, so the base is a right extension, the next k-mer is the text's k-mer at t + 1, and its class
// is a property of the text position. 32 bases are compared per step (XOR of two 64-bit windows), and the AND of the classes of a run of
// text positions [t, t + n) comes from a sparse table with two loads whatever n is (AND is idempotent: two overlapping power-of-two
// blocks). Only where the read leaves its anchor sequence (a sequencing error, an allele difference, an N, the end of the sequence) does
// the kernel look a k-mer up again — ~20-25 lookups per 150-bp read instead of 131.
//   text      the library sequences concatenated, 2 bits per base, 16 bases per u32 (base j at bits [2(j & 15), +2) of word j >> 4);
//             N is stored as A and never compared: `room` stops a run before it
//   AEntry    same slot, key and pay as the PHEntry of the k-mer (same PHash, same pilots); tpos = text position of the anchor
//             occurrence; room (low 16 bits) = how many times the walk may step to the NEXT text k-mer from there: the following k-mers
//             are valid (no N, same sequence), do not occur on both strands, and the run stays below 2^ANCHOR_LEVELS positions; bits [8, 15)
//             = the same towards the PREVIOUS text k-mers (left extension from the first seed, at most 127); bit 15 =
//             the component of the anchor sequence spans more class-row words than a sparse-table entry holds (the read is then mapped
//             by the wave-per-read kernel); the high 16 bits = first class-row word of that component
//   sparse    level l, position t: AND of the classes of the k-mers at text positions [t, t + 2^l), as NW = anchor_nw words (1 or 2)
//             counted from the first word of the sequence's component (internal numbering: every class of a k-mer of a sequence lies
//             inside that component's words, so the AND is plain word arithmetic); level-major, then position, then word
struct AEntry { uint32_t key, pay, tpos, room; };
static_assert(sizeof(AEntry) == 16, "anchor entry must be 16 bytes");
constexpr int ANCHOR_LEVELS = 8;                        // runs of up to 255 k-mer positions
constexpr uint32_t ANCHOR_MAX_ROOM = (1u << ANCHOR_LEVELS) - 2;
constexpr uint32_t ANCHOR_ROOM_MASK = 0xFFu;            // AEntry.room: bits [0,8) room, [8,15) left room, bit 15 ANCHOR_WIDE, bits [16,32) first class-row word of the component
constexpr uint32_t ANCHOR_MAX_LROOM = 127u;             // left room: how many times the LEFT extension of SPEC 3 may step to the previous text k-mer (same rules as room)
constexpr uint32_t ANCHOR_WIDE = 1u << 15;              // the anchor sequence's component is wider than the sparse table's entries
constexpr int64_t ANCHOR_MAX_TEXT = 1ll << 31;
constexpr int64_t ANCHOR_LOOKAHEAD = 64;                // text k-mers ahead that decide which occurrence of a k-mer becomes its anchor (nb_index.cpp)
constexpr int64_t ANCHOR_MAX_SPARSE_BYTES = 8ll << 30;  // 8 levels x 8 (16) bytes per library base: libraries up to ~130 (65) M bases
constexpr int64_t PH_LARGE_MAX_TABLE_BYTES = 4ll << 30; // perfect-hash table with its first level in global memory

// ---- PRESENCE FILTER of the scan kernel (nb_anchor.hip): "can any k-mer that CONTAINS this m-mer be in the library?" ---------------------
// The scan kernel must show every seed-scan position of both orientations of a read absent from the table (88 positions of a 150-bp
// read at k = 20); an exact test per position costs the reverse complement, the canonical choice, the two-level hash and a load, and ran
// at the rate at which the L2s accept requests. Two facts make most of that unnecessary:
//  * a k-mer of the table contains only m-mers of the library (m = pf_m <= k), so ONE absent m-mer rules out every k-mer that contains
//    it: the m-mer at base q clears the k-mer positions [q - (k - m), q]. With m = k + 1 - S, S a power of two, the tests sit on a grid
//    of S bases (S = 8 for k = 20: m = 13, seventeen tests for the 131 positions of a 150-bp read);
//  * the filter holds ORIENTED m-mers — those of every k-mer of the table as it is and reverse-complemented — so a query needs neither
//    the reverse complement nor the canonical form.
// Two stages: a one-bit-per-slot table of 2^20 bits that the scan kernel copies into LDS (lf_index: a fold of the m-mer's bits, no
// multiply; built only while it stays at most half full), and a blocked Bloom filter in global memory (two bits per m-mer inside one
// 32-bit word: < 1 % of absent m-mers pass at the sizes pf_words_log2 picks) that is asked only for what the LDS stage lets through.
// Only the k-mers of tiles that pass both take the exact lookup. No false negatives by construction, so results cannot change.
NB_HD inline uint32_t pf_hash(uint64_t x) {
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  return (lo ^ ((hi << 15) | (hi >> 17))) * 0x9E3779B1u;
}
NB_HD inline uint32_t pf_word(uint32_t h, uint32_t wbits) { return h >> (32u - wbits); }
NB_HD inline uint32_t pf_mask(uint32_t h) { const uint32_t g = h ^ (h >> 15); return (1u << (g & 31u)) | (1u << ((g >> 5) & 31u)); }
constexpr uint32_t LF_BITS_LOG2 = 20;
NB_HD inline uint32_t lf_index(uint64_t x) { return ((uint32_t)x ^ (uint32_t)(x >> 20) ^ (uint32_t)(x >> 40)) & ((1u << LF_BITS_LOG2) - 1u); }
constexpr double LF_MAX_FILL = 0.5;                   // beyond that the LDS stage passes too much to be worth it (no lfilter is built)
inline uint32_t pf_words_log2(int64_t n_canonical) {      // ~20+ bits per oriented m-mer, 4 KB .. 16 MB
  uint32_t b = 10; while (b < 22 && ((int64_t)32 << b) < 40 * n_canonical) ++b; return b;
}
inline int pf_stride(int k) {                         // S: the largest power of two <= 8 that leaves m = k + 1 - S >= 12 (the anchor structures exist for k <= 23 only:
  int s = 8; while (s > 1 && k + 1 - s < 12) s >>= 1; return s;   //  m <= 16). Measured (profiles/r03_tuning_notes.md): at k = 20 one in a hundred 13-mers of a foreign
}                                                     //  read is a library 13-mer by chance and its tile takes the exact lookups, yet S = 4 (m = 17) is slower: twice the tiles

// A colour (equivalence class of the graph) for the wave-per-read kernels: the span of u64 words of its bitset that can be non-zero, with
// the first two words inline (32 bytes, two dwordx4 issued together). Bit positions are INTERNAL sequence numbers: the index builder
// renumbers the sequences so that every connected component of the "share a k-mer" relation (a locus, for a gene-family library) starts
// at a 64-bit word boundary or lies inside one word (nb_index.cpp, step 3a). A class is a subset of one component, so classes of
// components of <= 64 sequences occupy exactly ONE word and those of components of <= 128 sequences two: intersecting a read's classes
// needs nothing but the descriptors. Wider classes keep their span in `packed` at [off, off + nw). (The lane-per-read kernels do not
// load descriptors at all: their table entries carry the class word, PHEntry above.) Rows of the class table are in internal numbering
// too; the C-ABI converts them back (nb_classes_read), so callers only ever see original sequence indices.
// (Measured: a 16-byte form with only one word inline cost the 10 000-sequence library, whose 100-allele loci span two words, a
// dependent `packed` load per visited position: 314 vs 285 ms per 100 M reads.)
struct ColDesc {
  uint32_t w0;     // first word of the span (internal numbering)
  uint32_t nw;     // words in the span (>= 1)
  uint32_t off;    // index into packed[] of the span (valid when nw > 2)
  uint32_t size;   // members (popcount)
  uint64_t m0, m1; // words w0 and w0 + 1
};
static_assert(sizeof(ColDesc) == 32, "descriptor must be 32 bytes");

inline uint32_t make_payload(uint32_t colour, uint32_t lext, uint32_t rext, uint32_t r_int, uint32_t l_int) {
  return (colour << 10) | (l_int << 9) | (r_int << 8) | (lext << 4) | rext;
}

inline uint64_t kmer_mask(int k) { return (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1); }

inline uint64_t rc_kmer_host(uint64_t x, int k) {
  uint64_t v = ~x, r = 0;
  for (int j = 0; j < k; ++j) { r = (r << 2) | (v & 3); v >>= 2; }
  return r;
}


struct HostIndex {
  int k = 20;
  bool compact = false;               // true: ctable/chash are the table (k <= COMPACT_MAX_K); false: buckets/hash
  CHash chash{};
  std::vector<CBucket> ctable;        // chash.nb buckets
  int n_seqs = 0;
  int words = 0;                      // u64 words per colour row in ORIGINAL numbering = ceil(n_seqs / 64): eq_rows, the C-ABI
  int words_dev = 0;                  // u64 words per row in INTERNAL numbering (device descriptors, class table rows)
  bool has_ph = false;                // perfect-hash table of the lane-per-read kernels (built next to the compact table when it fits)
  PHash ph{};
  std::vector<PHEntry> ph_table;      // ph.n_slots entries
  std::vector<uint8_t> ph_pilots;     // ph.n_groups bytes
  bool has_anchor = false;            // anchor structures of nb_anchor.hip (built next to the perfect-hash table)
  int anchor_nw = 1;                  // words per sparse-table entry
  int max_comp_words = 1;             // class-row words of the widest connected component (internal numbering)
  std::vector<uint32_t> comp_w0;      // per sequence: first class-row word of its component
  std::vector<uint32_t> comp_nw;      // per sequence: class-row words its component spans
  int64_t text_len = 0;               // bases in `text`
  std::vector<uint32_t> text;         // text_len bases + 64 bases of padding
  std::vector<AEntry> atable;         // ph.n_slots entries, slot for slot the k-mers of ph_table
  std::vector<uint32_t> akeys;        // the key words of atable alone (4 bytes per slot): what a presence / strand test needs (probes, the
                                      // other-orientation check, the scan kernel) from an array a quarter the size of the table
  WordsNoInit sparse;                 // ANCHOR_LEVELS * text_len words (GB-sized for large libraries: sized without touching the pages, filled by threads)
  std::vector<uint32_t> text_rint;    // one bit per text position: r_internal of the k-mer that starts there (tolerated mismatches of the anchor walk)
  std::vector<uint32_t> pfilter;      // presence filter over the oriented k-mers of the table (above): 2^pf_wbits words
  uint32_t pf_wbits = 0;
  int pf_m = 0, pf_s = 1;             // m-mer length of the filters, stride S = k + 1 - m of the tests
  std::vector<uint32_t> lfilter;      // first, LDS-sized stage of it (2^LF_BITS_LOG2 bits), empty when it would be more than half full
  int max_desc_nw = 1;                // widest class span in words (1 = every class inside one word: the lane-per-read kernels apply)
  std::vector<int32_t> perm;          // original sequence index -> internal bit position
  std::vector<int32_t> inv_perm;      // internal bit position -> original sequence index, -1 = padding
  KmerHash hash{};                    // hash.n_buckets buckets of SLOTS_PER_BUCKET slots
  int64_t n_kmers = 0;                // distinct stranded k-mers
  int64_t n_canonical = 0;
  std::vector<Bucket> buckets;        // hash.n_buckets
  std::vector<uint64_t> eq_rows;      // n_colours * words, dense bitsets (HOST only: nb_index_query, tests)
  std::vector<uint32_t> eq_sizes;     // n_colours
  std::vector<ColDesc> desc;          // n_colours descriptors (device)
  std::vector<uint64_t> packed;       // spans of the colours wider than two words (device)
  // device copies (owned by the ctx that uploaded; recorded here for stats)
  int uploaded = 0;
};

// sorted runs on worker threads, then rounds of pairwise merges (the keys are total orders, so the result is unique)
template <class T, class Less>
void parallel_sort(std::vector<T>& v, int threads, Less less) {
  const size_t n = v.size();
  int runs = 1; while (runs * 2 <= threads && n / (size_t)(runs * 2) >= ((size_t)1 << 16)) runs *= 2;
  if (runs == 1) { std::sort(v.begin(), v.end(), less); return; }
  std::vector<size_t> cut((size_t)runs + 1);
  for (int r = 0; r <= runs; ++r) cut[(size_t)r] = n * (size_t)r / (size_t)runs;
  { std::vector<std::thread> th;
    for (int r = 0; r < runs; ++r) th.emplace_back([&, r] { std::sort(v.begin() + (std::ptrdiff_t)cut[(size_t)r], v.begin() + (std::ptrdiff_t)cut[(size_t)r + 1], less); });
    for (auto& x : th) x.join(); }
  std::vector<T> tmp(n);
  std::vector<T>* src = &v; std::vector<T>* dst = &tmp;
  for (int width = 1; width < runs; width *= 2) {
    std::vector<std::thread> th;
    for (int r = 0; r < runs; r += 2 * width)
      th.emplace_back([&, r] {
        const size_t lo = cut[(size_t)r], mid = cut[(size_t)std::min(r + width, runs)], hi = cut[(size_t)std::min(r + 2 * width, runs)];
        std::merge(src->begin() + (std::ptrdiff_t)lo, src->begin() + (std::ptrdiff_t)mid, src->begin() + (std::ptrdiff_t)mid, src->begin() + (std::ptrdiff_t)hi,
                   dst->begin() + (std::ptrdiff_t)lo, less); });
    for (auto& x : th) x.join();
    std::swap(src, dst);
  }
  if (src != &v) v.swap(tmp);
}

void set_error(const std::string& msg);
int build_host_index(int n_seqs, const char* const* seqs, const int64_t* lens, int k, HostIndex** out);

}  // namespace nb
