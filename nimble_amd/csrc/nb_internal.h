// nb_internal.h — layouts shared by the host index builder and the HIP kernels (not part of the C-ABI).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "nimble_amd.h"

namespace nb {

// ---- k-mer encoding (product side): base j of the k-mer sits at bits [2j, 2j+2): FIRST base LEAST significant,
// so that a k-mer is a plain bit-field of the packed read (nimble_amd.h, nb_pack_reads). ----

// One slot of the canonical-k-mer table, 16 B, read with one dwordx4 load.
//   key    canonical k-mer = min(x, rc(x)); NB_SLOT_EMPTY when free
//   pay[0] payload of the k-mer equal to `key` as it occurs in the reference (stranded), or NB_PAY_ABSENT
//   pay[1] payload of rc(key) as it occurs in the reference, or NB_PAY_ABSENT (palindromes: pay[1] == pay[0])
// payload bits: [0,4) rext mask  [4,8) lext mask  8 r_internal  9 l_internal  [10,32) colour id
struct Slot {
  uint64_t key;
  uint32_t pay[2];
};
static_assert(sizeof(Slot) == 16, "slot must be 16 bytes");

constexpr uint64_t SLOT_EMPTY = ~0ull;
constexpr uint32_t PAY_ABSENT = 0xFFFFFFFFu;
constexpr uint32_t MAX_COLOURS = (1u << 22) - 1;  // colour ids are 22 bits; all-ones reserved by PAY_ABSENT
constexpr uint64_t HASH_MULT = 0x9E3779B97F4A7C15ull;

inline uint32_t make_payload(uint32_t colour, uint32_t lext, uint32_t rext, uint32_t r_int, uint32_t l_int) {
  return (colour << 10) | (l_int << 9) | (r_int << 8) | (lext << 4) | rext;
}

inline uint64_t kmer_mask(int k) { return (k >= 32) ? ~0ull : ((1ull << (2 * k)) - 1); }

inline uint64_t rc_kmer_host(uint64_t x, int k) {
  uint64_t v = ~x, r = 0;
  for (int j = 0; j < k; ++j) { r = (r << 2) | (v & 3); v >>= 2; }
  return r;
}

inline uint64_t slot_of(uint64_t canon, int cap_log2) { return (canon * HASH_MULT) >> (64 - cap_log2); }

struct HostIndex {
  int k = 20;
  int n_seqs = 0;
  int words = 0;                      // u64 words per colour row
  int cap_log2 = 0;
  int64_t n_kmers = 0;                // distinct stranded k-mers
  int64_t n_canonical = 0;
  std::vector<Slot> slots;            // 1 << cap_log2
  std::vector<uint64_t> eq_rows;      // n_colours * words; colour ids ascend by set size
  std::vector<uint32_t> eq_sizes;     // n_colours
  // device copies (owned by the ctx that uploaded; recorded here for stats)
  int uploaded = 0;
};

void set_error(const std::string& msg);
int build_host_index(int n_seqs, const char* const* seqs, const int64_t* lens, int k, HostIndex** out);

}  // namespace nb
