// anchor_emul.cpp — the device code of nimble_amd/csrc/nb_anchor.hip compiled for the HOST with one lane per wave (tests/native/hipshim)
// and run on the real index structures (nb_index.cpp): the walk logic of the anchor-extend kernel under AddressSanitizer / UBSan, without a
// GPU. tests/test_anchor_emul_cpu.py writes the input file, runs this program and compares every read the kernels finish themselves with
// the oracle (the redo list — reads the GPU hands to the wave-per-read kernel — is reported as such and must stay short).
//
//   anchor_emul <in.bin> <out.bin>
//   in : i32 k, n_seqs, n_reads, max_len | per sequence: i32 len, bytes | u64 packed[n_reads][stride] | u16 lens[n_reads]
//        | i32 score_threshold, score_filter, discard_multiple, strand_filter | f64 score_percent
//   out: i32 n_reads, words | per read: i32 status (0 final, 1 redo), reason, score, orientation, mates | u64 cov | u64 set[words]
#define NB_WAIT_VMCNT0() ((void)0)
#define __HIPCC__ 1                     // the product headers keep their per-lane device helpers behind it
#include <hip/hip_runtime.h>

dim3 threadIdx, blockIdx, blockDim, gridDim;

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../nimble_amd/csrc/nb_anchor.hip"

namespace nb {
unsigned char s_dyn[160 << 10] __attribute__((aligned(16)));      // the kernels' dynamic LDS
static std::string g_error;
void set_error(const std::string& m) { g_error = m; }
int64_t lane_lds_bytes(const nb_ctx*, int, int*) { return 0; }
}  // namespace nb

using namespace nb;

template <class T> static T rd(FILE* f) { T v; if (std::fread(&v, sizeof v, 1, f) != 1) { std::fprintf(stderr, "short input\n"); std::exit(2); } return v; }

static unsigned long long g_retried = 0;
static int g_pf_s = 1;
template <int NR>
static void run_scan(const AnchorHead& h, const LaneTail* tail) {
  switch (g_pf_s) {
    case 8: nb_anchor_scan_kernel<NR, 8, false>(h, tail); break;
    case 4: nb_anchor_scan_kernel<NR, 4, false>(h, tail); break;
    case 2: nb_anchor_scan_kernel<NR, 2, false>(h, tail); break;
    default: nb_anchor_scan_kernel<NR, 1, false>(h, tail); break;
  }
}

template <int NR, int NW>
static void run_kernels(const AnchorHead& h_in, const LaneTail* tail, long long blocks, std::vector<uint32_t>* retry) {
  // one wave per workgroup, one lane per wave: block b maps read 64 * b, then (last pass) scans its own segment of the scan list.
  // retry: every other read is handed to the second pass of nb_anchor_kernel after NB_EMUL_RETRY rounds (retry_lanes = 1 with one lane per
  // wave, switched per block), with the state of its walk; the second pass resumes it.
  blockDim = dim3(64); gridDim = dim3((unsigned)blocks); threadIdx = dim3(0);
  AnchorHead h = h_in;
  std::vector<uint32_t> counts((size_t)blocks, 0);
  std::vector<uint32_t> state(retry ? ((size_t)blocks * 64 + 64) * RetryShape<NR, NW>::DW : 0, 0xDEADBEEFu);
  h.retry_state = state.data(); h.retry_seg = 64;
  h.retry_rounds = retry ? std::atoi(std::getenv("NB_EMUL_RETRY")) : 0;          // NB_EMUL_RETRY=<rounds>: hand over after that many rounds (0 = before the first)
  for (int pass = 0; pass < 3; ++pass)
    for (long long b = 0; b < blocks; ++b) {
      blockIdx = dim3((unsigned)b);
      if (pass == 0) {
        h.retry_in = nullptr; h.retry_out = (retry && (b & 1)) ? retry->data() : nullptr; h.retry_counts = counts.data(); h.retry_lanes = 1;
        (h.M > 0 ? nb_anchor_kernel<NR, NW, false, false, true>(h, tail) : nb_anchor_kernel<NR, NW, false, false, false>(h, tail));
        if (h.retry_out) g_retried += counts[(size_t)b];
      } else if (pass == 1) {
        if (!retry) continue;
        h.retry_out = nullptr; h.retry_in = retry->data(); h.retry_counts = counts.data();
        (h.M > 0 ? nb_anchor_kernel<NR, NW, false, false, true>(h, tail) : nb_anchor_kernel<NR, NW, false, false, false>(h, tail));
      } else {
        // (the workgroup's copy of the LDS filter stage is a loop over all 1024 threads: with one lane per wave it is made here)
        if (h.lfilter) std::memcpy(s_dyn, h.lfilter, (size_t)SCAN_LF_WORDS * 4);
        h.retry_in = nullptr; h.retry_out = nullptr; run_scan<NR>(h, tail);
      }
    }
}

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: anchor_emul in.bin out.bin\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 2; }
  const int k = rd<int32_t>(f), n_seqs = rd<int32_t>(f), n_reads = rd<int32_t>(f), max_len = rd<int32_t>(f);
  std::vector<std::string> seqs((size_t)n_seqs);
  for (auto& s : seqs) { const int len = rd<int32_t>(f); s.resize((size_t)len); if (len && std::fread(&s[0], 1, (size_t)len, f) != (size_t)len) return 2; }
  const int nwb = (max_len + 31) / 32, nwn = (max_len + 63) / 64, stride = nwb + nwn;
  // one lane per wave: the lane-0 reads of the emulated waves sit at indices 0, 64, 128, ...
  const long long n_emul = 64LL * n_reads;
  std::vector<uint64_t> reads((size_t)n_emul * (size_t)stride, 0);
  std::vector<uint16_t> lens((size_t)n_emul, 0);
  for (int i = 0; i < n_reads; ++i) if (std::fread(&reads[(size_t)i * 64 * (size_t)stride], 8, (size_t)stride, f) != (size_t)stride) return 2;
  for (int i = 0; i < n_reads; ++i) lens[(size_t)i * 64] = rd<uint16_t>(f);
  BatchArgs a{};
  a.score_threshold = rd<int32_t>(f); a.score_filter = rd<int32_t>(f); a.discard_multiple = rd<int32_t>(f); a.strand_filter = rd<int32_t>(f);
  const double score_percent = rd<double>(f);
  std::fclose(f);

  std::vector<const char*> sp; std::vector<int64_t> sl;
  for (auto& s : seqs) { sp.push_back(s.data()); sl.push_back((int64_t)s.size()); }
  HostIndex* ix = nullptr;
  if (build_host_index(n_seqs, sp.data(), sl.data(), k, &ix) != NB_OK) { std::fprintf(stderr, "index: %s\n", g_error.c_str()); return 3; }
  if (!ix->has_anchor) { std::fprintf(stderr, "index has no anchor structures\n"); return 4; }
  const int W = ix->words_dev;
  g_pf_s = ix->pf_s;

  // class table on the host (same layout as nb_ctx_create makes on the device)
  const unsigned cap_log2 = 18; const size_t cap = (size_t)1 << cap_log2; const unsigned max_classes = 1u << 16;
  std::vector<unsigned long long> keys(cap, 0), rows((size_t)max_classes * (size_t)W, 0), counts(max_classes, 0), reasons(NB_NUM_REASONS, 0);
  std::vector<unsigned int> ids(cap, 0), spans(max_classes, 0);
  unsigned int n_classes = 0, err = 0;
  DevClasses ct{keys.data(), ids.data(), rows.data(), counts.data(), spans.data(), &n_classes, &err, reasons.data(), cap_log2, max_classes};
  std::vector<uint16_t> pct((size_t)max_len + 1, 0);
  for (int L = 1; L <= max_len; ++L) { int cov = 0; while (cov <= 65534 && (double)cov / (double)L < score_percent) ++cov; pct[(size_t)L] = (uint16_t)cov; }
  std::vector<uint32_t> out_class((size_t)n_emul, 0xDEADBEEFu), out_info((size_t)n_emul, 0xDEADBEEFu), lists((size_t)n_emul * 2, 0), counters(2, 0);
  std::vector<uint64_t> out_cov((size_t)n_emul, 0);
  a.reads = reads.data(); a.lens = lens.data(); a.max_len = max_len; a.stride = stride; a.nwb = nwb; a.n = n_emul;
  a.out_class = out_class.data(); a.out_info = out_info.data(); a.out_cov = out_cov.data(); a.pct_min = pct.data(); a.pct_fixed = pct[(size_t)max_len];
  LaneLists ll{lists.data(), lists.data() + n_emul, counters.data()};
  LaneTail tail{ct, a, ll, W};
  std::vector<uint32_t> scan_counts((size_t)n_reads, 0);
  AnchorHead h{ix->atable.data(), ix->akeys.data(), ix->ph_pilots.data(), ix->ph, k, ix->text.data(), ix->sparse.data(), (long long)ix->text_len, reads.data(), lens.data(), max_len, stride, nwb,
               n_emul, 0, 2 * (nwb + nwn) + 2, (long long)ix->text.size(), (long long)ix->sparse.size(), W, nullptr, 0, scan_counts.data(), 64, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, ix->pfilter.data(), ix->pf_wbits, ix->pf_s, ix->lfilter.empty() ? nullptr : ix->lfilter.data(),
               ix->text_rint.data(), std::getenv("NB_EMUL_MM") ? std::atoi(std::getenv("NB_EMUL_MM")) : 0};   // NB_EMUL_MM = num_mismatches (the redo reads are then the general kernel's)
  std::vector<uint32_t> retry_list((size_t)n_reads * 64 + 64, 0);
  std::vector<uint32_t>* retry = std::getenv("NB_EMUL_RETRY") ? &retry_list : nullptr;
  const int positions = max_len - k + 1, nr = positions < 1 ? 1 : (positions + 63) / 64;
  const long long blocks = n_reads;
#define RUN(NR) (ix->anchor_nw == 1 ? run_kernels<NR, 1>(h, &tail, blocks, retry) : run_kernels<NR, 2>(h, &tail, blocks, retry))
  switch (nr) { case 1: RUN(1); break; case 2: RUN(2); break; case 3: RUN(3); break; case 4: RUN(4); break; default: RUN(5); break; }
  if (err) { std::fprintf(stderr, "class table error %u\n", err); return 5; }

  std::vector<char> redo((size_t)n_emul, 0);
  for (uint32_t i = 0; i < counters[1]; ++i) redo[ll.redo_list[i]] = 1;
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) { std::perror("open"); return 2; }
  const int32_t words = ix->words;
  std::fwrite(&n_reads, 4, 1, o); std::fwrite(&words, 4, 1, o);
  std::vector<uint64_t> set((size_t)words);
  for (int i = 0; i < n_reads; ++i) {
    const size_t r = (size_t)i * 64;
    const uint32_t info = out_info[r];
    int32_t rec[5] = {redo[r], (int32_t)((info >> 16) & 0xFF), (int32_t)(info & 0xFFFF), (int32_t)((info >> 24) & 1), (int32_t)((info >> 25) & 3)};
    if (!redo[r] && info == 0xDEADBEEFu) { std::fprintf(stderr, "read %d was neither finished nor put on the redo list\n", i); return 6; }
    std::fill(set.begin(), set.end(), 0ull);
    if (!redo[r] && rec[1] == NB_CALLED) {
      const uint32_t cls = out_class[r];
      if (cls >= n_classes) { std::fprintf(stderr, "read %d: class id out of range\n", i); return 6; }
      for (int w = 0; w < W; ++w) for (uint64_t v = rows[(size_t)cls * (size_t)W + (size_t)w]; v; v &= v - 1) {
          const int32_t orig = ix->inv_perm[(size_t)w * 64 + (size_t)__builtin_ctzll(v)];
          if (orig >= 0) set[(size_t)(orig >> 6)] |= 1ull << (orig & 63);
        }
    }
    std::fwrite(rec, 4, 5, o); std::fwrite(&out_cov[r], 8, 1, o); std::fwrite(set.data(), 8, (size_t)words, o);
  }
  std::fclose(o);
  uint32_t n_scan = 0; for (uint32_t v : scan_counts) n_scan += v;
  std::printf("reads %d, scan list %u, redo list %u, retried %llu, classes %u\n", n_reads, n_scan, counters[1], g_retried, n_classes);
  delete ix;
  return 0;
}
