// AddressSanitizer / UBSan run of the host-side C++ (index build, FASTQ ingest, BAM reader, report reducer, whitelist,
// row formatter) and of the oracle. Built with g++ -fsanitize=address,undefined by tests/test_sanitize_cpu.py.
// GPU sanitizers are not available on the pool; the kernels are covered by the bit-exact parity tests instead.
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "nb_internal.h"

namespace nb { static std::string g; void set_error(const std::string& m) { g = m; } }
extern "C" int64_t nb_packed_stride_words(int32_t max_len) { return (int64_t)((max_len + 31) / 32 + (max_len + 63) / 64); }

extern "C" {
void* nbo_index_build(int32_t, const char* const*, const int64_t*, int32_t);
void nbo_index_destroy(void*);
int64_t nbo_index_num_kmers(void*);
struct OCfg { int32_t k, st, sf, mm, dm, il, rv, sfilt; double pct; };
int32_t nbo_align(void*, const OCfg*, int64_t, const char*, const int64_t*, const char*, const int64_t*, uint64_t*, int32_t*, uint8_t*, uint8_t*, uint8_t*, int32_t*, int32_t);
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  std::mt19937_64 rng(3);
  // ---- index build (product) and oracle on a small polymorphic library ----
  std::vector<std::string> seqs;
  for (int l = 0; l < 4; ++l) { std::string anc(300, 'A'); for (auto& c : anc) c = "ACGT"[rng() & 3];
    for (int a = 0; a < 9; ++a) { std::string s = anc; for (auto& c : s) if (rng() % 100 < 2) c = "ACGTN"[rng() % 5]; seqs.push_back(s); } }
  seqs.push_back("ACG"); seqs.push_back("");
  std::vector<const char*> p; std::vector<int64_t> L;
  for (auto& s : seqs) { p.push_back(s.c_str()); L.push_back((int64_t)s.size()); }
  for (int k : {5, 20, 31}) {
    nb::HostIndex* ix = nullptr;
    if (nb::build_host_index((int)seqs.size(), p.data(), L.data(), k, &ix) != NB_OK) { std::printf("index build failed: %s\n", nb::g.c_str()); return 1; }
    void* o = nbo_index_build((int)seqs.size(), p.data(), L.data(), k);
    if (!o || nbo_index_num_kmers(o) != ix->n_kmers) { std::printf("k-mer count mismatch at k=%d\n", k); return 1; }
    if (k == 20) {
      std::string blob; std::vector<int64_t> off{0};
      for (int i = 0; i < 300; ++i) { const std::string& s = seqs[rng() % 36]; const size_t st = rng() % 150; std::string r = s.substr(st, 20 + rng() % 130); blob += r; off.push_back((int64_t)blob.size()); }
      const int64_t n = (int64_t)off.size() - 1; const int W = ((int)seqs.size() + 63) / 64;
      std::vector<uint64_t> bits((size_t)n * W); std::vector<int32_t> score((size_t)n), cov((size_t)n * 4); std::vector<uint8_t> rs((size_t)n), orr((size_t)n), mt((size_t)n);
      OCfg c{20, 20, 25, 1, 0, 1, 0, 0, 0.5};
      if (nbo_align(o, &c, n, blob.data(), off.data(), blob.data(), off.data(), bits.data(), score.data(), rs.data(), orr.data(), mt.data(), cov.data(), 3) != 0) return 1;
    }
    nbo_index_destroy(o); delete ix;
  }
  // ---- FASTQ ingest (plain file written here), 3 threads ----
  {
    const std::string fq = dir + "/r.fastq";
    FILE* f = std::fopen(fq.c_str(), "w");
    for (int i = 0; i < 20000; ++i) { std::string r(40 + rng() % 60, 'A'); for (auto& c : r) c = "ACGTN"[rng() % 5];
      std::fprintf(f, "@r%d\n%s\n+\n%s\n", i, r.c_str(), std::string(r.size(), 'I').c_str()); }
    std::fclose(f);
    nb_fastq* q = nullptr;
    if (nb_fastq_open(fq.c_str(), &q) != NB_OK) return 1;
    nb_fastq_set_threads(q, 3);
    const int64_t stride = nb_packed_stride_words(100);
    std::vector<uint64_t> w((size_t)(8192 * stride)); std::vector<uint16_t> ln(8192); std::vector<char> cb(8192 * 16), um(8192 * 12), cq(8192 * 16);
    int64_t n = 0, total = 0;
    do { if (nb_fastq_next_batch(q, 8192, 100, w.data(), ln.data(), &n, cb.data(), 16, um.data(), 12, cq.data()) != NB_OK) { std::printf("fastq: %s\n", nb::g.c_str()); return 1; } total += n; } while (n);
    nb_fastq_close(q);
    if (total != 20000) { std::printf("fastq count %lld\n", (long long)total); return 1; }
  }
  // ---- BAM reader on the file the Python test wrote ----
  {
    nb_bam* b = nullptr;
    if (nb_bam_open((dir + "/t.bam").c_str(), &b) != NB_OK) { std::printf("bam: %s\n", nb::g.c_str()); return 1; }
    const int64_t stride = nb_packed_stride_words(120);
    std::vector<uint64_t> w1((size_t)(64 * stride)), w2((size_t)(64 * stride)); std::vector<uint16_t> l1(64), l2(64); std::vector<char> cb(64 * 20), ub(64 * 20);
    int64_t n = 0, total = 0; int32_t paired = 0;
    do { if (nb_bam_next_batch(b, 64, 120, w1.data(), l1.data(), w2.data(), l2.data(), cb.data(), ub.data(), 20, &n, &paired) != NB_OK) { std::printf("bam: %s\n", nb::g.c_str()); return 1; } total += n; } while (n);
    nb_bam_close(b);
    if (total != 300) { std::printf("bam templates %lld\n", (long long)total); return 1; }
  }
  // ---- gzip decoder: the good file, then 90 corrupted copies (must fail or succeed cleanly, never touch memory out of bounds) ----
  {
    const std::string gz = dir + "/t.fastq.gz";
    FILE* f = std::fopen(gz.c_str(), "rb"); if (!f) return 1;
    std::vector<unsigned char> good; { unsigned char tmp[65536]; size_t k; while ((k = std::fread(tmp, 1, sizeof tmp, f)) > 0) good.insert(good.end(), tmp, tmp + k); } std::fclose(f);
    std::vector<char> out((size_t)8 << 20); int64_t n = 0;
    if (nb_gunzip_file(gz.c_str(), out.data(), (int64_t)out.size(), &n) != NB_OK || n < 100000) { std::printf("gunzip: %s\n", nb::g.c_str()); return 1; }
    const std::string bad = dir + "/fuzz.gz";
    for (int it = 0; it < 90; ++it) {
      std::vector<unsigned char> v = good;
      const int kind = it % 3;
      if (kind == 0) v[rng() % v.size()] ^= (unsigned char)(1u << (rng() % 8));
      else if (kind == 1) v.resize(rng() % v.size());
      else { const size_t a = rng() % v.size(); for (size_t j = a; j < v.size() && j < a + 1 + rng() % 40; ++j) v[j] = (unsigned char)rng(); }
      FILE* o = std::fopen(bad.c_str(), "wb"); std::fwrite(v.data(), 1, v.size(), o); std::fclose(o);
      (void)nb_gunzip_file(bad.c_str(), out.data(), (int64_t)out.size(), &n);
    }
    nb_fastq* q = nullptr; int64_t total = 0;
    if (nb_fastq_open(gz.c_str(), &q) != NB_OK) return 1;
    nb_fastq_set_threads(q, 3);
    std::vector<uint64_t> w((size_t)40000 * nb_packed_stride_words(160)); std::vector<uint16_t> ln(40000);
    for (;;) { int32_t ml = 0; if (nb_fastq_next_batch_tight(q, 40000, 160, w.data(), ln.data(), &n, &ml, nullptr, 0, nullptr, 0, nullptr) != NB_OK) { std::printf("gz fastq: %s\n", nb::g.c_str()); return 1; } if (!n) break; total += n; }
    nb_fastq_close(q);
    if (total != 10000) { std::printf("gz fastq: %lld records\n", (long long)total); return 1; }
  }
  // ---- gzip writer + class feature strings ----
  {
    std::string text; for (int i = 0; i < 200000; ++i) text += "L0*01,L0*02\t1\tACGTACGTACGTACGT\tAACCGGTTAACC\t91\t0\t0\t0\n";
    std::vector<char> out(text.size() + (text.size() >> 9) + 64 * ((text.size() >> 20) + 1) + 64); int64_t ol = 0;
    if (nb_gzip_members(text.data(), (int64_t)text.size(), 4, 3, out.data(), (int64_t)out.size(), &ol) != NB_OK) return 1;
    FILE* o = std::fopen((dir + "/w.gz").c_str(), "wb"); std::fwrite(out.data(), 1, (size_t)ol, o); std::fclose(o);
    std::vector<char> back(text.size() + 16); int64_t n = 0;
    if (nb_gunzip_file((dir + "/w.gz").c_str(), back.data(), (int64_t)back.size(), &n) != NB_OK || (size_t)n != text.size() || std::memcmp(back.data(), text.data(), text.size())) { std::printf("gzip round trip failed\n"); return 1; }
    const char* names[3] = {"A", "B*02", "C"}; int32_t sf[70]; for (int i = 0; i < 70; ++i) sf[i] = i % 3;
    uint64_t rows[4] = {0x5ull, 0x0ull, ~0ull, 0x3Full}; char blob[256], sblob[256]; int64_t off[3], soff[3]; int32_t rs[2];
    if (nb_class_features(2, 2, rows, 70, sf, names, 3, 2, 0, blob, 256, off, sblob, 256, soff, rs) != NB_OK || rs[0] != NB_CALLED || rs[1] != NB_MAX_HITS) return 1;
  }
  // ---- report reducer on the TSV the Python test wrote ----
  {
    int64_t st[2];
    if (nb_report_tsv((dir + "/in.tsv").c_str(), (dir + "/out.tsv").c_str(), 0.05, 0, st) != NB_OK) { std::printf("report: %s\n", nb::g.c_str()); return 1; }
  }
  // ---- whitelist + row formatter ----
  {
    nb_whitelist* wl = nullptr;
    if (nb_whitelist_load((dir + "/wl.txt").c_str(), &wl) != NB_OK) return 1;
    const int n = 1000; std::vector<char> raw(n * 8), q(n * 8, 'I'), out(n * 8); std::vector<uint8_t> ok(n); int64_t stt[3];
    for (auto& c : raw) c = "ACGTN"[rng() % 5];
    if (nb_whitelist_correct(wl, raw.data(), q.data(), n, 8, out.data(), ok.data(), stt) != NB_OK) return 1;
    int64_t fst[8];
    if (nb_fastq_to_bam((dir + "/t.fastq.gz").c_str(), (dir + "/t.fastq.gz").c_str(), wl, (dir + "/conv.bam").c_str(), 3, 8, 4, fst) != NB_OK || fst[0] != 10000) { std::printf("fastq-to-bam: %s\n", nb::g.c_str()); return 1; }
    nb_whitelist_destroy(wl);
    if (fst[1] > 0) {
      nb_bam* cb = nullptr; int64_t total = 0, got = 0; int32_t pr = 0, l1 = 0, l2 = 0;
      if (nb_bam_open((dir + "/conv.bam").c_str(), &cb) != NB_OK) { std::printf("conv bam: %s\n", nb::g.c_str()); return 1; }
      nb_bam_set_threads(cb, 3);
      std::vector<uint64_t> a((size_t)4096 * nb_packed_stride_words(160)), c2((size_t)4096 * nb_packed_stride_words(160)); std::vector<uint16_t> la(4096), lb(4096); std::vector<char> t1(4096 * 20), t2(4096 * 20);
      do { if (nb_bam_next_batch_tight(cb, 4096, 160, a.data(), la.data(), c2.data(), lb.data(), t1.data(), t2.data(), 20, &got, &l1, &l2, &pr) != NB_OK) { std::printf("conv bam: %s\n", nb::g.c_str()); return 1; } total += got; } while (got);
      nb_bam_close(cb);
      if (total != fst[1] || !pr) { std::printf("conv bam: %lld templates, %lld written\n", (long long)total, (long long)fst[1]); return 1; }
    }
    std::vector<uint32_t> cls(n), info(n); std::vector<uint64_t> cov(n);
    for (int i = 0; i < n; ++i) { cls[i] = i % 5 == 4 ? NB_NO_CLASS : i % 3; info[i] = (i % 7 ? 0u : 2u << 16) | (1u << 25) | 100; cov[i] = 100; }
    const char* feats[3] = {"A*01", nullptr, "B*02,B*03"};
    std::vector<char> buf(n * 64); int64_t ol = 0, nr = 0;
    if (nb_format_tenx_rows(n, cls.data(), info.data(), cov.data(), feats, 3, out.data(), 8, ok.data(), raw.data(), 8, buf.data(), (int64_t)buf.size(), &ol, &nr) != NB_OK) return 1;
    if (nb_format_tenx_rows(n, cls.data(), info.data(), cov.data(), feats, 3, out.data(), 8, ok.data(), raw.data(), 8, buf.data(), 50, &ol, &nr) != NB_ERR_OVERFLOW) return 1;
  }
  std::printf("sanitize ok\n");
  return 0;
}
