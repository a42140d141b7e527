/* nimble_amd.h — C-ABI of the MI355X pseudoalignment engine (libnimble_amd.so).
 *
 * Drop-in boundary. In the reference the hot path is entered through ONE process boundary:
 *   subprocess.Popen([<pkgdir>/aligner] + argv); proc.wait()      /root/reference/nimble/__main__.py:195-198
 * with argv = --input F.. -c N --strand_filter S {-r LIB -o OUT}.. [-t TRIM]   (__main__.py:177-192)
 * and the exit code as the only status (__main__.py:198,211). The reference has no FFI for this path
 * (ctypes is imported at __main__.py:5 only for csv.field_size_limit, :374), so the entry points below are what a
 * ctypes binding placed at __main__.py:195 binds instead of spawning the Rust binary; INTEGRATION.md shows it.
 *
 * Conventions: plain C, no C++/torch types. Every function returns int32 status (0 = NB_OK, <0 = error) unless
 * stated; nb_last_error() returns a thread-local message for the last failing call on this thread. Host buffers are
 * caller-allocated. Device memory of the index and of the class table is owned by the library behind the opaque
 * handles; read batches and per-read outputs are caller-provided DEVICE pointers (e.g. torch tensors' data_ptr) and
 * `stream` is a hipStream_t passed as void* (NULL = default stream). One nb_ctx per host thread / GPU; no globals.
 */
#ifndef NIMBLE_AMD_H
#define NIMBLE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_OK 0
#define NB_ERR_INVALID (-1)     /* bad argument */
#define NB_ERR_HIP (-2)         /* a HIP runtime call failed (message has the hipError string) */
#define NB_ERR_UNSUPPORTED (-3) /* valid request this build cannot serve (e.g. read longer than NB_MAX_READ_LEN) */
#define NB_ERR_OVERFLOW (-4)    /* class table full: raise max_classes */
#define NB_ERR_NOGPU (-5)       /* no usable HIP device */

#define NB_ABI_VERSION 4
#define NB_MAX_K 31
#define NB_MAX_POSITIONS 320    /* k-mer positions per read the kernels handle (5 wave-rounds of 64) */
#define NB_NO_CLASS 0xFFFFFFFFu

/* strand_filter values; strings accepted by the CLI at /root/reference/nimble/__main__.py:399 */
#define NB_STRAND_UNSTRANDED 0
#define NB_STRAND_FIVEPRIME 1
#define NB_STRAND_THREEPRIME 2
#define NB_STRAND_NONE 3

/* reason codes, docs/SPEC.md §4 */
enum {
  NB_CALLED = 0, NB_TOO_SHORT = 1, NB_NO_HIT = 2, NB_SCORE_THRESHOLD = 3, NB_SCORE_PERCENT = 4, NB_EMPTY_EQCLASS = 5,
  NB_MULTIPLE_MATCH = 6, NB_INVALID_PAIR = 7, NB_STRAND = 8, NB_SCORE_FILTER = 9, NB_MAX_HITS = 10, NB_MULTI_HITS = 11,
  NB_NUM_REASONS = 12
};

/* Aligner parameters evaluated on the device. Replaces the numeric part of the library JSON's config object that
 * the Rust binary parses: /root/reference/nimble/types.py:10-25 (field for field where named the same).
 * group_on / discard_multi_hits / max_hits_to_report act on feature NAMES and stay on the host (nimble_amd/align.py). */
typedef struct nb_config {
  int32_t k;                        /* k-mer length, 2..31; docs/SPEC.md §1 (default 20) */
  int32_t score_threshold;          /* types.py:12 */
  int32_t score_filter;             /* types.py:13 */
  int32_t num_mismatches;           /* types.py:15 */
  int32_t discard_multiple_matches; /* types.py:16 (bool) */
  int32_t intersect_level;          /* types.py:17: 0 union, 1 intersect with fallback, 2 force intersect */
  int32_t require_valid_pair;       /* types.py:20 (bool) */
  int32_t strand_filter;            /* NB_STRAND_*; --strand_filter, __main__.py:180,399 */
  double score_percent;             /* types.py:14 */
} nb_config;

typedef struct nb_index_stats_t {
  int64_t n_seqs;
  int64_t n_kmers;        /* distinct stranded k-mers (|S|) */
  int64_t n_canonical;    /* distinct canonical k-mers */
  int64_t table_slots;    /* entry capacity of the k-mer table */
  int64_t table_bytes;    /* bytes of the k-mer table: 8-byte entries, (2,2) cuckoo (k <= 23) or 16-byte slots (k > 23) */
  int64_t n_colours;      /* equivalence classes of the graph */
  int64_t words_per_row;  /* u64 words per colour bitset = ceil(n_seqs/64) */
  int64_t device_bytes;   /* bytes the index occupies in HBM once uploaded */
  int64_t ph_table_bytes; /* perfect-hash table of the lane-per-read kernels (16-byte entries; 0 = not built) and its first level,  */
  int64_t ph_pilot_bytes; /* one byte per group, which the kernels stage in LDS (csrc/nb_internal.h, PHEntry)                    */
  int64_t anchor_bytes;   /* anchor structures of the anchor-extend kernel: library text, anchor entries, sparse AND table (0 = not built) */
  int32_t k;
  int32_t uploaded;
  int32_t compact;        /* 1: compact cuckoo table (csrc/nb_internal.h), 0: wide table */
  int32_t words_internal; /* u64 words per class row on the device: sequences are renumbered internally so that every class of a
                             component of <= 64 sequences lies in one word (csrc/nb_internal.h, ColDesc); callers never see it */
} nb_index_stats_t;

typedef struct nb_ctx nb_ctx;
typedef struct nb_index nb_index;

int32_t nb_version(void);
const char* nb_last_error(void);

/* ---- context: one per GPU. max_classes bounds the number of distinct result sets (class table capacity). ---- */
int32_t nb_ctx_create(int32_t device_id, int64_t max_classes, nb_ctx** out);
int32_t nb_ctx_destroy(nb_ctx* ctx);
int32_t nb_ctx_sync(nb_ctx* ctx, void* stream);

/* ---- device memory, streams, copies: so that a caller needs no GPU array library (PyTorch, hip-python ...) to drive
 * nb_align_batch: INTEGRATION.md §B uses only these. Buffers from torch / any other allocator on the same device work as
 * well (the hot call takes raw device pointers). `stream` = a stream from nb_stream_create, any hipStream_t, or NULL. ---- */
int32_t nb_device_alloc(nb_ctx* ctx, int64_t bytes, void** d_out);
int32_t nb_device_free(nb_ctx* ctx, void* d_ptr);
int32_t nb_host_alloc_pinned(int64_t bytes, void** out);     /* page-locked host memory: async copies are true DMA */
int32_t nb_host_free_pinned(void* ptr);
int32_t nb_stream_create(nb_ctx* ctx, void** stream_out);
int32_t nb_stream_destroy(nb_ctx* ctx, void* stream);
int32_t nb_copy_h2d_async(nb_ctx* ctx, void* d_dst, const void* src, int64_t bytes, void* stream);
int32_t nb_copy_d2h_async(nb_ctx* ctx, void* dst, const void* d_src, int64_t bytes, void* stream);

/* ---- index: replaces the Rust binary's "-r LIB" load + graph build (__main__.py:182-189).
 * Built on the host from the library's `sequence` column (types.py:29-32, headers[3]); host-only, needs no GPU. ---- */
int32_t nb_index_build(int32_t n_seqs, const char* const* seqs, const int64_t* seq_lens, int32_t k, nb_index** out);
int32_t nb_index_destroy(nb_index* index);
int32_t nb_index_stats(const nb_index* index, nb_index_stats_t* out);
/* host query of one stranded k-mer (ASCII, length k): returns 1 if present and fills exts/flags/colour bitset
 * (words_per_row u64), 0 if absent, <0 on error. Used by index-parity tests. */
int32_t nb_index_query(const nb_index* index, const char* kmer_ascii, uint8_t* lext, uint8_t* rext, uint8_t* r_internal,
                       uint8_t* l_internal, uint64_t* colour_bits);

/* host query of the presence filter of the scan kernel (csrc/nb_internal.h) for a sequence of len >= m bases AS GIVEN (no canonical
 * form, no reverse complement): 1 = every m-mer of it passes both stages, i.e. a k-mer containing them may be in the library (the kernel
 * then takes the exact lookup); 0 = some m-mer is certainly in no k-mer of the library on either strand; 2 = the index carries no
 * filter (no anchor structures); len = 0 only reports the parameters. m_out = the filter's m, lds_stage_out = 1 if the LDS stage was built. An m-mer with a base other than
 * ACGT counts as passing. No false negatives: nb_index_query fails if an m-mer of a k-mer of the table is rejected. Test / inspection
 * helper; the reference has no counterpart (its index lives inside the Rust binary, /root/reference/nimble/__main__.py:195). */
int32_t nb_index_filter_maybe(const nb_index* index, const char* seq_ascii, int32_t len, int32_t* m_out, int32_t* lds_stage_out);
/* host query of the anchor structures (csrc/nb_internal.h, AEntry) for one stranded k-mer: returns 1 and fills the text position of its
 * anchor occurrence, `room` (steps the walk may take along the text from there) and, in colour_and (words_per_row u64, original sequence
 * numbering), the AND of the classes of the k-mers at text positions [tpos, tpos + min(run, room)]; 2 (tpos and room only) if the anchor
 * sequence belongs to a component wider than the sparse table's entries (such reads are mapped by the wave-per-read kernel); 0 if the
 * k-mer is absent on this strand, occurs on both strands or the index has no anchor structures. Used by index-parity tests. */
int32_t nb_index_anchor_query(const nb_index* index, const char* kmer_ascii, int32_t run, int64_t* tpos, int32_t* room, uint64_t* colour_and);
/* copy the index into HBM of ctx's GPU and bind it to ctx */
int32_t nb_index_upload(nb_ctx* ctx, nb_index* index);

/* ---- read packing (host helper): replaces the Rust binary's FASTQ/BAM record decode for the device format.
 * Layout per read: bases_words = ceil(max_len/32) u64 of 2-bit codes (base i at bits [2*(i%32), +2) of word i/32,
 * A C G T = 0 1 2 3, N packed as 0) followed by nmask_words = ceil(max_len/64) u64 with bit i set iff base i is N.
 * stride (u64 words) = bases_words + nmask_words. `ascii` holds the reads back to back, read i = [off[i], off[i+1]). */
int64_t nb_packed_stride_words(int32_t max_len);
int32_t nb_pack_reads(const char* ascii, const int64_t* offsets, int64_t n_reads, int32_t max_len, uint64_t* out_words,
                      uint16_t* out_lens);

/* ---- the hot path: replaces the per-read work of the Rust binary (one call per batch, any number of batches).
 * d_reads/d_lens (+ optional mates) are DEVICE pointers in the packed layout above; d_lens may be NULL when every
 * read has length max_len. Outputs (device, caller-allocated, any may be NULL to skip):
 *   d_out_class[n]  u32  class id (index into the ctx class table) or NB_NO_CLASS
 *   d_out_info[n]   u32  bits 0-15 score, 16-23 reason, 24 orientation (0 = F, 1 = R), 25-26 mates
 *   d_out_cov[n]    u64  four u16 coverage scores: r1 fwd | r1 rev << 16 | r2 fwd << 32 | r2 rev << 48
 * Also accumulates count[class] += 1 per called template and per-reason counters in ctx. Asynchronous on `stream`. */
int32_t nb_align_batch(nb_ctx* ctx, const nb_config* cfg, const uint64_t* d_reads, const uint16_t* d_lens, int32_t max_len,
                       const uint64_t* d_mates, const uint16_t* d_mate_lens, int32_t mate_max_len, int64_t n_reads,
                       uint32_t* d_out_class, uint32_t* d_out_info, uint64_t* d_out_cov, void* stream);

/* ---- results held in ctx (synchronise the stream first) ---- */
int32_t nb_classes_count(nb_ctx* ctx, int64_t* n_classes);
/* rows_out: n_classes * words_per_row u64 (bitset over sequences); counts_out: n_classes i64 */
int32_t nb_classes_read(nb_ctx* ctx, int64_t cap, uint64_t* rows_out, int64_t* counts_out);
/* classes [first, first + count) only (align.py reads the classes a batch added, not the whole table) */
int32_t nb_classes_read_range(nb_ctx* ctx, int64_t first, int64_t count, uint64_t* rows_out, int64_t* counts_out);
int32_t nb_reasons_read(nb_ctx* ctx, int64_t* counts_out /* NB_NUM_REASONS */);
int32_t nb_results_reset(nb_ctx* ctx, void* stream);
/* device pointer + length of the per-class count vector (u64), for an RCCL reduce by the caller (bench.py, align.py) */
int32_t nb_counts_device_ptr(nb_ctx* ctx, void** d_counts, int64_t* capacity);
/* the class table for a caller that reduces it ON THE DEVICE (several ranks: nimble_amd/merge.py): copies rows [0, n) — in the engine's
 * INTERNAL sequence numbering, words_internal u64 per row (nb_index_stats) — and their counts into the caller's device buffers, on `stream`;
 * nb_rows_to_external turns such rows (host memory, e.g. the merged table on the root) into bitsets over the original sequence indices,
 * words_per_row u64 per row. Internal numbering depends on the library alone: equal on every rank that built the same index. */
int32_t nb_classes_copy_device(nb_ctx* ctx, int64_t n, uint64_t* d_rows_out, int64_t* d_counts_out, void* stream);
int32_t nb_rows_to_external(nb_ctx* ctx, const uint64_t* rows_internal, int64_t n, uint64_t* rows_out);

/* ---- FASTQ(.gz) ingest (host): replaces the Rust binary's decode of each "--input" file (__main__.py:177-179) and,
 * for 10x R1, the CB/UMI split of /root/reference/nimble/fastq_barcode_processor.py:131-141. Packs straight into the
 * device layout. cb_out/umi_out/cbqual_out (fixed width cb_len/umi_len/cb_len per record; cbqual = quality characters
 * of the CB bases) and out_words/out_lens may each be NULL. ---- */
typedef struct nb_fastq nb_fastq;
int32_t nb_fastq_open(const char* path, nb_fastq** out);
int32_t nb_fastq_close(nb_fastq* fq);
/* several ranks, one input: a byte range [begin, end) (end < 0: to the end) of a PLAIN FASTQ file that starts at a record boundary
 * (NB_ERR_UNSUPPORTED for compressed files: they cannot be entered in the middle); the helpers below find such boundaries:
 * newlines in a byte range (+ its last byte, -1 if empty) and the offset just behind the n-th newline at or after `begin` (-1: fewer) */
int32_t nb_fastq_open_range(const char* path, int64_t begin, int64_t end, nb_fastq** out);
int32_t nb_file_count_newlines(const char* path, int64_t begin, int64_t end, int64_t* count_out, int32_t* last_byte_out);
int32_t nb_file_nth_newline(const char* path, int64_t begin, int64_t n, int64_t* offset_out);
/* worker threads used to pack a batch (the reference's -c/--num_cores, __main__.py:398); default 1 */
int32_t nb_fastq_set_threads(nb_fastq* fq, int32_t threads);
int32_t nb_fastq_next_batch(nb_fastq* fq, int64_t max_reads, int32_t max_len, uint64_t* out_words, uint16_t* out_lens,
                            int64_t* n_out, char* cb_out, int32_t cb_len, char* umi_out, int32_t umi_len, char* cbqual_out);
/* the same in two steps, so that the caller can size the packed layout to the batch: index finds the next records and
 * reports the longest sequence line; pack writes them (max_len >= that length) and consumes the batch. */
int32_t nb_fastq_index_batch(nb_fastq* fq, int64_t max_reads, int64_t* n_out, int32_t* max_seq_len_out);
int32_t nb_fastq_pack_batch(nb_fastq* fq, int32_t max_len, uint64_t* out_words, uint16_t* out_lens, char* cb_out, int32_t cb_len,
                            char* umi_out, int32_t umi_len, char* cbqual_out);
/* streaming form used by align.py: cache-sized sub-batches, all packed at the layout of *max_len_out = the longest read
 * of the first sub-batch (the batch ends early before a longer read); out_words holds max_reads *
 * nb_packed_stride_words(cap_len) words, cap_len = longest read accepted. */
int32_t nb_fastq_next_batch_tight(nb_fastq* fq, int64_t max_reads, int32_t cap_len, uint64_t* out_words, uint16_t* out_lens,
                                  int64_t* n_out, int32_t* max_len_out, char* cb_out, int32_t cb_len, char* umi_out, int32_t umi_len,
                                  char* cbqual_out);

/* a whole gzip file (any number of members, CRC-32 / ISIZE verified) decoded with the engine's own inflater
 * (csrc/nb_inflate.cpp: what reads .gz FASTQ input); NB_ERR_OVERFLOW if cap is too small */
int32_t nb_gunzip_file(const char* path, char* out, int64_t cap, int64_t* n_out);

/* ---- BAM input (host, zlib only): replaces the Rust binary's BAM decode for `--input x.bam` (__main__.py:171-175).
 * No sort is needed (the reference sorts by UR only so that its aligner sees one UMI's records together,
 * __main__.py:344-347; this engine emits one row per template and `report` groups them). r1/r2 are packed at
 * nb_packed_stride_words(max_len); cb_out/ub_out receive the Z tags CB / UB of r1, tag_width bytes per template, NUL
 * padded. *paired_out = 1 when the file holds pairs (mates adjacent or not: a record waits for its mate by read name; an orphan
 * comes out with an empty partner). *n_out = templates read (0 = end of file). ---- */
typedef struct nb_bam nb_bam;
int32_t nb_bam_open(const char* path, nb_bam** out);
int32_t nb_bam_close(nb_bam* bam);
int32_t nb_bam_next_batch(nb_bam* bam, int64_t max_templates, int32_t max_len, uint64_t* r1_words, uint16_t* r1_lens, uint64_t* r2_words,
                          uint16_t* r2_lens, char* cb_out, char* ub_out, int32_t tag_width, int64_t* n_out, int32_t* paired_out);
/* worker threads for BGZF inflate and packing (the reference's -c/--num_cores); default 1 */
int32_t nb_bam_set_threads(nb_bam* bam, int32_t threads);
/* streaming form used by align.py (see nb_fastq_next_batch_tight): r1 packed at the layout of *len1_out, r2 at *len2_out;
 * the word buffers hold max_templates * nb_packed_stride_words(cap_len) words. */
int32_t nb_bam_next_batch_tight(nb_bam* bam, int64_t max_templates, int32_t cap_len, uint64_t* r1_words, uint16_t* r1_lens,
                                uint64_t* r2_words, uint16_t* r2_lens, char* cb_out, char* ub_out, int32_t tag_width, int64_t* n_out,
                                int32_t* len1_out, int32_t* len2_out, int32_t* paired_out);

/* ---- 10x cell-barcode whitelist correction (host): restates correct_cell_barcode,
 * /root/reference/nimble/fastq_barcode_processor.py:73-128 (exact -> Hamming-1 candidates -> lowest quality at the
 * differing base; per-raw-barcode cache). raw/qual/out are fixed-width cb_len per record; ok[i] = 0 drops the pair
 * (:176-179). stats (may be NULL) receives cumulative {perfect, corrected, none}. ---- */
typedef struct nb_whitelist nb_whitelist;
int32_t nb_whitelist_load(const char* path, nb_whitelist** out);
int32_t nb_whitelist_destroy(nb_whitelist* wl);
int64_t nb_whitelist_size(const nb_whitelist* wl);
/* worker threads for the exact-match pass of nb_whitelist_correct (order-independent); default 1 */
int32_t nb_whitelist_set_threads(nb_whitelist* wl, int32_t threads);
int32_t nb_whitelist_correct(nb_whitelist* wl, const char* raw, const char* qual, int64_t n, int32_t cb_len, char* out, uint8_t* ok,
                             int64_t* stats);

/* ---- class bitsets -> feature strings (host; docs/SPEC.md §5): `group_on` naming (/root/reference/nimble/types.py:18) and the
 * feature-count filters max_hits_to_report / discard_multi_hits (types.py:19,22). Row r's distinct feature names in ascending
 * sequence order, comma-joined, go to blob[offsets[r] .. offsets[r+1]); the same names sorted (the form `report` groups by,
 * __main__.py:248) to sorted_blob; reasons[r] = NB_CALLED / NB_MAX_HITS / NB_MULTI_HITS. ---- */
int32_t nb_class_features(int64_t n_rows, int32_t words, const uint64_t* rows, int32_t n_seqs, const int32_t* seq_feature,
                          const char* const* feature_names, int32_t n_features, int32_t max_hits, int32_t multi_hits, char* blob,
                          int64_t blob_cap, int64_t* offsets, char* sorted_blob, int64_t sorted_cap, int64_t* sorted_offsets, int32_t* reasons);

/* ---- `nimble fastq-to-bam` (host): 10x R1/R2 FASTQ(.gz) -> unaligned BAM with the whitelist-corrected cell barcode (CB:Z)
 * and the raw UMI (UB:Z) on both mates; restates fastq_to_bam_with_barcodes / process_pair,
 * /root/reference/nimble/fastq_barcode_processor.py:145-320 (filters in the same order, flags 77 / 141, header text, tag
 * order), records in file order. stats[8] = total_pairs, written_pairs, cb_perfect_match, cb_corrected, cb_no_correction,
 * name_mismatch, too_short, no_remaining_seq (what the reference prints, :290-317). ---- */
int32_t nb_fastq_to_bam(const char* r1_path, const char* r2_path, nb_whitelist* wl, const char* out_bam, int32_t threads, int32_t cb_len,
                        int32_t umi_len, int64_t* stats);

/* ---- per-read TSV rows of one batch (host): replaces the Rust binary's TSV writer for barcoded input; columns
 * nimble_features, nimble_score, r1_CB, r1_UB (consumed at /root/reference/nimble/__main__.py:237-245) + the four
 * coverage scores (report_generation.py:275). class_feature[c] = feature string of class c or NULL if rejected. ---- */
int32_t nb_format_tenx_rows(int64_t n, const uint32_t* cls, const uint32_t* info, const uint64_t* cov, const char* const* class_feature,
                            int64_t n_classes, const char* cb, int32_t cb_len, const uint8_t* cb_ok, const char* umi, int32_t umi_len,
                            char* out, int64_t out_cap, int64_t* out_len, int64_t* n_rows);

/* gzip for that table (the reference's aligner writes .gz tables, validate_gzip at __main__.py:35-42): BGZF blocks (gzip
 * members of <= 64 KiB input with a 'BC' size field), deflated on `threads` workers; the concatenation is one valid gzip file
 * that block-aware readers (nb_report_tsv, bgzip tools) can also inflate in parallel. out_cap >= n + n/512 + 64 per MiB + 64. */
int32_t nb_gzip_members(const char* in, int64_t n, int32_t level, int32_t threads, char* out, int64_t out_cap, int64_t* out_len);

/* ---- `nimble report` reducer (host, no GPU needed): per-read TSV -> feature<TAB>count<TAB>cell_barcode.
 * Replaces /root/reference/nimble/__main__.py:254-302 + nimble/utils.py:119-224 for large tables; byte-identical output
 * (same golden fixtures as nimble_amd/report.py). status_out[0] = 1 if the empty output was written, status_out[1] =
 * number of UMIs dropped for an empty intersection. ---- */
int32_t nb_report_tsv(const char* input_path, const char* output_path, double threshold, int32_t disable_thresholding,
                      int64_t* status_out);
/* the same reduction from arrays, skipping the per-read TSV: row i = (cb_strings[cb_ids[i]], umi_keys[i] (any key that
 * identifies the UMI), feat_strings[feat_ids[i]] (features sorted and comma-joined), scores[i]); rows need not be
 * merged or ordered; empty cb / feature strings mark null rows. Same output text as nb_report_tsv on the equivalent TSV. */
int32_t nb_report_rows(int64_t n, const uint32_t* cb_ids, const uint64_t* umi_keys, const uint32_t* feat_ids, const int64_t* scores,
                       int64_t n_cb, const char* const* cb_strings, int64_t n_feat, const char* const* feat_strings, double threshold,
                       int32_t disable_thresholding, const char* output_path, int64_t* status_out);

/* kernels the LAST nb_align_batch on ctx launched, ';'-separated, dominant first (for matching rocprofv3 kernel-trace rows); the
 * string is owned by ctx and valid until its next nb_align_batch */
const char* nb_last_kernel_names(nb_ctx* ctx);
/* the wave-per-read kernel nb_align_batch would launch for this shape, guessed from the shape alone (the table kind is chosen by size) */
const char* nb_kernel_name(int32_t max_len, int32_t k, int32_t n_seqs, int32_t paired);

#ifdef __cplusplus
}
#endif
#endif /* NIMBLE_AMD_H */
