/*
 * taxor_oracle.h -- CPU ORACLE for the `taxor search` hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm (JensUweUlrich/Taxor @ 2025-05-23), written
 * from the reference's readable sources; every function cites the reference file:line it follows.  It is
 * the checker for the HIP path -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.  The product library (libtaxor_gpu.so) never links, loads or calls anything in oracle/.
 *
 * PINNING STATUS
 *   The reference has no tests, golden vectors or fixtures, and its search path cannot be
 *   built in this image: syncmer.cpp, hierarchical_interleaved_xor_filter.hpp, threshold.hpp and taxor_search.cpp
 *   include seqan3 / cereal / ankerl headers that are fetched by git at configure time and are absent here
 *   (src/seqan/CMakeLists.txt.in:7-62, src/hashing/CMakeLists.txt.in:6-15); building them against hand-written
 *   stand-ins would not be a reference build.
 *   PINNED against the reference's own code (oracle/_ref/libtaxor_ref.so, built by `make ref` straight from the
 *   reference's files that need nothing but the standard library; tests/test_oracle_ref.py):
 *     - orc_syncmer_match_ratio / orc_threshold        = src/hixf/search/syncmer_model.hpp, every (k, error rate)
 *     - orc_nmut_kmer_ci_high, orc_containment_index_ci_low, orc_normal_cdf_inverse (and so orc_threshold_model)
 *                                                      = kmer_model.cpp, fracminhash_model.cpp, gaussian_inverse.cpp,
 *                                                        including the NaN / negative casts of very short reads
 *     - orc_adjust_seed                                = src/hixf/build/adjust_seed.hpp
 *     - orc_ixf_probe / orc_ixf_seg_len / lookup rule  = src/main/xorfilter.hpp + hashutil.hpp, the in-repo XOR-filter
 *                                                        prototype (same rows, fingerprint, sizing; a filter it builds
 *                                                        answers identically through orc_ixf_bulk_count)
 *   "parity unpinned" at three boundaries that live in absent third-party code:
 *     (1) orc_wyhash_u64   -- restates the PUBLISHED algorithm of martinus/unordered_dense v3.0.1
 *         (`detail::wyhash::hash(uint64_t)` = mix(x, 0x9E3779B97F4A7C15), mix = lo64 ^ hi64 of the
 *         128-bit product); call site src/hashing/syncmer.cpp:73-77.                 parity unpinned
 *     (2) orc_ixf_* as the INTERLEAVED filter of the author's seqan3 fork (JensUweUlrich/seqan3@master, un-vendored,
 *         un-pinned): that its per-bin arithmetic is the prototype's (pinned above), that rows are interleaved as
 *         data[row*stride + bin] with stride = ceil(bins/64)*64, and how it serialises.  parity unpinned
 *     (3) orc_minimiser_hash -- restates seqan3::views::minimiser_hash (same un-vendored fork), used for
 *         indexes built without --use-syncmer; call sites src/main/taxor_search.cpp:210-212,241-256.
 *         Window > k: which of several equal minima is kept is recalled, not read.   parity unpinned
 *   Everything else (syncmer selector, HIXF traversal/tally, classification call, TSV) is restated from code
 *   that IS in /root/reference and is cross-checked against a second, independent pure-Python restatement
 *   (tests/golden/make_golden.py) whose outputs are committed under tests/golden/.
 */
#ifndef TAXOR_ORACLE_H
#define TAXOR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashing ---------------------------------------------------------------------------------- */

/* ankerl::unordered_dense::detail::wyhash::hash(uint64_t), v3.0.1; call site syncmer.cpp:73-77 */
uint64_t orc_wyhash_u64(uint64_t x);

/* seqan3::dna4 char->rank->char mapping applied by sequence_file_input<dna4_traits>
 * (src/hixf/build/dna4_traits.hpp:15-18, taxor_search.cpp:181-182) [RECALL of upstream seqan3 dna4:
 * IUPAC codes map to their first base, U->T, anything else -> A].  In-place; returns 0, or -1 if a
 * character is outside the dna15 legal alphabet (the reference reader throws there). */
int orc_dna4_normalise(char *seq, size_t len);

/* hashing::seq_to_syncmers (syncmer.cpp:80-165): distinct wyhash(canonical k-mer) of the open canonical
 * syncmers of `seq`, in first-insertion order.  Writes at most `cap` values to `out`; returns the number
 * of distinct hashes (may exceed cap; then out is truncated). */
size_t orc_seq_to_syncmers(const char *seq, size_t len, int k, int s, int t, uint64_t *out, size_t cap);

/* ---- thresholds ------------------------------------------------------------------------------- */

/* get_min_syncmer_match_ratio (syncmer_model.hpp:38-50).  Returns -1.0 where the reference's asserts
 * (compiled out in Release) would fire / it would read out of bounds. */
double orc_syncmer_match_ratio(size_t kmer_size, double error_rate);

/* threshold::get, syncmer and percentage branches (threshold.hpp:22-47,51-81).
 * percentage in (0,1] selects the percentage model exactly as the constructor does. */
size_t orc_threshold(size_t hash_count, size_t kmer_size, double error_rate, double percentage);

/* threshold::threshold kind selection (threshold.hpp:22-47) and threshold::get for every kind (:51-81), including
 * the k-mer model (kmer_model.cpp:10-23, gaussian_inverse.cpp:13-50) and the FracMinHash containment model
 * (fracminhash_model.cpp:8-33).  scaling_factor = hash_count / (read_len - k + 1) as at taxor_search.cpp:263. */
enum { ORC_THR_PERCENTAGE = 0, ORC_THR_SYNCMER = 1, ORC_THR_KMER = 2, ORC_THR_FRACMINHASH = 3 };
int orc_threshold_kind(int use_syncmer, size_t kmer_size, size_t window_size, double percentage);
size_t orc_threshold_model(int kind, size_t minimiser_count, size_t kmer_size, double error_rate, double percentage,
                           double scaling_factor);
double orc_normal_cdf_inverse(double p);
/* the two model components, exported so that tests can hold them against the reference's own translation units
 * (oracle/_ref, see ref_driver.cpp): calculate_nmut_kmer_CI(...).second and calculate_containment_index_CI(...).first */
size_t orc_nmut_kmer_ci_high(double r, size_t kmer_size, size_t kmer_count, double confidence);
double orc_containment_index_ci_low(double r, size_t kmer_size, size_t kmer_count, double scaling_factor, double confidence);

/* hixf::adjust_seed (adjust_seed.hpp:40-44) and seqan3::views::minimiser_hash as the reference calls it
 * (taxor_search.cpp:210-212): values in emission order, duplicates kept.  See the .c for the pinning status. */
uint64_t orc_adjust_seed(int k);
size_t orc_minimiser_hash(const char *seq, size_t len, int k, int w, uint64_t *out, size_t cap);

/* ---- interleaved XOR filter (un-vendored; see header comment) ---------------------------------- */

typedef struct {
    uint64_t bins;        /* user-visible bin count (= counting_vector size)            */
    uint64_t stride;      /* bytes per fingerprint row = ceil(bins/64)*64               */
    uint64_t seg_len;     /* rows per segment; rows = 3*seg_len                         */
    uint64_t seed;        /* per-IXF hash seed (construct_ixf.cpp:100-108 may redraw it) */
    const uint8_t *data;  /* rows*stride fingerprints, data[row*stride + bin]           */
    uint32_t arith;       /* 0 = the reading below (xorfilter.hpp / hashutil.hpp).  The IXF arithmetic is un-vendored in the
                             reference, so a published index may follow another reading; the checker is parametrised the same
                             way as the product (taxor_amd/csrc/ixf_arith.h, its own code): bits 0-1 key hash (0 murmur64
                             finaliser, 1 none, 2 wyhash mix, 3 splitmix64), 2-3 seed entry (0 h(key+seed), 1 h(key^seed),
                             2 h(key)+seed, 3 unused), 4-5 range reduction (0 (u32)rot*seg>>32, 1 (u32)rot%seg, 2 mulhi64),
                             6-7 fingerprint (0 (u8)(h^h>>32), 1 (u8)h, 2 (u8)(h>>56), 3 (u8)(h>>32)), 8-15 rotation step ^ 21 */
} orc_ixf;

/* seg_len for a filter built for `max_bin_elements` keys per bin: (32 + 1.23*n)/3 (xorfilter.hpp:67-68) */
uint64_t orc_ixf_seg_len(uint64_t max_bin_elements);

/* key -> rows h0,h1,h2 and 8-bit fingerprint (xorfilter.hpp:36-45,60-62,338-347; hashutil.hpp:50-61) */
void orc_ixf_probe(const orc_ixf *f, uint64_t key, uint64_t rows[3], uint8_t *fp);

/* counting_agent<uint32_t>::bulk_count (call site hierarchical_interleaved_xor_filter.hpp:307-309):
 * counts[j] = #{hashes : fp == D[h0][j]^D[h1][j]^D[h2][j]}, j < bins */
void orc_ixf_bulk_count(const orc_ixf *f, const uint64_t *hashes, size_t n, uint32_t *counts);

/* Checker of built filters: the synthetic keys first .. first + n - 1 (key i = splitmix64 finaliser of i + salt) looked up in
 * column `bin` of f with bulk_count's rule -> number found (n = none missing); every sample_step-th key also through
 * orc_ixf_bulk_count over all bins (counts[bins] accumulates, *sampled += keys that did).  OpenMP over the keys, `threads` of them. */
void orc_synth_keys(uint64_t first, uint64_t n, uint64_t salt, uint64_t *out);
uint64_t orc_ixf_synth_keys_found(const orc_ixf *f, uint64_t bin, uint64_t first, uint64_t n, uint64_t salt, uint64_t sample_step,
                                  uint64_t *counts, uint64_t *sampled, int threads);

/* ---- hierarchical IXF -------------------------------------------------------------------------- */

typedef struct {
    size_t n_ixf;
    const orc_ixf *ixf;
    const int64_t *const *next_ixf;   /* next_ixf_id[i][bin]                (hixf.hpp:115-122) */
    const int64_t *const *fname_idx;  /* ixf_bin_to_filename_position[i][bin], -1 = merged (:172-178) */
} orc_hixf;

/* membership_agent::bulk_contains (hixf.hpp:303-340,381-406): (user_bin, count) tuples in DFS order.
 * Returns the number of tuples (may exceed cap; output truncated).  *visited_bytes (optional) receives
 * sum over visited IXFs of n*3*bins -- the algorithmic gather bytes of SURVEY.md section 8(d). */
size_t orc_bulk_contains(const orc_hixf *h, const uint64_t *hashes, size_t n, size_t threshold,
                         int64_t *user_bin, uint32_t *count, size_t cap, uint64_t *visited_bytes);

/* ---- per-read driver (taxor_search.cpp:196-313) -------------------------------------------------- */

typedef struct {
    int k, s, t;
    double error_rate;   /* --error-rate, default 0.04 (taxor_search_configuration.hpp:16) */
    double percentage;   /* --percentage, default -1.0                                    */
    int scaling;         /* index scaling (FracMinHash down-sampling), 1 = off            */
    int window;          /* 0: syncmer index (use_syncmer); >= k: index built without --use-syncmer, this window size */
} orc_search_params;

/* One read: dna4-normalised ASCII in, tuples (before the 0.8*max filter) out. Returns #tuples. */
size_t orc_search_read(const orc_hixf *h, const orc_search_params *p, const char *seq, size_t len,
                       uint32_t *n_hashes, int64_t *user_bin, uint32_t *count, size_t cap,
                       uint64_t *visited_bytes);

/* Batch driver with the reference's threading shape (do_parallel.hpp:17-36: `threads` contiguous
 * slices).  offsets has n_reads+1 entries into `bases`.  Results are written CSR-style:
 * out_off[n_reads+1], tuples of read r at [out_off[r], out_off[r+1]).  cap = capacity of the tuple
 * arrays; returns 0 on success, -1 if cap was too small (out_off still holds the needed sizes). */
int orc_search_batch(const orc_hixf *h, const orc_search_params *p, const char *bases,
                     const uint64_t *offsets, uint64_t n_reads, int threads, uint32_t *n_hashes,
                     uint64_t *out_off, int64_t *user_bin, uint32_t *count, uint64_t cap,
                     uint64_t *visited_bytes);

/* The same batch with the scheduler left to the caller: begin, then orc_batch_worker(ctx, start, end) from any number of
 * threads over slices that tile [0, n_reads) exactly once (the worker lambda of taxor_search.cpp:196-313 over one slice), then
 * finish (assembles the CSR in read order and frees the context; -1 = cap too small, -2 = the slices did not tile the batch).
 * oracle/ref_driver.cpp drives it with the reference's own hixf::do_parallel over 1024-record chunks. */
typedef struct orc_batch_ctx orc_batch_ctx;
orc_batch_ctx *orc_batch_begin(const orc_hixf *h, const orc_search_params *p, const char *bases, const uint64_t *offsets,
                               uint64_t n_reads, uint32_t *n_hashes);
void orc_batch_worker(orc_batch_ctx *c, uint64_t start, uint64_t end);
int orc_batch_finish(orc_batch_ctx *c, uint64_t *out_off, int64_t *user_bin, uint32_t *count, uint64_t cap, uint64_t *visited_bytes);

/* Classification call (taxor_search.cpp:268-306): keep[i]=1 iff double(cnt) >= double(max)*0.8 */
void orc_classify_filter(const uint32_t *count, size_t n, uint8_t *keep);

#ifdef __cplusplus
}
#endif
#endif
