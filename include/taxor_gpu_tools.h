/*
 * taxor_gpu_tools.h -- everything libtaxor_gpu.so exports BESIDE the drop-in seam (taxor_gpu.h): the batch call split into its
 * phases for callers that keep a batch resident, run statistics and measurement aids, the stage entry points
 * the parity tests check one by one against the oracle, index construction on the device (8(f) #3), diagnosis of indexes this
 * library did not write (record schema probe, arithmetic / layout variant scan; 8(f) #2), the synthetic workload generator and
 * the device side of the .gz reader.  Same conventions as taxor_gpu.h; a binding of the reference needs none of this.
 */
#ifndef TAXOR_GPU_TOOLS_H
#define TAXOR_GPU_TOOLS_H

#include "taxor_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- more about a resident index */
/* current hash seed of one IXF (construction on the device may have redrawn it) */
uint64_t taxor_gpu_index_ixf_seed(const taxor_gpu_index *idx, uint64_t ixf);
/* number of leaf runs (= tuples a threshold-0 read produces) and IXF tree depth */
uint64_t taxor_gpu_index_leaf_runs(const taxor_gpu_index *idx);
uint32_t taxor_gpu_index_depth(const taxor_gpu_index *idx);
/* Measurement aid (SURVEY.md 8(d) "measured gather ceiling"): read about want_bytes of IXF `ixf` as whole rows at
 * random row indices with the access shape of the query kernel's dense phase (16 B per lane, neighbouring lanes on
 * one row) and nothing else, `reps` times; reports the requested-bytes rate and the bytes read per row. */
int taxor_gpu_gather_ceiling(taxor_gpu_index *idx, uint64_t ixf, uint64_t want_bytes, int reps, double *gb_per_s,
                             uint64_t *row_bytes);
/* the same over up to n_ixf consecutive, equally shaped IXFs starting at `ixf` (e.g. all children of a synthetic index:
 * one 128-bin IXF of 68 MB sits in the caches, a thousand of them do not); *span_used = how many were covered */
int taxor_gpu_gather_ceiling_span(taxor_gpu_index *idx, uint64_t ixf, uint64_t n_ixf, uint64_t want_bytes, int reps,
                                  double *gb_per_s, uint64_t *row_bytes, uint64_t *span_used);
/* Calibration aid for the traffic counter (rocprofv3 --pmc FETCH_SIZE is calibrated for wide coalesced reads only):
 * launches with a KNOWN request count in the two access shapes of the query kernel, nothing else.  pattern 0 = whole rows
 * at random row indices (dense phase), pattern 1 = one 16-B load per lane, every lane on a row of its own (sparse phase);
 * nt = non-temporal loads.  One warm-up launch plus `reps` timed ones, all of the same size; reports the requested-bytes
 * rate and, per launch, the requested bytes (pattern 0: rows x row bytes; pattern 1: loads x 16) and the request count
 * (rows / loads). */
int taxor_gpu_gather_pattern(taxor_gpu_index *idx, uint64_t ixf, int pattern, int nt, uint64_t want_bytes, int reps,
                             double *gb_per_s, uint64_t *bytes_per_launch, uint64_t *requests_per_launch);
/* Index construction helpers for synthetic / planted indexes (what a GPU builder would use):
 * fill one IXF with seeded pseudo-random fingerprints (behaves like non-matching bins, FPR 2^-8),
 * overwrite one bin column (rows = 3*seg_len bytes), read an IXF back (to hand the same bytes to a
 * checker). */
int taxor_gpu_index_fill_random(taxor_gpu_index *idx, uint64_t ixf, uint64_t seed);
int taxor_gpu_index_upload_bin(taxor_gpu_index *idx, uint64_t ixf, uint64_t bin, const uint8_t *column,
                               uint64_t rows);
int taxor_gpu_index_download_ixf(const taxor_gpu_index *idx, uint64_t ixf, uint8_t *data, uint64_t len);
/* GPU construction of the fingerprint columns of one IXF, in place (the reference builds on
 * the CPU: src/hixf/build/construct_ixf.cpp:50-165, add_bin_elements + reseed loop).  keys = the bins' key lists
 * concatenated (distinct within a bin), key_off[bins+1]; bins without keys keep their content.  All bins are peeled
 * at once in synchronous rounds (taxor_amd/csrc/builder.hip); if a bin does not peel the IXF is re-seeded and rebuilt,
 * like the reference.  On success the IXF carries *seed_out (also written into the resident index); *rounds_out =
 * peeling rounds of the slowest chunk.  The columns are a function of (keys, seed) alone: two builds are byte-identical.
 * Every key is looked up in the finished columns before the call returns.  The index must not be searched meanwhile.  The
 * builder's scratch (peeling state of one chunk, <= 3 GB unless one bin needs more; the union table; mark bytes) stays with the
 * index for its next build and is released by taxor_gpu_index_destroy. */
int taxor_gpu_index_build_ixf(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, const uint64_t *key_off,
                              uint64_t seed0, uint64_t *seed_out, uint32_t *rounds_out);
/* The whole hierarchy at once (the back end of hierarchical_build.cpp:27-236): key_off[total_bins + 1] indexes `keys`
 * per technical bin in the index's bin order (all bins of IXF 0, then IXF 1, ...); LEAF bins bring their keys
 * (distinct within a bin; a split user bin brings one part per technical bin), MERGED bins bring none -- their key set
 * is the union of everything in their child IXF, computed on the device (a hash set in HBM), level by level from the leaves
 * up; the IXFs of one level share peeling chunks.  An IXF that does not peel is redone under a redrawn seed, it alone.
 * Unions are limited to 2^32 keys. */
int taxor_gpu_index_build_hixf(taxor_gpu_index *idx, const uint64_t *keys, const uint64_t *key_off, uint64_t seed0,
                               uint32_t *rounds_out);
/* The same two with the keys already ON THE INDEX'S DEVICE (keys_on_device != 0: `keys` is a device pointer; key_off
 * stays a host array) and with the run's figures. */
typedef struct taxor_build_stats {
    uint64_t keys_inserted;   /* key insertions done (a key below a merged bin counts once per level it is inserted at) */
    uint64_t scratch_bytes;   /* peeling scratch at its largest */
    uint32_t rounds_max;      /* peeling rounds of the slowest chunk */
    uint32_t reseeds;         /* IXFs redone under a new seed */
    uint32_t chunks;          /* peeling chunks */
    uint32_t reserved;
    double seconds_peel;      /* count + seed scan + rounds */
    double seconds_assign;    /* clearing, assignment in reverse, verification */
    double seconds_union;     /* duplicate-free unions of the merged bins' key sets */
    double seconds_total;     /* from the call to the last IXF built and verified (key upload and scratch allocation inside) */
    double seconds_release;   /* handing keys, unions and scratch back to the driver afterwards (not in seconds_total) */
    double seconds_count;     /* GPU time of k_count (3 atomic adds per key), HIP events on the builder's stream */
    double seconds_rounds;    /* GPU time of the seed scan and the peeling rounds (2 atomic subs per key), HIP events */
    double seconds_upload;    /* keys from host memory to the device (allocation + copy; 0 when they were there already); in seconds_total */
    double seconds_alloc;     /* hipMalloc / hipFree of the peeling scratch (in seconds_total; the driver's time, erratic for GB-sized blocks) */
    uint64_t keys_counted_in_lds; /* of keys_inserted: keys whose bin's degree words were built in LDS (no global atomic adds for them) */
} taxor_build_stats;
int taxor_gpu_index_build_ixf_ex(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, int keys_on_device,
                                 const uint64_t *key_off, uint64_t seed0, uint64_t *seed_out, taxor_build_stats *stats);
int taxor_gpu_index_build_hixf_ex(taxor_gpu_index *idx, const uint64_t *keys, int keys_on_device, const uint64_t *key_off,
                                  uint64_t seed0, taxor_build_stats *stats);
/* The same with GENERATED keys for some leaf bins: bin g (index bin order) holds gen_count[g] keys synth_key(gen_first[g] + k, gen_salt),
 * k < gen_count[g], instead of keys from `keys` (its key_off range is then empty).  Generated keys need no memory -- the kernels
 * compute them -- and a merged bin above an IXF of generated bins with consecutive index ranges is itself such a range, so an
 * index far larger than its keys would be (a 113-GB GTDB-class index has 7e10 of them: 560 GB) can be built with EVERY bin a
 * real filter.  Bins with real keys (planted genomes) and generated decoys mix freely. */
int taxor_gpu_index_build_hixf_gen(taxor_gpu_index *idx, const uint64_t *keys, int keys_on_device, const uint64_t *key_off,
                                   const uint64_t *gen_first, const uint64_t *gen_count, uint64_t gen_salt, uint64_t seed0,
                                   taxor_build_stats *stats);
/* Synthetic key sets for the build bench and tests: key i = a bijection of (i + salt) (distinct without a table);
 * taxor_gpu_synth_keys writes keys first .. first + n - 1 to the DEVICE array d_out, taxor_synth_key is the same
 * function on the host.  taxor_gpu_malloc / _free / _memcpy_to_host / _from_host: plain device memory for such arrays. */
uint64_t taxor_synth_key(uint64_t i, uint64_t salt);
int taxor_gpu_synth_keys(int device, uint64_t *d_out, uint64_t first, uint64_t n, uint64_t salt);
int taxor_gpu_malloc(int device, uint64_t bytes, void **out);
void taxor_gpu_free(void *p);
int taxor_gpu_memcpy_to_host(void *dst, const void *d_src, uint64_t bytes);
int taxor_gpu_memcpy_from_host(void *d_dst, const void *src, uint64_t bytes);

/* ---- taxor_gpu_search_batch split into its three phases so that a caller can keep a batch resident in HBM
 * (upload once, run many times) and overlap transfers with compute:
 *   upload : H2D of the ASCII bases + on-device dna4 mapping and 2-bit packing
 *   run    : all kernels (syncmers -> dedup -> threshold -> level-synchronous HIXF query -> DFS order),
 *            asynchronous on the searcher's stream
 *   fetch  : wait + D2H of the CSR results */
int taxor_gpu_batch_upload(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets,
                           uint64_t n_reads);
int taxor_gpu_batch_run(taxor_gpu_searcher *s);
int taxor_gpu_batch_sync(taxor_gpu_searcher *s);
int taxor_gpu_batch_fetch(taxor_gpu_searcher *s, taxor_gpu_results *out);
/* Device-resident results of the last run (for an RCCL gather): sizes, then D2D copy into caller-provided
 * DEVICE buffers (read_off u64[n_reads+1], user_bin i64[n_tuples], count u32[n_tuples], n_hashes
 * u32[n_reads]); any pointer may be NULL to skip it.  Synchronises the searcher's stream. */
int taxor_gpu_batch_result_sizes(taxor_gpu_searcher *s, uint64_t *n_reads, uint64_t *n_tuples);
int taxor_gpu_batch_export_device(taxor_gpu_searcher *s, void *d_read_off, void *d_user_bin, void *d_count,
                                  void *d_n_hashes);


/* ---- communicator statistics and the one-GPU test hook */
typedef struct {
    int32_t transport;
    uint32_t n_devices;
    uint64_t index_bytes;            /* fingerprint bytes of one replica                                  */
    uint64_t index_upload_bytes;     /* bytes that crossed PCIe host -> device for the replicas           */
    uint64_t index_broadcast_bytes;  /* bytes delivered device -> device by ncclBroadcast                 */
    double index_seconds;            /* wall time of taxor_gpu_index_create_replicated                    */
    uint64_t gathers, gather_bytes;  /* gather calls; result bytes that left a peer device                */
    double gather_seconds;           /* wall time inside taxor_gpu_gather_results (sync of the runs included) */
    uint64_t index_broadcast_calls;  /* grouped ncclBroadcast rounds issued behind the upload (RCCL transport)  */
    uint64_t self_exchange_bytes;    /* result bytes rank 0 sent to itself through ncclSend/ncclRecv (test hook below) */
    int32_t rccl_version;            /* ncclGetVersion of the RCCL bound at run time, 0 = none loaded           */
    uint64_t selftest_bytes;         /* known bytes verified through ncclBroadcast + ncclSend/ncclRecv at creation (RCCL) */
} taxor_gpu_comm_stats;
int taxor_gpu_comm_info(const taxor_gpu_comm *c, taxor_gpu_comm_stats *out);
/* Test hook for boxes with ONE GPU: with on != 0, rank 0's own part of every gather travels through the grouped
 * ncclSend / ncclRecv (to itself) like a peer's instead of a device-to-device copy, so a communicator of one rank executes
 * the exchange code of a larger run line by line.  Results are unchanged.  RCCL transport only. */
int taxor_gpu_comm_set_self_exchange(taxor_gpu_comm *c, int on);

/* Measurement of the last taxor_gpu_batch_run (valid after sync).  algorithmic_bytes follows SURVEY.md
 * section 8(d): sum over reads of ceil(L/4) + sum over visited IXFs n_h*3*bins + 8 + 12*tuples;
 * query_* are the dominant kernel (k_query_level) only: launches, HIP-event milliseconds on the searcher's
 * stream (0 unless time_kernels), and its gather bytes sum n_h*3*bins. */
typedef struct {
    uint64_t n_reads, n_bases, n_hashes, n_tuples, n_work_items;
    uint64_t algorithmic_bytes;
    uint64_t query_bytes;
    uint64_t query_touched_bytes; /* bytes k_query_level actually requested: threshold-aware pruning skips row
                                     segments of bin runs that provably cannot reach the threshold       */
    uint32_t query_launches;
    float query_ms;
    float syncmer_ms;
    float finalize_ms;
    float total_ms;
    /* k_query_level per HIXF level (level 7 collects everything deeper): HIP-event milliseconds, requested bytes, and
     * fingerprint-row reads (levels of rows <= 128 B are bound by DRAM row activations, not by bytes) */
    float level_ms[8];
    uint64_t level_requested_bytes[8];
    uint64_t level_row_reads[8];
    uint64_t level_sparse_loads[8];  /* of level_row_reads: 16-B loads of the pruned (sparse) phase, one fingerprint row each;
                                        level_requested_bytes bills each as one 64-B sector */
    uint32_t tree_stalls_recovered;  /* pieces of a small call whose one-launch traversal gave up waiting (its watchdog fired) and
                                        were classified again level by level; results are unaffected */
} taxor_gpu_run_stats;
int taxor_gpu_batch_stats(taxor_gpu_searcher *s, taxor_gpu_run_stats *out);
/* Measurement aid: a searcher created while TAXOR_PROFILE_PHASES=1 is set launches instrumented instantiations of the
 * two big kernels (s_memtime marks at their phase boundaries, summed over blocks).  Returns and clears 16 cycle sums:
 * [0..7] k_syncmers (cursor, staging, s-mer values, window argmins, selection, hash emit, dedup, copy-out),
 * [8..15] k_query_level (cursor+flush, metadata+probe staging, dense gathers, prune check, sparse gathers, tally,
 * final flush, -).  Results are unchanged; throughput is not (the marks cost a few percent). */
int taxor_gpu_phase_profile(taxor_gpu_searcher *s, uint64_t *cycles16);

/* ------------------------------------------------------------------------------------------------
 * Stage entry points (used by the parity tests; each stage is checked on its own against the oracle).
 * ---------------------------------------------------------------------------------------------- */
/* hashing::seq_to_syncmers (src/hashing/syncmer.hpp:23) for a batch: distinct hashes of read r, in first-
 * insertion order, at hashes[hash_off[r] .. hash_off[r+1]) -- after the FracMinHash filter of
 * taxor_search.cpp:223-233 when the index has scaling > 1.  Pointers valid until the next call. */
int taxor_gpu_syncmers(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                       const uint64_t **hash_off, const uint64_t **hashes);
/* ixf.counting_agent<uint32_t>().bulk_count(values) for one IXF of the index
 * (call site hierarchical_interleaved_xor_filter.hpp:307-309): counts[bins]. */
int taxor_gpu_ixf_bulk_count(taxor_gpu_searcher *s, uint64_t ixf, const uint64_t *hashes, uint64_t n,
                             uint32_t *counts);
/* membership_agent::bulk_contains(values, threshold) (:381-406) for one hash list. */
int taxor_gpu_bulk_contains(taxor_gpu_searcher *s, const uint64_t *hashes, uint64_t n, uint64_t threshold,
                            taxor_gpu_results *out);


/* ---- Diagnosis of an index this library did not write (`taxor verify --variants`, `taxor pin`).  Both the
 * arithmetic of seqan3::interleaved_xor_filter and the way its serialiser lays the fingerprints out are un-vendored in the
 * reference; taxor_amd/csrc/ixf_arith.h holds this library's reading of the former (evidence: src/main/xorfilter.hpp:36-45,
 * 60-68,338-350, src/main/hashutil.hpp:50-61), taxor_amd/csrc/ixf_layout.h the layouts a file may follow.  A variant is one
 * reading of the same RAW bytes; the scan probes one IXF's raw bytes under every variant with hash lists cut from sequences that
 * are in the index and reports, per (variant, list), the best-bin match ratio: ~1.0 under the file's true reading, ~2^-8
 * otherwise.  best_ratio[n_variants * n_lists], variant-major. */
typedef struct {
    uint64_t seed;
    uint64_t seg_len;   /* rows per hash segment */
    uint64_t stride;    /* the source's row pitch in bytes (row-interleaved) / bin columns stored (bin-major); unused for bit-sliced */
    uint8_t key_hash;   /* 0 murmur64 finaliser (hashutil.hpp:50-57), 1 none, 2 wyhash mix, 3 splitmix64 finaliser */
    uint8_t seed_mode;  /* 0 h(key + seed) (hashutil.hpp:59-61), 1 h(key ^ seed), 2 h(key) + seed, 3 seed unused */
    uint8_t rot;        /* row i uses rotl64(h, rot * i); 21 in xorfilter.hpp:42-45 */
    uint8_t reduce;     /* 0 ((u32)rot * seg_len) >> 32 (xorfilter.hpp:36-40), 1 (u32)rot % seg_len, 2 mulhi64(rot, seg_len) */
    uint8_t fp_mode;    /* 0 (u8)(h ^ h>>32) (xorfilter.hpp:60-62), 1 (u8)h, 2 (u8)(h>>56), 3 (u8)(h>>32) */
    uint8_t pad;
    uint16_t layout;    /* layout code (taxor_hixf_view::ixf_layout): kind and row order; the pitch bits say how `stride` was
                           derived (0 bins padded to 64, 1 exactly bins, 2 the record's stored scalar) */
} taxor_ixf_variant;
/* The arithmetic part of a variant (key hash, seed entry, rotation step, range reduction, fingerprint fold) as the code an index
 * carries (taxor_hixf_view::ixf_arith); 0 for this library's reading.  _decode fills those five fields and leaves the others. */
uint32_t taxor_ixf_arith_code(const taxor_ixf_variant *v);
void taxor_ixf_arith_decode(uint32_t code, taxor_ixf_variant *out);
/* this library's reading for the given seed / segment length / stride */
void taxor_ixf_variant_default(taxor_ixf_variant *out, uint64_t seed, uint64_t seg_len, uint64_t stride);
/* raw = the IXF's bytes as the file holds them (host memory, e.g. taxor_hixf_get_view()->ixf[i].data), raw_len of them */
int taxor_gpu_ixf_variant_scan(int device, const uint8_t *raw, uint64_t raw_len, uint64_t bins, const taxor_ixf_variant *variants,
                               uint32_t n_variants, const uint64_t *hashes, const uint64_t *hash_off, uint64_t n_lists, float *best_ratio);
/* one-line description of a variant; returns the length written */
uint64_t taxor_ixf_variant_describe(const taxor_ixf_variant *v, char *buf, uint64_t cap);
/* "bin-major,unpadded,position-major" <-> layout code; tokens: interleaved | bin-major | bit-sliced, padded | unpadded |
 * stored-pitch, segment-major | position-major; what is left out keeps the search layout's choice.  _parse returns 0 or TAXOR_E_ARG */
int taxor_ixf_layout_parse(const char *spec, uint32_t *code);
uint64_t taxor_ixf_layout_describe(uint32_t code, char *buf, uint64_t cap);

/* ---- .hixf: the record of one seqan3::interleaved_xor_filter inside the file (UN-VENDORED in the reference; this library's own
 * is documented in taxor_amd/csrc/hixf_io.cpp): n_before u64 scalars, the fingerprint vector (u64 length + bytes), n_after u64
 * scalars.  idx_* select the scalar (counted over before-then-after) that holds a field, -1 = not stored: bins then come from
 * next_ixf_id's inner sizes, the pitch from the layout's rule, seg_len = rows / 3, seed = default_seed. */
typedef struct {
    uint32_t n_before, n_after;
    int32_t idx_bins, idx_stride, idx_seg_len, idx_seed;
    uint32_t seg_len_is_rows;   /* 1: the idx_seg_len scalar holds rows = 3*seg_len */
    uint64_t default_seed;      /* 13572355802537770549 = the fixed start seed of src/main/xorfilter.hpp:153 */
    uint32_t layout;            /* how the fingerprint vector is laid out (taxor_hixf_view::ixf_layout); 0 = the search layout */
    uint32_t len_unit;          /* what the vector's u64 length word counts: 0 / 1 bytes (cereal's std::vector<uint8_t>), 8 = 64-bit words
                                   (std::vector<uint64_t>), 64 = BITS held in whole 64-bit words (an sdsl int_vector / bit_vector) */
    uint32_t skip_before_len, skip_after_len; /* bytes between the scalars and the length word / between it and the data (an int_vector's
                                   u8 width); the scalars themselves are u64 */
} taxor_ixf_schema;
/* this library's own schema: bins | technical_bins | seg_len | bin_words | seed | ftype | data, layout 0 */
void taxor_ixf_schema_default(taxor_ixf_schema *out);
/* `hixf-probe`: walk a real file with every (n_before, n_after) until the records re-parse n times and the pinned tail
 * (next_ixf_id, user_bins) lands exactly on end-of-file, then infer which scalar is which and which layouts the array lengths
 * admit (the layout itself is decided by the variant scan).  Writes a human-readable report (NUL-terminated, truncated to cap). */
int taxor_hixf_probe(const char *path, taxor_ixf_schema *out, char *report, uint64_t cap);
int taxor_hixf_load_schema(const char *path, const taxor_ixf_schema *schema, taxor_hixf **out);
/* writes view's IXFs (host bytes in the SEARCH layout, view->ixf_layout == 0) under schema->layout: what another writer's file
 * would look like (tests of the re-layout; export) */
int taxor_hixf_store_schema(const char *path, const taxor_hixf_view *view, const taxor_hixf_meta *meta, const taxor_ixf_schema *schema);
/* bytes of IXF i's fingerprint vector as the file holds it */
uint64_t taxor_hixf_ixf_raw_bytes(const taxor_hixf *h, uint64_t ixf);

/* ---- host-side XOR-filter construction and the synthetic workload */
/* seg_len of an IXF sized for max_bin_elements keys per bin: (size_t)(32 + 1.23*n) / 3 */
uint64_t taxor_ixf_seg_len(uint64_t max_bin_elements);
/* XOR-filter construction of one bin column (3*seg_len bytes) for `keys` under (seed, seg_len); returns 0,
 * or 1 if peeling failed for this seed (caller redraws the seed like construct_ixf.cpp:100-108). */
int taxor_ixf_build_bin(const uint64_t *keys, uint64_t n, uint64_t seed, uint64_t seg_len, uint8_t *column);
/* the same under another arithmetic code (taxor_ixf_arith_code) */
int taxor_ixf_build_bin_arith(const uint64_t *keys, uint64_t n, uint64_t seed, uint64_t seg_len, uint32_t arith, uint8_t *column);
/* Seeded synthetic long reads: read i is drawn from genome g_i at a uniform start
 * (reverse-complemented with probability frac_reverse) with ONT-like errors at rate e (40/30/30
 * sub/ins/del), or uniformly random with probability frac_random.  Note: with the reference's
 * t = ceil((k-s+1)/2) in INTEGER division (taxor_build.cpp:509-510; 5 at k22/s12) open-syncmer selection
 * is not strand-symmetric, so a reverse-strand read shares no syncmers with a forward-indexed genome.  genomes = concatenated ACGT, genome_off[n_genomes+1].  Writes ASCII into bases (capacity
 * cap) and offsets[n_reads+1]; origin[i] = genome index or -1.  Deterministic in (seed, i). */
int taxor_synth_reads(const char *genomes, const uint64_t *genome_off, uint64_t n_genomes, uint64_t n_reads,
                      uint32_t read_len, double error_rate, double frac_random, double frac_reverse,
                      uint64_t seed, int threads, char *bases, uint64_t cap, uint64_t *offsets, int32_t *origin);

#ifdef __cplusplus
}
#endif
#endif
