/*
 * taxor_gpu.h -- C ABI of the MI355X-native `taxor search` hot path (libtaxor_gpu.so).
 *
 * The reference (JensUweUlrich/Taxor @ 2025-05-23) has no FFI; its seam for this path is the chunk loop
 * of search_single():  hixf::do_parallel(worker, records.size(), threads, compute_time)
 * (src/main/taxor_search.cpp:315-326, worker = :196-313).  A batch of (id, seq) records goes in, per-read
 * (user_bin, count) tuples come out.  This header replaces exactly that seam; every entry point cites
 * the reference interface it stands in for.  Plain pointers and sizes only, no C++/torch types, no
 * exceptions: every function returns 0 on success or a negative taxor_status, and
 * taxor_gpu_last_error() returns the message of the calling thread's last failure (the reference prints "[TAXOR SEARCH ERROR] ..." and
 * returns -1, taxor_search.cpp:380-384).
 *
 * Threading: an index is immutable after creation and may be shared; a searcher is single-caller, like
 * the reference's membership_agent (hierarchical_interleaved_xor_filter.hpp:371-379).  Result pointers
 * returned by a searcher stay valid until the next call on that searcher (same convention as
 * bulk_contains' reference return, :381-406).
 */
#ifndef TAXOR_GPU_H
#define TAXOR_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TAXOR_OK = 0,
    TAXOR_E_ARG = -1,       /* bad argument / unsupported parameter                          */
    TAXOR_E_HIP = -2,       /* HIP runtime failure                                           */
    TAXOR_E_ALPHABET = -3,  /* read character outside the dna15 alphabet                     */
    TAXOR_E_INTERNAL = -4,  /* internal capacity invariant violated (never silent)           */
    TAXOR_E_IO = -5,        /* file could not be read/written or is inconsistent             */
    TAXOR_E_NOMEM = -6
} taxor_status;

const char *taxor_gpu_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Index hand-off.  Replaces: taxor_index<hixf_t> loaded by cereal (src/main/load_index.hpp:27-38) and
 * held by pointer in the agents (hierarchical_interleaved_xor_filter.hpp:300).
 * The view is what a loader parsed from a .hixf: per IXF the interleaved fingerprint array plus the two
 * bookkeeping vectors next_ixf_id[i] (:115-122) and ixf_bin_to_filename_position[i] (:172-178).
 * The library copies everything into HBM; the caller keeps ownership of the host arrays.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t bins;            /* bin count == counting_vector size                                  */
    uint64_t stride;          /* bytes per fingerprint row, multiple of 64, >= bins                 */
    uint64_t seg_len;         /* rows per hash segment; the array has 3*seg_len rows                */
    uint64_t seed;            /* per-IXF seed (src/hixf/build/construct_ixf.cpp:100-108 redraws it) */
    const uint8_t *data;      /* host, 3*seg_len*stride bytes, data[row*stride+bin]; NULL = leave
                                 the device rows uninitialised (use ..._fill_random / _upload_bin)  */
    const int64_t *next_ixf;  /* [bins]                                                             */
    const int64_t *fname_idx; /* [bins], -1 = merged bin                                            */
} taxor_ixf_view;

/* Fingerprint bytes that are NOT in host memory (a .hixf on disk or in tmpfs): the library pulls them piece by piece,
 * from several of its own threads at once, straight into page-locked staging buffers on their way to HBM -- the 113 GB
 * of a GTDB index then never get host page tables of their own (mapping the file and letting the runtime copy from the
 * mapping costs a page fault per 4 KiB on the way in and seconds of munmap on the way out).  read() copies `len` bytes of
 * IXF `ixf`'s array, starting at byte `offset` of it, to dst; returns 0 on success; must be thread-safe. */
typedef struct {
    int (*read)(void *ctx, uint64_t ixf, uint64_t offset, uint64_t len, void *dst);
    void *ctx;
} taxor_ixf_source;

typedef struct {
    uint64_t n_ixf;
    const taxor_ixf_view *ixf;
    uint64_t n_user_bins;
    uint8_t kmer_size, syncmer_size, t_syncmer; /* src/main/index.hpp:219-221 */
    uint8_t use_syncmer;                        /* :223; 1 = open canonical syncmers (syncmer.cpp:80-165); 0 = index built
                                                   without --use-syncmer: seqan3 minimiser_hash over window_size
                                                   (taxor_search.cpp:210-212,239-260), every emitted value counts */
    uint16_t scaling;                           /* :224; >1 = FracMinHash down-sampling of the hashes
                                                   (taxor_search.cpp:223-233,243-249), applied on the device */
    uint64_t window_size;                       /* :212; used when use_syncmer == 0: window_size == kmer_size selects
                                                   every canonical k-mer, larger windows select minimisers */
    uint32_t ixf_arith;                         /* 0 = this library's reading of seqan3::interleaved_xor_filter's un-vendored
                                                   arithmetic (taxor_amd/csrc/ixf_arith.h); otherwise the code of another
                                                   reading, taxor_ixf_arith_code(variant): chosen at run time, e.g. the one
                                                   `taxor verify --variants` found a foreign file to follow */
    const taxor_ixf_source *source;             /* NULL: the fingerprint bytes are at ixf[i].data.  Otherwise index creation
                                                   reads them through the source and ignores ixf[i].data (taxor_hixf_load
                                                   sets it to a pread() reader of the file) */
} taxor_hixf_view;

typedef struct taxor_gpu_index taxor_gpu_index;

int taxor_gpu_index_create(const taxor_hixf_view *view, int device, taxor_gpu_index **out);
void taxor_gpu_index_destroy(taxor_gpu_index *idx);
/* current hash seed of one IXF (construction on the device may have redrawn it) */
uint64_t taxor_gpu_index_ixf_seed(const taxor_gpu_index *idx, uint64_t ixf);
/* bytes of fingerprint data resident in HBM */
uint64_t taxor_gpu_index_data_bytes(const taxor_gpu_index *idx);
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
 * in parallel rounds; if a bin does not peel the IXF is re-seeded and rebuilt, like the reference.  On success the
 * IXF carries *seed_out (also written into the resident index); *rounds_out = peeling rounds of the slowest chunk. */
int taxor_gpu_index_build_ixf(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, const uint64_t *key_off,
                              uint64_t seed0, uint64_t *seed_out, uint32_t *rounds_out);
/* The whole hierarchy at once (the back end of hierarchical_build.cpp:27-236): key_off[total_bins + 1] indexes `keys`
 * per technical bin in the index's bin order (all bins of IXF 0, then IXF 1, ...); LEAF bins bring their keys
 * (distinct within a bin; a split user bin brings one part per technical bin), MERGED bins bring none -- their key set
 * is the union of everything in their child IXF, computed on the device (sort + unique), bottom-up.  Every IXF is
 * then constructed as by taxor_gpu_index_build_ixf (its seed may be redrawn).  Unions are limited to 2^32 keys. */
int taxor_gpu_index_build_hixf(taxor_gpu_index *idx, const uint64_t *keys, const uint64_t *key_off, uint64_t seed0,
                               uint32_t *rounds_out);

/* ------------------------------------------------------------------------------------------------
 * Searcher = one GPU-side "membership agent" + the per-read driver state.
 * Replaces: the worker lambda's locals (taxor_search.cpp:198-203) and hixf::threshold::threshold
 * (src/hixf/search/threshold.hpp:22-81).  For the syncmer and percentage kinds `ratio` is what threshold::get
 * multiplies the hash count by -- get_min_syncmer_match_ratio(k, error_rate) or --percentage
 * (taxor_threshold_ratio()).  For indexes built without --use-syncmer the kind is the k-mer model (window == k) or
 * the FracMinHash containment model (window > k): those thresholds are evaluated per read on the HOST, in the same
 * double expressions as the reference (taxor_threshold_model()), between the hashing and the query kernels.
 * taxor_threshold_select() fills `ratio`, `model` and `error_rate` the way threshold::threshold's constructor picks.
 * ---------------------------------------------------------------------------------------------- */
enum { TAXOR_THR_PERCENTAGE = 0, TAXOR_THR_SYNCMER = 1, TAXOR_THR_KMER = 2, TAXOR_THR_FRACMINHASH = 3 };

typedef struct {
    double ratio;             /* threshold = (size_t)(n_hashes * ratio), threshold.hpp:60,76-79 */
    uint32_t sub_batch_reads; /* reads per internal launch group (0 = default 32768)           */
    uint64_t sub_batch_bases; /* bases per internal launch group (0 = default 2^29)            */
    uint32_t time_kernels;    /* 1 = bracket the dominant kernel with HIP events               */
    uint32_t model;           /* TAXOR_THR_*; PERCENTAGE and SYNCMER use `ratio`               */
    double error_rate;        /* --error-rate, used by the KMER and FRACMINHASH models         */
    uint32_t flags;           /* TAXOR_SEARCH_* below; 0 = the defaults.  None changes a result      */
} taxor_gpu_search_params;

/* taxor_gpu_search_params::flags.  Choices a caller (a test, a benchmark) makes per searcher; the library reads nothing of
 * the kind from the environment unless TAXOR_TUNING=1 is set (taxor_amd/csrc/tuning.h). */
enum {
    TAXOR_SEARCH_NO_PRUNE = 1u,      /* count every hash against every bin of every visited IXF, the reference's formulation
                                        (hierarchical_interleaved_xor_filter.hpp:307-309) -- the default stops counting bin runs
                                        that provably cannot reach the threshold; tuples are identical either way */
    TAXOR_SEARCH_GROUP_ALWAYS = 2u,  /* group each level's work items by IXF whatever the sub-batch size (default: from 4096 reads) */
    TAXOR_SEARCH_NO_SMALL_PATH = 4u, /* calls of a few thousand reads take the level-synchronous pipeline of large batches */
    TAXOR_SEARCH_SPLIT_ALWAYS = 8u   /* root work items are split over column ranges (several blocks per read) whatever the batch size */
};

typedef struct taxor_gpu_searcher taxor_gpu_searcher;

int taxor_gpu_searcher_create(taxor_gpu_index *idx, const taxor_gpu_search_params *prm,
                              taxor_gpu_searcher **out);
void taxor_gpu_searcher_destroy(taxor_gpu_searcher *s);

/* Per-read results, CSR: tuples of read r are [read_off[r], read_off[r+1]) in DFS order of the HIXF
 * traversal (hierarchical_interleaved_xor_filter.hpp:313-338), BEFORE the 0.8*max filter of
 * taxor_search.cpp:275-286 (apply taxor_classify_filter or the host formatter).  n_hashes[r] is
 * QHASH_COUNT (taxor_search.cpp:261,298). */
typedef struct {
    uint64_t n_reads;
    uint64_t n_tuples;
    const uint64_t *read_off; /* [n_reads+1] */
    const int64_t *user_bin;  /* [n_tuples]  */
    const uint32_t *count;    /* [n_tuples]  */
    const uint32_t *n_hashes; /* [n_reads]   */
} taxor_gpu_results;

/* The drop-in batch call (host buffers in, host results out).  Replaces
 * hixf::do_parallel(worker, n, threads, compute_time) for one chunk of records:
 * bases = concatenated read sequences as read from FASTA/FASTQ (any dna15 character; the dna4 mapping of
 * src/hixf/build/dna4_traits.hpp:15-18 is applied on the device), offsets[n_reads+1] into bases. */
int taxor_gpu_search_batch(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets,
                           uint64_t n_reads, taxor_gpu_results *out);

/* The same call in two halves, for a host that wants to read / parse its next chunk while this one is classified
 * (the reference's chunk loop is synchronous, taxor_search.cpp:315-326): _begin enqueues everything and returns --
 * immediately when `bases` is registered memory (below), after the last host-to-device copy otherwise; `bases` and
 * `offsets` must stay valid until _end, which waits and hands out the results like taxor_gpu_search_batch. */
/* Exception: with the FRACMINHASH threshold model (minimiser indexes with window > k, or scaling > 1) _begin is NOT
 * asynchronous -- each sub-batch's minimiser counts come back to the host, the model is evaluated there in the
 * reference's double arithmetic, and the thresholds go up again before that sub-batch's query is enqueued, so _begin
 * returns only after the last sub-batch's hashing has finished (the GPU keeps classifying the previous sub-batch
 * meanwhile).  The KMER model (window == k) depends on the read length alone and is evaluated before anything is
 * enqueued. */
int taxor_gpu_search_batch_begin(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads);
int taxor_gpu_search_batch_end(taxor_gpu_searcher *s, taxor_gpu_results *out);

/* The same call over reads that live in SEVERAL host buffers (the reference's chunk is a vector of records, each with
 * its own sequence storage, taxor_search.cpp:319): one batch over the reads of all segments in segment order -- a host
 * whose parser threads each fill their own buffer hands the GPU a batch of useful size without copying them together.
 * Segment j holds n_reads reads, read i at bases[offsets[i] .. offsets[i+1]).  Everything must stay valid until
 * taxor_gpu_search_batch_end, which hands out one CSR over all reads (segment 0's first). */
typedef struct {
    const char *bases;
    const uint64_t *offsets;   /* [n_reads + 1] */
    uint64_t n_reads;
} taxor_read_segment;
int taxor_gpu_search_segments_begin(taxor_gpu_searcher *s, const taxor_read_segment *segs, uint64_t n_segs);

/* Optional: pin a host buffer that the caller passes to taxor_gpu_search_batch / taxor_gpu_batch_upload again and
 * again (a recycled staging buffer, like the reference's per-chunk `records` vector, taxor_search.cpp:319).  Copies
 * from registered memory are direct DMA; for pageable memory the runtime locks and unlocks the pages on every call
 * (~2.4 ms per 64 MB on MI355X).  The registration covers all devices; unregister before freeing or resizing. */
int taxor_gpu_host_register(void *ptr, uint64_t bytes);
int taxor_gpu_host_unregister(void *ptr);

/* The same call split into its three phases so that a caller can keep a batch resident in HBM
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

/* ------------------------------------------------------------------------------------------------
 * Several GPUs of one node, driven by ONE host process (taxor search --gpus N).  Reads are independent
 * (taxor_search.cpp:214): the index is replicated, every device classifies its own batches, and the path has two
 * exchange steps, both behind a communicator:
 *   taxor_gpu_index_create_replicated : one PCIe upload into devices[0], ncclBroadcast of the fingerprint slab to
 *                                       the others behind it (instead of N uploads of the same index);
 *   taxor_gpu_gather_results          : after a round in which searcher i classified its own batch on devices[i],
 *                                       the per-read results of all of them on devices[0] (grouped ncclSend /
 *                                       ncclRecv, every peer on its own xGMI link), handed out as ONE CSR in device
 *                                       order -- reads of searcher 0 first -- with offsets rebased.
 * Replaces: the reference has one address space; its workers write into one result stream under a mutex
 * (taxor_search.cpp:311, sync_out.hpp:24-29) and all read the one loaded index (taxor_search.cpp:323).
 * Transports: TAXOR_COMM_RCCL (RCCL bound at run time; one rank per device, a device may not repeat; creation sends known
 * bytes through a broadcast and a grouped send/recv between all ranks and fails if they arrive wrong) or
 * TAXOR_COMM_HOST (same calls, every transfer staged through host memory over each device's own PCIe link).  A
 * communicator never changes transport by itself: a failing RCCL call is an error return.
 * Single-caller: create / replicate / gather of one communicator are called from one host thread at a time.  A gather
 * waits for the runs of the searchers it is given (they may still be in flight when it is called) and touches nothing
 * else: OTHER searchers on the same devices -- a second set working on the next round, which is how `taxor search --gpus N`
 * overlaps one round's gather with the next round's kernels -- may keep running batches meanwhile.  Result pointers stay
 * valid until the next gather on the communicator.
 * ---------------------------------------------------------------------------------------------- */
enum { TAXOR_COMM_RCCL = 0, TAXOR_COMM_HOST = 1 };
typedef struct taxor_gpu_comm taxor_gpu_comm;
int taxor_gpu_comm_create(const int *devices, uint32_t n_devices, int transport, taxor_gpu_comm **out);
void taxor_gpu_comm_destroy(taxor_gpu_comm *c);
/* out[n_devices]: out[i] is the replica on devices[i]; each is destroyed with taxor_gpu_index_destroy */
int taxor_gpu_index_create_replicated(taxor_gpu_comm *c, const taxor_hixf_view *view, taxor_gpu_index **out);
/* searchers[n_devices]: searcher i was created on out[i] / devices[i] and has a run in flight or finished */
int taxor_gpu_gather_results(taxor_gpu_comm *c, taxor_gpu_searcher *const *searchers, taxor_gpu_results *out);
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

/* ------------------------------------------------------------------------------------------------
 * Diagnosis of an index this library did not write (`taxor verify --variants`).  The arithmetic of
 * seqan3::interleaved_xor_filter is un-vendored in the reference; taxor_amd/csrc/ixf_arith.h holds this library's
 * reading (evidence: src/main/xorfilter.hpp:36-45,60-68,338-350, src/main/hashutil.hpp:50-61).  A variant describes
 * another reading of the same raw fingerprint bytes; the scan probes IXF `ixf` of a resident index under every variant
 * with hash lists cut from a genome that is in the index and reports, per (variant, list), the best-bin match ratio:
 * ~1.0 under the file's true arithmetic, ~2^-8 otherwise.  best_ratio[n_variants * n_lists], variant-major.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed;
    uint64_t seg_len;   /* rows per hash segment */
    uint64_t stride;    /* bytes per fingerprint row */
    uint8_t key_hash;   /* 0 murmur64 finaliser (hashutil.hpp:50-57), 1 none, 2 wyhash mix, 3 splitmix64 finaliser */
    uint8_t seed_mode;  /* 0 h(key + seed) (hashutil.hpp:59-61), 1 h(key ^ seed), 2 h(key) + seed, 3 seed unused */
    uint8_t rot;        /* row i uses rotl64(h, rot * i); 21 in xorfilter.hpp:42-45 */
    uint8_t reduce;     /* 0 ((u32)rot * seg_len) >> 32 (xorfilter.hpp:36-40), 1 (u32)rot % seg_len, 2 mulhi64(rot, seg_len) */
    uint8_t fp_mode;    /* 0 (u8)(h ^ h>>32) (xorfilter.hpp:60-62), 1 (u8)h, 2 (u8)(h>>56), 3 (u8)(h>>32) */
    uint8_t layout;     /* 0 data[row*stride + bin] (interleaved), 1 data[bin*rows + row] */
    uint8_t pad[2];
} taxor_ixf_variant;
/* The arithmetic part of a variant (key hash, seed entry, rotation step, range reduction, fingerprint fold) as the code
 * an index carries (taxor_hixf_view::ixf_arith); 0 for this library's reading.  layout, seed, seg_len and stride are not
 * part of it: the last three come from the file per IXF, and only the interleaved layout can be searched.  _decode fills
 * the five arithmetic fields of *out and leaves the others alone. */
uint32_t taxor_ixf_arith_code(const taxor_ixf_variant *v);
void taxor_ixf_arith_decode(uint32_t code, taxor_ixf_variant *out);
/* this library's reading for the given seed / segment length / stride */
void taxor_ixf_variant_default(taxor_ixf_variant *out, uint64_t seed, uint64_t seg_len, uint64_t stride);
int taxor_gpu_ixf_variant_scan(taxor_gpu_index *idx, uint64_t ixf, const taxor_ixf_variant *variants, uint32_t n_variants,
                               const uint64_t *hashes, const uint64_t *hash_off, uint64_t n_lists, float *best_ratio);
/* one-line description of a variant; returns the length written */
uint64_t taxor_ixf_variant_describe(const taxor_ixf_variant *v, char *buf, uint64_t cap);

/* ------------------------------------------------------------------------------------------------
 * .hixf on-disk format (drop-in): cereal BinaryOutputArchive of taxor_index<hixf_t>, native little endian,
 * no header (src/main/store_index.hpp:24-27).  Envelope order is pinned by src/main/index.hpp:208-244,
 * src/taxonomy/Species.hpp:40-50 and hierarchical_interleaved_xor_filter.hpp:152-158,277-282; the record of
 * one seqan3::interleaved_xor_filter is UN-VENDORED -- this library's schema for it is documented in
 * taxor_amd/csrc/hixf_io.cpp (one place to change).  The loader fails loudly on truncated or inconsistent
 * files (the reference swallows read errors, index.hpp:235-238 -- deliberate behavioural improvement).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const char *organism_name, *accession_id, *taxid, *taxnames_string, *taxid_string; /* Species.hpp:43-47 */
    uint64_t user_bin, seq_len;                                                       /* :48-49           */
} taxor_species;

typedef struct {
    uint64_t window_size;     /* index.hpp:217 */
    uint8_t parts;            /* :222 */
    uint8_t compressed;       /* :225 */
    uint64_t n_species;
    const taxor_species *species;            /* :227 */
    uint64_t n_user_bin_filenames;
    const char *const *user_bin_filenames;   /* hixf.hpp:280; bin_path (index.hpp:226) is written as one
                                                single-element vector per filename, like taxor_build.cpp:519-523 */
    uint8_t foreign_schema;                  /* set by taxor_hixf_load: the IXF records did not follow this library's
                                                own layout and were read through the probed one -- the file was written
                                                by other software, whose IXF arithmetic this library has not been
                                                verified against (run `taxor verify`) */
} taxor_hixf_meta;

typedef struct taxor_hixf taxor_hixf;  /* a parsed .hixf held in host memory (mmap) */

/* Layout of one seqan3::interleaved_xor_filter record inside the file (UN-VENDORED in the reference): n_before
 * u64 scalars, the fingerprint vector (u64 length + bytes), n_after u64 scalars.  idx_* select the scalar (counted
 * over before-then-after) that holds a field, -1 = not stored: bins then come from next_ixf_id's inner sizes,
 * stride = ceil(bins/64)*64, seg_len = rows/3 with rows = length/stride, seed = default_seed. */
typedef struct {
    uint32_t n_before, n_after;
    int32_t idx_bins, idx_stride, idx_seg_len, idx_seed;
    uint32_t seg_len_is_rows;   /* 1: the idx_seg_len scalar holds rows = 3*seg_len */
    uint64_t default_seed;      /* 13572355802537770549 = the fixed start seed of src/main/xorfilter.hpp:153 */
} taxor_ixf_schema;

/* this library's own schema: bins | technical_bins | seg_len | bin_words | seed | ftype | data */
void taxor_ixf_schema_default(taxor_ixf_schema *out);
/* `hixf-probe`: walk a real file with every (n_before, n_after) until the records re-parse n times and the
 * pinned tail (next_ixf_id, user_bins) lands exactly on end-of-file, then infer which scalar is which.  Writes a
 * human-readable report (NUL-terminated, truncated to cap).  SURVEY.md 8(f) #2. */
int taxor_hixf_probe(const char *path, taxor_ixf_schema *out, char *report, uint64_t cap);
int taxor_hixf_load_schema(const char *path, const taxor_ixf_schema *schema, taxor_hixf **out);
int taxor_hixf_store_schema(const char *path, const taxor_hixf_view *view, const taxor_hixf_meta *meta,
                            const taxor_ixf_schema *schema);
/* load with the default schema; if the records do not fit it, probe the file and load with what was found */
int taxor_hixf_load(const char *path, taxor_hixf **out);
void taxor_hixf_free(taxor_hixf *h);
/* Once the index is resident on the devices: give the pages of the file mapping that hold fingerprint bytes back to the
 * system, in slices, from whatever thread the caller likes (the search can run meanwhile).  The metadata (species,
 * filenames, bin tables) stays; view->ixf[i].data must not be read by the caller afterwards.  taxor_hixf_free is then
 * cheap. */
void taxor_hixf_release_data(taxor_hixf *h);
const taxor_hixf_view *taxor_hixf_get_view(const taxor_hixf *h);
/* the file does not say which reading of the IXF arithmetic its writer followed: a loaded file starts at 0 (this library's);
 * the caller sets what `taxor verify --variants` found (taxor_ixf_arith_code) before creating the index from the view */
void taxor_hixf_set_arith(taxor_hixf *h, uint32_t arith);
const taxor_hixf_meta *taxor_hixf_get_meta(const taxor_hixf *h);
int taxor_hixf_store(const char *path, const taxor_hixf_view *view, const taxor_hixf_meta *meta);

/* Per-read output text (taxor_search.cpp:268-305): appends the line(s) of one read to buf (capacity cap) and
 * returns the number of bytes the text needs (call again with a larger buffer if > cap).  Species lookup
 * follows the reference: user_bin -> first species with that user_bin, species[0] if none (:172-178,289). */
uint64_t taxor_format_read(const taxor_hixf *h, const char *id, uint64_t id_len, uint64_t read_len,
                           uint32_t n_hashes, const int64_t *user_bin, const uint32_t *count, uint64_t n_tuples,
                           char *buf, uint64_t cap);

/* the same for a whole chunk of reads (what a formatter thread of the host calls): read r has id ids[r] (id_len[r] bytes),
 * read_len[r] bases, n_hashes[r] hashes and the tuples [read_off[r], read_off[r+1]) of user_bin / count.  Returns the bytes
 * the text needs; nothing is written unless it fits cap. */
uint64_t taxor_format_reads(const taxor_hixf *h, uint64_t n_reads, const char *const *ids, const uint64_t *id_len,
                            const uint64_t *read_len, const uint32_t *n_hashes, const uint64_t *read_off,
                            const int64_t *user_bin, const uint32_t *count, char *buf, uint64_t cap);

/* ------------------------------------------------------------------------------------------------
 * Host-side scalars of the path (no GPU needed).
 * ---------------------------------------------------------------------------------------------- */
/* threshold::threshold + get(): ratio by which the hash count is multiplied.  percentage in (0,1] selects
 * the percentage model (threshold.hpp:27-32), otherwise get_min_syncmer_match_ratio(k, error_rate)
 * (syncmer_model.hpp:38-50).  Returns a negative value where the reference would read out of bounds. */
double taxor_threshold_ratio(uint32_t kmer_size, double error_rate, double percentage);
/* threshold::threshold's choice of model (threshold.hpp:22-47) -> TAXOR_THR_* */
int taxor_threshold_kind(int use_syncmer, uint32_t kmer_size, uint64_t window_size, double percentage);
/* threshold::get for every kind (threshold.hpp:51-81; kmer_model.cpp:10-23, fracminhash_model.cpp:8-33,
 * gaussian_inverse.cpp:13-50); scaling_factor = count / (read_len - k + 1) as at taxor_search.cpp:263.  size_t
 * arithmetic wraps exactly like the reference's (a short read's threshold can be unreachable). */
uint64_t taxor_threshold_model(int kind, uint64_t count, uint32_t kmer_size, double error_rate, double percentage,
                               double scaling_factor);
/* fills prm->ratio / model / error_rate for an index and the command-line values; leaves the other fields alone.
 * TAXOR_E_ARG where the syncmer model has no entry (k odd or outside 12..30, error rate outside [0, 0.2]). */
int taxor_threshold_select(const taxor_hixf_view *view, double error_rate, double percentage, taxor_gpu_search_params *prm);
/* (size_t)(hash_count * ratio) */
uint64_t taxor_threshold(uint64_t hash_count, double ratio);
/* keep[i] = !(double(count[i]) < double(max)*0.8), taxor_search.cpp:275-286 */
void taxor_classify_filter(const uint32_t *count, uint64_t n, uint8_t *keep);
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

/* ---- Deflate chunks decoded on the device (taxor_amd/csrc/inflate.hip): the reader of single-member .gz query files (the
 * reference reads .gz through seqan3's stream layer, one zlib stream on one thread, src/main/taxor_search.cpp:181-184).  The host
 * (taxor_amd/csrc/pgz.h) cuts the member's deflate stream into chunks, finds a block start in each, and hands a batch over:
 * every chunk is decoded from its start bit to the first block boundary at or behind its stop bit into 16-bit symbols -- a byte,
 * or 256 + w for "byte w of the 32 KiB before this chunk".  The host then checks that every chunk starts where its predecessor
 * ended; a chunk that does not, or that the device gave up on (status != 0), is decoded on the host and its symbols are put in
 * its place (taxor_gpu_inflate_replace).  taxor_gpu_inflate_resolve chains the 32-KiB windows from chunk to chunk, turns every
 * symbol into its byte and copies the bytes of chunk first + i to out[i].  Bits are counted from the first byte of `in`.  One
 * caller at a time per inflater. */
typedef struct taxor_gpu_inflater taxor_gpu_inflater;
typedef struct taxor_inflate_chunk {
    uint64_t start_bit, stop_bit; /* start == stop: nothing to decode (the host will put the chunk's symbols in place) */
    uint64_t weight;              /* compressed bits the chunk stands for: the arena is shared out by it */
} taxor_inflate_chunk;
typedef struct taxor_inflate_result {
    uint64_t end_bit;     /* the block boundary the chunk ended at (>= stop_bit), or the end of the member's final block */
    uint64_t n_out;       /* symbols = bytes of output */
    uint32_t status;      /* 0 decoded; 2 invalid deflate data from this start; 3 more output than the chunk's share of the arena; 4 ran past the input */
    uint32_t final_block; /* the member's last block ended this chunk */
} taxor_inflate_result;
/* max_symbols: 16-bit symbols the arena holds for one batch (every chunk needs 32768 + its output + 256) */
int taxor_gpu_inflater_create(int device, uint64_t max_in_bytes, uint32_t max_chunks, uint64_t max_symbols, taxor_gpu_inflater **out);
void taxor_gpu_inflater_destroy(taxor_gpu_inflater *h);
int taxor_gpu_inflate_decode(taxor_gpu_inflater *h, const uint8_t *in, uint64_t in_bytes, const taxor_inflate_chunk *chunks, uint32_t n_chunks,
                             taxor_inflate_result *results);
/* the same in two halves: _begin returns when the input is on its way and the kernel queued, _end waits for the results */
int taxor_gpu_inflate_decode_begin(taxor_gpu_inflater *h, const uint8_t *in, uint64_t in_bytes, const taxor_inflate_chunk *chunks, uint32_t n_chunks);
int taxor_gpu_inflate_decode_end(taxor_gpu_inflater *h, taxor_inflate_result *results);
int taxor_gpu_inflate_replace(taxor_gpu_inflater *h, uint32_t chunk, const uint16_t *symbols, uint64_t n_out, uint64_t end_bit, uint32_t final_block);
int taxor_gpu_inflate_resolve(taxor_gpu_inflater *h, const uint8_t *window_in /* 32768 bytes */, uint32_t first, uint32_t count, uint8_t *const *out,
                              uint8_t *window_out /* 32768 bytes, may be NULL */);
/* a decoded chunk's symbols, n_out of them (parity tests against the host decoder) */
int taxor_gpu_inflate_symbols(taxor_gpu_inflater *h, uint32_t chunk, uint16_t *out);

#ifdef __cplusplus
}
#endif
#endif
