/*
 * taxor_gpu.h -- the drop-in C ABI of the MI355X-native `taxor search` hot path (libtaxor_gpu.so).
 *
 * The reference (JensUweUlrich/Taxor @ 2025-05-23) has no FFI; its seam for this path is the chunk loop of search_single():
 *   hixf::do_parallel(worker, records.size(), threads, compute_time)   (src/main/taxor_search.cpp:315-326, worker = :196-313)
 * A batch of (id, seq) records goes in, per-read (user_bin, count) tuples come out.  THIS header is that seam and what a binding
 * needs around it (index hand-off, thresholds, .hixf loading, the output text, several GPUs); measurement aids, stage entry
 * points of the parity tests, index construction and diagnosis live in taxor_gpu_tools.h.  Plain pointers and sizes, no C++
 * types, no exceptions: every int function returns 0 or a negative taxor_status, and taxor_gpu_last_error() is the calling
 * thread's last message (the reference prints "[TAXOR SEARCH ERROR] ..." and returns -1, taxor_search.cpp:380-384).
 * Threading: an index is immutable and shareable; a searcher is single-caller like a membership_agent
 * (hierarchical_interleaved_xor_filter.hpp:371-379); result pointers stay valid until the next call on that searcher (:381-406).
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
    TAXOR_E_ARG = -1,      /* bad argument / unsupported parameter */
    TAXOR_E_HIP = -2,      /* HIP or RCCL failure */
    TAXOR_E_ALPHABET = -3, /* read character outside the dna15 alphabet */
    TAXOR_E_INTERNAL = -4, /* internal capacity invariant violated (never silent) */
    TAXOR_E_IO = -5,       /* file unreadable / unwritable / inconsistent */
    TAXOR_E_NOMEM = -6
} taxor_status;

const char *taxor_gpu_last_error(void);

/* ---- Index hand-off.  Replaces taxor_index<hixf_t> loaded by cereal (src/main/load_index.hpp:27-38) and held by pointer in the
 * agents (hierarchical_interleaved_xor_filter.hpp:300).  Per IXF: the fingerprint array plus next_ixf_id[i] (:115-122) and
 * ixf_bin_to_filename_position[i] (:172-178).  The library copies everything into HBM; the caller keeps its host arrays. */
typedef struct {
    uint64_t bins;            /* bin count == counting_vector size */
    uint64_t stride;          /* bytes per fingerprint row in HBM: multiple of 64, >= bins */
    uint64_t seg_len;         /* rows per hash segment; 3 * seg_len rows */
    uint64_t seed;            /* per-IXF seed (src/hixf/build/construct_ixf.cpp:100-108 redraws it) */
    const uint8_t *data;      /* host bytes in the layout taxor_hixf_view::ixf_layout names (0: data[row * stride + bin],
                                 3 * seg_len * stride bytes); NULL = rows left uninitialised (tools: fill_random / upload_bin) */
    const int64_t *next_ixf;  /* [bins] */
    const int64_t *fname_idx; /* [bins], -1 = merged bin */
    uint64_t src_stride;      /* row pitch in bytes (row-interleaved) / bin columns stored (bin-major) of the SOURCE bytes;
                                 0 = stride */
} taxor_ixf_view;

/* Fingerprint bytes that are NOT in host memory (a .hixf on disk): index creation pulls them piece by piece from several of its
 * own threads into page-locked staging.  read() copies `len` bytes of IXF `ixf`'s array from byte `offset`; 0 = ok; thread-safe. */
typedef struct {
    int (*read)(void *ctx, uint64_t ixf, uint64_t offset, uint64_t len, void *dst);
    void *ctx;
} taxor_ixf_source;

typedef struct {
    uint64_t n_ixf;
    const taxor_ixf_view *ixf;
    uint64_t n_user_bins;
    uint8_t kmer_size, syncmer_size, t_syncmer; /* src/main/index.hpp:219-221 */
    uint8_t use_syncmer;    /* :223; 1 = open canonical syncmers (syncmer.cpp:80-165), 0 = seqan3 minimiser_hash over window_size
                               (taxor_search.cpp:210-212,239-260) */
    uint16_t scaling;       /* :224; > 1 = FracMinHash down-sampling (taxor_search.cpp:223-233), applied on the device */
    uint64_t window_size;   /* :212; use_syncmer == 0 only: == kmer_size selects every canonical k-mer */
    uint32_t ixf_arith;     /* which reading of seqan3::interleaved_xor_filter's un-vendored arithmetic the fingerprints follow:
                               0 = this library's (taxor_amd/csrc/ixf_arith.h), else taxor_ixf_arith_code() of another (tools) */
    const taxor_ixf_source *source; /* NULL: bytes at ixf[i].data; else read through it (taxor_hixf_load sets a pread() reader) */
    uint32_t ixf_layout;    /* how the SOURCE stores each IXF's bytes (taxor_amd/csrc/ixf_layout.h): 0 = the search layout
                               data[row * stride + bin]; bits 0-7 kind (1 bin-major data[bin * rows + row], 2 bit-sliced 64-bin
                               words), bit 8 rows position-major (pos * 3 + segment), bits 9-10 pitch (0 padded to 64, 1 exactly
                               bins, 2 stored scalar).  Anything but 0 is transposed into the search layout on the device while
                               the index is uploaded (the serialiser is un-vendored: hierarchical_interleaved_xor_filter.hpp:152-158) */
} taxor_hixf_view;

typedef struct taxor_gpu_index taxor_gpu_index;
int taxor_gpu_index_create(const taxor_hixf_view *view, int device, taxor_gpu_index **out);
void taxor_gpu_index_destroy(taxor_gpu_index *idx);
uint64_t taxor_gpu_index_data_bytes(const taxor_gpu_index *idx); /* fingerprint bytes resident in HBM */

/* ---- Searcher = one GPU-side membership agent + the worker lambda's locals (taxor_search.cpp:198-203) + hixf::threshold
 * (src/hixf/search/threshold.hpp:22-81).  PERCENTAGE / SYNCMER thresholds are (size_t)(n_hashes * ratio) on the device; the KMER
 * and FRACMINHASH models (indexes built without --use-syncmer) are evaluated per read on the host in the reference's doubles. */
enum { TAXOR_THR_PERCENTAGE = 0, TAXOR_THR_SYNCMER = 1, TAXOR_THR_KMER = 2, TAXOR_THR_FRACMINHASH = 3 };
typedef struct {
    double ratio;             /* threshold.hpp:60,76-79 */
    uint32_t sub_batch_reads; /* reads per internal launch group (0 = default 32768) */
    uint64_t sub_batch_bases; /* bases per internal launch group (0 = default 2^29) */
    uint32_t time_kernels;    /* 1 = bracket the dominant kernel with HIP events (taxor_gpu_batch_stats, tools) */
    uint32_t model;           /* TAXOR_THR_* */
    double error_rate;        /* --error-rate, used by the KMER and FRACMINHASH models */
    uint32_t flags;           /* TAXOR_SEARCH_*; 0 = defaults.  None changes a result */
} taxor_gpu_search_params;
enum {
    TAXOR_SEARCH_NO_PRUNE = 1u,        /* count every hash against every bin of every visited IXF, the reference's formulation
                                          (:307-309); the default skips bin runs that provably cannot reach the threshold */
    TAXOR_SEARCH_GROUP_ALWAYS = 2u,    /* group each level's work items by IXF whatever the sub-batch size */
    TAXOR_SEARCH_NO_SMALL_PATH = 4u,   /* calls of a few thousand reads take the pipeline of large batches */
    TAXOR_SEARCH_SPLIT_ALWAYS = 8u,    /* root work items split over column ranges whatever the batch size */
    TAXOR_SEARCH_FORCE_TREE_STALL = 16u /* test hook: a small call's one-launch traversal gives up at its first empty poll, so
                                          every piece takes the recovery path (rerun level by level) */
};
typedef struct taxor_gpu_searcher taxor_gpu_searcher;
int taxor_gpu_searcher_create(taxor_gpu_index *idx, const taxor_gpu_search_params *prm, taxor_gpu_searcher **out);
void taxor_gpu_searcher_destroy(taxor_gpu_searcher *s);

/* Per-read results, CSR: tuples of read r at [read_off[r], read_off[r+1]) in DFS order of the HIXF traversal
 * (hierarchical_interleaved_xor_filter.hpp:313-338), BEFORE the 0.8*max filter of taxor_search.cpp:275-286 (taxor_classify_filter
 * or the formatter below apply it).  n_hashes[r] is QHASH_COUNT (:261,298). */
typedef struct {
    uint64_t n_reads, n_tuples;
    const uint64_t *read_off; /* [n_reads + 1] */
    const int64_t *user_bin;  /* [n_tuples] */
    const uint32_t *count;    /* [n_tuples] */
    const uint32_t *n_hashes; /* [n_reads] */
} taxor_gpu_results;

/* THE drop-in call: replaces hixf::do_parallel(worker, n, threads, compute_time) for one chunk.  bases = the reads' sequences
 * concatenated as read from FASTA/FASTQ (any dna15 character; the dna4 mapping of src/hixf/build/dna4_traits.hpp:15-18 is applied
 * on the device), offsets[n_reads + 1]. */
int taxor_gpu_search_batch(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads, taxor_gpu_results *out);
/* The same in two halves, so the host reads its next chunk meanwhile (the reference's loop is synchronous, :315-326): _begin
 * enqueues and returns (after the last host-to-device copy unless `bases` is registered memory); inputs stay valid until _end.
 * With the FRACMINHASH model _begin returns after the last sub-batch's hashing (its thresholds are computed on the host). */
int taxor_gpu_search_batch_begin(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads);
int taxor_gpu_search_batch_end(taxor_gpu_searcher *s, taxor_gpu_results *out);
/* The same over reads in SEVERAL host buffers (the reference's chunk is a vector of records with their own storage, :319): one
 * batch over all segments in order, one CSR from taxor_gpu_search_batch_end. */
typedef struct {
    const char *bases;
    const uint64_t *offsets; /* [n_reads + 1] */
    uint64_t n_reads;
} taxor_read_segment;
int taxor_gpu_search_segments_begin(taxor_gpu_searcher *s, const taxor_read_segment *segs, uint64_t n_segs);
/* Optional: page-lock a buffer the caller passes again and again (copies become direct DMA); unregister before freeing it. */
int taxor_gpu_host_register(void *ptr, uint64_t bytes);
int taxor_gpu_host_unregister(void *ptr);

/* ---- Several GPUs of one node from ONE process (`taxor search --gpus N`).  Reads are independent (taxor_search.cpp:214): the
 * index is replicated, every device classifies its own batches, and the path has two exchange steps behind a communicator:
 * _index_create_replicated (one PCIe upload into devices[0], ncclBroadcast of the slab behind it) and _gather_results (the
 * per-read results of searcher i on devices[i] -> devices[0] by grouped ncclSend/ncclRecv, one CSR in device order).  Replaces:
 * one address space, one result stream under a mutex (taxor_search.cpp:311, sync_out.hpp:24-29).  TAXOR_COMM_RCCL binds RCCL at
 * run time and verifies known bytes through both collectives at creation; TAXOR_COMM_HOST stages through host memory.  A
 * communicator never changes transport by itself.  Single-caller; a gather waits for the runs of the searchers it is given and
 * touches nothing else; its result pointers stay valid until the next gather. */
enum { TAXOR_COMM_RCCL = 0, TAXOR_COMM_HOST = 1 };
typedef struct taxor_gpu_comm taxor_gpu_comm;
int taxor_gpu_comm_create(const int *devices, uint32_t n_devices, int transport, taxor_gpu_comm **out);
void taxor_gpu_comm_destroy(taxor_gpu_comm *c);
int taxor_gpu_index_create_replicated(taxor_gpu_comm *c, const taxor_hixf_view *view, taxor_gpu_index **out /* [n_devices] */);
int taxor_gpu_gather_results(taxor_gpu_comm *c, taxor_gpu_searcher *const *searchers /* [n_devices] */, taxor_gpu_results *out);

/* ---- .hixf on-disk format: cereal BinaryOutputArchive of taxor_index<hixf_t>, native little endian, no header
 * (src/main/store_index.hpp:24-27).  Envelope order pinned by src/main/index.hpp:208-244, src/taxonomy/Species.hpp:40-50,
 * hierarchical_interleaved_xor_filter.hpp:152-158,277-282; the record of one seqan3::interleaved_xor_filter is UN-VENDORED: the
 * loader tries this library's schema, then probes the file (tools: taxor_hixf_probe, taxor_ixf_schema).  It fails loudly on
 * truncated or inconsistent files (the reference swallows read errors, index.hpp:235-238). */
typedef struct {
    const char *organism_name, *accession_id, *taxid, *taxnames_string, *taxid_string; /* Species.hpp:43-47 */
    uint64_t user_bin, seq_len;                                                       /* :48-49 */
} taxor_species;
typedef struct {
    uint64_t window_size; /* index.hpp:217 */
    uint8_t parts;        /* :222 */
    uint8_t compressed;   /* :225 */
    uint64_t n_species;
    const taxor_species *species;          /* :227 */
    uint64_t n_user_bin_filenames;
    const char *const *user_bin_filenames; /* hixf.hpp:280; bin_path (index.hpp:226) = one single-element vector per filename */
    uint8_t foreign_schema;                /* set by taxor_hixf_load: the IXF records were read through a probed layout -- written
                                              by other software; run `taxor verify` / `taxor pin` */
} taxor_hixf_meta;
typedef struct taxor_hixf taxor_hixf; /* a parsed .hixf (metadata in host memory, fingerprints read on demand) */
int taxor_hixf_load(const char *path, taxor_hixf **out);
void taxor_hixf_free(taxor_hixf *h);
/* once the index is resident: give the file mapping's fingerprint pages back (view->ixf[i].data must not be read afterwards) */
void taxor_hixf_release_data(taxor_hixf *h);
const taxor_hixf_view *taxor_hixf_get_view(const taxor_hixf *h);
const taxor_hixf_meta *taxor_hixf_get_meta(const taxor_hixf *h);
/* the file does not say which reading of the IXF arithmetic / which fingerprint layout its writer followed: a loaded file starts
 * at 0 / at what its array lengths admit; set what `taxor verify --variants` found before creating the index from the view.
 * _set_layout recomputes every IXF's stride / seg_len / src_stride and fails if an array length contradicts the layout. */
void taxor_hixf_set_arith(taxor_hixf *h, uint32_t arith);
int taxor_hixf_set_layout(taxor_hixf *h, uint32_t layout);
int taxor_hixf_store(const char *path, const taxor_hixf_view *view, const taxor_hixf_meta *meta);

/* ---- Per-read output text (taxor_search.cpp:268-305): appends the line(s) of one read to buf and returns the bytes the text
 * needs (call again with a larger buffer if > cap).  Species lookup: user_bin -> first species with it, species[0] if none. */
uint64_t taxor_format_read(const taxor_hixf *h, const char *id, uint64_t id_len, uint64_t read_len, uint32_t n_hashes,
                           const int64_t *user_bin, const uint32_t *count, uint64_t n_tuples, char *buf, uint64_t cap);
/* the same for a chunk of reads; nothing is written unless everything fits cap */
uint64_t taxor_format_reads(const taxor_hixf *h, uint64_t n_reads, const char *const *ids, const uint64_t *id_len,
                            const uint64_t *read_len, const uint32_t *n_hashes, const uint64_t *read_off,
                            const int64_t *user_bin, const uint32_t *count, char *buf, uint64_t cap);

/* ---- Host-side scalars of the path (no GPU needed). */
/* threshold::threshold + get(): percentage in (0,1] selects the percentage model (threshold.hpp:27-32), otherwise
 * get_min_syncmer_match_ratio(k, error_rate) (syncmer_model.hpp:38-50); negative where the reference reads out of bounds */
double taxor_threshold_ratio(uint32_t kmer_size, double error_rate, double percentage);
/* threshold::threshold's choice of model (threshold.hpp:22-47) -> TAXOR_THR_* */
int taxor_threshold_kind(int use_syncmer, uint32_t kmer_size, uint64_t window_size, double percentage);
/* threshold::get for every kind (threshold.hpp:51-81; kmer_model.cpp:10-23, fracminhash_model.cpp:8-33, gaussian_inverse.cpp:13-50);
 * scaling_factor as at taxor_search.cpp:263; size_t arithmetic wraps like the reference's */
uint64_t taxor_threshold_model(int kind, uint64_t count, uint32_t kmer_size, double error_rate, double percentage, double scaling_factor);
/* fills prm->ratio / model / error_rate for an index and the command-line values; TAXOR_E_ARG where the syncmer model has no
 * entry (k odd or outside 12..30, error rate outside [0, 0.2]) */
int taxor_threshold_select(const taxor_hixf_view *view, double error_rate, double percentage, taxor_gpu_search_params *prm);
uint64_t taxor_threshold(uint64_t hash_count, double ratio); /* (size_t)(hash_count * ratio) */
/* keep[i] = !(double(count[i]) < double(max) * 0.8), taxor_search.cpp:275-286 */
void taxor_classify_filter(const uint32_t *count, uint64_t n, uint8_t *keep);

#ifdef __cplusplus
}
#endif
#endif
