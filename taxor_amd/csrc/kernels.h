// kernels.h -- device data layout + launch wrappers of the taxor search hot path (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace taxor {

// ---- index layout in HBM --------------------------------------------------------------------------
// One slab holds every IXF's fingerprint array, 4 KiB aligned: row r of IXF v at data + r*stride,
// stride = bins rounded up to 64 B, so a probe (one hash, one IXF) is three contiguous rows and a lane
// reads one aligned 16-B unit of 16 bins.
struct IxfDesc {
    const uint8_t *data;
    uint64_t seed;
    uint32_t seg_len;
    uint32_t bins;
    uint32_t stride;   // bytes per row
    uint32_t units;    // stride / 16
    uint32_t bin_base; // first entry of this IXF in the per-bin tables
    uint32_t arith;    // arithmetic code of the index (ixf_arith.h); 0 = this library's reading
};

// per-bin tables, indexed by bin_base + bin:
//   binfo   : bit31 = merged bin, bit30 = last technical bin of a (possibly split) user bin,
//             bits 0..29 = child IXF id for merged bins
//   ubin    : user bin (filename index) of a leaf bin
//   dfs_key : rank of the bin in the full depth-first traversal of the hierarchy; sorting a read's
//             tuples by it restores the reference's emission order
static constexpr uint32_t BINFO_MERGED = 0x80000000u;
static constexpr uint32_t BINFO_END = 0x40000000u;

// ---- flags raised by kernels (never silent) ---------------------------------------------------------
enum : uint32_t {
    FLAG_ALPHABET = 1u,       // character outside dna15 in the input
    FLAG_CAND_OVERFLOW = 2u,  // syncmer candidate capacity bound violated (internal invariant)
    FLAG_QUEUE_OVERFLOW = 4u, // work queue too small  -> host grows and reruns
    FLAG_HITS_OVERFLOW = 8u,  // hit buffer too small   -> host grows and reruns
    FLAG_TUPLE_OVERFLOW = 16u,// batch tuple arrays too small -> host grows and reruns
    FLAG_DEDUP_OVERFLOW = 32u,// dedup scratch too small (internal invariant)
    FLAG_TREE_STALL = 64u     // the one-launch traversal of a small batch stopped waiting for work that never came -> host reruns the piece level by level
};

// counters block (one per searcher).  Hot words sit on their own 128-B lines: returning atomics on one line are
// serialised by a single L2 channel (~90 per microsecond), and a level with tens of thousands of short work items
// would otherwise queue its cursor, queue-append, hit-append and statistics atomics behind each other.
struct alignas(128) PaddedU32 {
    uint32_t v;
    uint32_t pad[31];
};
struct Counters {
    uint32_t flags;
    uint32_t reserved0;
    uint32_t n_big;
    uint32_t pad0[29];
    PaddedU32 n_hits;
    PaddedU32 q_n[16];                // work items per level
    PaddedU32 q_cursor[16];           // dynamic work cursor per level
    PaddedU32 q_xcur[16][8];          // levels below the root, queue grouped by IXF: one cursor per eighth of the queue (QueryArgs::xcd_slices)
    alignas(128) unsigned long long tuple_total;   // tuples emitted so far in this batch run
    unsigned long long n_hashes;                   // distinct hashes so far
    alignas(128) unsigned long long query_bytes;   // sum n_h*3*bins over work items
    unsigned long long n_work;                     // work items processed
    unsigned long long touched_bytes;              // bytes the query kernel actually requested (after pruning)
    unsigned long long pad1[13];
    // per HIXF level (levels >= 7 share the last entry): requested bytes and fingerprint-row reads (one per hash and row
    // in the dense phase, one per hash, row and surviving unit in the sparse phase) -- a narrow row is bound by the
    // number of DRAM rows opened, not by its bytes
    unsigned long long lvl_touched[8];
    unsigned long long lvl_rows[8];
    unsigned long long lvl_sparse[8];   // of lvl_rows: 16-B loads of the sparse phase (one row each, billed as a 64-B sector in lvl_touched)
};
static constexpr int MAX_LEVELS = 16;

// reads with at most this many selected syncmers dedup in LDS (partitioned passes); longer ones need SyncmerArgs::gtab
static constexpr uint32_t SYNC_LDS_DEDUP_MAX = 66816;
// reads with at most this many candidate slots (hcap) may go to the wave-per-read syncmer kernel
static constexpr uint32_t SYNC_WAVE_CAND = 512;

struct SyncmerArgs {
    const uint32_t *packed;   // 2-bit bases, 16 per word, first base in the top bits
    const uint64_t *poff;     // word offset of read r
    const uint32_t *rlen;     // bases in read r
    const uint64_t *hoff;     // first candidate/hash slot of read r
    const uint32_t *hcap;     // slots of read r
    uint64_t *cand;           // selected syncmer hashes in window order (with duplicates)
    uint64_t *hashes;         // distinct hashes, first-occurrence order
    uint32_t *nh;             // distinct count per read
    uint64_t *thr;            // (size_t)(nh * ratio)
    double ratio;
    double scaling_limit;     // > 0: keep a hash only if (double)wyhash(hash) <= limit (FracMinHash, taxor_search.cpp:223-233)
    uint32_t *gtab;           // per-block dedup scratch for reads whose table does not fit LDS
    uint32_t gtab_stride;     // slots per block (power of two), 0 = none
    Counters *ctr;
    uint32_t *cursor;         // dynamic work cursor of this launch (zeroed by the host)
    const uint32_t *order;    // processing order: longest reads first, so that no long read is left for the tail
    uint32_t n_reads;
    uint32_t chunk;           // reads per cursor atomic (short reads: 8, so that their metadata and words are prefetched)
    int k, s, t;
    int w_min;                // > 0: minimiser / k-mer mode with this window size (index built without --use-syncmer)
    int thr_on_device;        // minimiser mode: 1 = thr = (size_t)(nh * ratio) here; 0 = the host applies a model
    unsigned long long *prof; // measurement aid (TAXOR_PROFILE_PHASES=1): 16 per-phase cycle sums, else nullptr
};

struct QueryArgs {
    const IxfDesc *ixf;
    const uint32_t *binfo;
    const uint64_t *hashes;
    const uint64_t *hoff;
    const uint32_t *nh;
    const uint64_t *thr;
    const uint2 *q_in;        // (read, ixf) work items; nullptr = level 0: item i is (i, 0)
    uint2 *q_out;
    uint4 *hits;              // (read, global bin, count, -)
    uint32_t *read_hits;      // tuples per read
    uint32_t *counts_out;     // optional: raw per-bin counts of the (single) work item
    Counters *ctr;
    uint32_t level;
    uint32_t n_level0;        // number of items when q_in == nullptr
    const uint32_t *order0;   // level 0: item i is read order0[i] (longest first); nullptr = identity
    uint32_t q_cap, hit_cap;
    uint32_t map_words;       // words of the alive-unit bitmap in LDS (query_lds_map_words(max_stride))
    uint32_t max_stride;      // widest row of the index: sizes the per-bin LDS arrays
    uint32_t prune;           // 1 = threshold-aware pruning of dead bin runs (off for raw bulk_count)
    uint32_t cursor_chunk;    // work items taken per cursor atomic (0 = 1)
    float prune_margin;       // constant term of the pruning margin mu + 4 sqrt(mu) + c (0 = 3.5)
    uint32_t sparse_stages;   // stages of the pruned phase for long hash lists (0 = 3); between stages the alive set is re-evaluated
    uint32_t xcd_slices;      // 8 = the (IXF-grouped) queue is cut into eight slices and block b starts in slice b % 8 -- the XCD the
                              // dispatcher places it on -- so an IXF's items meet in one XCD's L2 instead of all eight; 0 = one cursor
    uint32_t parts;           // level 0 only (q_in == nullptr): > 1 = every root item is split into `parts` column ranges, each a work
                              // item of its own (item i = part i / n_level0 of read order0[i % n_level0]); the ranges are cut at unit
                              // boundaries where a bin run ends (part_cut, in 16-bin units), so a part sees whole runs and prunes,
                              // tallies and reports on its own -- small batches then fill the chip and finish sooner (api.hip)
    uint16_t part_cut[10];
    uint32_t tree_polls;      // TREE launch: polls of an empty queue slot before the watchdog gives up (FLAG_TREE_STALL; the host then
                              // classifies the piece level by level); 0 = at the first empty poll (test hook)
    uint32_t sort_units;      // 1 = the alive units of a sparse stage are put in ascending order, so that units sharing a 128-B line of a
                              // row are fetched by neighbouring lanes of one load instruction (one request to the memory side, not several)
    uint32_t dense_max_stride;// rows of at most this many bytes are counted densely to the end (a 16-B load of the sparse phase costs HBM
                              // the same 128-B line as the row itself): no alive-unit bookkeeping for them; 0 = prune every width
    uint32_t tally_mode;      // measurement aid (TAXOR_QUERY_TALLY): bit 0 = tally walks every bin, bit 1 = bin info fetched per item
    unsigned long long *prof; // measurement aid (TAXOR_PROFILE_PHASES=1): 16 per-phase cycle sums, else nullptr
};

struct FinalizeArgs {
    const uint4 *hits;
    uint32_t *read_hits;
    uint32_t *cursor;          // zeroed scratch, per read
    uint32_t *roff;            // exclusive scan of read_hits (sub-batch local)
    uint32_t *biglist;
    uint32_t *block_sums;      // scratch of the offset scan: ceil(n_reads / 4096) + 2 words
    const uint32_t *dfs_key;
    const int64_t *ubin;
    uint64_t *read_off;        // batch CSR, at the sub-batch's first read
    int64_t *out_ub;
    uint32_t *out_cnt;
    uint32_t *out_key;
    Counters *ctr;
    uint32_t n_reads;          // reads in this sub-batch
    uint64_t tuple_cap;
    uint32_t hit_cap;
    int is_last;
};

// CSR assembly of a SMALL batch (<= SMALL_FIN_MAX reads) in ONE launch of one block, results written straight into host
// memory the device can address (hipHostMalloc): scan of the per-read tuple counts, scatter of the hit records, per-read sort by
// DFS key, and -- so that the next batch on the same lane needs no memset launches -- the counters block, the per-read hit counts
// and the syncmer cursors are cleared at the end.  h_status: [0] flags, [1] tuples, [2] distinct hashes, [3] work items, [4] query
// bytes (algorithmic), [5] bytes requested.
static constexpr uint32_t SMALL_FIN_MAX = 4096;
struct SmallFinalizeArgs {
    const uint4 *hits;
    uint32_t *read_hits;
    const uint32_t *dfs_key;
    const int64_t *ubin;
    const uint32_t *nh;
    uint32_t *key, *cnt;       // device scratch for the tuples, tuple_cap entries each
    int64_t *ub;
    Counters *ctr;
    uint32_t *sync_cursor;     // two words
    uint32_t n_reads, tuple_cap, hit_cap;
    uint64_t *h_read_off;      // [n_reads + 1], batch-local
    uint32_t *h_nh;            // [n_reads]
    int64_t *h_ub;             // [tuple_cap]
    uint32_t *h_cnt;           // [tuple_cap]
    uint64_t *h_status;        // [8]
};
void launch_finalize_small(const SmallFinalizeArgs &a, hipStream_t st);

// launch wrappers (all asynchronous on `st`)
void launch_pack_dna4(const uint8_t *ascii, const uint64_t *aoff, const uint64_t *poff, uint32_t *packed,
                      uint32_t n_reads, Counters *ctr, hipStream_t st, int max_grid = 0);
void launch_syncmers(const SyncmerArgs &a, int grid, hipStream_t st);
// short reads (hcap <= SYNC_WAVE_CAND), k - s + 1 == 11: one wavefront per read
void launch_syncmers_wave(const SyncmerArgs &a, int grid, hipStream_t st);
int syncmers_wave_grid(int device, int want_per_cu);
bool syncmers_wave_applies(int k, int s);
int syncmers_grid(int device);
// small = the single-wave instantiation for launches of tiny items (IXFs of <= 512 bins under short reads)
void launch_query_level(const QueryArgs &a, int grid, size_t lds_bytes, hipStream_t st, bool small = false, bool root_streams = true);
// the whole traversal of a small batch in one launch (k_query_level<..., TREE>): a.q_in = nullptr, a.q_out = the one queue, whose
// slots hold ~0 (all bytes 0xFF) before and after; a.level = 0
void launch_query_tree(const QueryArgs &a, int grid, size_t lds_bytes, hipStream_t st, int unroll = 2);
int query_grid(int device, size_t lds_bytes, int want_per_cu);
int query_grid_small(int device, size_t lds_bytes);
size_t query_lds_bytes(uint32_t max_stride, bool small = false);
uint32_t query_map_words(uint32_t max_stride);
void launch_finalize(const FinalizeArgs &a, hipStream_t st);
// counting sort of a level's work queue by IXF id (q -> out; hist = n_ixf words of scratch); the item count is read on the device
void launch_queue_group_by_ixf(const uint2 *q, const Counters *ctr, uint32_t lvl, uint32_t q_cap, uint32_t *hist, uint32_t n_ixf,
                               uint2 *out, hipStream_t st);
void launch_fill_random(uint8_t *data, uint64_t n_bytes, uint64_t seed, hipStream_t st);
// random whole-row reads of one IXF, nothing else; returns the bytes the launch requests
// one 16-B load per lane, every lane on a row (and unit) of its own: the access shape of the query kernel's sparse phase;
// returns the number of loads the launch issues
uint64_t launch_gather_sparse(const uint8_t *data, uint64_t rows, uint32_t stride, uint32_t bins, uint64_t want_loads, uint64_t seed,
                              uint32_t *sink, bool nt, hipStream_t st);
uint64_t launch_gather_ceiling(const uint8_t *data, uint64_t rows, uint32_t stride, uint32_t bins, uint64_t want_bytes,
                               uint64_t seed, uint32_t *sink, bool nt, hipStream_t st, uint32_t n_ixf = 1, uint64_t spacing = 0);
void launch_scatter_column(uint8_t *data, uint64_t stride, uint64_t bin, const uint8_t *col, uint64_t rows,
                           hipStream_t st);

} // namespace taxor
