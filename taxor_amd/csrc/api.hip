// api.hip -- C ABI of libtaxor_gpu.so (include/taxor_gpu.h): index residency in HBM, the per-GPU searcher,
// the batch pipeline (upload -> syncmers -> level-synchronous HIXF query -> DFS-ordered CSR) and the stage
// entry points the parity tests use.  No CPU fallback exists: every compute entry point runs HIP kernels.
#include "../../include/taxor_gpu_tools.h"
#include "ixf_arith.h"
#include "ixf_layout.h"
#include "kernels.h"
#include "tuning.h"

#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace taxor;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(TAXOR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                              \
    } while (0)

// growable device buffer
template <typename T> struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 8 + 64;
        HIP_TRY(hipMalloc((void **)&p, want * sizeof(T)));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

inline uint64_t round_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

} // namespace

struct taxor_gpu_index {
    int device = 0;
    uint8_t *d_slab = nullptr;
    uint64_t slab_bytes = 0, data_bytes = 0;
    IxfDesc *d_ixf = nullptr;
    uint32_t *d_binfo = nullptr;
    int64_t *d_ubin = nullptr;
    uint32_t *d_dfs_key = nullptr;
    std::vector<IxfDesc> h_ixf; // data pointers are device pointers
    std::vector<uint64_t> rows;
    uint64_t n_user_bins = 0, total_bins = 0, leaf_runs = 0;
    uint32_t depth = 0, max_stride = 0;
    uint32_t lvl_max_stride[MAX_LEVELS] = {};   // widest row among the IXFs of each hierarchy level (root = level 0)
    int k = 0, s = 0, t = 0;
    uint32_t scaling = 1;
    int w_min = 0;           // > 0: index built without --use-syncmer, minimiser window size (== k: every k-mer)
    std::vector<uint32_t> h_binfo, h_bin_base;   // host copies for the hierarchical builder (bin_base has n_ixf + 1 entries)
    std::vector<int64_t> h_ubin;                 // host copies of the other per-bin tables: a replica on another device
    std::vector<uint32_t> h_dfs;                 // (comm.hip) gets them from here, its fingerprint slab over RCCL
    std::vector<uint64_t> slab_off;              // byte offset of every IXF inside the slab
    // column parts of the ROOT's rows (QueryArgs::parts): root_pmax = 8, 4, 2 or 1 (no valid cut); root_cut[j] = first 16-bin unit of
    // part j when the row is cut into root_pmax parts (every boundary is a unit boundary at which a bin run ends)
    uint32_t root_pmax = 1;
    uint16_t root_cut[9] = {};
    // the builder's scratch (builder.hip: peeling state, union table, mark bytes), kept from one build of this index to the next --
    // hipMalloc / hipFree of GB-sized blocks takes the driver anything between nothing and a second -- and released with the index
    void *build_ctx = nullptr;
    void (*build_ctx_free)(void *) = nullptr;
};

struct SubBatch {
    uint32_t first, n;
    uint32_t n_long = 0;     // the first n_long reads of the processing order go to the block-per-read syncmer kernel, the rest
                             // (candidate capacity <= SYNC_WAVE_CAND, ~2.5 kb) to the wave-per-read one
    uint64_t slots;          // candidate/hash slots of this sub-batch
    uint64_t a_begin, a_end; // ASCII byte range of its reads within the batch
};

struct taxor_gpu_searcher {
    taxor_gpu_index *idx = nullptr;
    taxor_gpu_search_params prm{};
    std::vector<uint32_t> h_rlen, h_nh_sub;   // k-mer / FracMinHash threshold models are evaluated on the host
    std::vector<uint64_t> lay_poff, lay_hoff, lay_aoff;   // host side of the per-read layout arrays of the current batch (h_rlen is one of them)
    std::vector<uint32_t> lay_hcap, lay_order;
    std::vector<uint64_t> h_thr_sub;
    bool thr_precomputed = false;             // k-mer model: the count is L-k+1, so the thresholds went up with the batch
    std::vector<uint64_t> thr_memo;           // k-mer model: threshold by k-mer count (index 0 unused marker = ~0)
    hipStream_t st = nullptr;       // query + CSR assembly; the stream callers synchronise on
    hipStream_t st_sync = nullptr;  // syncmer kernel of the next sub-batch, overlapped with the query of this one
    hipStream_t st_sync2 = nullptr; // its short reads (k_syncmers_wave), concurrent with the long ones on st_sync
    hipEvent_t ev_wave = nullptr;
    hipStream_t st_copy = nullptr;  // H2D of the sub-batches' bases, nothing else (streamed search_batch)
    std::vector<hipEvent_t> ev_sync_done, ev_query_done, ev_pack_done, ev_copy_done;
    hipEvent_t ev_reset = nullptr;
    DBuf<uint32_t> d_sync_cursor;
    Counters *d_ctr = nullptr;
    unsigned long long *d_prof = nullptr;   // TAXOR_PROFILE_PHASES=1: per-phase cycle sums of the two big kernels
    Counters h_ctr{};
    int grid_query_small[MAX_LEVELS] = {};      // single-wave query blocks for levels of narrow IXFs under short reads (0 = not eligible)
    size_t lds_query_small[MAX_LEVELS] = {};
    int grid_wave = 0, grid_wave_overlap = 0;   // k_syncmers_wave: full occupancy / beside a query kernel
    int grid_sync = 0, grid_sync_overlap = 0, grid_query = 0, grid_query_short = 0;   // query blocks: 3 per CU, 4 for short reads
    uint32_t first_div = 1; // first sub-batch = 1/first_div of the others (its syncmer kernel is not hidden)
    bool auto_sub_reads = true; // sub_batch_reads was left to the library: short reads get more of them per sub-batch
    bool prune = true;   // taxor_gpu_search_params::flags & TAXOR_SEARCH_NO_PRUNE disables the threshold-aware pruning (A/B measurements)
    bool group_always = false;   // TAXOR_SEARCH_GROUP_ALWAYS: the queue grouping also for sub-batches of a few thousand reads
    bool split_always = false;   // TAXOR_SEARCH_SPLIT_ALWAYS: root items in column parts whatever the batch size (parity tests)
    bool force_tree_stall = false; // TAXOR_SEARCH_FORCE_TREE_STALL: the one-launch traversal's watchdog fires at once (recovery-path test)
    bool small_path = true;      // !TAXOR_SEARCH_NO_SMALL_PATH: calls of up to a few thousand reads go through the lanes below
    size_t lds_query = 0;

    // batch-resident input
    struct HostSpan { uint64_t vbegin, len; const char *ptr; };   // streamed batch: where the bases of [vbegin, vbegin+len) of the
    std::vector<HostSpan> host_spans;                              // (virtually concatenated) ASCII input live in host memory
    uint64_t n_reads = 0, n_bases = 0, mean_read_len = 1u << 20;
    DBuf<uint8_t> d_ascii;
    DBuf<uint64_t> d_aoff, d_poff, d_hoff;
    DBuf<uint32_t> d_packed, d_rlen, d_hcap, d_nh, d_order;
    DBuf<uint64_t> d_thr;
    std::vector<SubBatch> subs;
    uint64_t max_slots = 0, max_read_slots = 0;
    uint32_t max_sub_reads = 0;
    uint64_t packed_word_count = 0, packed_in_bytes = 0;

    // per-sub-batch scratch
    DBuf<uint64_t> d_cand[2], d_hashes[2];   // double-buffered across sub-batches
    DBuf<uint2> d_q[2], d_qs;     // work queues of two consecutive levels; the next level's queue grouped by IXF
    DBuf<uint32_t> d_qhist;
    DBuf<uint4> d_hits;
    DBuf<uint32_t> d_read_hits, d_cursor, d_roff, d_biglist, d_gtab, d_scan;
    uint32_t q_cap = 0, hit_cap = 0, gtab_stride = 0;

    // batch-resident output
    DBuf<uint64_t> d_read_off;
    DBuf<int64_t> d_out_ub;
    DBuf<uint32_t> d_out_cnt, d_out_key;
    uint64_t tuple_cap = 0;
    bool ran = false, synced = false;

    // host mirrors handed out through taxor_gpu_results
    std::vector<uint64_t> h_read_off, h_hash_off, h_hashes;
    std::vector<int64_t> h_ub;
    std::vector<uint32_t> h_cnt, h_nh;
    void *h_small = nullptr;       // page-locked landing area for the results of SMALL batches (see taxor_gpu_batch_fetch)
    void *h_small_in = nullptr;    // page-locked staging of a small batch's per-read arrays (prepare_batch)

    // small batches (a call of up to SMALL_MAX_READS reads, e.g. the reference's 1024-record chunk): pieces of the call run on
    // LANES -- internal searchers with one stream each -- see small_begin()
    struct SmallLane {
        taxor_gpu_searcher *c = nullptr;    // the lane: streams, counters, queues, hit buffers of its own
        hipEvent_t done = nullptr, copied = nullptr;   // piece finished / its bases are on the device
        DBuf<uint8_t> d_in;                 // aoff | poff | hoff | rlen | hcap | order of the piece: ONE host-to-device copy
        void *h_in = nullptr;               // its page-locked source
        size_t h_in_cap = 0;
        void *h_bases = nullptr;            // TAXOR_SMALL_STAGE=1 (experiment, profiles/r05/small_calls.txt): the piece's bases copied here by the
        size_t h_bases_cap = 0;             // calling thread, so that their host-to-device copy is asynchronous
        void *h_out = nullptr, *d_out = nullptr;   // results in host memory the device writes: status | read_off | nh | ub | cnt
        uint32_t out_reads = 0, out_tuples = 0;    // capacities of that area
        bool fresh = true;                  // counters / hit counts not known to be zero: clear them before the next piece
        bool q_dirty = false;               // the queue's slots are not known to be empty (a level-by-level piece used it): ~0 before a tree launch
    };
    struct SmallPiece { uint32_t lane; uint64_t first, n; };
    std::vector<SmallLane> lanes;
    std::vector<SmallPiece> small_pieces;
    const char *small_bases = nullptr;
    const uint64_t *small_offsets = nullptr;
    bool lane_mode = false;                 // this searcher IS a lane
    bool st_borrowed = false;               // ... working on a stream that belongs to the searcher it serves
    bool small_active = false, small_done = false;   // the call in flight went through the lanes / its results are in the host arrays
    size_t small_harvested = 0;             // pieces whose results have been appended to the host arrays
    uint64_t small_tbase = 0;
    // (Tried: a second host thread enqueueing the odd pieces, so that one blocking copy of pageable bases page-locks the caller's
    // pages while the other's bytes are on the wire -- the runtime serialises pageable copies process-wide; the helper's first
    // copy returned after both of the caller's, 350 us into a 390-us enqueue.)
    bool dev_results_stale = false;         // ... and not (yet) in the device-resident CSR that export_device / the communicator read

    // timing
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, int>> ev_spans; // (start event index, kind 0=syncmer 2=finalize 3=whole run, 16+l = query level l)
    taxor_gpu_run_stats stats{};
};

// The HIP runtime multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues per device (default 4), and two streams
// that share a queue execute in submission order.  A searcher's pipeline lives on concurrency between its streams -- the
// copy stream ahead of everything, pack + syncmers of sub-batch i+1 beside the query of sub-batch i -- and with the null
// stream a single streamed searcher already has five: measured on the streamed single call, the copy stream shared the
// query stream's queue, every copy waited for the previous sub-batch's query chain and every syncmer launch ran exposed
// (60 -> 55 ms per 1.31 Gbp with 8 or 16 queues, profiles/r03/single_call_hw_queues.txt; with 32 or 64 the whole pipeline
// loses a third -- more queues than the hardware has slots are time-sliced).  The runtime reads the variable
// once, when it initialises: set it here, before this library's first HIP call, unless the user has chosen a value.  In a
// process whose runtime is already up (another library used HIP first) this has no effect; export it there.
static void runtime_env_once()
{
    static const bool done = [] { setenv("GPU_MAX_HW_QUEUES", "8", 0); return true; }();   // 32 and more: measured harmful (queues time-sliced)
    (void)done;
}

extern "C" __attribute__((visibility("hidden"))) void taxor_runtime_env_once() { runtime_env_once(); }

extern "C" int taxor_gpu_host_register(void *ptr, uint64_t bytes)
{
    runtime_env_once();
    if (!ptr || !bytes) return fail(TAXOR_E_ARG, "taxor_gpu_host_register: null buffer");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return TAXOR_OK;
}
extern "C" int taxor_gpu_host_unregister(void *ptr)
{
    if (!ptr) return fail(TAXOR_E_ARG, "taxor_gpu_host_unregister: null buffer");
    HIP_TRY(hipHostUnregister(ptr));
    return TAXOR_OK;
}

extern "C" const char *taxor_gpu_last_error(void) { return g_err.c_str(); }
extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg) { g_err = msg ? msg : ""; }

// =========================================================================================================
// index
// =========================================================================================================
// ---------------------------------------------------------------------------------------------------------
// Fingerprint bytes -> the slab.  Two routes:
//   * host pointers (view->source == NULL): one hipMemcpy per IXF piece from the caller's memory, in slab order (the
//     runtime's pageable path; page-locked staging of our own was measured slower for such memory in round 2);
//   * a source (a .hixf on disk / tmpfs, taxor_hixf_load): `threads` workers, each with two page-locked staging
//     buffers and a stream of its own, take pieces off a shared cursor: read() into one buffer while the other is in
//     flight.  No host mapping of the data is ever touched, so there is nothing to fault in and nothing to unmap.
// progress(ctx, b): every byte of the slab below offset b is final (pieces complete out of order; b is the contiguous
// prefix).  Used by the communicator to broadcast behind the upload.
// ---------------------------------------------------------------------------------------------------------
extern "C" __attribute__((visibility("hidden"))) int taxor_index_upload_relayout(taxor_gpu_index *idx, const taxor_hixf_view *v);   // relayout.hip

// the source's bytes are in the search layout (data[row * stride + bin] at the index's own strides): uploaded as they are
static bool view_is_search_layout(const taxor_hixf_view *v)
{
    if (taxor::ixf_layout_kind(v->ixf_layout) != taxor::IXF_KIND_ROWS || (v->ixf_layout & taxor::IXF_ROWS_POSITION_MAJOR)) return false;
    for (uint64_t i = 0; i < v->n_ixf; ++i)       // (the pitch a CODE names counts too: 0x200 "unpadded" with src_stride left 0 is not the search layout)
        if (taxor::ixf_src_pitch(v->ixf_layout, v->ixf[i].src_stride, v->ixf[i].stride, v->ixf[i].bins) != v->ixf[i].stride) return false;
    return true;
}

static int index_upload(taxor_gpu_index *idx, const taxor_hixf_view *v, void (*progress)(void *, uint64_t), void *pctx)
{
    if (!view_is_search_layout(v)) {
        // another writer's layout (ixf_layout.h): transposed on the device while it is uploaded (relayout.hip).  The slab is final
        // only when the last chunk has landed (a bin-major chunk writes a column range of every row), so progress is one step
        if (int rc = taxor_index_upload_relayout(idx, v)) return rc;
        if (progress) progress(pctx, idx->slab_bytes);
        return 0;
    }
    struct Piece { uint64_t ixf, off, len, slab_end; };
    static const uint64_t piece_bytes = [] { const char *e = tune_env("TAXOR_UPLOAD_PIECE_MB"); const long m = e ? atol(e) : 0; return (uint64_t)(m > 0 ? m : 8) << 20; }();
    const uint64_t n = v->n_ixf;
    std::vector<Piece> pieces;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t bytes = idx->rows[i] * idx->h_ixf[i].stride;
        const uint64_t next = i + 1 < n ? idx->slab_off[i + 1] : idx->slab_bytes;
        if (!v->source && !v->ixf[i].data) continue;                     // left to fill_random / upload_bin / the builder
        const uint64_t step = v->source ? piece_bytes : (1ull << 30);
        for (uint64_t o = 0; o < bytes; o += step) {
            const uint64_t len = std::min(step, bytes - o);
            pieces.push_back({i, o, len, o + len == bytes ? next : idx->slab_off[i] + o + len});
        }
    }
    if (pieces.empty()) { if (progress) progress(pctx, idx->slab_bytes); return 0; }
    static const bool trace_up = tune_env("TAXOR_TRACE_UPLOAD") != nullptr;
    const auto up_t0 = std::chrono::steady_clock::now();
    struct UpTrace {
        bool on; std::chrono::steady_clock::time_point t0; const std::vector<Piece> &pc; bool src;
        ~UpTrace()
        {
            if (!on) return;
            uint64_t b = 0;
            for (const Piece &p : pc) b += p.len;
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "[upload] %.2f GB in %.3f s = %.1f GB/s (%s)\n", b / 1e9, dt, b / 1e9 / dt, src ? "source reader, page-locked staging" : "host pointers, runtime pageable path");
        }
    } up_trace{trace_up, up_t0, pieces, v->source != nullptr};
    if (!v->source) {
        // The runtime copies large pageable buffers by pinning the caller's pages in place and letting the DMA engines read
        // them -- no CPU copy, but every page has to be present in this process's page tables first, and the runtime's
        // copy thread takes those faults one 4-KiB page at a time.  For a mapped file of tens of gigabytes that is most of
        // the time; a few helper threads populate the page tables ahead of the copy (MADV_POPULATE_READ, Linux >= 5.14;
        // where the kernel does not know it the call fails and the copy faults the pages in as before).
        static const int pf_threads = [] { const char *e = tune_env("TAXOR_UPLOAD_PREFAULT"); const int t = e ? atoi(e) : -1; return t >= 0 && t <= 64 ? t : 8; }();
        uint64_t total = 0;
        for (const Piece &p : pieces) total += p.len;
        std::atomic<size_t> pf_cursor{0};
        std::atomic<bool> pf_stop{false};
        std::vector<std::thread> pf;
        struct Slice { const uint8_t *p; uint64_t len; };
        std::vector<Slice> slices;
        if (pf_threads > 0 && total >= (256ull << 20)) {
            const uint64_t sl = 64ull << 20;
            for (const Piece &p : pieces)
                for (uint64_t o = 0; o < p.len; o += sl) slices.push_back({v->ixf[p.ixf].data + p.off + o, std::min(sl, p.len - o)});
            for (int t = 0; t < pf_threads; ++t)
                pf.emplace_back([&] {
                    for (;;) {
                        const size_t i = pf_cursor.fetch_add(1);
                        if (i >= slices.size() || pf_stop.load()) break;
                        const uintptr_t a = (uintptr_t)slices[i].p & ~(uintptr_t)4095, e = ((uintptr_t)slices[i].p + slices[i].len + 4095) & ~(uintptr_t)4095;
#ifdef MADV_POPULATE_READ
                        if (madvise((void *)a, e - a, MADV_POPULATE_READ) != 0) { pf_stop.store(true); break; }
#else
                        (void)a; (void)e; pf_stop.store(true); break;
#endif
                    }
                });
        }
        auto join_pf = [&] { pf_stop.store(true); for (auto &t : pf) t.join(); };
        for (const Piece &p : pieces) {
            const hipError_t e = hipMemcpy(idx->d_slab + idx->slab_off[p.ixf] + p.off, v->ixf[p.ixf].data + p.off, p.len, hipMemcpyHostToDevice);
            if (e != hipSuccess) { join_pf(); return fail(TAXOR_E_HIP, "index upload of IXF %llu failed: %s", (unsigned long long)p.ixf, hipGetErrorString(e)); }
            if (progress) progress(pctx, p.slab_end);
        }
        join_pf();
        if (progress) progress(pctx, idx->slab_bytes);
        return 0;
    }
    static const int n_threads = [] { const char *e = tune_env("TAXOR_UPLOAD_THREADS"); const int t = e ? atoi(e) : 0; return t >= 1 && t <= 64 ? t : 8; }();
    const int T = (int)std::min<size_t>((size_t)n_threads, pieces.size());
    std::atomic<size_t> cursor{0};
    std::vector<std::atomic<uint8_t>> done(pieces.size());
    for (auto &d : done) d.store(0);
    std::atomic<int> failed{0};
    std::mutex mu;                      // progress bookkeeping + the first error text
    size_t prefix = 0;
    std::string err;
    auto mark = [&](size_t i) {
        done[i].store(1);
        std::lock_guard<std::mutex> lk(mu);
        bool moved = false;
        while (prefix < pieces.size() && done[prefix].load()) { ++prefix; moved = true; }
        if (moved && progress) progress(pctx, prefix == pieces.size() ? idx->slab_bytes : pieces[prefix - 1].slab_end);
    };
    auto set_err = [&](const std::string &m) {
        std::lock_guard<std::mutex> lk(mu);
        if (err.empty()) err = m;
        failed.store(1);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&] {
            hipStream_t st = nullptr;
            void *buf[2] = {nullptr, nullptr};
            hipEvent_t ev[2] = {nullptr, nullptr};
            size_t inflight[2] = {(size_t)-1, (size_t)-1};
            hipError_t e = hipSetDevice(idx->device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) {
                e = hipHostMalloc(&buf[b], piece_bytes, hipHostMallocDefault);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
            }
            if (e != hipSuccess) set_err(std::string("index upload: staging buffers: ") + hipGetErrorString(e));
            int b = 0;
            while (!failed.load()) {
                const size_t i = cursor.fetch_add(1);
                if (i >= pieces.size()) break;
                const Piece &p = pieces[i];
                if (inflight[b] != (size_t)-1) {                 // this buffer's previous copy must have left it
                    if ((e = hipEventSynchronize(ev[b])) != hipSuccess) { set_err(std::string("index upload: ") + hipGetErrorString(e)); break; }
                    mark(inflight[b]);
                    inflight[b] = (size_t)-1;
                }
                if (v->source->read(v->source->ctx, p.ixf, p.off, p.len, buf[b]) != 0) {
                    set_err("index upload: reading IXF " + std::to_string(p.ixf) + " at byte " + std::to_string(p.off) + " from its source failed");
                    break;
                }
                e = hipMemcpyAsync(idx->d_slab + idx->slab_off[p.ixf] + p.off, buf[b], p.len, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) e = hipEventRecord(ev[b], st);
                if (e != hipSuccess) { set_err(std::string("index upload: ") + hipGetErrorString(e)); break; }
                inflight[b] = i;
                b ^= 1;
            }
            if (st) (void)hipStreamSynchronize(st);
            for (int k = 0; k < 2; ++k) {
                if (inflight[k] != (size_t)-1 && !failed.load()) mark(inflight[k]);
                if (ev[k]) (void)hipEventDestroy(ev[k]);
                if (buf[k]) (void)hipHostFree(buf[k]);
            }
            if (st) (void)hipStreamDestroy(st);
        });
    for (auto &t : th) t.join();
    if (failed.load()) return fail(TAXOR_E_IO, "%s", err.c_str());
    return 0;
}

// upload = false: everything but the fingerprint bytes (the slab is allocated, its rows are left as they are) -- for a
// replica that receives them over RCCL, or through a pipelined upload (comm.hip)
static int index_create_impl(const taxor_hixf_view *v, int device, bool upload, taxor_gpu_index **out)
{
    runtime_env_once();
    if (!v || !out || v->n_ixf == 0 || !v->ixf) return fail(TAXOR_E_ARG, "index_create: empty view");
    if (!v->use_syncmer) { // seqan3 minimiser_hash over window_size (taxor_search.cpp:210-212)
        if (v->kmer_size < 1 || v->kmer_size > 32)
            return fail(TAXOR_E_ARG, "index_create: k-mer size %u outside [1,32]", (unsigned)v->kmer_size);
        if (v->window_size < v->kmer_size || v->window_size - v->kmer_size + 1 > 512)
            return fail(TAXOR_E_ARG, "index_create: window size %llu must be in [k, k+511]", (unsigned long long)v->window_size);
    }
    const int k = v->kmer_size, s = v->syncmer_size, t = v->t_syncmer;
    if (v->use_syncmer && (k < 2 || k > 32 || s < 1 || s > 16 || s >= k || k - s + 1 > 32 || t < 1))
        return fail(TAXOR_E_ARG, "index_create: unsupported k=%d s=%d t=%d (need k<=32, s<=16, s<k, t>=1)", k, s, t);
    if (v->n_ixf >= (1u << 30)) return fail(TAXOR_E_ARG, "index_create: too many IXFs");
    if (!taxor::ixf_layout_valid(v->ixf_layout)) return fail(TAXOR_E_ARG, "index_create: unknown fingerprint layout code %u", v->ixf_layout);
    HIP_TRY(hipSetDevice(device));

    auto idx = new taxor_gpu_index();
    idx->device = device;
    idx->k = k;
    idx->s = s;
    idx->t = t;
    idx->scaling = v->scaling ? v->scaling : 1;
    idx->w_min = v->use_syncmer ? 0 : (int)v->window_size;
    idx->n_user_bins = v->n_user_bins;
    const uint64_t n = v->n_ixf;
    idx->h_ixf.resize(n);
    idx->rows.resize(n);
    std::vector<uint64_t> slab_off(n);
    uint64_t off = 0, tb = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const taxor_ixf_view &f = v->ixf[i];
        if (f.bins == 0 || f.stride < f.bins || f.stride % 64 != 0 || f.seg_len == 0 || !f.next_ixf || !f.fname_idx) {
            delete idx;
            return fail(TAXOR_E_ARG, "index_create: IXF %llu malformed (bins=%llu stride=%llu seg_len=%llu)",
                        (unsigned long long)i, (unsigned long long)f.bins, (unsigned long long)f.stride,
                        (unsigned long long)f.seg_len);
        }
        const uint64_t src_pitch = taxor::ixf_src_pitch(v->ixf_layout, f.src_stride, f.stride, f.bins);
        if (src_pitch < f.bins && taxor::ixf_layout_kind(v->ixf_layout) != taxor::IXF_KIND_BIT_SLICED) {
            delete idx;
            if (src_pitch == 0)
                return fail(TAXOR_E_ARG, "index_create: IXF %llu: the layout code names the record's stored pitch but src_stride is 0", (unsigned long long)i);
            return fail(TAXOR_E_ARG, "index_create: IXF %llu: source pitch %llu below its %llu bins", (unsigned long long)i, (unsigned long long)src_pitch,
                        (unsigned long long)f.bins);
        }
        if (3 * f.seg_len >= (1ull << 32) || f.stride > (1u << 20)) {
            delete idx;
            return fail(TAXOR_E_ARG, "index_create: IXF %llu too large (rows or stride)", (unsigned long long)i);
        }
        idx->rows[i] = 3 * f.seg_len;
        slab_off[i] = off;
        off += round_up(idx->rows[i] * f.stride, 4096);
        idx->data_bytes += idx->rows[i] * f.stride;
        IxfDesc &d = idx->h_ixf[i];
        d.seed = f.seed;
        d.seg_len = (uint32_t)f.seg_len;
        d.bins = (uint32_t)f.bins;
        d.stride = (uint32_t)f.stride;
        d.units = (uint32_t)(f.stride / 16);
        d.bin_base = (uint32_t)tb;
        d.arith = v->ixf_arith;
        tb += f.bins;
        idx->max_stride = std::max(idx->max_stride, d.stride);
    }
    if (tb >= (1ull << 32)) {
        delete idx;
        return fail(TAXOR_E_ARG, "index_create: more than 2^32 technical bins");
    }
    idx->total_bins = tb;
    idx->slab_bytes = off + 4096;

    // per-bin tables + DFS keys + depth; validates that the IXFs form a tree rooted at 0
    std::vector<uint32_t> binfo(tb), dfs(tb, 0);
    std::vector<int64_t> ubin(tb, -1);
    for (uint64_t i = 0; i < n; ++i) {
        const taxor_ixf_view &f = v->ixf[i];
        const uint32_t bb = idx->h_ixf[i].bin_base;
        for (uint64_t b = 0; b < f.bins; ++b) {
            const int64_t cur = f.fname_idx[b];
            uint32_t info = 0;
            if (cur < 0) { // merged bin (hixf.hpp:172-178); child = next_ixf_id[i][bin] (:115-122)
                const int64_t ch = f.next_ixf[b];
                if (ch <= 0 || (uint64_t)ch >= n || (uint64_t)ch == i) {
                    delete idx;
                    return fail(TAXOR_E_ARG, "index_create: IXF %llu bin %llu: bad child %lld", (unsigned long long)i,
                                (unsigned long long)b, (long long)ch);
                }
                info = BINFO_MERGED | BINFO_END | (uint32_t)ch;
            } else {
                if ((uint64_t)cur >= v->n_user_bins) {
                    delete idx;
                    return fail(TAXOR_E_ARG, "index_create: IXF %llu bin %llu: user bin %lld out of range",
                                (unsigned long long)i, (unsigned long long)b, (long long)cur);
                }
                ubin[bb + b] = cur;
                if (b + 1 == f.bins || cur != f.fname_idx[b + 1]) { // hixf.hpp:325-326
                    info = BINFO_END;
                    idx->leaf_runs++;
                }
            }
            binfo[bb + b] = info;
        }
    }
    {   // where the root's rows may be cut into column parts: at a multiple of 16 bins whose predecessor ends a run (a merged bin
        // is a run of its own), as close to the equal division as the layout allows; parts narrower than 128 B are not made
        const uint32_t U = idx->h_ixf[0].units, B = idx->h_ixf[0].bins;
        for (uint32_t P = 8; P >= 2 && idx->root_pmax == 1; P >>= 1) {
            if (U / P < 8 || U > 60000) continue;
            uint16_t cut[9];
            cut[0] = 0;
            cut[P] = (uint16_t)U;
            bool ok = true;
            for (uint32_t j = 1; j < P && ok; ++j) {
                const uint32_t ideal = (uint32_t)((uint64_t)j * U / P);
                const uint32_t slack = U / P / 4;
                ok = false;
                for (uint32_t d = 0; d <= slack && !ok; ++d)
                    for (int sgn = -1; sgn <= 1 && !ok; sgn += 2) {
                        const uint32_t u = sgn < 0 ? ideal - d : ideal + d;
                        if (u <= cut[j - 1] || u >= U || (uint64_t)u * 16 >= B) continue;
                        if (binfo[(size_t)u * 16 - 1] & BINFO_END) { cut[j] = (uint16_t)u; ok = true; }
                    }
            }
            if (ok) {
                idx->root_pmax = P;
                for (uint32_t j = 0; j <= P; ++j) idx->root_cut[j] = cut[j];
            }
        }
    }
    idx->h_binfo = binfo;
    idx->h_bin_base.resize(n + 1);
    for (uint64_t i = 0; i < n; ++i) idx->h_bin_base[i] = idx->h_ixf[i].bin_base;
    idx->h_bin_base[n] = (uint32_t)tb;
    {
        std::vector<uint8_t> seen(n, 0);
        struct Frame { uint64_t ixf, bin; uint32_t depth; };
        std::vector<Frame> stack;
        stack.push_back({0, 0, 1});
        seen[0] = 1;
        uint32_t key = 0;
        while (!stack.empty()) {
            Frame &fr = stack.back();
            const taxor_ixf_view &f = v->ixf[fr.ixf];
            if (fr.bin == f.bins) { stack.pop_back(); continue; }
            const uint64_t b = fr.bin++;
            const uint32_t g = idx->h_ixf[fr.ixf].bin_base + (uint32_t)b;
            dfs[g] = key++;
            idx->depth = std::max(idx->depth, fr.depth);
            if (fr.depth - 1 < (uint32_t)MAX_LEVELS)
                idx->lvl_max_stride[fr.depth - 1] = std::max(idx->lvl_max_stride[fr.depth - 1], idx->h_ixf[fr.ixf].stride);
            if (binfo[g] & BINFO_MERGED) {
                const uint64_t ch = binfo[g] & 0x3FFFFFFFu;
                if (seen[ch]) {
                    delete idx;
                    return fail(TAXOR_E_ARG, "index_create: IXF %llu is referenced twice (not a tree)", (unsigned long long)ch);
                }
                seen[ch] = 1;
                const uint32_t d = fr.depth + 1;
                stack.push_back({ch, 0, d});
            }
        }
        if (idx->depth >= (uint32_t)MAX_LEVELS - 1) {
            delete idx;
            return fail(TAXOR_E_ARG, "index_create: hierarchy deeper than %d levels", MAX_LEVELS - 2);
        }
    }

    hipError_t e = hipMalloc((void **)&idx->d_slab, idx->slab_bytes);
    if (e != hipSuccess) {
        delete idx;
        return fail(TAXOR_E_NOMEM, "index_create: hipMalloc of %llu bytes failed: %s", (unsigned long long)idx->slab_bytes,
                    hipGetErrorString(e));
    }
    for (uint64_t i = 0; i < n; ++i) idx->h_ixf[i].data = idx->d_slab + slab_off[i];
    idx->slab_off = slab_off;
    if (upload)
        if (int rc = index_upload(idx, v, nullptr, nullptr)) {
            taxor_gpu_index_destroy(idx);
            return rc;
        }
    bool ok = hipMalloc((void **)&idx->d_ixf, n * sizeof(IxfDesc)) == hipSuccess &&
              hipMalloc((void **)&idx->d_binfo, tb * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc((void **)&idx->d_ubin, tb * sizeof(int64_t)) == hipSuccess &&
              hipMalloc((void **)&idx->d_dfs_key, tb * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemcpy(idx->d_ixf, idx->h_ixf.data(), n * sizeof(IxfDesc), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(idx->d_binfo, binfo.data(), tb * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(idx->d_ubin, ubin.data(), tb * sizeof(int64_t), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(idx->d_dfs_key, dfs.data(), tb * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        taxor_gpu_index_destroy(idx);
        return fail(TAXOR_E_HIP, "index_create: table upload failed");
    }
    idx->h_ubin = std::move(ubin);
    idx->h_dfs = std::move(dfs);
    *out = idx;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_index_create(const taxor_hixf_view *v, int device, taxor_gpu_index **out)
{
    return index_create_impl(v, device, true, out);
}

// library-internal (comm.hip): an index on `device` with every table in place and an allocated but unwritten slab
extern "C" __attribute__((visibility("hidden"))) int taxor_index_create_empty(const taxor_hixf_view *v, int device, taxor_gpu_index **out)
{
    return index_create_impl(v, device, false, out);
}

// library-internal (comm.hip): the upload alone, into an index made by taxor_index_create_empty, reporting progress
extern "C" __attribute__((visibility("hidden"))) int taxor_index_upload(taxor_gpu_index *idx, const taxor_hixf_view *v,
                                                                        void (*progress)(void *, uint64_t), void *ctx)
{
    if (!idx || !v || v->n_ixf != idx->h_ixf.size()) return fail(TAXOR_E_ARG, "index_upload: view does not match the index");
    HIP_TRY(hipSetDevice(idx->device));
    return index_upload(idx, v, progress, ctx);
}

extern "C" __attribute__((visibility("hidden"))) int taxor_index_slab(taxor_gpu_index *idx, uint8_t **slab, uint64_t *slab_bytes,
                                                                      const uint64_t **ixf_off, uint64_t *n_ixf, int *device)
{
    if (!idx) return -1;
    *slab = idx->d_slab;
    *slab_bytes = idx->slab_bytes;
    *ixf_off = idx->slab_off.data();
    *n_ixf = idx->h_ixf.size();
    *device = idx->device;
    return 0;
}

// library-internal (builder.hip): the builder's per-index context
extern "C" __attribute__((visibility("hidden"))) void *taxor_index_build_ctx(taxor_gpu_index *idx) { return idx ? idx->build_ctx : nullptr; }
extern "C" __attribute__((visibility("hidden"))) void taxor_index_set_build_ctx(taxor_gpu_index *idx, void *ctx, void (*free_fn)(void *))
{
    if (!idx) return;
    if (idx->build_ctx && idx->build_ctx_free && idx->build_ctx != ctx) idx->build_ctx_free(idx->build_ctx);
    idx->build_ctx = ctx;
    idx->build_ctx_free = free_fn;
}

extern "C" void taxor_gpu_index_destroy(taxor_gpu_index *idx)
{
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    if (idx->build_ctx && idx->build_ctx_free) idx->build_ctx_free(idx->build_ctx);
    idx->build_ctx = nullptr;
    if (idx->d_slab) (void)hipFree(idx->d_slab);
    if (idx->d_ixf) (void)hipFree(idx->d_ixf);
    if (idx->d_binfo) (void)hipFree(idx->d_binfo);
    if (idx->d_ubin) (void)hipFree(idx->d_ubin);
    if (idx->d_dfs_key) (void)hipFree(idx->d_dfs_key);
    delete idx;
}

// library-internal accessors for builder.hip
extern "C" __attribute__((visibility("hidden"))) int taxor_index_ixf_info(taxor_gpu_index *idx, uint64_t ixf, uint8_t **data,
                                                                          uint64_t *stride, uint64_t *seg_len, uint64_t *bins,
                                                                          int *device)
{
    if (!idx || ixf >= idx->h_ixf.size()) return -1;
    *data = const_cast<uint8_t *>(idx->h_ixf[ixf].data);
    *stride = idx->h_ixf[ixf].stride;
    *seg_len = idx->h_ixf[ixf].seg_len;
    *bins = idx->h_ixf[ixf].bins;
    *device = idx->device;
    return 0;
}

extern "C" __attribute__((visibility("hidden"))) int taxor_index_tree(taxor_gpu_index *idx, uint64_t *n_ixf, const uint32_t **bin_base,
                                                                      const uint32_t **binfo)
{
    if (!idx) return -1;
    *n_ixf = idx->h_ixf.size();
    *bin_base = idx->h_bin_base.data();
    *binfo = idx->h_binfo.data();
    return 0;
}

extern "C" __attribute__((visibility("hidden"))) int taxor_index_set_seed(taxor_gpu_index *idx, uint64_t ixf, uint64_t seed)
{
    if (!idx || ixf >= idx->h_ixf.size()) return -1;
    idx->h_ixf[ixf].seed = seed;
    return hipMemcpy(&idx->d_ixf[ixf], &idx->h_ixf[ixf], sizeof(IxfDesc), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

extern "C" __attribute__((visibility("hidden"))) uint32_t taxor_index_arith(const taxor_gpu_index *idx)
{
    return idx && !idx->h_ixf.empty() ? idx->h_ixf[0].arith : 0u;
}

extern "C" uint64_t taxor_gpu_index_data_bytes(const taxor_gpu_index *idx) { return idx ? idx->data_bytes : 0; }
extern "C" uint64_t taxor_gpu_index_ixf_seed(const taxor_gpu_index *idx, uint64_t ixf)
{
    return idx && ixf < idx->h_ixf.size() ? idx->h_ixf[ixf].seed : 0;
}
extern "C" uint64_t taxor_gpu_index_leaf_runs(const taxor_gpu_index *idx) { return idx ? idx->leaf_runs : 0; }
extern "C" uint32_t taxor_gpu_index_depth(const taxor_gpu_index *idx) { return idx ? idx->depth : 0; }

extern "C" int taxor_gpu_index_fill_random(taxor_gpu_index *idx, uint64_t ixf, uint64_t seed)
{
    if (!idx || ixf >= idx->h_ixf.size()) return fail(TAXOR_E_ARG, "fill_random: bad IXF id");
    HIP_TRY(hipSetDevice(idx->device));
    const uint64_t bytes = idx->rows[ixf] * idx->h_ixf[ixf].stride; // multiple of 64
    launch_fill_random(const_cast<uint8_t *>(idx->h_ixf[ixf].data), bytes, seed, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return TAXOR_OK;
}

extern "C" int taxor_gpu_gather_ceiling(taxor_gpu_index *idx, uint64_t ixf, uint64_t want_bytes, int reps, double *gb_per_s,
                                        uint64_t *row_bytes)
{
    return taxor_gpu_gather_ceiling_span(idx, ixf, 1, want_bytes, reps, gb_per_s, row_bytes, nullptr);
}

extern "C" int taxor_gpu_gather_ceiling_span(taxor_gpu_index *idx, uint64_t ixf, uint64_t n_ixf, uint64_t want_bytes, int reps,
                                             double *gb_per_s, uint64_t *row_bytes, uint64_t *span_used)
{
    if (!idx || ixf >= idx->h_ixf.size() || !gb_per_s || !n_ixf) return fail(TAXOR_E_ARG, "gather_ceiling: bad argument");
    const IxfDesc &f = idx->h_ixf[ixf];
    // the longest run of IXFs from `ixf` on that share its shape and lie equally spaced in the slab
    uint64_t span = 1, spacing = 0;
    if (n_ixf > 1 && ixf + 1 < idx->h_ixf.size()) {
        spacing = (uint64_t)(idx->h_ixf[ixf + 1].data - f.data);
        while (span < n_ixf && ixf + span < idx->h_ixf.size()) {
            const IxfDesc &g = idx->h_ixf[ixf + span];
            if (g.bins != f.bins || g.stride != f.stride || g.seg_len != f.seg_len || (uint64_t)(g.data - f.data) != span * spacing) break;
            ++span;
        }
    }
    if (span_used) *span_used = span;
    const uint32_t units = (f.bins + 15) / 16;
    if (units > 256) return fail(TAXOR_E_ARG, "gather_ceiling: rows wider than 4096 bins");
    if (reps < 1) reps = 1;
    HIP_TRY(hipSetDevice(idx->device));
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc((void **)&sink, 4));
    HIP_TRY(hipMemset(sink, 0, 4));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    static const bool nt = [] { const char *e = tune_env("TAXOR_QUERY_NT"); return !e || atoi(e) != 0; }();
    uint64_t bytes = launch_gather_ceiling(f.data, idx->rows[ixf], (uint32_t)f.stride, f.bins, want_bytes, 1, sink, nt, nullptr, (uint32_t)span, spacing);   // warm-up
    HIP_TRY(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r)
        bytes = launch_gather_ceiling(f.data, idx->rows[ixf], (uint32_t)f.stride, f.bins, want_bytes, 2 + r, sink, nt, nullptr, (uint32_t)span, spacing);
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    *gb_per_s = ms > 0.f ? (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
    if (row_bytes) *row_bytes = (uint64_t)units * 16;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_gather_pattern(taxor_gpu_index *idx, uint64_t ixf, int pattern, int nt, uint64_t want_bytes, int reps,
                                        double *gb_per_s, uint64_t *bytes_per_launch, uint64_t *requests_per_launch)
{
    if (!idx || ixf >= idx->h_ixf.size() || !gb_per_s || (pattern != 0 && pattern != 1)) return fail(TAXOR_E_ARG, "gather_pattern: bad argument");
    const IxfDesc &f = idx->h_ixf[ixf];
    const uint32_t units = (f.bins + 15) / 16;
    if (units > 256) return fail(TAXOR_E_ARG, "gather_pattern: rows wider than 4096 bins");
    if (reps < 1) reps = 1;
    HIP_TRY(hipSetDevice(idx->device));
    uint32_t *sink = nullptr;
    HIP_TRY(hipMalloc((void **)&sink, 4));
    HIP_TRY(hipMemset(sink, 0, 4));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    uint64_t bytes = 0, reqs = 0;
    auto launch = [&](uint64_t seed) {
        if (pattern == 0) {
            bytes = launch_gather_ceiling(f.data, idx->rows[ixf], (uint32_t)f.stride, f.bins, want_bytes, seed, sink, nt != 0, nullptr);
            reqs = bytes / ((uint64_t)units * 16);
        } else {
            reqs = launch_gather_sparse(f.data, idx->rows[ixf], (uint32_t)f.stride, f.bins, want_bytes / 16, seed, sink, nt != 0, nullptr);
            bytes = reqs * 16;
        }
    };
    launch(1);   // warm-up
    HIP_TRY(hipEventRecord(e0, nullptr));
    for (int r = 0; r < reps; ++r) launch(2 + (uint64_t)r);
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipGetLastError());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    *gb_per_s = ms > 0.f ? (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
    if (bytes_per_launch) *bytes_per_launch = bytes;
    if (requests_per_launch) *requests_per_launch = reqs;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_index_upload_bin(taxor_gpu_index *idx, uint64_t ixf, uint64_t bin, const uint8_t *column,
                                          uint64_t rows)
{
    if (!idx || ixf >= idx->h_ixf.size() || bin >= idx->h_ixf[ixf].bins || rows != idx->rows[ixf] || !column)
        return fail(TAXOR_E_ARG, "upload_bin: bad arguments");
    HIP_TRY(hipSetDevice(idx->device));
    uint8_t *d_col = nullptr;
    HIP_TRY(hipMalloc((void **)&d_col, rows));
    hipError_t e = hipMemcpy(d_col, column, rows, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_scatter_column(const_cast<uint8_t *>(idx->h_ixf[ixf].data), idx->h_ixf[ixf].stride, bin, d_col, rows, nullptr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_col);
    if (e != hipSuccess) return fail(TAXOR_E_HIP, "upload_bin: %s", hipGetErrorString(e));
    return TAXOR_OK;
}

extern "C" int taxor_gpu_index_download_ixf(const taxor_gpu_index *idx, uint64_t ixf, uint8_t *data, uint64_t len)
{
    if (!idx || ixf >= idx->h_ixf.size() || !data || len != idx->rows[ixf] * idx->h_ixf[ixf].stride)
        return fail(TAXOR_E_ARG, "download_ixf: bad arguments");
    HIP_TRY(hipSetDevice(idx->device));
    HIP_TRY(hipMemcpy(data, idx->h_ixf[ixf].data, len, hipMemcpyDeviceToHost));
    return TAXOR_OK;
}

// =========================================================================================================
// searcher
// =========================================================================================================
static int searcher_create_impl(taxor_gpu_index *idx, const taxor_gpu_search_params *prm, taxor_gpu_searcher **out, hipStream_t lane_stream);

extern "C" int taxor_gpu_searcher_create(taxor_gpu_index *idx, const taxor_gpu_search_params *prm,
                                         taxor_gpu_searcher **out)
{
    return searcher_create_impl(idx, prm, out, nullptr);
}

// lane_stream != nullptr: the searcher becomes a lane of another one (small batches) and works on that stream, which stays the
// other searcher's; it gets no second stream of its own (every stream a process holds is a candidate to share a hardware queue
// with -- runtime_env_once -- and four lanes with two streams each were measured to run two of their pieces one after the other)
static int searcher_create_impl(taxor_gpu_index *idx, const taxor_gpu_search_params *prm, taxor_gpu_searcher **out, hipStream_t lane_stream)
{
    if (!idx || !prm || !out) return fail(TAXOR_E_ARG, "searcher_create: null argument");
    const bool by_ratio = prm->model == TAXOR_THR_PERCENTAGE || prm->model == TAXOR_THR_SYNCMER;
    if (by_ratio && (!(prm->ratio >= 0.0) || !(prm->ratio <= 1.0)))
        return fail(TAXOR_E_ARG, "searcher_create: threshold ratio %g outside [0,1] (error rate / k outside the model?)", prm->ratio);
    if (prm->model > TAXOR_THR_FRACMINHASH) return fail(TAXOR_E_ARG, "searcher_create: unknown threshold model %u", prm->model);
    if (!by_ratio && (!(prm->error_rate > 0.0) || !(prm->error_rate < 1.0)))
        return fail(TAXOR_E_ARG, "searcher_create: the k-mer / FracMinHash threshold models need an error rate in (0,1), got %g", prm->error_rate);
    if (prm->model == TAXOR_THR_SYNCMER && idx->w_min > 0)
        return fail(TAXOR_E_ARG, "searcher_create: syncmer threshold model on an index built without syncmers");
    HIP_TRY(hipSetDevice(idx->device));
    auto s = new taxor_gpu_searcher();
    s->idx = idx;
    s->prm = *prm;
    s->auto_sub_reads = s->prm.sub_batch_reads == 0;
    if (s->prm.sub_batch_reads == 0) s->prm.sub_batch_reads = 32768;
    if (s->prm.sub_batch_bases == 0) s->prm.sub_batch_bases = 1ull << 29;
    if (s->prm.sub_batch_reads > (1u << 20)) s->prm.sub_batch_reads = 1u << 20;
    hipError_t e = hipSuccess;
    if (lane_stream) {
        s->st = lane_stream;
        s->st_borrowed = s->lane_mode = true;
    } else {
        e = hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->st_sync, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_wave, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_reset, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_ctr, sizeof(Counters));
    if (e != hipSuccess) {
        delete s;
        return fail(TAXOR_E_HIP, "searcher_create: %s", hipGetErrorString(e));
    }
    s->lds_query = query_lds_bytes(idx->max_stride);
    if (s->lds_query > 160 * 1024) {
        taxor_gpu_searcher_destroy(s);
        return fail(TAXOR_E_ARG, "searcher_create: an IXF with %u-byte rows does not fit the LDS tally", idx->max_stride);
    }
    s->grid_sync = syncmers_grid(idx->device);
    if (const char *e = tune_env("TAXOR_PROFILE_PHASES")) {
        if (atoi(e) != 0 && (hipMalloc((void **)&s->d_prof, 16 * sizeof(unsigned long long)) != hipSuccess ||
                             hipMemset(s->d_prof, 0, 16 * sizeof(unsigned long long)) != hipSuccess)) {
            taxor_gpu_searcher_destroy(s);
            return fail(TAXOR_E_HIP, "searcher_create: profile buffer");
        }
    }
    s->prune = !(prm->flags & TAXOR_SEARCH_NO_PRUNE);
    s->group_always = (prm->flags & TAXOR_SEARCH_GROUP_ALWAYS) != 0;
    s->split_always = (prm->flags & TAXOR_SEARCH_SPLIT_ALWAYS) != 0;
    s->force_tree_stall = (prm->flags & TAXOR_SEARCH_FORCE_TREE_STALL) != 0;
    s->small_path = !(prm->flags & TAXOR_SEARCH_NO_SMALL_PATH);
    if (const char *e = tune_env("TAXOR_QUERY_PRUNE")) s->prune = atoi(e) != 0;
    if (const char *e = tune_env("TAXOR_FIRST_DIV")) { const int v = atoi(e); if (v >= 1 && v <= 64) s->first_div = (uint32_t)v; }
    if (const char *e = tune_env("TAXOR_SUB_READS")) { const long v = atol(e); if (v >= 1 && v <= (1 << 20)) { s->prm.sub_batch_reads = (uint32_t)v; s->auto_sub_reads = false; } }
    {   // syncmer launches that run beside a query kernel keep to two blocks per CU: at full occupancy (three) the
        // query kernel stalls for as long as the syncmer kernel runs (measured); with one or two it is not slowed
        hipDeviceProp_t p;
        int per = 2;
        if (const char *e = tune_env("TAXOR_SYNC_BPC_OVERLAP")) { const int v = atoi(e); if (v >= 1 && v <= 8) per = v; }
        s->grid_sync_overlap = hipGetDeviceProperties(&p, idx->device) == hipSuccess ? p.multiProcessorCount * per : s->grid_sync;
        if (s->grid_sync_overlap > s->grid_sync) s->grid_sync_overlap = s->grid_sync;
    }
    s->grid_wave = syncmers_wave_grid(idx->device, 8);
    {
        int per = 2;
        if (const char *e = tune_env("TAXOR_SYNC_BPC_OVERLAP")) { const int v = atoi(e); if (v >= 1 && v <= 8) per = v; }
        s->grid_wave_overlap = syncmers_wave_grid(idx->device, per);
    }
    s->grid_query = query_grid(idx->device, s->lds_query, 3);
    s->grid_query_short = query_grid(idx->device, s->lds_query, 4);
    {
        static const bool small_off = [] { const char *e = tune_env("TAXOR_QUERY_SMALL"); return e && atoi(e) == 0; }();
        for (uint32_t l = 0; l < idx->depth && l < (uint32_t)MAX_LEVELS && !small_off; ++l) {
            static const uint32_t max_stride_small = [] { const char *e = tune_env("TAXOR_QUERY_SMALL_MAXSTRIDE"); return e ? (uint32_t)atoi(e) : 512u; }();
            if (idx->lvl_max_stride[l] == 0 || idx->lvl_max_stride[l] > max_stride_small) continue;      // rows of up to 512 bins (1024-bin roots gain nothing: 34.4 vs 34.7 ms)
            s->lds_query_small[l] = query_lds_bytes(idx->lvl_max_stride[l], true);
            s->grid_query_small[l] = query_grid_small(idx->device, s->lds_query_small[l]);
        }
    }
    *out = s;
    return TAXOR_OK;
}

extern "C" void taxor_gpu_searcher_destroy(taxor_gpu_searcher *s)
{
    if (!s) return;
    (void)hipSetDevice(s->idx->device);
    for (auto &L : s->lanes) {
        if (L.c) { (void)hipStreamSynchronize(L.c->st); taxor_gpu_searcher_destroy(L.c); }
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.copied) (void)hipEventDestroy(L.copied);
        L.d_in.release();
        if (L.h_in) (void)hipHostFree(L.h_in);
        if (L.h_bases) (void)hipHostFree(L.h_bases);
        if (L.h_out) (void)hipHostFree(L.h_out);
    }
    s->lanes.clear();
    if (s->st_copy) (void)hipStreamSynchronize(s->st_copy);
    if (s->st_sync2) (void)hipStreamSynchronize(s->st_sync2);
    if (s->st_sync) (void)hipStreamSynchronize(s->st_sync);
    if (s->st) (void)hipStreamSynchronize(s->st);
    s->d_ascii.release(); s->d_aoff.release(); s->d_poff.release(); s->d_hoff.release();
    s->d_packed.release(); s->d_rlen.release(); s->d_hcap.release(); s->d_nh.release(); s->d_thr.release();
    s->d_order.release();
    for (int b = 0; b < 2; ++b) { s->d_cand[b].release(); s->d_hashes[b].release(); }
    s->d_sync_cursor.release();
    for (auto ev : s->ev_sync_done) (void)hipEventDestroy(ev);
    for (auto ev : s->ev_query_done) (void)hipEventDestroy(ev);
    for (auto ev : s->ev_pack_done) (void)hipEventDestroy(ev);
    for (auto ev : s->ev_copy_done) (void)hipEventDestroy(ev);
    if (s->st_copy) (void)hipStreamDestroy(s->st_copy);
    if (s->h_small) (void)hipHostFree(s->h_small);
    if (s->h_small_in) (void)hipHostFree(s->h_small_in);
    if (s->ev_reset) (void)hipEventDestroy(s->ev_reset);
    if (s->st_sync) (void)hipStreamDestroy(s->st_sync);
    if (s->st_sync2) (void)hipStreamDestroy(s->st_sync2);
    if (s->ev_wave) (void)hipEventDestroy(s->ev_wave);
    s->d_q[0].release(); s->d_q[1].release(); s->d_qs.release(); s->d_qhist.release(); s->d_hits.release();
    s->d_read_hits.release(); s->d_cursor.release(); s->d_roff.release(); s->d_biglist.release(); s->d_gtab.release(); s->d_scan.release();
    s->d_read_off.release(); s->d_out_ub.release(); s->d_out_cnt.release(); s->d_out_key.release();
    for (auto ev : s->ev) (void)hipEventDestroy(ev);
    if (s->d_ctr) (void)hipFree(s->d_ctr);
    if (s->d_prof) (void)hipFree(s->d_prof);
    if (s->st && !s->st_borrowed) (void)hipStreamDestroy(s->st);
    delete s;
}

namespace {

int ev_reserve(taxor_gpu_searcher *s)
{
    if (s->ev_used + 2 > s->ev.size()) {
        for (int i = 0; i < 64; ++i) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            s->ev.push_back(ev);
        }
    }
    return 0;
}

// bracket helpers: begin returns the index of the start event
int ev_begin(taxor_gpu_searcher *s, int kind, size_t *slot, hipStream_t st = nullptr)
{
    *slot = (size_t)-1;
    if (!s->prm.time_kernels) return 0;
    if (ev_reserve(s)) return TAXOR_E_HIP;
    *slot = s->ev_used;
    s->ev_used += 2;
    s->ev_spans.push_back({*slot, kind});
    HIP_TRY(hipEventRecord(s->ev[*slot], st ? st : s->st));
    return 0;
}

int ev_end(taxor_gpu_searcher *s, size_t slot, hipStream_t st = nullptr)
{
    if (slot == (size_t)-1) return 0;
    HIP_TRY(hipEventRecord(s->ev[slot + 1], st ? st : s->st));
    return 0;
}

// host-side layout of a batch: packed offsets, candidate slots, sub-batch partition
int layout_batch(taxor_gpu_searcher *s, const uint64_t *offsets, uint64_t n_reads, std::vector<uint64_t> &poff,
                 std::vector<uint32_t> &rlen, std::vector<uint64_t> &hoff, std::vector<uint32_t> &hcap,
                 std::vector<uint32_t> &order, uint32_t first_div, bool ramp)
{
    const taxor_gpu_index *idx = s->idx;
    const int w = idx->k - idx->s + 1;
    int gap = std::min(idx->t, w - idx->t + 1); // minimum distance between two open syncmers of one read
    if (gap < 1) gap = 1;
    poff.resize(n_reads);
    rlen.resize(n_reads);
    hoff.resize(n_reads);
    hcap.resize(n_reads);
    s->subs.clear();
    s->max_slots = 0;
    s->max_read_slots = 0;
    s->max_sub_reads = 0;
    uint64_t words = 0, sub_slots = 0, sub_bases = 0;
    uint32_t sub_first = 0;
    // a sub-batch of 32768 reads of 1 kb is 33 Mbases: three query launches and four finalize kernels per millisecond
    // of work.  With the size left to the library, shorter reads get more of them per sub-batch, up to 524288 and as long as
    // the sub-batch stays within sub_batch_bases (2^29): the fixed cost per level and sub-batch is paid less often and a
    // level's work items, grouped by IXF, re-read their children out of the caches more often (1-kb reads, 131072 -> 524288
    // per sub-batch: GTDB-class 22.0 -> 23.1 Gbp/s, RefSeq-class 30.0 -> 33.7; profiles/r03/sub_reads_1kb.txt).  10-kb reads
    // keep 32768.
    uint64_t full_reads = s->prm.sub_batch_reads;
    if (s->auto_sub_reads && n_reads) {
        const uint64_t mean_len = std::max<uint64_t>(1, (offsets[n_reads] - offsets[0]) / n_reads);
        while (full_reads < 524288 && 2 * full_reads * mean_len <= s->prm.sub_batch_bases) full_reads *= 2;
    }
    uint64_t lim_reads = 0, lim_bases = 0;
    size_t lim_for = (size_t)-1;
    for (uint64_t r = 0; r < n_reads; ++r) {
        if (offsets[r + 1] < offsets[r]) return fail(TAXOR_E_ARG, "offsets not monotone at read %llu", (unsigned long long)r);
        const uint64_t len = offsets[r + 1] - offsets[r];
        if (len >= (1ull << 31)) return fail(TAXOR_E_ARG, "read %llu longer than 2^31 bases", (unsigned long long)r);
        const uint64_t nwin = len >= (uint64_t)idx->k ? len - idx->k + 1 : 0;
        // 128-B aligned regions: no line shared between reads.  Minimiser mode emits at most one value per window.
        const uint64_t cap = round_up((idx->w_min > 0 ? nwin : (uint64_t)((uint32_t)nwin / (uint32_t)gap)) + 2, 16);   // len < 2^31: a 32-bit divide
        // the first sub-batch's syncmer kernel has nothing to hide behind: keep it a quarter the size
        // Resident batch: only the first sub-batch may be smaller (first_div).  Streamed batch: sub-batch i+1 is ready
        // when its bases have crossed PCIe (serially, behind all earlier ones) and its syncmer kernel has run, and it
        // should be ready before sub-batch i is classified -- which allows a growth of ~1.27x per sub-batch at 48 GB/s
        // of PCIe against ~35 Gbp/s of classification, so the sizes ramp 1/first_div, x1.25, x1.25, ... up to the full
        // size instead of jumping there (a jump leaves the GPU idle for most of the second sub-batch's copy).
        if (lim_for != s->subs.size()) {          // the limits depend on the sub-batch's number alone: once per sub-batch, not per read
            lim_for = s->subs.size();
            lim_reads = full_reads;
            lim_bases = s->prm.sub_batch_bases;
            if (ramp) {
                double f = 1.0 / (double)first_div;
                static const double growth = [] { const char *e = tune_env("TAXOR_RAMP_GROWTH"); const double v = e ? atof(e) : 0.0; return v > 1.0 ? v : 1.25; }();
                for (size_t i = 0; i < s->subs.size() && f < 1.0; ++i) f *= growth;
                if (f < 1.0) {
                    lim_reads = std::max<uint64_t>((uint64_t)((double)lim_reads * f), 1);
                    lim_bases = std::max<uint64_t>((uint64_t)((double)lim_bases * f), 1);
                }
            } else if (s->subs.empty()) {
                lim_reads = std::max<uint64_t>(lim_reads / first_div, 1);
                lim_bases = std::max<uint64_t>(lim_bases / first_div, 1);
            }
        }
        if (r > sub_first && (r - sub_first >= lim_reads || sub_bases + len > lim_bases)) {
            s->subs.push_back({sub_first, (uint32_t)(r - sub_first), (uint32_t)(r - sub_first), sub_slots, offsets[sub_first] - offsets[0], offsets[r] - offsets[0]});
            s->max_slots = std::max(s->max_slots, sub_slots);
            s->max_sub_reads = std::max(s->max_sub_reads, (uint32_t)(r - sub_first));
            sub_first = (uint32_t)r;
            sub_slots = 0;
            sub_bases = 0;
        }
        poff[r] = words;
        rlen[r] = (uint32_t)len;
        hoff[r] = sub_slots;
        hcap[r] = (uint32_t)cap;
        words += round_up((len + 15) / 16, 4);
        sub_slots += cap;
        sub_bases += len;
        s->max_read_slots = std::max(s->max_read_slots, cap);
    }
    if (n_reads > sub_first) {
        s->subs.push_back({sub_first, (uint32_t)(n_reads - sub_first), (uint32_t)(n_reads - sub_first), sub_slots, offsets[sub_first] - offsets[0], offsets[n_reads] - offsets[0]});
        s->max_slots = std::max(s->max_slots, sub_slots);
        s->max_sub_reads = std::max(s->max_sub_reads, (uint32_t)(n_reads - sub_first));
    }
    // processing order inside each sub-batch: longest reads first (stable), so the dynamic work cursors hand out
    // the expensive items early and no long read is left alone at the tail of a launch
    order.resize(n_reads);
    static const bool no_order = tune_env("TAXOR_NO_ORDER") != nullptr; // A/B knob for measurements
    std::vector<uint32_t> tmp;
    const bool wave_ok = idx->w_min == 0 && syncmers_wave_applies(idx->k, idx->s);
    auto split_long_short = [&](SubBatch &sb) {   // the order is longest first: the short reads are a suffix of it
        const uint32_t *o = order.data() + sb.first;
        const uint32_t *cp = hcap.data() + sb.first;
        sb.n_long = sb.n;
        if (!wave_ok) return;
        uint32_t nl = 0;
        while (nl < sb.n && cp[o[nl]] > SYNC_WAVE_CAND) ++nl;
        for (uint32_t i = nl; i < sb.n; ++i)
            if (cp[o[i]] > SYNC_WAVE_CAND) return;       // not sorted by length (TAXOR_NO_ORDER): block kernel for all
        sb.n_long = nl;
    };
    for (SubBatch &sb : s->subs) {
        uint32_t *o = order.data() + sb.first;
        const uint32_t *len = rlen.data() + sb.first;
        for (uint32_t i = 0; i < sb.n; ++i) o[i] = i;
        if (no_order || sb.n < 2) continue;
        uint32_t lo = len[0], hi = len[0];
        for (uint32_t i = 1; i < sb.n; ++i) { lo = std::min(lo, len[i]); hi = std::max(hi, len[i]); }
        if (lo == hi) continue;                                       // equal lengths: input order is the order
        if (sb.n < 4096) {
            std::stable_sort(o, o + sb.n, [&](uint32_t a, uint32_t b) { return len[a] > len[b]; });
            continue;
        }
        // stable LSD radix sort by descending length (two 16-bit digits of hi - len): O(n), this runs per batch on the
        // host before anything is enqueued
        tmp.resize(sb.n);
        uint32_t *src = o, *dst = tmp.data();
        for (int pass = 0; pass < 2; ++pass) {
            const int shift = 16 * pass;
            if (pass == 1 && ((hi - lo) >> 16) == 0) break;
            static thread_local std::vector<uint32_t> cnt;
            cnt.assign(65537, 0u);
            for (uint32_t i = 0; i < sb.n; ++i) ++cnt[(((hi - len[src[i]]) >> shift) & 0xFFFFu) + 1];
            for (uint32_t d = 0; d < 65536; ++d) cnt[d + 1] += cnt[d];
            for (uint32_t i = 0; i < sb.n; ++i) dst[cnt[((hi - len[src[i]]) >> shift) & 0xFFFFu]++] = src[i];
            std::swap(src, dst);
        }
        if (src != o) std::copy(src, src + sb.n, o);
    }
    for (SubBatch &sb : s->subs) split_long_short(sb);
    s->packed_word_count = words + 16;
    s->packed_in_bytes = 0;
    for (uint64_t r = 0; r < n_reads; ++r) s->packed_in_bytes += (rlen[r] + 3u) / 4u; // ceil(L/4), SURVEY 8(d)
    return 0;
}

int ensure_scratch(taxor_gpu_searcher *s)
{
    const taxor_gpu_index *idx = s->idx;
    const uint32_t R = std::max<uint32_t>(s->max_sub_reads, 1);
    const bool syncmer_mode = idx->w_min == 0;  // minimiser mode writes its hashes directly: no candidates, no dedup
    for (int b = 0; b < 2; ++b)
        if ((syncmer_mode && s->d_cand[b].reserve(s->max_slots + 64)) || s->d_hashes[b].reserve(s->max_slots + 64)) return TAXOR_E_HIP;
    if (s->d_sync_cursor.reserve(2 * (s->subs.size() + 1))) return TAXOR_E_HIP;      // block kernel + wave kernel per sub-batch
    while (s->ev_sync_done.size() < s->subs.size() + 1) {
        hipEvent_t a, b, c;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c, hipEventDisableTiming));
        s->ev_sync_done.push_back(a);
        s->ev_query_done.push_back(b);
        s->ev_pack_done.push_back(c);
        hipEvent_t d;
        HIP_TRY(hipEventCreateWithFlags(&d, hipEventDisableTiming));
        s->ev_copy_done.push_back(d);
    }
    // Initial capacities: a read of a clade with many indexed relatives reports a tuple per relative and enters several
    // child IXFs (8 tuples / 3 children per read on the family workload), and an overflow costs a rerun of the whole
    // batch; these buffers are small (8 B per queue entry, 16 B per hit), so start generously.
    const uint64_t qmin = std::max<uint64_t>(8ull * R, idx->h_ixf.size() + 64);
    const uint64_t hmin = std::max<uint64_t>(16ull * R, idx->leaf_runs + 64);
    if (s->q_cap < qmin) s->q_cap = (uint32_t)std::min<uint64_t>(qmin, 0x7FFFFFFFu);
    if (s->hit_cap < hmin) s->hit_cap = (uint32_t)std::min<uint64_t>(hmin, 0x7FFFFFFFu);
    if (s->d_q[0].reserve(s->q_cap) || s->d_q[1].reserve(s->q_cap) || s->d_qs.reserve(s->q_cap) ||
        s->d_qhist.reserve(idx->h_ixf.size() + 1))
        return TAXOR_E_HIP;
    if (s->d_hits.reserve(s->hit_cap)) return TAXOR_E_HIP;
    if (s->d_read_hits.reserve(R) || s->d_cursor.reserve(R) || s->d_roff.reserve(R + 1) || s->d_biglist.reserve(R) || s->d_scan.reserve(R / 4096 + 8))
        return TAXOR_E_HIP;
    // dedup scratch in global memory, only for reads that could select more syncmers than the LDS passes cover
    if (syncmer_mode && s->max_read_slots > SYNC_LDS_DEDUP_MAX) {
        uint64_t ts = 64;
        while (ts < 2 * s->max_read_slots) ts <<= 1;
        if (ts > (1ull << 31)) return fail(TAXOR_E_ARG, "read too long for the dedup table");
        if (s->gtab_stride < ts) s->gtab_stride = (uint32_t)ts;
        if (s->d_gtab.reserve((size_t)s->gtab_stride * (size_t)s->grid_sync)) return TAXOR_E_HIP;
    }
    const uint64_t tmin = std::max<uint64_t>(12 * s->n_reads + 1024, idx->leaf_runs + 64);   // 16 B per tuple
    if (s->tuple_cap < tmin) s->tuple_cap = tmin;
    if (s->d_out_ub.reserve(s->tuple_cap) || s->d_out_cnt.reserve(s->tuple_cap) || s->d_out_key.reserve(s->tuple_cap))
        return TAXOR_E_HIP;
    if (s->d_read_off.reserve(s->n_reads + 1)) return TAXOR_E_HIP;
    return 0;
}

// level loop + CSR assembly for one group of reads whose hashes / thresholds are already on the device
int run_query(taxor_gpu_searcher *s, const uint64_t *d_hashes, const uint64_t *d_hoff, const uint32_t *d_nh,
              const uint64_t *d_thr, uint32_t n_reads, uint64_t *d_read_off, int is_last, uint32_t *d_counts_out,
              int only_ixf, const uint32_t *d_order = nullptr, uint32_t root_parts = 0, bool finalize = true, bool tree = false)
{
    const taxor_gpu_index *idx = s->idx;
    if (finalize) {            // (a lane's small finalize leaves the hit counts cleared and needs no scatter cursors)
        HIP_TRY(hipMemsetAsync(s->d_read_hits.p, 0, (size_t)n_reads * sizeof(uint32_t), s->st));
        HIP_TRY(hipMemsetAsync(s->d_cursor.p, 0, (size_t)n_reads * sizeof(uint32_t), s->st));
    }
    QueryArgs q{};
    q.ixf = idx->d_ixf;
    q.binfo = idx->d_binfo;
    q.hashes = d_hashes;
    q.hoff = d_hoff;
    q.nh = d_nh;
    q.thr = d_thr;
    q.hits = s->d_hits.p;
    q.read_hits = s->d_read_hits.p;
    q.counts_out = d_counts_out;
    q.ctr = s->d_ctr;
    q.q_cap = s->q_cap;
    q.hit_cap = s->hit_cap;
    q.map_words = query_map_words(idx->max_stride);
    q.max_stride = idx->max_stride;
    q.prune = (d_counts_out == nullptr && s->prune) ? 1u : 0u;
    q.prof = s->d_prof;
    static const uint32_t stages_env = [] { const char *e = tune_env("TAXOR_QUERY_STAGES"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? (uint32_t)v : 0u; }();
    q.sparse_stages = stages_env;
    static const float margin_env = [] { const char *e = tune_env("TAXOR_QUERY_MARGIN"); const double v = e ? atof(e) : 0.0; return v > 0.0 && v < 64.0 ? (float)v : 0.f; }();
    q.prune_margin = margin_env;
    static const uint32_t tally_env = [] { const char *e = tune_env("TAXOR_QUERY_TALLY"); return e ? (uint32_t)atoi(e) & 3u : 0u; }();
    q.tally_mode = tally_env;
    static const uint32_t sort_env = [] { const char *e = tune_env("TAXOR_QUERY_SORT_UNITS"); return e ? (uint32_t)(atoi(e) != 0) : 0u; }();     // measured: no fewer requests (profiles/r04/sparse_lines.txt)
    q.sort_units = sort_env;
    static const uint32_t dense_stride_env = [] { const char *e = tune_env("TAXOR_QUERY_DENSE_STRIDE"); return e ? (uint32_t)atoi(e) : 0u; }();
    q.dense_max_stride = dense_stride_env;
    // root items in column parts (QueryArgs::parts): what the caller asks for (the small-batch lanes), or the widest valid
    // division when the searcher was created with TAXOR_SEARCH_SPLIT_ALWAYS; never for the raw per-IXF entry points
    if (root_parts == 0) root_parts = s->split_always ? idx->root_pmax : 1u;
    if (d_counts_out || only_ixf >= 0 || root_parts > idx->root_pmax || idx->root_pmax % root_parts != 0) root_parts = 1;   // (a count that does not
                                                                                      // divide the widest division would leave the root's last columns out)
    q.parts = root_parts;
    for (uint32_t j = 0; j <= root_parts; ++j) q.part_cut[j] = idx->root_cut[j * (idx->root_pmax / root_parts)];
    q.part_cut[root_parts] = idx->root_cut[idx->root_pmax];
    if (tree) {        // a lane's piece: the whole traversal in one launch (k_query_level<..., TREE>), children through the one queue d_q[0]
        q.level = 0;
        q.q_in = nullptr;
        q.q_out = s->d_q[0].p;
        q.n_level0 = n_reads;
        q.order0 = d_order;
        q.cursor_chunk = 1;
        q.xcd_slices = 0;
        // the watchdog of a block that waits for its queue slot: ~2^22 polls of >= 0.5 us are seconds, far beyond any run of a piece
        // of <= 1024 reads; TAXOR_SEARCH_FORCE_TREE_STALL (or TAXOR_TREE_POLLS=0) makes it fire at the first empty poll
        const char *pe = tune_env("TAXOR_TREE_POLLS");
        q.tree_polls = s->force_tree_stall ? 0u : pe ? (uint32_t)std::max(0, atoi(pe)) : (1u << 22);
        const uint64_t items0 = (uint64_t)n_reads * root_parts;
        // Blocks beyond the items a piece can have at one time only poll (the root's items, or a couple per read below it) -- and a
        // block that polls holds its place on a CU until ITS launch is complete, so a launch is also kept to half the chip: two
        // lanes' traversals then run side by side, one's tail under the other's body, instead of one after the other
        static const int grid_env = [] { const char *e = tune_env("TAXOR_TREE_GRID"); return e ? atoi(e) : 0; }();
        static const int unroll_env = [] { const char *e = tune_env("TAXOR_TREE_UNROLL"); return e ? atoi(e) : 0; }();
        const uint64_t grid_cap = grid_env > 0 ? (uint64_t)grid_env : (uint64_t)s->grid_query_short / 2;
        const int grid = (int)std::min<uint64_t>(grid_cap, std::max<uint64_t>(items0, 2ull * n_reads) + 64);
        launch_query_tree(q, grid, s->lds_query, s->st, unroll_env == 2 ? 2 : 4);
        s->stats.query_launches++;
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const uint32_t levels = only_ixf >= 0 ? 1u : idx->depth;
    static const bool group_queue = [] { const char *e = tune_env("TAXOR_QUERY_GROUP"); return !e || atoi(e) != 0; }();
    for (uint32_t lvl = 0; lvl < levels; ++lvl) {
        q.level = lvl;
        q.q_in = (lvl == 0 && only_ixf < 0) ? nullptr : s->d_q[lvl & 1].p;
        // (not for small sub-batches -- a few thousand items have no cache locality to win, and the grouping's five small
        // launches per level are 5 % of a call of 1024 reads)
        const char *gm = tune_env("TAXOR_QUERY_GROUP_MIN");        // read per call: the parity test of the grouping itself sets it to 0
        const uint32_t group_min = gm ? (uint32_t)atoi(gm) : (s->group_always ? 0u : 4096u);
        q.xcd_slices = 0;
        if (lvl >= 1 && only_ixf < 0 && group_queue && n_reads >= group_min) {
            static const uint32_t xcd_env = [] { const char *e = tune_env("TAXOR_QUERY_XCD"); const int v = e ? atoi(e) : 8;
                                                 return (v == 1 || v == 8) ? 8u : (v == 2 || v == 4) ? (uint32_t)v : 0u; }();   // 0 = one cursor; 2 / 4 = XCD pairs / quads share a slice
            q.xcd_slices = xcd_env;
            // this level's items, pushed by the previous one in no particular order, grouped by IXF: blocks that run at
            // the same time then read the same few child IXFs (cache-resident) instead of rows all over the slab
            launch_queue_group_by_ixf(s->d_q[lvl & 1].p, s->d_ctr, lvl, s->q_cap, s->d_qhist.p, (uint32_t)idx->h_ixf.size(), s->d_qs.p, s->st);
            q.q_in = s->d_qs.p;
        }
        q.q_out = s->d_q[(lvl + 1) & 1].p;
        q.n_level0 = n_reads;
        q.order0 = d_order;
        static const uint32_t chunk_env = [] { const char *e = tune_env("TAXOR_QUERY_CHUNK"); return e ? (uint32_t)atoi(e) : 0u; }();
        // levels below the root: small items, four per cursor atomic; the root level too when the reads are short (a
        // long read is ~30 us of work and chunks of those would leave blocks idle at the tail of the launch)
        // (1-kb reads at the root, 2 / 4 / 8 per atomic: 23.69 / 23.85 / 23.98 Gbp/s -- profiles/r03/chunk_ab.txt)
        q.cursor_chunk = chunk_env ? chunk_env : (lvl >= 1 ? 4u : (s->mean_read_len < 1500 ? 8u : (s->mean_read_len < 3000 ? 4u : 1u)));
        size_t slot;
        if (ev_begin(s, 16 + (int)std::min(lvl, 7u), &slot)) return TAXOR_E_HIP;
        // four blocks per CU for short reads and for every level below the root (small items: half their time is spent
        // outside the gather loop, and by then the next sub-batch's syncmer kernel has left the CUs), three otherwise
        static const int bpc_l1 = [] { const char *e = tune_env("TAXOR_QUERY_BPC_L1"); return e ? atoi(e) : 4; }();
        // (and for a root level whose items all fit ONE round of the wide grid but not of the narrow one -- the reference's
        // chunk of 1024 reads on 256 CUs; two searchers in flight gain ~10 %, one nothing)
        const uint64_t items0 = (uint64_t)n_reads * root_parts;
        const bool wide_grid = s->mean_read_len < 6000 || (lvl >= 1 && bpc_l1 >= 4) ||
                               (lvl == 0 && items0 > (uint64_t)s->grid_query && items0 <= (uint64_t)s->grid_query_short);
        const bool root_streams = idx->rows[0] * (uint64_t)idx->h_ixf[0].stride > (16ull << 30);   // root table beyond any cache
        // tiny items (a level of IXFs with <= 512 bins, reads short enough that their probes fit 256 LDS slots): the
        // single-wave instantiation, sixteen blocks per CU; raw bulk_count calls (d_counts_out) stay on the general one
        const bool small = !d_counts_out && only_ixf < 0 && lvl < (uint32_t)MAX_LEVELS && s->grid_query_small[lvl] > 0 &&
                           s->mean_read_len < 2600 && !s->d_prof;
        if (small) {
            QueryArgs qs = q;
            qs.max_stride = idx->lvl_max_stride[lvl];
            qs.map_words = query_map_words(idx->lvl_max_stride[lvl]);
            qs.cursor_chunk = chunk_env ? chunk_env : 4;
            launch_query_level(qs, s->grid_query_small[lvl], s->lds_query_small[lvl], s->st, true, root_streams);
        } else
            launch_query_level(q, wide_grid ? s->grid_query_short : s->grid_query, s->lds_query, s->st, false, root_streams);
        if (ev_end(s, slot)) return TAXOR_E_HIP;
        s->stats.query_launches++;
    }
    HIP_TRY(hipGetLastError());
    if (d_counts_out || !finalize) return 0;
    FinalizeArgs f{};
    f.hits = s->d_hits.p;
    f.read_hits = s->d_read_hits.p;
    f.cursor = s->d_cursor.p;
    f.roff = s->d_roff.p;
    f.biglist = s->d_biglist.p;
    f.block_sums = s->d_scan.p;
    f.dfs_key = idx->d_dfs_key;
    f.ubin = idx->d_ubin;
    f.read_off = d_read_off;
    f.out_ub = s->d_out_ub.p;
    f.out_cnt = s->d_out_cnt.p;
    f.out_key = s->d_out_key.p;
    f.ctr = s->d_ctr;
    f.n_reads = n_reads;
    f.tuple_cap = s->tuple_cap;
    f.hit_cap = s->hit_cap;
    f.is_last = is_last;
    size_t slot;
    if (ev_begin(s, 2, &slot)) return TAXOR_E_HIP;
    launch_finalize(f, s->st);
    if (ev_end(s, slot)) return TAXOR_E_HIP;
    HIP_TRY(hipGetLastError());
    return 0;
}

// zero the per-sub-batch part of the counters block, keep the running totals
int reset_sub_counters(taxor_gpu_searcher *s, bool whole)
{
    const size_t n = whole ? sizeof(Counters) : offsetof(Counters, tuple_total);
    // flags must survive across sub-batches: they sit first, so skip them unless the whole block is reset
    if (whole) HIP_TRY(hipMemsetAsync(s->d_ctr, 0, n, s->st));
    else HIP_TRY(hipMemsetAsync(reinterpret_cast<uint8_t *>(s->d_ctr) + sizeof(uint32_t), 0, n - sizeof(uint32_t), s->st));
    return 0;
}

// syncmer kernel of sub-batch `sub_i` into scratch buffer `buf`, on stream `st`
int launch_syncmers_sub(taxor_gpu_searcher *s, const SubBatch &sb, size_t sub_i, int buf, hipStream_t st,
                        bool overlapped = false, hipStream_t st_short = nullptr)
{
    if (!st_short) st_short = st;   // the wave-per-read kernel of the short reads: its own stream when the caller has one
    const taxor_gpu_index *idx = s->idx;
    SyncmerArgs a{};
    a.packed = s->d_packed.p;
    a.poff = s->d_poff.p + sb.first;
    a.rlen = s->d_rlen.p + sb.first;
    a.hoff = s->d_hoff.p + sb.first;
    a.hcap = s->d_hcap.p + sb.first;
    a.cand = s->d_cand[buf].p;
    a.hashes = s->d_hashes[buf].p;
    a.cursor = s->d_sync_cursor.p + 2 * sub_i;
    a.order = s->d_order.p ? s->d_order.p + sb.first : nullptr;
    a.nh = s->d_nh.p + sb.first;
    a.thr = s->d_thr.p + sb.first;
    a.ratio = s->prm.ratio;
    // same expression as the reference: double(UINT64_MAX) / double(scaling)  (taxor_search.cpp:228)
    a.scaling_limit = idx->scaling > 1 ? (double)UINT64_MAX / (double)idx->scaling : 0.0;
    a.gtab = s->d_gtab.p;
    a.gtab_stride = s->gtab_stride;
    a.ctr = s->d_ctr;
    a.n_reads = sb.n;
    {   // short reads: eight per cursor atomic (metadata + first words prefetched); long reads one (load balance at the tail)
        static const uint32_t chunk_env = [] { const char *e = tune_env("TAXOR_SYNC_CHUNK"); return e ? (uint32_t)atoi(e) : 0u; }();
        a.chunk = chunk_env ? chunk_env : (s->mean_read_len < 2500 ? 8u : (s->mean_read_len < 6000 ? 4u : 1u));
    }
    a.k = idx->k;
    a.s = idx->s;
    a.t = idx->t;
    a.w_min = idx->w_min;
    a.thr_on_device = s->prm.model == TAXOR_THR_PERCENTAGE ? 1 : 0;
    a.prof = s->d_prof;
    size_t slot;
    if (ev_begin(s, 0, &slot, st)) return TAXOR_E_HIP;
    const uint32_t n_long = idx->w_min > 0 ? sb.n : sb.n_long;
    if (n_long) {
        a.n_reads = n_long;
        launch_syncmers(a, overlapped ? s->grid_sync_overlap : s->grid_sync, st);
    }
    if (n_long < sb.n) {      // short reads: one wavefront per read (k_syncmers_wave), behind the long ones on the same stream
        SyncmerArgs b = a;
        b.order = a.order + n_long;
        b.n_reads = sb.n - n_long;
        b.cursor = a.cursor + 1;
        b.chunk = 4;
        b.prof = nullptr;
        // beside the block kernel, not behind it: the few very long reads of an ONT-like mix keep single blocks busy
        // for milliseconds, and the short reads fill the rest of the chip meanwhile
        launch_syncmers_wave(b, overlapped ? s->grid_wave_overlap : s->grid_wave, st_short);
        if (st_short != st) {
            HIP_TRY(hipEventRecord(s->ev_wave, st_short));
            HIP_TRY(hipStreamWaitEvent(st, s->ev_wave, 0));
        }
    }
    if (ev_end(s, slot, st)) return TAXOR_E_HIP;
    HIP_TRY(hipGetLastError());
    if ((s->prm.model == TAXOR_THR_KMER || s->prm.model == TAXOR_THR_FRACMINHASH) && !s->thr_precomputed) {
        // threshold::get of the k-mer / FracMinHash models (threshold.hpp:62-75) is a page of double arithmetic with
        // pow / log / sqrt whose result is truncated to an integer: evaluated on the host, with the host's libm, so
        // that it is the reference's value bit for bit.  The GPU keeps classifying the previous sub-batch meanwhile.
        s->h_nh_sub.resize(sb.n);
        s->h_thr_sub.resize(sb.n);
        HIP_TRY(hipMemcpyAsync(s->h_nh_sub.data(), s->d_nh.p + sb.first, sb.n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint32_t i = 0; i < sb.n; ++i) {
            const uint64_t n = s->h_nh_sub[i];
            const double sf = (double)n / ((double)s->h_rlen[sb.first + i] - (double)idx->k + 1.0);   // taxor_search.cpp:263
            s->h_thr_sub[i] = taxor_threshold_model((int)s->prm.model, n, (uint32_t)idx->k, s->prm.error_rate, -1.0, sf);
        }
        HIP_TRY(hipMemcpyAsync(s->d_thr.p + sb.first, s->h_thr_sub.data(), sb.n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st)); // h_thr_sub is reused by the next sub-batch
    }
    return 0;
}

int check_flags(taxor_gpu_searcher *s, bool *rerun)
{
    *rerun = false;
    HIP_TRY(hipMemcpyAsync(&s->h_ctr, s->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, s->st));
    HIP_TRY(hipStreamSynchronize(s->st));
    const uint32_t f = s->h_ctr.flags;
    if (f & FLAG_ALPHABET) return fail(TAXOR_E_ALPHABET, "a read contains a character outside the dna15 alphabet");
    if (f & FLAG_CAND_OVERFLOW) return fail(TAXOR_E_INTERNAL, "syncmer candidate capacity bound violated");
    if (f & FLAG_DEDUP_OVERFLOW) return fail(TAXOR_E_INTERNAL, "dedup scratch too small");
    if (f & FLAG_QUEUE_OVERFLOW) {
        uint32_t need = 0;
        for (int i = 0; i < MAX_LEVELS; ++i) need = std::max(need, s->h_ctr.q_n[i].v);
        s->q_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * s->q_cap, (uint64_t)need + 1024), 0x7FFFFFFFu);
        *rerun = true;
    }
    if (f & FLAG_HITS_OVERFLOW) {
        s->hit_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * s->hit_cap, (uint64_t)s->h_ctr.n_hits.v + 1024), 0x7FFFFFFFu);
        *rerun = true;
    }
    if (f & FLAG_TUPLE_OVERFLOW) {
        s->tuple_cap = std::max<uint64_t>(2 * s->tuple_cap, s->h_ctr.tuple_total + 1024);
        *rerun = true;
    }
    return 0;
}

} // namespace

namespace {

// streamed batches: the first sub-batch is 1/n of a full one (TAXOR_STREAM_FIRST_DIV, default 8)
uint32_t stream_first_div(const taxor_gpu_searcher *s)
{
    static const uint32_t env = [] { const char *e = tune_env("TAXOR_STREAM_FIRST_DIV"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? (uint32_t)v : 0u; }();
    return env ? env : std::max(s->first_div, 8u);
}

// host-side layout + device copies of the per-read arrays (everything except the bases themselves)
int prepare_batch(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads, bool streamed)
{
    if (!s || !offsets || (!bases && n_reads && offsets[n_reads] != offsets[0]))
        return fail(TAXOR_E_ARG, "batch_upload: null argument");
    if (n_reads >= (1ull << 32)) return fail(TAXOR_E_ARG, "batch_upload: more than 2^32 reads in one batch");
    HIP_TRY(hipSetDevice(s->idx->device));
    s->ran = s->synced = false;
    // the per-read layout arrays live in the searcher: a batch of a million short reads is 45 MB of them, and fresh vectors
    // per call cost more in page faults than the loop that fills them
    std::vector<uint64_t> &poff = s->lay_poff, &hoff = s->lay_hoff, &aoff = s->lay_aoff;
    std::vector<uint32_t> &rlen = s->h_rlen, &hcap = s->lay_hcap, &order = s->lay_order;
    // streamed: the first sub-batch's PCIe copy has nothing to hide behind either, so it is a quarter the size
    static const bool trace = tune_env("TAXOR_TRACE_BATCH") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    if (int rc = layout_batch(s, offsets, n_reads, poff, rlen, hoff, hcap, order, streamed ? stream_first_div(s) : s->first_div, streamed))
        return rc;
    const double t_layout = ms_since(t0);
    s->n_reads = n_reads;
    const uint64_t a0 = offsets[0], nb = offsets[n_reads] - a0;
    s->n_bases = nb;
    s->mean_read_len = n_reads ? nb / n_reads : (1u << 20);
    if (s->d_ascii.reserve(nb + 64) || s->d_aoff.reserve(n_reads + 1) || s->d_poff.reserve(n_reads + 1) ||
        s->d_hoff.reserve(n_reads + 1) || s->d_rlen.reserve(n_reads + 1) || s->d_hcap.reserve(n_reads + 1) ||
        s->d_nh.reserve(n_reads + 1) || s->d_thr.reserve(n_reads + 1) || s->d_order.reserve(n_reads + 1) ||
        s->d_packed.reserve(s->packed_word_count))
        return TAXOR_E_HIP;
    aoff.resize(n_reads + 1);
    for (uint64_t r = 0; r <= n_reads; ++r) aoff[r] = offsets[r] - a0;
    HIP_TRY(hipMemsetAsync(s->d_ctr, 0, sizeof(Counters), s->st));
    // A small batch (the reference's chunk is 1024 reads): six copies out of pageable vectors are six blocking host round
    // trips, ~70 us of a 1-ms call.  Its arrays go through one page-locked staging block instead: the copies are enqueued
    // and the host moves on; every other stream orders itself behind them through ev_reset (run_pipeline).
    constexpr uint64_t kSmallIn = 1ull << 20;
    const uint64_t need_in = (n_reads + 1) * 8 + n_reads * (8 + 8 + 4 + 4 + 4) + 64;
    bool staged_in = false;
    if (n_reads && need_in <= kSmallIn && (s->h_small_in || hipHostMalloc(&s->h_small_in, kSmallIn, hipHostMallocDefault) == hipSuccess)) {
        char *p = (char *)s->h_small_in;
        auto put = [&](void *dst, const void *src, uint64_t bytes) -> int {
            memcpy(p, src, bytes);
            HIP_TRY(hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s->st));
            p += (bytes + 7) & ~7ull;
            return 0;
        };
        if (put(s->d_aoff.p, aoff.data(), (n_reads + 1) * 8) || put(s->d_poff.p, poff.data(), n_reads * 8) || put(s->d_hoff.p, hoff.data(), n_reads * 8) ||
            put(s->d_rlen.p, rlen.data(), n_reads * 4) || put(s->d_hcap.p, hcap.data(), n_reads * 4) || put(s->d_order.p, order.data(), n_reads * 4))
            return TAXOR_E_HIP;
        staged_in = true;
    } else {
        (void)hipGetLastError();
        HIP_TRY(hipMemcpyAsync(s->d_aoff.p, aoff.data(), (n_reads + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s->st));
        if (n_reads) {
            HIP_TRY(hipMemcpyAsync(s->d_poff.p, poff.data(), n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s->st));
            HIP_TRY(hipMemcpyAsync(s->d_hoff.p, hoff.data(), n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s->st));
            HIP_TRY(hipMemcpyAsync(s->d_rlen.p, rlen.data(), n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, s->st));
            HIP_TRY(hipMemcpyAsync(s->d_hcap.p, hcap.data(), n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, s->st));
            HIP_TRY(hipMemcpyAsync(s->d_order.p, order.data(), n_reads * sizeof(uint32_t), hipMemcpyHostToDevice, s->st));
        }
    }
    // k-mer model (window == k, scaling 1): every k-mer of the read counts, so hash_count = L - k + 1 is known here and
    // threshold::get (threshold.hpp:62-66) -- a function of the count alone -- is evaluated for the whole batch now,
    // memoised per count; nothing blocks between a sub-batch's hashing and its query.  (The FracMinHash model needs the
    // number of minimisers the device finds: launch_syncmers_sub fetches it per sub-batch.)
    s->thr_precomputed = false;
    std::vector<uint64_t> thr_h;
    if (s->prm.model == TAXOR_THR_KMER && s->idx->w_min == s->idx->k && s->idx->scaling <= 1 && n_reads) {
        thr_h.resize(n_reads);
        const uint64_t k = (uint64_t)s->idx->k;
        for (uint64_t r = 0; r < n_reads; ++r) {
            const uint64_t n = rlen[r] >= k ? rlen[r] - k + 1 : 0;
            // taxor_search.cpp:263; the k-mer model does not read the factor (threshold.hpp:62-66), and for a read shorter
            // than k (n = 0) the quotient is 0/0 or -0: keep that out of the memo, whose key is n alone
            const double sf = n ? (double)n / ((double)rlen[r] - (double)k + 1.0) : 0.0;
            if (n == 0 || n >= (1u << 22)) {      // no k-mer at all, or chromosome-sized "reads": not worth a table entry each
                thr_h[r] = taxor_threshold_model(TAXOR_THR_KMER, n, (uint32_t)k, s->prm.error_rate, -1.0, sf);
                continue;
            }
            if (n >= s->thr_memo.size()) s->thr_memo.resize(n + 1, ~0ull);
            uint64_t &m = s->thr_memo[n];
            if (m == ~0ull) {
                m = taxor_threshold_model(TAXOR_THR_KMER, n, (uint32_t)k, s->prm.error_rate, -1.0, sf);
                if (m == ~0ull) m = ~0ull - 1;   // keep the marker free (a threshold that large is unreachable either way)
            }
            thr_h[r] = m;
        }
        HIP_TRY(hipMemcpyAsync(s->d_thr.p, thr_h.data(), n_reads * sizeof(uint64_t), hipMemcpyHostToDevice, s->st));
        s->thr_precomputed = true;
    }
    const double t_enq = ms_since(t0);
    if (!staged_in || s->thr_precomputed) HIP_TRY(hipStreamSynchronize(s->st)); // the pageable host vectors above may now die
    const double t_sync = ms_since(t0);
    const int rc = ensure_scratch(s);
    if (trace)
        fprintf(stderr, "[prepare_batch] %llu reads: layout %.2f ms, device buffers + per-read arrays enqueued at %.2f, on the device at %.2f, scratch ready at %.2f\n",
                (unsigned long long)n_reads, t_layout, t_enq, t_sync, ms_since(t0));
    return rc;
}

// The whole pipeline for the uploaded (host_ascii == nullptr) or streaming (host_ascii = first base of the batch)
// case.  Streams: st_copy (H2D of the sub-batches' bases, back to back), st_sync (pack + syncmers of i+1), st (query + CSR of i).
// streams a searcher does not always need are made at first use: every stream the process holds is one more candidate
// to share a hardware queue with (runtime_env_once above)
int ensure_stream(hipStream_t *st)
{
    if (!*st) HIP_TRY(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    return 0;
}

bool host_pointer_is_pinned(const void *p)
{
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc'd memory: unknown to the runtime
    return a.type == hipMemoryTypeHost;
}

int run_pipeline(taxor_gpu_searcher *s, bool host_ascii)
{
    if (int rc = ensure_scratch(s)) return rc;
    bool need_wave = false;
    for (const SubBatch &sb : s->subs) need_wave = need_wave || (s->idx->w_min == 0 && sb.n_long < sb.n);
    if (need_wave && ensure_stream(&s->st_sync2)) return TAXOR_E_HIP;
    // pageable input: each copy blocks the host anyway, so the copy stream buys nothing -- the bases go in on the hashing
    // stream, in front of their sub-batch's pack kernel (no event hop between two streams: ~50 us of a call of 1024 reads)
    bool pinned_input = host_ascii;
    for (const auto &sp : s->host_spans) pinned_input = pinned_input && host_pointer_is_pinned(sp.ptr);
    const bool inline_copy = host_ascii && !pinned_input && s->subs.size() == 1;
    if (host_ascii && !inline_copy && ensure_stream(&s->st_copy)) return TAXOR_E_HIP;
    s->ev_used = 0;
    s->ev_spans.clear();
    s->stats = taxor_gpu_run_stats{};
    size_t tot_slot;
    if (reset_sub_counters(s, true)) return TAXOR_E_HIP;
    HIP_TRY(hipMemsetAsync(s->d_sync_cursor.p, 0, 2 * (s->subs.size() + 1) * sizeof(uint32_t), s->st));
    HIP_TRY(hipEventRecord(s->ev_reset, s->st));
    HIP_TRY(hipStreamWaitEvent(s->st_sync, s->ev_reset, 0));
    if (s->st_sync2) HIP_TRY(hipStreamWaitEvent(s->st_sync2, s->ev_reset, 0));
    if (ev_begin(s, 3, &tot_slot)) return TAXOR_E_HIP;
    // TAXOR_NO_OVERLAP=1 (measurement knob): hashing on the query's own stream, at full occupancy, nothing concurrent
    static const bool no_overlap = [] { const char *e = tune_env("TAXOR_NO_OVERLAP"); return e && atoi(e) != 0; }();
    // the bases of sub-batch i, from wherever the caller keeps them: one buffer, or several segments in turn
    size_t span_i = 0;
    auto enqueue_copy = [&](size_t i) -> int {
        const SubBatch &sb = s->subs[i];
        hipStream_t cs = inline_copy ? (no_overlap ? s->st : s->st_sync) : s->st_copy;       // inline: the stream the pack kernel runs on
        for (; span_i < s->host_spans.size() && s->host_spans[span_i].vbegin + s->host_spans[span_i].len <= sb.a_begin; ++span_i) {}
        for (size_t j = span_i; j < s->host_spans.size() && s->host_spans[j].vbegin < sb.a_end; ++j) {
            const auto &sp = s->host_spans[j];
            const uint64_t lo = std::max(sp.vbegin, sb.a_begin), hi = std::min(sp.vbegin + sp.len, sb.a_end);
            if (hi > lo)
                HIP_TRY(hipMemcpyAsync(s->d_ascii.p + lo, sp.ptr + (lo - sp.vbegin), hi - lo, hipMemcpyHostToDevice, cs));
        }
        if (!inline_copy) HIP_TRY(hipEventRecord(s->ev_copy_done[i], cs));
        return 0;
    };
    // Page-locked input: every copy of the batch is enqueued now, before the first kernel -- they then run back to back at
    // PCIe speed far ahead of the kernels even when the runtime has put the copy stream on a hardware queue it shares
    // with one of the kernel streams (in-order per queue: what is submitted first runs first).  Pageable input: each
    // copy blocks the host until it is done, so it is issued just before its sub-batch's kernels (the GPU works on the
    // previous sub-batch meanwhile).
    const bool copies_first = pinned_input;
    if (copies_first)
        for (size_t i = 0; i < s->subs.size(); ++i)
            if (int rc = enqueue_copy(i)) return rc;
    if (s->subs.empty()) { // zero reads: CSR = [0]
        HIP_TRY(hipMemsetAsync(s->d_read_off.p, 0, sizeof(uint64_t), s->st));
    }
    // The syncmer kernel (VALU/LDS bound) of sub-batch i+1 runs beside the HBM-bound query of sub-batch i;
    // candidate/hash scratch is double buffered, events carry the dependencies.  When streaming, the bases of
    // sub-batch i+1 cross PCIe (pageable copy: the host blocks in it, the GPU keeps working) and are packed on
    // a third stream meanwhile.
    for (size_t i = 0; i < s->subs.size(); ++i) {
        const SubBatch &sb = s->subs[i];
        const int buf = (int)(i & 1);
        if (host_ascii && !copies_first)
            if (int rc = enqueue_copy(i)) return rc;
        hipStream_t ss = no_overlap ? s->st : s->st_sync;
        hipStream_t ss2 = (no_overlap || !s->st_sync2) ? ss : s->st_sync2;
        if (host_ascii) {
            // The copy stream carries copies only; the pack kernel sits on the hashing stream in front of its sub-batch's
            // syncmer kernel: with the copy long done, pack + syncmers of sub-batch i+1 both run beside the query of
            // sub-batch i.
            if (!inline_copy) HIP_TRY(hipStreamWaitEvent(ss, s->ev_copy_done[i], 0));
            launch_pack_dna4(s->d_ascii.p, s->d_aoff.p + sb.first, s->d_poff.p + sb.first, s->d_packed.p, sb.n, s->d_ctr, ss, s->grid_sync_overlap);
            HIP_TRY(hipGetLastError());
            if (ss2 != ss) {
                HIP_TRY(hipEventRecord(s->ev_pack_done[i], ss));
                HIP_TRY(hipStreamWaitEvent(ss2, s->ev_pack_done[i], 0));
            }
        }
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(ss, s->ev_query_done[i - 2], 0));
        if (i >= 2 && ss2 != ss) HIP_TRY(hipStreamWaitEvent(ss2, s->ev_query_done[i - 2], 0));
        if (int rc = launch_syncmers_sub(s, sb, i, buf, ss, i > 0 && !no_overlap, ss2)) return rc;
        HIP_TRY(hipEventRecord(s->ev_sync_done[i], ss));
        HIP_TRY(hipStreamWaitEvent(s->st, s->ev_sync_done[i], 0));
        if (i && reset_sub_counters(s, false)) return TAXOR_E_HIP;
        if (int rc = run_query(s, s->d_hashes[buf].p, s->d_hoff.p + sb.first, s->d_nh.p + sb.first, s->d_thr.p + sb.first,
                               sb.n, s->d_read_off.p + sb.first, i + 1 == s->subs.size(), nullptr, -1, s->d_order.p + sb.first))
            return rc;
        HIP_TRY(hipEventRecord(s->ev_query_done[i], s->st));
    }
    if (ev_end(s, tot_slot)) return TAXOR_E_HIP;
    s->ran = true;
    s->synced = false;
    return TAXOR_OK;
}

} // namespace


// =========================================================================================================
// small batches
// =========================================================================================================
// The reference hands its workers 1024 records per call (taxor_search.cpp:315); a binding that replaces do_parallel one to one
// calls this library with ~10 Mbp at a time.  Through the pipeline of large batches such a call was 1.0 ms for 0.39 ms of
// kernel work at the resident rate (profiles/r03/small_calls.txt): ~35 runtime calls, the bases' copy (0.19 ms at PCIe speed)
// exposed in front of everything, three levels that each end when their slowest item ends, six finalize launches and five
// result copies.  Calls of up to SMALL_MAX_READS reads therefore take another route:
//   * the call is cut into up to four PIECES; piece p runs on LANE p -- an internal searcher with ONE stream, its own
//     counters, queues and hit buffers -- so piece p+1's bases cross PCIe (the blocking copy of pageable memory) while piece p
//     is hashed and classified, and the tail of one piece's level overlaps the body of another's;
//   * per piece ~10 runtime calls: one copy of the per-read arrays (laid out back to back in one device block, from
//     page-locked staging), one copy of the bases, pack, syncmers, one launch per level, ONE finalize launch that writes the CSR
//     straight into host memory the device can address and leaves the lane's counters cleared (no memsets, no result copies);
//   * the root's work items are split into column parts (QueryArgs::parts) until a piece has ~a grid's worth of them: 256
//     reads are 1024 parts, the chip is full and an item lasts a quarter as long.
// Same kernels, same arithmetic, same results; anything unusual (a capacity overflow, reads too long for the LDS dedup, the
// host-evaluated threshold models, kernel timing) goes through the pipeline above instead -- for a piece or for the call.
namespace {

constexpr uint64_t SMALL_MAX_READS = 16384, SMALL_MAX_BASES = 1ull << 28, SMALL_PIECE_MIN = 256, SMALL_TUPLES = 1u << 16;
constexpr uint32_t SMALL_LANES = 4;

bool small_applicable(const taxor_gpu_searcher *s, const uint64_t *offsets, uint64_t n_reads)
{
    static const bool off = [] { const char *e = tune_env("TAXOR_SMALL_PATH"); return e && atoi(e) == 0; }();
    if (off || !s->small_path || s->lane_mode || s->prm.time_kernels || s->d_prof || s->idx->w_min != 0) return false;
    if (s->prm.model != TAXOR_THR_PERCENTAGE && s->prm.model != TAXOR_THR_SYNCMER) return false;
    if (n_reads == 0 || n_reads > SMALL_MAX_READS || offsets[n_reads] - offsets[0] > SMALL_MAX_BASES) return false;
    return true;
}

int small_lane_ready(taxor_gpu_searcher *s, uint32_t li, uint32_t n_reads, uint32_t tuples)
{
    if (s->lanes.size() <= li) s->lanes.resize(li + 1);
    taxor_gpu_searcher::SmallLane &L = s->lanes[li];
    if (!L.c) {
        taxor_gpu_search_params p = s->prm;
        p.flags |= TAXOR_SEARCH_NO_SMALL_PATH;
        p.time_kernels = 0;
        // the four streams this searcher has (or would make for its pipeline of large batches) carry the four lanes: no new ones
        hipStream_t *slot[SMALL_LANES] = {&s->st, &s->st_sync, &s->st_copy, &s->st_sync2};
        if (ensure_stream(slot[li])) return TAXOR_E_HIP;
        if (int rc = searcher_create_impl(s->idx, &p, &L.c, *slot[li])) return rc;
        L.c->prune = s->prune;
        HIP_TRY(hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&L.copied, hipEventDisableTiming));
        L.fresh = true;
    }
    if (L.out_reads < n_reads || L.out_tuples < tuples) {
        if (L.h_out) { HIP_TRY(hipStreamSynchronize(L.c->st)); (void)hipHostFree(L.h_out); L.h_out = nullptr; }
        L.out_reads = std::max<uint32_t>(n_reads, 1024);
        L.out_tuples = std::max<uint32_t>(tuples, SMALL_TUPLES);
        const size_t bytes = 64 + ((size_t)L.out_reads + 2) * 8 + (size_t)L.out_reads * 4 + 8 + (size_t)L.out_tuples * 12;
        HIP_TRY(hipHostMalloc(&L.h_out, bytes, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&L.d_out, L.h_out, 0));
    }
    return 0;
}

// the result area of a lane: status[8] | read_off[out_reads + 1] | ub[out_tuples] | cnt[out_tuples] | nh[out_reads]
struct SmallOut { uint64_t *status, *read_off; int64_t *ub; uint32_t *cnt, *nh; };
SmallOut small_out(const taxor_gpu_searcher::SmallLane &L, void *base)
{
    SmallOut o;
    char *p = (char *)base;
    o.status = (uint64_t *)p; p += 64;
    o.read_off = (uint64_t *)p; p += ((size_t)L.out_reads + 2) * 8;
    o.ub = (int64_t *)p; p += (size_t)L.out_tuples * 8;
    o.cnt = (uint32_t *)p; p += (size_t)L.out_tuples * 4;
    o.nh = (uint32_t *)p;
    return o;
}

// everything of one piece -- reads [first, first + n) of the call -- enqueued on its lane's stream
int small_enqueue(taxor_gpu_searcher *s, uint32_t li, const char *bases, const uint64_t *offsets, uint64_t first, uint32_t n, hipEvent_t copy_after = nullptr)
{
    const taxor_gpu_index *idx = s->idx;
    static const bool trace = tune_env("TAXOR_TRACE_BATCH") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    double t_ready = 0, t_layout = 0, t_scratch = 0, t_arrays = 0, t_bases = 0, t_hash = 0, t_query = 0;
    if (int rc = small_lane_ready(s, li, n, SMALL_TUPLES)) return rc;
    t_ready = us();
    taxor_gpu_searcher::SmallLane &L = s->lanes[li];
    taxor_gpu_searcher *c = L.c;
    const uint64_t *off = offsets + first;
    const uint64_t a0 = off[0], nb = off[n] - a0;
    // ---- layout of the piece, straight into the page-locked block the one copy reads (cf. layout_batch)
    const size_t o_aoff = 0, o_poff = o_aoff + ((size_t)n + 1) * 8, o_hoff = o_poff + (size_t)n * 8, o_rlen = o_hoff + (size_t)n * 8,
                 o_hcap = o_rlen + (((size_t)n * 4 + 7) & ~(size_t)7), o_order = o_hcap + (((size_t)n * 4 + 7) & ~(size_t)7),
                 in_bytes = o_order + (((size_t)n * 4 + 7) & ~(size_t)7);
    if (L.h_in_cap < in_bytes) {
        if (L.h_in) { HIP_TRY(hipStreamSynchronize(c->st)); (void)hipHostFree(L.h_in); L.h_in = nullptr; }
        L.h_in_cap = std::max<size_t>(in_bytes + in_bytes / 2, 64 << 10);
        HIP_TRY(hipHostMalloc(&L.h_in, L.h_in_cap, hipHostMallocDefault));
    }
    char *hi = (char *)L.h_in;
    uint64_t *aoff = (uint64_t *)(hi + o_aoff), *poff = (uint64_t *)(hi + o_poff), *hoff = (uint64_t *)(hi + o_hoff);
    uint32_t *rlen = (uint32_t *)(hi + o_rlen), *hcap = (uint32_t *)(hi + o_hcap), *order = (uint32_t *)(hi + o_order);
    const int w = idx->k - idx->s + 1;
    const int gap = std::max(1, std::min(idx->t, w - idx->t + 1));
    uint64_t words = 0, slots = 0, max_cap = 0;
    uint32_t lo = ~0u, hi_len = 0;
    for (uint32_t r = 0; r < n; ++r) {
        if (off[r + 1] < off[r]) return fail(TAXOR_E_ARG, "offsets not monotone at read %llu", (unsigned long long)(first + r));
        const uint64_t len = off[r + 1] - off[r];
        if (len >= (1ull << 31)) return fail(TAXOR_E_ARG, "read %llu longer than 2^31 bases", (unsigned long long)(first + r));
        const uint64_t nwin = len >= (uint64_t)idx->k ? len - idx->k + 1 : 0;
        const uint64_t cap = round_up((uint64_t)((uint32_t)nwin / (uint32_t)gap) + 2, 16);
        aoff[r] = off[r] - a0;
        poff[r] = words;
        hoff[r] = slots;
        rlen[r] = (uint32_t)len;
        hcap[r] = (uint32_t)cap;
        order[r] = r;
        words += round_up((len + 15) / 16, 4);
        slots += cap;
        max_cap = std::max(max_cap, cap);
        lo = std::min(lo, (uint32_t)len);
        hi_len = std::max(hi_len, (uint32_t)len);
    }
    aoff[n] = nb;
    if (max_cap > SYNC_LDS_DEDUP_MAX) return 1;                      // a read whose dedup needs the global table: the other pipeline
    if (lo != hi_len) std::stable_sort(order, order + n, [&](uint32_t a, uint32_t b) { return rlen[a] > rlen[b]; });   // longest first
    uint32_t n_long = n;
    if (syncmers_wave_applies(idx->k, idx->s)) {
        n_long = 0;
        while (n_long < n && hcap[order[n_long]] > SYNC_WAVE_CAND) ++n_long;
    }
    t_layout = us();
    // ---- the lane's buffers (allocation only when a piece is larger than any before it)
    c->n_reads = n;
    c->n_bases = nb;
    c->mean_read_len = std::max<uint64_t>(1, nb / n);
    c->max_slots = slots;
    c->max_read_slots = max_cap;
    c->max_sub_reads = n;
    c->subs.assign(1, SubBatch{0, n, n_long, slots, 0, nb});
    const size_t cap_before = c->d_read_hits.cap, qcap_before = c->d_q[0].cap;
    Counters *ctr_before = c->d_ctr;
    if (L.d_in.reserve(in_bytes) || c->d_ascii.reserve(nb + 64) || c->d_packed.reserve(words + 16) || c->d_nh.reserve(n + 1) || c->d_thr.reserve(n + 1))
        return TAXOR_E_HIP;
    if (int rc = ensure_scratch(c)) return rc;
    if (c->d_read_hits.cap != cap_before || c->d_ctr != ctr_before || c->d_q[0].cap != qcap_before) L.fresh = true;
    hipStream_t st = c->st;
    t_scratch = us();
    if (L.fresh) {            // first use, or the hit counts moved to a new allocation: clear what the small finalize otherwise leaves cleared
        HIP_TRY(hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), st));
        HIP_TRY(hipMemsetAsync(c->d_read_hits.p, 0, c->d_read_hits.cap * sizeof(uint32_t), st));
        HIP_TRY(hipMemsetAsync(c->d_sync_cursor.p, 0, 2 * sizeof(uint32_t), st));
        L.q_dirty = true;
        L.fresh = false;
    }
    static const int tree_max = [] { const char *e = tune_env("TAXOR_SMALL_TREE"); return e ? atoi(e) : 1024; }();
    const bool tree = (int)n <= tree_max;
    if (tree && L.q_dirty) {
        HIP_TRY(hipMemsetAsync(c->d_q[0].p, 0xFF, c->d_q[0].cap * sizeof(uint2), st));      // the one queue of the tree launch: every slot empty (~0)
        L.q_dirty = false;
    }
    if (!tree) L.q_dirty = true;
    // ---- two copies
    HIP_TRY(hipMemcpyAsync(L.d_in.p, L.h_in, in_bytes, hipMemcpyHostToDevice, st));
    t_arrays = us();
    // Page-locked input: the copies of all pieces are enqueued within microseconds, and on four streams they would cross PCIe side
    // by side and all arrive late; each waits for its predecessor's instead, so the pieces arrive -- and start -- one after the
    // other as from pageable memory (where the call itself blocks until the bytes are over)
    if (copy_after) HIP_TRY(hipStreamWaitEvent(st, copy_after, 0));
    // (Experiment of round 5, off by default: TAXOR_SMALL_STAGE=1 has the CALLING thread memcpy the piece's bases into a page-locked
    // buffer of the lane, so that no copy enters the runtime's blocking pageable path.  A single thread copies ~10 GB/s: the 10 MB of a
    // 1024 x 10 kb call cost more than the runtime's pin-in-place copy saves -- measured in profiles/r05/small_calls.txt.)
    static const bool stage_bases = [] { const char *e = tune_env("TAXOR_SMALL_STAGE"); return e && atoi(e) != 0; }();
    const char *src = bases + a0;
    if (stage_bases && nb) {
        if (L.h_bases_cap < nb) {
            if (L.h_bases) { HIP_TRY(hipStreamSynchronize(st)); (void)hipHostFree(L.h_bases); }
            L.h_bases = nullptr;
            L.h_bases_cap = round_up(nb + (nb >> 2), 1u << 20);
            HIP_TRY(hipHostMalloc(&L.h_bases, L.h_bases_cap, hipHostMallocDefault));
        }
        memcpy(L.h_bases, src, nb);
        src = (const char *)L.h_bases;
    }
    if (nb) HIP_TRY(hipMemcpyAsync(c->d_ascii.p, src, nb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(L.copied, st));
    t_bases = us();
    const uint64_t *d_aoff = (const uint64_t *)(L.d_in.p + o_aoff), *d_poff = (const uint64_t *)(L.d_in.p + o_poff),
                   *d_hoff = (const uint64_t *)(L.d_in.p + o_hoff);
    const uint32_t *d_rlen = (const uint32_t *)(L.d_in.p + o_rlen), *d_hcap = (const uint32_t *)(L.d_in.p + o_hcap),
                   *d_order = (const uint32_t *)(L.d_in.p + o_order);
    // ---- pack, syncmers
    launch_pack_dna4(c->d_ascii.p, d_aoff, d_poff, c->d_packed.p, n, c->d_ctr, st);
    SyncmerArgs a{};
    a.packed = c->d_packed.p;
    a.poff = d_poff;
    a.rlen = d_rlen;
    a.hoff = d_hoff;
    a.hcap = d_hcap;
    a.cand = c->d_cand[0].p;
    a.hashes = c->d_hashes[0].p;
    a.cursor = c->d_sync_cursor.p;
    a.order = d_order;
    a.nh = c->d_nh.p;
    a.thr = c->d_thr.p;
    a.ratio = s->prm.ratio;
    a.scaling_limit = idx->scaling > 1 ? (double)UINT64_MAX / (double)idx->scaling : 0.0;
    a.gtab = nullptr;
    a.gtab_stride = 0;
    a.ctr = c->d_ctr;
    a.chunk = 1;                       // few reads: one per cursor atomic, every block gets one as early as possible
    a.k = idx->k;
    a.s = idx->s;
    a.t = idx->t;
    a.w_min = 0;
    a.thr_on_device = 1;
    if (n_long) {
        a.n_reads = n_long;
        launch_syncmers(a, std::min<int>(c->grid_sync, (int)n_long), st);
    }
    if (n_long < n) {
        SyncmerArgs b = a;
        b.order = d_order + n_long;
        b.n_reads = n - n_long;
        b.cursor = a.cursor + 1;
        b.chunk = 4;
        launch_syncmers_wave(b, c->grid_wave, st);
    }
    HIP_TRY(hipGetLastError());
    t_hash = us();
    // ---- the levels; root items in column parts until the piece has about a grid's worth of them
    uint32_t parts = 1;
    static const int parts_env = [] { const char *e = tune_env("TAXOR_SMALL_PARTS"); return e ? atoi(e) : 0; }();
    while (parts < idx->root_pmax && (uint64_t)n * parts * 2 <= (uint64_t)c->grid_query_short / 2) parts *= 2;
    if (parts_env >= 1) {          // rounded down to a power of two: the divisions are halvings of the widest one (root_pmax)
        parts = 1;
        while (parts * 2 <= (uint32_t)parts_env && parts * 2 <= idx->root_pmax) parts *= 2;
    }
    if (s->split_always) parts = idx->root_pmax;
    c->stats = taxor_gpu_run_stats{};
    // one launch for the whole traversal up to TREE_MAX reads; larger pieces keep the chip busy level by level, and from 4096 reads
    // on the grouping of the work items by IXF pays (run_query)
    if (int rc = run_query(c, c->d_hashes[0].p, d_hoff, c->d_nh.p, c->d_thr.p, n, nullptr, 1, nullptr, -1, d_order, tree ? parts : 1u, false, tree)) return rc;
    t_query = us();
    // ---- CSR assembly into host memory, counters cleared for the lane's next piece
    const SmallOut o = small_out(L, L.d_out);
    SmallFinalizeArgs f{};
    f.hits = c->d_hits.p;
    f.read_hits = c->d_read_hits.p;
    f.dfs_key = idx->d_dfs_key;
    f.ubin = idx->d_ubin;
    f.nh = c->d_nh.p;
    f.key = c->d_out_key.p;
    f.cnt = c->d_out_cnt.p;
    f.ub = c->d_out_ub.p;
    f.ctr = c->d_ctr;
    f.sync_cursor = c->d_sync_cursor.p;
    f.n_reads = n;
    f.tuple_cap = (uint32_t)std::min<uint64_t>({c->tuple_cap, (uint64_t)L.out_tuples, 0x7FFFFFFFull});
    f.hit_cap = c->hit_cap;
    f.h_read_off = o.read_off;
    f.h_nh = o.nh;
    f.h_ub = o.ub;
    f.h_cnt = o.cnt;
    f.h_status = o.status;
    launch_finalize_small(f, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(L.done, st));
    if (trace)
        fprintf(stderr, "[small piece %u: %u reads] lane ready %.0f us, layout %.0f, buffers %.0f, arrays copy issued %.0f, bases copied %.0f, pack+syncmers launched %.0f, "
                        "traversal launched %.0f, finalize + event %.0f\n", li, n, t_ready, t_layout, t_scratch, t_arrays, t_bases, t_hash, t_query, us());
    return 0;
}

// the results of the next piece in line -> the host arrays (waits for its lane); reruns it through the other pipeline if a
// queue, the hit buffer or the result area was too small for it
int small_harvest_one(taxor_gpu_searcher *s)
{
    const auto &pc = s->small_pieces[s->small_harvested];
    taxor_gpu_searcher::SmallLane &L = s->lanes[pc.lane];
    taxor_gpu_run_stats &st = s->stats;
    {   // the caller waits for a fraction of a millisecond: poll (a blocking wait's wake-up is tens of microseconds of it)
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e;
        while ((e = hipEventQuery(L.done)) == hipErrorNotReady)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) { e = hipEventSynchronize(L.done); break; }
        if (e != hipSuccess) return fail(TAXOR_E_HIP, "small batch: %s", hipGetErrorString(e));
    }
    const SmallOut o = small_out(L, L.h_out);
    const uint32_t f = (uint32_t)o.status[0];
    if (f & FLAG_ALPHABET) { L.fresh = true; return fail(TAXOR_E_ALPHABET, "a read contains a character outside the dna15 alphabet"); }
    if (f & FLAG_CAND_OVERFLOW) { L.fresh = true; return fail(TAXOR_E_INTERNAL, "syncmer candidate capacity bound violated"); }
    if (f & FLAG_DEDUP_OVERFLOW) { L.fresh = true; return fail(TAXOR_E_INTERNAL, "dedup scratch too small"); }
    const uint64_t *ro;
    const int64_t *ub;
    const uint32_t *cnt, *nh;
    uint64_t nt;
    uint32_t tree_stalls = 0;
    taxor_gpu_results r{};
    if (f & (FLAG_QUEUE_OVERFLOW | FLAG_HITS_OVERFLOW | FLAG_TUPLE_OVERFLOW | FLAG_TREE_STALL)) {
        // the lane classifies the piece once more through the pipeline of large batches, which grows its buffers and reruns until
        // everything fits (check_flags).  The same for a one-launch traversal whose watchdog fired (a block gave up waiting for
        // its queue slot): that launch's hits are partial, the level-by-level pipeline has no spin-wait, and L.fresh has the lane's
        // counters and queue cleared before its next piece -- a stall costs a rerun, never the batch
        L.fresh = true;
        if (f & FLAG_TREE_STALL) ++tree_stalls;
        if (ensure_stream(&L.c->st_sync)) return TAXOR_E_HIP;        // (a lane has no second stream of its own; that pipeline wants one)
        if (int rc = taxor_gpu_search_batch(L.c, s->small_bases, s->small_offsets + pc.first, pc.n, &r)) return rc;
        ro = r.read_off; ub = r.user_bin; cnt = r.count; nh = r.n_hashes; nt = r.n_tuples;
        st.n_hashes += L.c->stats.n_hashes;
        st.n_work_items += L.c->stats.n_work_items;
        st.query_bytes += L.c->stats.query_bytes;
        st.query_touched_bytes += L.c->stats.query_touched_bytes;
        st.tree_stalls_recovered += tree_stalls;
    } else {
        ro = o.read_off; ub = o.ub; cnt = o.cnt; nh = o.nh; nt = o.status[1];
        st.n_hashes += o.status[2];
        st.n_work_items += o.status[3];
        st.query_bytes += o.status[4];
        st.query_touched_bytes += o.status[5];
    }
    const uint64_t tbase = s->small_tbase;
    for (uint64_t i = 0; i < pc.n; ++i) s->h_read_off[pc.first + i + 1] = tbase + ro[i + 1];
    memcpy(s->h_nh.data() + pc.first, nh, pc.n * 4);
    s->h_ub.insert(s->h_ub.end(), ub, ub + nt);
    s->h_cnt.insert(s->h_cnt.end(), cnt, cnt + nt);
    s->small_tbase += nt;
    ++s->small_harvested;
    return 0;
}

int small_begin(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
    HIP_TRY(hipSetDevice(s->idx->device));
    s->ran = s->synced = false;
    s->small_active = s->small_done = false;
    s->small_pieces.clear();
    s->small_harvested = 0;
    s->small_tbase = 0;
    s->small_bases = bases;
    s->small_offsets = offsets;
    s->n_reads = n_reads;
    s->n_bases = offsets[n_reads] - offsets[0];
    s->h_read_off.resize(n_reads + 1);
    s->h_nh.resize(n_reads);
    s->h_ub.clear();
    s->h_cnt.clear();
    s->h_read_off[0] = 0;
    s->stats = taxor_gpu_run_stats{};
    // Pieces.  A blocking copy of pageable bases costs ~35 us beyond its bytes (the runtime page-locks the caller's pages around it)
    // and a piece's chain of launches ~270 us however small it is, so few pieces: one up to 512 reads; two -- five and three eighths
    // of the call -- up to 1280, four equal ones up to 2048 (1024 reads, A/B on one box: 640+384 0.770 ms, 512+512 0.770, 768+256 0.794, 4 x 256 0.795,
    // 384+384+256 0.800; profiles/r04/small_calls.txt).  Beyond: a first piece of an eighth of the call -- nothing runs until its
    // bases have crossed PCIe -- and the rest in equal pieces of at most SMALL_FIN_MAX reads; piece p runs on lane p mod 4, and a
    // lane is reused once its previous piece's results have been taken over.
    static const uint64_t piece_env = [] { const char *e = tune_env("TAXOR_SMALL_PIECE"); const long v = e ? atol(e) : 0; return (uint64_t)(v > 0 ? v : 0); }();
    std::vector<uint64_t> sizes;
    static const std::vector<uint64_t> sizes_env = [] {       // TAXOR_SMALL_SIZES=768,256: these pieces for a call of exactly their sum (experiments)
        std::vector<uint64_t> v;
        if (const char *e = tune_env("TAXOR_SMALL_SIZES"))
            for (const char *p = e; *p;) { char *q; const unsigned long x = strtoul(p, &q, 10); if (q == p) break; if (x) v.push_back(x); p = *q ? q + 1 : q; }
        return v;
    }();
    uint64_t env_sum = 0;
    for (uint64_t x : sizes_env) env_sum += x;
    if (env_sum == n_reads && !sizes_env.empty() && *std::max_element(sizes_env.begin(), sizes_env.end()) <= SMALL_FIN_MAX) sizes = sizes_env;
    else if (piece_env) {
        for (uint64_t f = 0; f < n_reads; f += std::min<uint64_t>(piece_env, SMALL_FIN_MAX)) sizes.push_back(std::min<uint64_t>(std::min<uint64_t>(piece_env, SMALL_FIN_MAX), n_reads - f));
    } else if (n_reads <= 512) {
        sizes.push_back(n_reads);
    } else if (n_reads <= 1280) {
        const uint64_t a = round_up(n_reads * 5 / 8, 64);
        sizes.push_back(std::min(a, n_reads));
        if (a < n_reads) sizes.push_back(n_reads - a);
    } else if (n_reads <= 2048) {
        const uint64_t per = round_up((n_reads + SMALL_LANES - 1) / SMALL_LANES, 64);          // (2048 reads: 4 x 512 1.24-1.30 ms, 1280 + 768 1.51)
        for (uint64_t f = 0; f < n_reads; f += per) sizes.push_back(std::min(per, n_reads - f));
    } else {
        const uint64_t first = std::max<uint64_t>(SMALL_PIECE_MIN, round_up(n_reads / 8, 64)), rest = n_reads - first;
        const uint64_t k = std::max<uint64_t>(3, (rest + SMALL_FIN_MAX - 1) / SMALL_FIN_MAX), per = std::min<uint64_t>(SMALL_FIN_MAX, round_up((rest + k - 1) / k, 64));
        sizes.push_back(first);
        for (uint64_t f = first; f < n_reads; f += per) sizes.push_back(std::min(per, n_reads - f));
    }
    if (s->lanes.size() < SMALL_LANES) s->lanes.resize(SMALL_LANES);
    uint64_t first = 0;
    for (size_t p = 0; p < sizes.size(); ++p) {
        const uint32_t li = (uint32_t)(p % SMALL_LANES);
        int rc = 0;
        while (rc == 0 && p >= SMALL_LANES && s->small_harvested + SMALL_LANES <= p) rc = small_harvest_one(s);   // the lane's previous piece first
        if (rc == 0) rc = small_enqueue(s, li, bases, offsets, first, (uint32_t)sizes[p], p ? s->lanes[(p - 1) % SMALL_LANES].copied : nullptr);
        if (rc) {                  // > 0: not a case for the lanes; wait for what is in flight, then the other pipeline (or the error)
            for (size_t q = s->small_harvested; q < s->small_pieces.size(); ++q) (void)hipEventSynchronize(s->lanes[s->small_pieces[q].lane].done);
            for (auto &L : s->lanes) L.fresh = true;
            s->small_pieces.clear();
            return rc;
        }
        s->small_pieces.push_back({li, first, sizes[p]});
        first += sizes[p];
    }
    s->small_active = true;
    s->ran = true;
    return 0;
}

// wait for the pieces, in order, and put the call's CSR together in the host arrays the results point at
int small_finish(taxor_gpu_searcher *s)
{
    if (s->small_done) return 0;
    while (s->small_harvested < s->small_pieces.size())
        if (int rc = small_harvest_one(s)) return rc;
    const uint64_t nr = s->n_reads, tbase = s->small_tbase;
    taxor_gpu_run_stats &st = s->stats;
    st.n_reads = nr;
    st.n_bases = s->n_bases;
    st.n_tuples = tbase;
    st.query_launches = (uint32_t)(s->small_pieces.size() * s->idx->depth);
    uint64_t packed_in = 0;
    for (uint64_t r = 0; r < nr; ++r) packed_in += (s->small_offsets[r + 1] - s->small_offsets[r] + 3) / 4;
    st.algorithmic_bytes = packed_in + st.query_bytes + 8 * nr + 12 * tbase;
    s->h_ctr.tuple_total = tbase;
    s->small_done = true;
    s->synced = true;
    s->dev_results_stale = true;
    return 0;
}

// export_device / the communicator read the device-resident CSR of the last run: after a call that went through the lanes it
// is put there from the host arrays (a few hundred kilobytes; the end-of-file batch of a multi-GPU CLI run)
int small_device_results(taxor_gpu_searcher *s)
{
    if (!s->dev_results_stale) return 0;
    const uint64_t nr = s->n_reads, nt = s->h_ctr.tuple_total;
    if (s->d_read_off.reserve(nr + 1) || s->d_nh.reserve(nr + 1) || s->d_out_ub.reserve(nt + 1) || s->d_out_cnt.reserve(nt + 1)) return TAXOR_E_HIP;
    HIP_TRY(hipMemcpyAsync(s->d_read_off.p, s->h_read_off.data(), (nr + 1) * 8, hipMemcpyHostToDevice, s->st));
    if (nr) HIP_TRY(hipMemcpyAsync(s->d_nh.p, s->h_nh.data(), nr * 4, hipMemcpyHostToDevice, s->st));
    if (nt) {
        HIP_TRY(hipMemcpyAsync(s->d_out_ub.p, s->h_ub.data(), nt * 8, hipMemcpyHostToDevice, s->st));
        HIP_TRY(hipMemcpyAsync(s->d_out_cnt.p, s->h_cnt.data(), nt * 4, hipMemcpyHostToDevice, s->st));
    }
    HIP_TRY(hipStreamSynchronize(s->st));
    s->dev_results_stale = false;
    return 0;
}

} // namespace

extern "C" int taxor_gpu_batch_upload(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
    if (s) s->small_active = false;
    if (int rc = prepare_batch(s, bases, offsets, n_reads, false)) return rc;
    const uint64_t a0 = offsets[0], nb = offsets[n_reads] - a0;
    if (nb) HIP_TRY(hipMemcpyAsync(s->d_ascii.p, bases + a0, nb, hipMemcpyHostToDevice, s->st));
    launch_pack_dna4(s->d_ascii.p, s->d_aoff.p, s->d_poff.p, s->d_packed.p, (uint32_t)n_reads, s->d_ctr, s->st);
    HIP_TRY(hipGetLastError());
    bool rerun;
    return check_flags(s, &rerun); // synchronises; reports a non-dna15 character
}

extern "C" int taxor_gpu_batch_run(taxor_gpu_searcher *s)
{
    if (!s) return fail(TAXOR_E_ARG, "batch_run: null searcher");
    HIP_TRY(hipSetDevice(s->idx->device));
    return run_pipeline(s, false);
}

extern "C" int taxor_gpu_batch_sync(taxor_gpu_searcher *s)
{
    if (!s) return fail(TAXOR_E_ARG, "batch_sync: null searcher");
    if (!s->ran) return fail(TAXOR_E_ARG, "batch_sync: no run in flight");
    HIP_TRY(hipSetDevice(s->idx->device));
    if (s->small_active) return small_finish(s);
    for (int attempt = 0; attempt < 40; ++attempt) {
        bool rerun;
        if (int rc = check_flags(s, &rerun)) return rc;
        if (!rerun) {
            s->synced = true;
            // stats
            taxor_gpu_run_stats &st = s->stats;
            st.n_reads = s->n_reads;
            st.n_bases = s->n_bases;
            st.n_hashes = s->h_ctr.n_hashes;
            st.n_tuples = s->h_ctr.tuple_total;
            st.n_work_items = s->h_ctr.n_work;
            st.query_bytes = s->h_ctr.query_bytes;
            st.query_touched_bytes = s->h_ctr.touched_bytes;
            st.algorithmic_bytes = s->packed_in_bytes + st.query_bytes + 8 * st.n_reads + 12 * st.n_tuples;
            st.query_ms = st.syncmer_ms = st.finalize_ms = st.total_ms = 0.f;
            for (int l = 0; l < 8; ++l) {
                st.level_ms[l] = 0.f;
                st.level_requested_bytes[l] = s->h_ctr.lvl_touched[l];
                st.level_row_reads[l] = s->h_ctr.lvl_rows[l];
                st.level_sparse_loads[l] = s->h_ctr.lvl_sparse[l];
            }
            for (auto &sp : s->ev_spans) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, s->ev[sp.first], s->ev[sp.first + 1]) != hipSuccess) continue;
                if (sp.second == 0) st.syncmer_ms += ms;
                else if (sp.second >= 16) { st.query_ms += ms; st.level_ms[sp.second - 16] += ms; }
                else if (sp.second == 2) st.finalize_ms += ms;
                else st.total_ms += ms;
            }
            return TAXOR_OK;
        }
        if (int rc = taxor_gpu_batch_run(s)) return rc; // capacities were grown; everything is deterministic
    }
    return fail(TAXOR_E_INTERNAL, "batch_sync: buffers kept overflowing");
}

extern "C" int taxor_gpu_batch_stats(taxor_gpu_searcher *s, taxor_gpu_run_stats *out)
{
    if (!s || !out || !s->synced) return fail(TAXOR_E_ARG, "batch_stats: no completed run");
    *out = s->stats;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_phase_profile(taxor_gpu_searcher *s, uint64_t *cycles16)
{
    if (!s || !cycles16) return fail(TAXOR_E_ARG, "phase_profile: null argument");
    if (!s->d_prof) return fail(TAXOR_E_ARG, "phase_profile: searcher was created without TAXOR_PROFILE_PHASES=1");
    HIP_TRY(hipSetDevice(s->idx->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(cycles16, s->d_prof, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(s->d_prof, 0, 16 * sizeof(uint64_t)));
    return TAXOR_OK;
}

extern "C" int taxor_gpu_batch_result_sizes(taxor_gpu_searcher *s, uint64_t *n_reads, uint64_t *n_tuples)
{
    if (!s) return fail(TAXOR_E_ARG, "result_sizes: null searcher");
    if (!s->synced)
        if (int rc = taxor_gpu_batch_sync(s)) return rc;
    if (n_reads) *n_reads = s->n_reads;
    if (n_tuples) *n_tuples = s->h_ctr.tuple_total;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_batch_export_device(taxor_gpu_searcher *s, void *d_read_off, void *d_user_bin, void *d_count,
                                             void *d_n_hashes)
{
    if (!s) return fail(TAXOR_E_ARG, "export_device: null searcher");
    if (!s->synced)
        if (int rc = taxor_gpu_batch_sync(s)) return rc;
    if (s->small_active)
        if (int rc = small_device_results(s)) return rc;
    const uint64_t nt = s->h_ctr.tuple_total, nr = s->n_reads;
    if (d_read_off) HIP_TRY(hipMemcpyAsync(d_read_off, s->d_read_off.p, (nr + 1) * 8, hipMemcpyDeviceToDevice, s->st));
    if (d_user_bin && nt) HIP_TRY(hipMemcpyAsync(d_user_bin, s->d_out_ub.p, nt * 8, hipMemcpyDeviceToDevice, s->st));
    if (d_count && nt) HIP_TRY(hipMemcpyAsync(d_count, s->d_out_cnt.p, nt * 4, hipMemcpyDeviceToDevice, s->st));
    if (d_n_hashes && nr) HIP_TRY(hipMemcpyAsync(d_n_hashes, s->d_nh.p, nr * 4, hipMemcpyDeviceToDevice, s->st));
    HIP_TRY(hipStreamSynchronize(s->st));
    return TAXOR_OK;
}

// library-internal (comm.hip): the device-resident CSR of the last run, for the RCCL gather.  Synchronises the run.
extern "C" __attribute__((visibility("hidden"))) int taxor_searcher_device_results(taxor_gpu_searcher *s, const uint64_t **d_read_off,
                                                                                   const int64_t **d_user_bin, const uint32_t **d_count,
                                                                                   const uint32_t **d_n_hashes, uint64_t *n_reads,
                                                                                   uint64_t *n_tuples, int *device)
{
    if (!s) return fail(TAXOR_E_ARG, "device_results: null searcher");
    if (!s->synced)
        if (int rc = taxor_gpu_batch_sync(s)) return rc;
    if (s->small_active)
        if (int rc = small_device_results(s)) return rc;
    *d_read_off = s->d_read_off.p;
    *d_user_bin = s->d_out_ub.p;
    *d_count = s->d_out_cnt.p;
    *d_n_hashes = s->d_nh.p;
    *n_reads = s->n_reads;
    *n_tuples = s->h_ctr.tuple_total;
    *device = s->idx->device;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_batch_fetch(taxor_gpu_searcher *s, taxor_gpu_results *out)
{
    if (!s || !out) return fail(TAXOR_E_ARG, "batch_fetch: null argument");
    if (!s->synced)
        if (int rc = taxor_gpu_batch_sync(s)) return rc;
    if (s->small_active) {              // the lanes' results are in the host arrays already (small_finish)
        out->n_reads = s->n_reads;
        out->n_tuples = s->h_ctr.tuple_total;
        out->read_off = s->h_read_off.data();
        out->user_bin = s->h_ub.data();
        out->count = s->h_cnt.data();
        out->n_hashes = s->h_nh.data();
        return TAXOR_OK;
    }
    const uint64_t nt = s->h_ctr.tuple_total, nr = s->n_reads;
    s->h_read_off.resize(nr + 1);
    s->h_ub.resize(nt);
    s->h_cnt.resize(nt);
    s->h_nh.resize(nr);
    // A copy into pageable memory is a host round trip of its own (~35 us): four of them are a tenth of a call of 1024 reads,
    // the reference's chunk size (taxor_search.cpp:315).  Small results land in a page-locked area first -- four copies in
    // flight, one wait -- and are moved to the result arrays by the host.
    constexpr uint64_t kSmall = 2ull << 20;
    const uint64_t b_off = (nr + 1) * 8, b_ub = nt * 8, b_cnt = nt * 4, b_nh = nr * 4;
    if (b_off + b_ub + b_cnt + b_nh + 64 <= kSmall && (s->h_small || hipHostMalloc(&s->h_small, kSmall, hipHostMallocDefault) == hipSuccess)) {
        char *p0 = (char *)s->h_small, *p1 = p0 + b_off, *p2 = p1 + b_ub, *p3 = p2 + ((b_cnt + 7) & ~7ull);
        HIP_TRY(hipMemcpyAsync(p0, s->d_read_off.p, b_off, hipMemcpyDeviceToHost, s->st));
        if (nt) {
            HIP_TRY(hipMemcpyAsync(p1, s->d_out_ub.p, b_ub, hipMemcpyDeviceToHost, s->st));
            HIP_TRY(hipMemcpyAsync(p2, s->d_out_cnt.p, b_cnt, hipMemcpyDeviceToHost, s->st));
        }
        if (nr) HIP_TRY(hipMemcpyAsync(p3, s->d_nh.p, b_nh, hipMemcpyDeviceToHost, s->st));
        HIP_TRY(hipStreamSynchronize(s->st));
        memcpy(s->h_read_off.data(), p0, b_off);
        if (nt) { memcpy(s->h_ub.data(), p1, b_ub); memcpy(s->h_cnt.data(), p2, b_cnt); }
        if (nr) memcpy(s->h_nh.data(), p3, b_nh);
    } else {
        (void)hipGetLastError();
        HIP_TRY(hipMemcpyAsync(s->h_read_off.data(), s->d_read_off.p, (nr + 1) * 8, hipMemcpyDeviceToHost, s->st));
        if (nt) {
            HIP_TRY(hipMemcpyAsync(s->h_ub.data(), s->d_out_ub.p, nt * 8, hipMemcpyDeviceToHost, s->st));
            HIP_TRY(hipMemcpyAsync(s->h_cnt.data(), s->d_out_cnt.p, nt * 4, hipMemcpyDeviceToHost, s->st));
        }
        if (nr) HIP_TRY(hipMemcpyAsync(s->h_nh.data(), s->d_nh.p, nr * 4, hipMemcpyDeviceToHost, s->st));
        HIP_TRY(hipStreamSynchronize(s->st));
    }
    out->n_reads = nr;
    out->n_tuples = nt;
    out->read_off = s->h_read_off.data();
    out->user_bin = s->h_ub.data();
    out->count = s->h_cnt.data();
    out->n_hashes = s->h_nh.data();
    return TAXOR_OK;
}

extern "C" int taxor_gpu_search_batch_begin(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads)
{
    if (s && bases && offsets && small_applicable(s, offsets, n_reads)) {
        const int rc = small_begin(s, bases, offsets, n_reads);
        if (rc <= 0) return rc;             // done, or an error; > 0: not a case for the lanes after all
    }
    if (s) s->small_active = false;
    // streamed: the bases of sub-batch i+1 are copied and packed while sub-batch i is being classified
    if (int rc = prepare_batch(s, bases, offsets, n_reads, true)) return rc;
    // (page-locking the caller's buffer for the duration of the call was measured and is slower: the registration
    // costs more than the pageable staging it saves -- 55 vs 50 ms for 1.3 GB; a caller that re-uses its buffer
    // registers it once with taxor_gpu_host_register, and then this call returns as soon as everything is enqueued)
    s->host_spans.clear();
    if (n_reads) s->host_spans.push_back({0, offsets[n_reads] - offsets[0], bases + offsets[0]});
    return run_pipeline(s, n_reads != 0);
}

extern "C" int taxor_gpu_search_segments_begin(taxor_gpu_searcher *s, const taxor_read_segment *segs, uint64_t n_segs)
{
    if (!s || (n_segs && !segs)) return fail(TAXOR_E_ARG, "search_segments_begin: null argument");
    // one batch over the reads of all segments, in segment order: the offsets are concatenated here (8 bytes per read),
    // the bases stay where they are and cross PCIe sub-batch by sub-batch straight from their segments
    uint64_t n_reads = 0;
    for (uint64_t j = 0; j < n_segs; ++j) {
        if (segs[j].n_reads && (!segs[j].offsets || !segs[j].bases)) return fail(TAXOR_E_ARG, "search_segments_begin: segment %llu is null", (unsigned long long)j);
        n_reads += segs[j].n_reads;
    }
    std::vector<uint64_t> off(n_reads + 1);
    std::vector<taxor_gpu_searcher::HostSpan> spans;
    uint64_t r = 0, vb = 0;
    off[0] = 0;
    for (uint64_t j = 0; j < n_segs; ++j) {
        const taxor_read_segment &g = segs[j];
        if (!g.n_reads) continue;
        const uint64_t a0 = g.offsets[0];
        for (uint64_t i = 0; i < g.n_reads; ++i) {
            if (g.offsets[i + 1] < g.offsets[i]) return fail(TAXOR_E_ARG, "search_segments_begin: offsets of segment %llu not monotone", (unsigned long long)j);
            off[r + i + 1] = vb + (g.offsets[i + 1] - a0);
        }
        const uint64_t len = g.offsets[g.n_reads] - a0;
        if (len) spans.push_back({vb, len, g.bases + a0});
        r += g.n_reads;
        vb += len;
    }
    static const char nothing = 0;
    s->small_active = false;
    if (int rc = prepare_batch(s, &nothing, off.data(), n_reads, true)) return rc;
    s->host_spans = std::move(spans);
    return run_pipeline(s, n_reads != 0);
}

extern "C" int taxor_gpu_search_batch_end(taxor_gpu_searcher *s, taxor_gpu_results *out) { return taxor_gpu_batch_fetch(s, out); }

extern "C" int taxor_gpu_search_batch(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                                      taxor_gpu_results *out)
{
    static const bool trace = tune_env("TAXOR_TRACE_BATCH") != nullptr;   // phase times of this call on stderr
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    if (int rc = taxor_gpu_search_batch_begin(s, bases, offsets, n_reads)) return rc;
    const double t_enq = ms_since(t0);
    const int rc = taxor_gpu_search_batch_end(s, out);
    if (trace)
        fprintf(stderr, "[search_batch] %llu reads: pipeline enqueued (layout, per-read arrays, blocking copies of pageable bases) at "
                        "%.2f ms, results on the host at %.2f ms\n", (unsigned long long)n_reads, t_enq, ms_since(t0));
    return rc;
}

// =========================================================================================================
// stage entry points
// =========================================================================================================
extern "C" int taxor_gpu_syncmers(taxor_gpu_searcher *s, const char *bases, const uint64_t *offsets, uint64_t n_reads,
                                  const uint64_t **hash_off, const uint64_t **hashes)
{
    if (!s || !hash_off || !hashes) return fail(TAXOR_E_ARG, "syncmers: null argument");
    if (int rc = taxor_gpu_batch_upload(s, bases, offsets, n_reads)) return rc;
    s->h_hash_off.assign(n_reads + 1, 0);
    s->h_hashes.clear();
    s->h_nh.resize(n_reads);
    std::vector<uint64_t> tmp, hoff_h;
    if (reset_sub_counters(s, true)) return TAXOR_E_HIP;
    for (size_t i = 0; i < s->subs.size(); ++i) {
        const SubBatch &sb = s->subs[i];
        if (i && reset_sub_counters(s, false)) return TAXOR_E_HIP;
        HIP_TRY(hipMemsetAsync(s->d_sync_cursor.p, 0, 2 * sizeof(uint32_t), s->st));
        if (int rc = launch_syncmers_sub(s, sb, 0, 0, s->st)) return rc;
        bool rerun;
        if (int rc = check_flags(s, &rerun)) return rc;
        tmp.resize(sb.slots);
        hoff_h.resize(sb.n);
        HIP_TRY(hipMemcpy(s->h_nh.data() + sb.first, s->d_nh.p + sb.first, sb.n * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hoff_h.data(), s->d_hoff.p + sb.first, sb.n * 8, hipMemcpyDeviceToHost));
        if (sb.slots) HIP_TRY(hipMemcpy(tmp.data(), s->d_hashes[0].p, sb.slots * 8, hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < sb.n; ++r) {
            const uint32_t nh = s->h_nh[sb.first + r];
            s->h_hash_off[sb.first + r + 1] = s->h_hash_off[sb.first + r] + nh;
            s->h_hashes.insert(s->h_hashes.end(), tmp.begin() + hoff_h[r], tmp.begin() + hoff_h[r] + nh);
        }
    }
    *hash_off = s->h_hash_off.data();
    *hashes = s->h_hashes.data();
    return TAXOR_OK;
}

namespace {

// stage a single pseudo-read whose hash list and threshold are given
int stage_hash_list(taxor_gpu_searcher *s, const uint64_t *hashes, uint64_t n, uint64_t threshold)
{
    if (n >= (1ull << 32)) return fail(TAXOR_E_ARG, "hash list too long");
    HIP_TRY(hipSetDevice(s->idx->device));
    s->ran = s->synced = false;
    s->small_active = false;
    s->n_reads = 1;
    s->n_bases = 0;
    s->mean_read_len = 1u << 20;
    s->subs.clear();
    s->max_slots = n + 64;
    s->max_read_slots = 16;
    s->max_sub_reads = 1;
    if (s->d_hoff.reserve(2) || s->d_nh.reserve(2) || s->d_thr.reserve(2)) return TAXOR_E_HIP;
    if (int rc = ensure_scratch(s)) return rc;
    const uint64_t zero = 0;
    const uint32_t nh = (uint32_t)n;
    if (n) HIP_TRY(hipMemcpy(s->d_hashes[0].p, hashes, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_hoff.p, &zero, 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_nh.p, &nh, 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_thr.p, &threshold, 8, hipMemcpyHostToDevice));
    return 0;
}

} // namespace

extern "C" int taxor_gpu_ixf_bulk_count(taxor_gpu_searcher *s, uint64_t ixf, const uint64_t *hashes, uint64_t n,
                                        uint32_t *counts)
{
    if (!s || !counts || (n && !hashes)) return fail(TAXOR_E_ARG, "ixf_bulk_count: null argument");
    if (ixf >= s->idx->h_ixf.size()) return fail(TAXOR_E_ARG, "ixf_bulk_count: bad IXF id");
    if (int rc = stage_hash_list(s, hashes, n, ~0ull)) return rc;
    const uint32_t bins = s->idx->h_ixf[ixf].bins;
    uint32_t *d_counts = nullptr;
    HIP_TRY(hipMalloc((void **)&d_counts, (size_t)bins * 4));
    int rc = reset_sub_counters(s, true);
    const uint2 item = make_uint2(0u, (uint32_t)ixf);
    const uint32_t one = 1;
    hipError_t e = hipMemcpyAsync(s->d_q[0].p, &item, sizeof item, hipMemcpyHostToDevice, s->st);
    if (e == hipSuccess) e = hipMemcpyAsync(&s->d_ctr->q_n[0].v, &one, 4, hipMemcpyHostToDevice, s->st);
    if (e == hipSuccess) e = hipStreamSynchronize(s->st);
    if (e != hipSuccess) rc = fail(TAXOR_E_HIP, "ixf_bulk_count: %s", hipGetErrorString(e));
    s->ev_used = 0;
    s->ev_spans.clear();
    if (!rc) rc = run_query(s, s->d_hashes[0].p, s->d_hoff.p, s->d_nh.p, s->d_thr.p, 1, nullptr, 1, d_counts, (int)ixf);
    if (!rc) {
        e = hipStreamSynchronize(s->st);
        if (e == hipSuccess) e = hipMemcpy(counts, d_counts, (size_t)bins * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TAXOR_E_HIP, "ixf_bulk_count: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_counts);
    return rc;
}

extern "C" int taxor_gpu_bulk_contains(taxor_gpu_searcher *s, const uint64_t *hashes, uint64_t n, uint64_t threshold,
                                       taxor_gpu_results *out)
{
    if (!s || !out || (n && !hashes)) return fail(TAXOR_E_ARG, "bulk_contains: null argument");
    if (int rc = stage_hash_list(s, hashes, n, threshold)) return rc;
    for (int attempt = 0; attempt < 40; ++attempt) {
        if (int rc = ensure_scratch(s)) return rc;
        if (reset_sub_counters(s, true)) return TAXOR_E_HIP;
        s->ev_used = 0;
        s->ev_spans.clear();
        if (int rc = run_query(s, s->d_hashes[0].p, s->d_hoff.p, s->d_nh.p, s->d_thr.p, 1, s->d_read_off.p, 1, nullptr, -1))
            return rc;
        bool rerun;
        if (int rc = check_flags(s, &rerun)) return rc;
        if (!rerun) {
            s->ran = s->synced = true;
            return taxor_gpu_batch_fetch(s, out);
        }
    }
    return fail(TAXOR_E_INTERNAL, "bulk_contains: buffers kept overflowing");
}
