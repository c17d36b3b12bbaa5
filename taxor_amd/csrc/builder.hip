// builder.hip -- GPU construction of IXF fingerprint columns, one IXF or a whole hierarchy.
//
// The reference builds an IXF on the CPU, bin by bin: add_bin_elements(bin, hashes) peels the 3-uniform hypergraph of the
// bin's keys and assigns fingerprints in reverse peeling order; if any bin fails to peel the whole IXF is cleared, re-seeded
// and rebuilt (src/hixf/build/construct_ixf.cpp:50-165, reseed loop :100-108; XOR-filter algorithm family of
// src/main/xorfilter.hpp:142-334; the hierarchy bottom-up, a merged bin holding everything below it:
// src/hixf/build/hierarchical_build.cpp:27-236).  Bins are independent, so a CHUNK of bins -- of one IXF or of many IXFs of one
// level -- is peeled at once.  What bounds this on MI355X is the rate of random read-modify-writes (18-27 G/s whatever their
// scope, width or return use: profiles/r06/atomics_bench.txt), so the design spends as few of them per key as it can, and none
// on a word that other lanes hit at the same time (a returning atomic on one address serialises at ~13 ns):
//
//   state   : ONE word per (bin,row) slot: degree in the low 8 bits, SUM of the incident keys' indices (position of the key in
//             its bin) above it.  One atomic add inserts a key into a row, one atomic sub removes it; a slot of degree 1 holds
//             the index of its only key.  32-bit words while every bin has < 2^26 keys (degree in 8 bits below 2^24 keys, in 6
//             beyond: the sum is kept modulo what is left, which is exact once one key remains), 64-bit words otherwise.
//   count   : 3 adds per key                                                                                 (k_count)
//   seed    : slots of degree 1 -> work list, one list append per BLOCK (entries collected in LDS)            (k_seed)
//   round t : the list is append-only; round t is its entries [end[t], end[t+1]), and every listed slot remembers the round it
//             was listed for (16 bits per slot).  A listed slot has degree 1 or 0 (only decrements happen).  Degree 1: the slot
//             names its key.  A key with several singleton rows in one round is peeled by the LOWEST-SEGMENT one of them: the
//             others see that a lower row of the key carries this round's mark and stand back -- no claim, no atomic, and the
//             choice does not depend on any race.  The peeling row logs the key's index at its list position and removes the
//             key from its OTHER two rows (its own slot is never looked at again); a row whose degree drops 2 -> 1 is appended
//             for round t+1 (it can never be appended twice).  Two read-modify-writes per key.  Appends: collected in LDS, one
//             returning atomic per block and 1024 entries.  The last WORKING block to finish a round (ticket counter; blocks
//             without entries leave at once) records where the list ends -- one launch per round, no launch in between, no
//             flags to set or retire                                                                              (k_round)
//   stop    : rounds are enqueued 32 at a time, two batches ahead of the host, which looks at the recorded ends of a batch
//             only when the next one is already queued: no device-to-host copy per round, the device never waits for the host.
//             Launches behind the last round find an empty range and return.
//   assign  : rounds in reverse.  A key peeled in round t is never incident to the singleton row of another key of round t
//             (that row had degree 1 when the round began), so a round is assigned in parallel: D[free] = fp ^ D[r'] ^ D[r''],
//             free = the row that peeled the key.  Which rows are listed for which round, and therefore which row peels which
//             key, is a function of (keys, seed) alone -- two builds of one index are byte-identical.  Small rounds (the long
//             plateau close to the peeling threshold) are assigned by ONE block that walks them with a barrier in between
//             instead of a launch each                                                                          (k_assign*)
//   verify  : every key is looked up in the finished columns (3 bytes per key); a mismatch is a bug and fails loudly.
//
// A bin that does not peel (its logged keys fall short) re-seeds ITS IXF only; the other IXFs of the chunk are
// finished, the failed one goes into the next chunk with a redrawn seed, like construct_ixf.cpp:100-108.
//
// taxor_gpu_index_build_hixf* builds a whole hierarchy level by level from the leaves up: the keys stay on the device, all IXFs of
// a level share chunks, and a merged bin's key set is the duplicate-free union of everything in its child IXF (keyset.hip: a hash
// set in HBM) -- for a child of leaf bins not a copy but the child's OWN key range plus one mark byte per key (BinJob::keep: the
// kernels skip unmarked keys), so that a level's unions cost an eighth of its keys in memory instead of all of them again.
#include "../../include/taxor_gpu_tools.h"
#include "ixf_arith.h"
#include "kernels.h"
#include "keyset.h"
#include "tuning.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace taxor;

extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);

namespace {

constexpr int BB = 256;                    // threads per block
constexpr uint32_t DEAD = 0xFFFFFFFFu;     // log entry: this list entry peeled nothing
constexpr uint32_t MAX_ROUNDS = 16000;     // (the pushed-round marks are 16 bits; 0xFFFF = never listed)
constexpr uint32_t KEYS_PER_BLOCK = 2048;  // k_count, k_verify
constexpr uint32_t SLOTS_PER_BLOCK = 4096; // k_seed
constexpr int ROUND_ITER = 4;              // list entries per thread and staging cycle in k_round
constexpr uint32_t ROUND_BATCH = 32;       // rounds enqueued between two looks at the recorded list ends
constexpr uint32_t TAIL_ENTRIES = 2048;    // rounds up to this size are assigned by the one-block walker
constexpr int TAIL_THREADS = 1024;
constexpr uint32_t SLICE_BLOCKS = 8192;    // blocks per launch of the key / slot sweeps (16 M keys, 32 M slots: ~2 ms each)

// one technical bin to construct
struct BinJob {
    const uint64_t *keys;   // device: the bin's keys (distinct among those that count); nullptr: GENERATED keys, key k = synth_key(gen_first + k, gen_salt)
    const uint8_t *keep;    // one byte per key, 0 = this key does not count (a duplicate inside a merged bin's key range); nullptr: all count
    uint64_t n_keys;        // length of the key range (a key's index is its position in it)
    uint64_t n_kept;        // keys that count
    uint64_t key_base;      // keys of the chunk's jobs before this one
    uint64_t slot_base;     // slots (3 * seg_len each) of the chunk's jobs before this one
    uint8_t *data;          // fingerprint array of the job's IXF
    uint64_t stride;
    uint64_t seed;
    uint32_t seg_len;
    uint32_t arith;         // arithmetic code of the index (ixf_arith.h)
    uint32_t bin;           // column
    uint32_t group;         // the chunk's IXF this bin belongs to: seed and failure are per IXF
    uint64_t gen_first;     // generated keys (keys == nullptr): index of the bin's first key in the synthetic key sequence ...
    uint64_t gen_salt;      // ... and its salt (ixf_arith.h synth_key: a bijection of the index, so the keys are distinct and need no memory)
    uint32_t lds_count;     // 1: the bin's degree words are built in LDS, range by range (k_count_lds), not by global atomics (k_count)
    uint32_t reserved;
};

// one piece of k_count_lds' work: rows [row0, row0 + LDS_ROWS) of segment `seg` of job `job`
struct CountItem {
    uint32_t job, seg, row0;
};
constexpr uint64_t WIDE_KEYS = 1ull << 26;   // a bin of this many keys needs 64-bit state words
constexpr uint32_t LDS_ROWS = 32768;       // 128 KB of 32-bit words: one block per CU
constexpr int LDS_THREADS = 1024;

// control block of a chunk; hot words on lines of their own
struct Ctl {
    uint32_t list_n;
    uint32_t pad0[31];
    unsigned long long peeled;
    unsigned long long pad1[15];
    unsigned long long mismatches;
    unsigned long long pad2[15];
    uint32_t done[MAX_ROUNDS + 2];       // blocks that finished launch i (0 = seed scan, t + 1 = round t)
    uint32_t round_end[MAX_ROUNDS + 2];  // round t = list entries [round_end[t], round_end[t + 1])
};

template <typename WT>
struct Peel {
    const BinJob *jobs;
    uint32_t n_jobs;
    uint64_t n_keys, n_slots;
    WT *w;               // [n_slots] degree | index sum
    uint64_t *list;      // [n_slots] job << 32 | row, append-only
    uint32_t *log;       // [n_slots] index of the key peeled by list entry i, or DEAD
    uint16_t *pushed;    // [n_slots] the round a slot was listed for, 0xFFFF = never
    const uint8_t *skip; // per group: 1 = not peeled under this seed, leave its columns alone (nullptr: none)
    Ctl *ctl;
    uint32_t dbits;      // width of a state word's degree field: 8, or 6 when a bin of the chunk has 2^24 .. 2^26 - 1 keys (32-bit words)
};

// last job whose first key (slot) is <= g, searched inside [lo, hi]
template <bool BY_KEY>
__device__ __forceinline__ uint32_t job_in(const BinJob *jobs, uint32_t lo, uint32_t hi, uint64_t g)
{
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if ((BY_KEY ? jobs[mid].key_base : jobs[mid].slot_base) <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The key and fingerprint pointers come out of a BinJob in memory, so the compiler cannot infer their address space and would emit
// flat loads / stores: cast to global.
__device__ __forceinline__ uint64_t ldg64(const uint64_t *p) { return *(const uint64_t __attribute__((address_space(1))) *)(uintptr_t)p; }
__device__ __forceinline__ uint8_t ldg8(const uint8_t *p) { return *(const uint8_t __attribute__((address_space(1))) *)(uintptr_t)p; }
__device__ __forceinline__ void stg8(uint8_t *p, uint8_t v) { *(uint8_t __attribute__((address_space(1))) *)(uintptr_t)p = v; }

__device__ __forceinline__ uint64_t job_key(const BinJob &J, uint64_t k) { return J.keys ? ldg64(J.keys + k) : synth_key(J.gen_first + k, J.gen_salt); }

// A slot of degree >= 2^dbits would carry into the sum: with distinct keys that is a 1e-60 event at 6 bits (mean degree 2.4), and a
// wrong peel it led to would be caught by k_verify (the IXF is redone under another seed).
template <typename WT>
__device__ __forceinline__ WT w_delta(uint64_t idx, uint32_t dbits) { return (WT)((WT)idx << dbits) + (WT)1; }
template <typename WT>
__device__ __forceinline__ WT w_degree(WT w, uint32_t dbits) { return w & (WT)(((WT)1 << dbits) - (WT)1); }

// (k_count, k_seed, k_verify are launched in slices of SLICE_BLOCKS blocks: block0 = first block of the slice)
template <typename WT>
__global__ __launch_bounds__(BB) void k_count(const Peel<WT> a, uint32_t block0)
{
    __shared__ uint32_t jr[2];
    const uint64_t g0 = (uint64_t)(block0 + blockIdx.x) * KEYS_PER_BLOCK;
    const uint64_t g1 = min(g0 + (uint64_t)KEYS_PER_BLOCK, a.n_keys);
    if (threadIdx.x == 0) {
        jr[0] = job_in<true>(a.jobs, 0, a.n_jobs - 1, g0);
        jr[1] = job_in<true>(a.jobs, jr[0], a.n_jobs - 1, g1 - 1);
    }
    __syncthreads();
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += BB) {
        const BinJob &J = a.jobs[job_in<true>(a.jobs, jr[0], jr[1], g)];
        if (J.lds_count) continue;                         // built by k_count_lds
        const uint64_t k = g - J.key_base;
        if (J.keep && !ldg8(J.keep + k)) continue;
        const ixf_probe p = ixf_probe_key_arith(job_key(J, k), J.seed, J.seg_len, J.arith);
        const WT d = w_delta<WT>(k, a.dbits);
#pragma unroll
        for (int j = 0; j < 3; ++j) __hip_atomic_fetch_add(&a.w[J.slot_base + p.row[j]], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The degree words of a bin of moderate size without a single global atomic: the rows of one segment are cut into ranges of 32 768
// (128 KB of LDS), a block takes one range, runs over ALL keys of the bin, hashes each and adds those that fall into its range with
// an LDS atomic, then writes the range out in one piece.  A key is read and hashed 3 x ceil(seg_len / 32768) times -- 15 times for a
// GTDB-class leaf bin of 422 k keys -- which is streaming and arithmetic, and cheaper than three random read-modify-writes in HBM
// (k_count: 27 G/s for the whole chip = 9 G keys/s; this: 19 G keys/s, bound by the hashing).  32-bit words only: a 54 M-key bin
// would need 2 000 passes.
__global__ __launch_bounds__(LDS_THREADS) void k_count_lds(const Peel<uint32_t> a, const CountItem *__restrict__ items)
{
    extern __shared__ uint32_t lds_w[];
    const CountItem it = items[blockIdx.x];
    const BinJob &J = a.jobs[it.job];
    const uint32_t n_rows = min(LDS_ROWS, J.seg_len - it.row0);
    for (uint32_t i = threadIdx.x; i < n_rows; i += LDS_THREADS) lds_w[i] = 0;
    __syncthreads();
    const uint32_t lo = it.seg * J.seg_len + it.row0;          // absolute row of the range's first row
    // (what bounds this loop is its arithmetic -- the key's hash, fifteen times per key of a GTDB-class leaf bin; more loads in flight
    //  per thread changed nothing, profiles/r06/build_lds_count.txt -- so only the ONE row this pass looks at is derived from the hash)
    for (uint64_t k = threadIdx.x; k < J.n_keys; k += LDS_THREADS) {
        if (J.keep && !ldg8(J.keep + k)) continue;
        const uint64_t h = ixf_key_hash_arith(job_key(J, k), J.seed, J.arith);
        const uint32_t r = (uint32_t)ixf_row_arith(h, (int)it.seg, J.seg_len, J.arith) - lo;     // (unsigned: rows below the range wrap to huge values)
        if (r < n_rows) atomicAdd(&lds_w[r], w_delta<uint32_t>(k, a.dbits));
    }
    __syncthreads();
    uint32_t *dst = a.w + J.slot_base + lo;
    for (uint32_t i = threadIdx.x; i < n_rows; i += LDS_THREADS) dst[i] = lds_w[i];
}

// entries of one wave into the block's LDS stage: one LDS atomic per wave
__device__ __forceinline__ void stage_push(bool pred, uint64_t e, uint64_t *stage, uint32_t *stage_n)
{
    const uint64_t m = __ballot(pred);
    if (m == 0) return;
    const int lane = (int)__lane_id(), leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(stage_n, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    if (pred) stage[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = e;
}

// the block's staged entries -> the list (one returning atomic), then the stage is empty again.  All threads.
__device__ __forceinline__ void stage_flush(uint64_t *stage, uint32_t *stage_n, uint32_t *gbase, uint64_t *list, uint32_t *list_n)
{
    __syncthreads();
    const uint32_t n = *stage_n;
    if (threadIdx.x == 0 && n) *gbase = atomicAdd(list_n, n);
    __syncthreads();
    if (n) {
        const uint32_t b = *gbase;
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) list[b + k] = stage[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) *stage_n = 0;
    __syncthreads();
}

// the last block of launch `which` records where the list ends now: round_end[which + 1]
__device__ __forceinline__ void finish_launch(Ctl *ctl, uint32_t which, uint32_t blocks_total)
{
    // (this thread's own list appends are returning atomics whose values it has used: they are complete before the ticket is
    //  drawn, and every block's are before ITS ticket -- the block that draws the last ticket sees every append)
    const uint32_t ticket = atomicAdd(&ctl->done[which], 1u);
    if (ticket == blocks_total - 1) ctl->round_end[which + 1] = atomicAdd(&ctl->list_n, 0u);
}

template <typename WT>
__global__ __launch_bounds__(BB) void k_seed(const Peel<WT> a, uint32_t block0, uint32_t blocks_total)
{
    __shared__ uint64_t stage[SLOTS_PER_BLOCK];
    __shared__ uint32_t stage_n, gbase, jr[2];
    const uint64_t s0 = (uint64_t)(block0 + blockIdx.x) * SLOTS_PER_BLOCK;
    const uint64_t s1 = min(s0 + (uint64_t)SLOTS_PER_BLOCK, a.n_slots);
    if (threadIdx.x == 0) {
        stage_n = 0;
        jr[0] = job_in<false>(a.jobs, 0, a.n_jobs - 1, s0);
        jr[1] = job_in<false>(a.jobs, jr[0], a.n_jobs - 1, s1 - 1);
    }
    __syncthreads();
    for (uint64_t s = s0 + threadIdx.x; s < s0 + SLOTS_PER_BLOCK; s += BB) {      // (uniform trip count: stage_push is a wave operation)
        bool single = false;
        uint64_t e = 0;
        if (s < s1 && w_degree<WT>(a.w[s], a.dbits) == (WT)1) {
            const uint32_t j = job_in<false>(a.jobs, jr[0], jr[1], s);
            e = ((uint64_t)j << 32) | (uint64_t)(s - a.jobs[j].slot_base);
            a.pushed[s] = 0;
            single = true;
        }
        stage_push(single, e, stage, &stage_n);
    }
    stage_flush(stage, &stage_n, &gbase, a.list, &a.ctl->list_n);
    if (threadIdx.x == 0) finish_launch(a.ctl, 0, blocks_total);
}

template <typename WT>
__global__ __launch_bounds__(BB) void k_round(const Peel<WT> a, uint32_t t)
{
    __shared__ uint64_t stage[2 * ROUND_ITER * BB];
    __shared__ uint32_t stage_n, gbase, peeled_blk;
    const uint32_t lo = a.ctl->round_end[t], hi = a.ctl->round_end[t + 1];
    const uint32_t step = (uint32_t)ROUND_ITER * BB;
    // Only blocks that have entries take part in the round's ticket (one returning atomic on ONE word each, ~13 ns: 1024 tickets
    // would be the whole cost of a plateau round of a few ten thousand entries).  A launch behind the last round finds an empty
    // range: the list cannot have grown, so its end is recorded as it is, without a ticket.
    const uint32_t n_work = (uint32_t)min((uint64_t)gridDim.x, ((uint64_t)(hi - lo) + step - 1) / step);
    if (blockIdx.x >= n_work) {
        if (n_work == 0 && blockIdx.x == 0 && threadIdx.x == 0) a.ctl->round_end[t + 2] = hi;
        return;
    }
    if (threadIdx.x == 0) { stage_n = 0; peeled_blk = 0; }
    __syncthreads();
    uint32_t peeled = 0;
    for (uint64_t base = (uint64_t)lo + (uint64_t)blockIdx.x * step; base < hi; base += (uint64_t)gridDim.x * step) {
#pragma unroll 1
        for (int it = 0; it < ROUND_ITER; ++it) {
            const uint64_t i = base + (uint64_t)it * BB + threadIdx.x;
            bool push[3] = {false, false, false};
            uint64_t ent[3] = {0, 0, 0};
            if (i < hi) {
                const uint64_t e = a.list[i];
                const uint32_t job = (uint32_t)(e >> 32), row = (uint32_t)e;
                const BinJob &J = a.jobs[job];
                const WT w = __hip_atomic_load(&a.w[J.slot_base + row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t logged = DEAD;
                const uint64_t k = (uint64_t)(w >> a.dbits);
                if (w_degree<WT>(w, a.dbits) == (WT)1 && k < J.n_keys) {
                    const ixf_probe p = ixf_probe_key_arith(job_key(J, k), J.seed, J.seg_len, J.arith);
                    // the key's lowest-segment row that is listed for this round peels it; this row, unless a lower one carries the mark
                    const int own = row == p.row[0] ? 0 : row == p.row[1] ? 1 : 2;
                    bool mine = true;
                    if (own >= 1 && a.pushed[J.slot_base + p.row[0]] == (uint16_t)t) mine = false;
                    if (own == 2 && mine && a.pushed[J.slot_base + p.row[1]] == (uint16_t)t) mine = false;
                    if (mine) {
                        logged = (uint32_t)k;
                        ++peeled;
                        const WT d = w_delta<WT>(k, a.dbits);
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            if (j == own) continue;                 // (this slot is listed once, here, and holds no other key: never read again)
                            const uint64_t s = J.slot_base + p.row[j];
                            const WT old = __hip_atomic_fetch_sub(&a.w[s], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (w_degree<WT>(old, a.dbits) == (WT)2) {
                                a.pushed[s] = (uint16_t)(t + 1);
                                push[j] = true;
                                ent[j] = ((uint64_t)job << 32) | p.row[j];
                            }
                        }
                    }
                }
                a.log[i] = logged;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) stage_push(push[j], ent[j], stage, &stage_n);
        }
        stage_flush(stage, &stage_n, &gbase, a.list, &a.ctl->list_n);
    }
    if (peeled) atomicAdd(&peeled_blk, peeled);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (peeled_blk) atomicAdd(&a.ctl->peeled, (unsigned long long)peeled_blk);
        finish_launch(a.ctl, t + 1, n_work);
    }
}

// list entry i: D[free row] = fp ^ D[other two rows]; the free row is the row that peeled the key, the entry's own
template <typename WT>
__device__ __forceinline__ void assign_entry(const Peel<WT> &a, uint64_t i)
{
    const uint32_t k = a.log[i];
    if (k == DEAD) return;
    const uint64_t e = a.list[i];
    const BinJob &J = a.jobs[(uint32_t)(e >> 32)];
    if (a.skip && a.skip[J.group]) return;
    const ixf_probe p = ixf_probe_key_arith(job_key(J, k), J.seed, J.seg_len, J.arith);
    const uint32_t row = (uint32_t)e;
    const int fr = row == p.row[0] ? 0 : row == p.row[1] ? 1 : 2;
    uint8_t v = (uint8_t)(p.fp4 & 0xFFu);
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (j != fr) v ^= ldg8(J.data + ((uint64_t)p.row[j] * J.stride + J.bin));
    stg8(J.data + ((uint64_t)p.row[fr] * J.stride + J.bin), v);
}

template <typename WT>
__global__ __launch_bounds__(BB) void k_assign(const Peel<WT> a, uint32_t t)
{
    const uint32_t lo = a.ctl->round_end[t], hi = a.ctl->round_end[t + 1];
    for (uint64_t i = (uint64_t)lo + (uint64_t)blockIdx.x * BB + threadIdx.x; i < hi; i += (uint64_t)gridDim.x * BB) assign_entry(a, i);
}

// rounds t_hi, t_hi - 1, ..., t_lo by ONE block (all of them small): a barrier between two rounds instead of a launch.  The block's
// waves share one L1, and __syncthreads() orders its stores before the next round's loads.
template <typename WT>
__global__ __launch_bounds__(TAIL_THREADS) void k_assign_tail(const Peel<WT> a, uint32_t t_hi, uint32_t t_lo)
{
    for (uint32_t t = t_hi + 1; t-- > t_lo;) {
        const uint32_t lo = a.ctl->round_end[t], hi = a.ctl->round_end[t + 1];
        for (uint64_t i = (uint64_t)lo + threadIdx.x; i < hi; i += TAIL_THREADS) assign_entry(a, i);
        __syncthreads();
    }
}

// every key against the finished columns
template <typename WT>
__global__ __launch_bounds__(BB) void k_verify(const Peel<WT> a, uint32_t block0)
{
    __shared__ uint32_t jr[2], bad_blk;
    const uint64_t g0 = (uint64_t)(block0 + blockIdx.x) * KEYS_PER_BLOCK;
    const uint64_t g1 = min(g0 + (uint64_t)KEYS_PER_BLOCK, a.n_keys);
    if (threadIdx.x == 0) {
        bad_blk = 0;
        jr[0] = job_in<true>(a.jobs, 0, a.n_jobs - 1, g0);
        jr[1] = job_in<true>(a.jobs, jr[0], a.n_jobs - 1, g1 - 1);
    }
    __syncthreads();
    uint32_t bad = 0;
    for (uint64_t g = g0 + threadIdx.x; g < g1; g += BB) {
        const BinJob &J = a.jobs[job_in<true>(a.jobs, jr[0], jr[1], g)];
        if (a.skip && a.skip[J.group]) continue;
        if (J.keep && !ldg8(J.keep + (g - J.key_base))) continue;
        const ixf_probe p = ixf_probe_key_arith(job_key(J, g - J.key_base), J.seed, J.seg_len, J.arith);
        uint8_t v = (uint8_t)(p.fp4 & 0xFFu);
#pragma unroll
        for (int j = 0; j < 3; ++j) v ^= ldg8(J.data + ((uint64_t)p.row[j] * J.stride + J.bin));
        bad += v != 0;
    }
    if (bad) atomicAdd(&bad_blk, bad);
    __syncthreads();
    if (threadIdx.x == 0 && bad_blk) atomicAdd(&a.ctl->mismatches, (unsigned long long)bad_blk);
}

// peeled keys per job, from the log (only looked at when a chunk fell short)
__global__ __launch_bounds__(BB) void k_job_peeled(const uint64_t *list, const uint32_t *log, uint32_t n_list, unsigned long long *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n_list; i += (uint64_t)gridDim.x * BB)
        if (log[i] != DEAD) atomicAdd(&out[(uint32_t)(list[i] >> 32)], 1ull);
}

// columns of the jobs [j0, j0 + nj) of one IXF -> 0 (bins that are not built keep their content)
__global__ __launch_bounds__(BB) void k_zero_columns(const BinJob *jobs, uint32_t j0, uint32_t nj, uint64_t rows)
{
    const uint64_t total = rows * nj;
    for (uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < total; i += (uint64_t)gridDim.x * BB) {
        const uint64_t r = i / nj;
        const BinJob &J = jobs[j0 + (uint32_t)(i - r * nj)];
        stg8(J.data + (r * J.stride + J.bin), 0);
    }
}

// keys[i] = a bijection of (first + i): distinct 64-bit keys without a table (synthetic key sets of the build bench and tests)
__global__ __launch_bounds__(BB) void k_synth_keys(uint64_t *out, uint64_t first, uint64_t n, uint64_t salt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BB) out[i] = synth_key(first + i, salt);
}

int bfail(int code, const std::string &m)
{
    taxor_set_last_error(m.c_str());
    return code;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

// library-internal accessors implemented in api.hip
extern "C" __attribute__((visibility("hidden"))) int taxor_index_ixf_info(taxor_gpu_index *idx, uint64_t ixf, uint8_t **data,
                                                                          uint64_t *stride, uint64_t *seg_len, uint64_t *bins,
                                                                          int *device);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_set_seed(taxor_gpu_index *idx, uint64_t ixf, uint64_t seed);
extern "C" __attribute__((visibility("hidden"))) uint32_t taxor_index_arith(const taxor_gpu_index *idx);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_tree(taxor_gpu_index *idx, uint64_t *n_ixf, const uint32_t **bin_base,
                                                                      const uint32_t **binfo);
extern "C" __attribute__((visibility("hidden"))) void *taxor_index_build_ctx(taxor_gpu_index *idx);
extern "C" __attribute__((visibility("hidden"))) void taxor_index_set_build_ctx(taxor_gpu_index *idx, void *ctx, void (*free_fn)(void *));

namespace {

#define E_TRY(expr)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return bfail(TAXOR_E_HIP, std::string("build: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

// The chunk engine: scratch for one chunk of bins, reused from chunk to chunk, on a stream of its own.
struct Engine {
    int device = 0;
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};      // timing: before k_count, after it, after the last round
    void *d_w = nullptr;
    uint64_t *d_list = nullptr, *d_job_peeled = nullptr;
    uint32_t *d_log = nullptr;
    uint16_t *d_pushed = nullptr;
    uint8_t *d_skip = nullptr;
    CountItem *d_items = nullptr;
    uint64_t cap_items = 0;
    bool lds_ready = false;
    Ctl *d_ctl = nullptr;
    BinJob *d_jobs = nullptr;
    uint32_t *h_round_end = nullptr;                   // page-locked
    unsigned long long *h_counts = nullptr;            // page-locked: peeled, mismatches
    uint64_t cap_slots = 0, cap_jobs = 0, w_bytes = 0, budget_bytes = 0, free_half = 0;
    taxor_build_stats stats{};

    ~Engine() { release(); }

    void release()
    {
        for (void *p : {(void *)d_w, (void *)d_list, (void *)d_job_peeled, (void *)d_log, (void *)d_pushed, (void *)d_skip, (void *)d_ctl, (void *)d_jobs, (void *)d_items})
            if (p) (void)hipFree(p);
        d_items = nullptr;
        cap_items = 0;
        d_w = nullptr; d_list = nullptr; d_job_peeled = nullptr; d_log = nullptr; d_pushed = nullptr; d_skip = nullptr; d_ctl = nullptr; d_jobs = nullptr;
        if (h_round_end) (void)hipHostFree(h_round_end);
        if (h_counts) (void)hipHostFree(h_counts);
        h_round_end = nullptr; h_counts = nullptr;
        for (auto &e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        for (auto &e : ev_t) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (st) (void)hipStreamDestroy(st);
        st = nullptr;
        cap_slots = cap_jobs = w_bytes = 0;
    }

    int open(int dev)
    {
        const double ta = now_s();
        struct Acc { double &s; double t0; ~Acc() { s += now_s() - t0; } } acc{stats.seconds_alloc, ta};
        if (st) { E_TRY(hipSetDevice(device)); return TAXOR_OK; }      // kept from an earlier build of the same index
        device = dev;
        E_TRY(hipSetDevice(device));
        E_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        for (auto &e : ev) E_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : ev_t) E_TRY(hipEventCreate(&e));
        E_TRY(hipHostMalloc((void **)&h_round_end, (MAX_ROUNDS + 2) * sizeof(uint32_t), hipHostMallocDefault));
        E_TRY(hipHostMalloc((void **)&h_counts, 2 * sizeof(unsigned long long), hipHostMallocDefault));
        E_TRY(hipMalloc((void **)&d_ctl, sizeof(Ctl)));
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) fr = (size_t)8 << 30;
        free_half = (uint64_t)fr / 2;
        budget_bytes = std::min<uint64_t>(free_half, 3ull << 30);
        if (const char *e = tune_env("TAXOR_BUILD_SCRATCH_MB")) budget_bytes = std::min<uint64_t>(free_half, strtoull(e, nullptr, 10) << 20);   // (tests, sweeps)
        return TAXOR_OK;
    }

    // scratch bytes a chunk of `slots` needs: 18 per slot (26 with 64-bit words): state word, list entry, log entry, round mark
    static uint64_t bytes_for(uint64_t slots, bool wide) { return slots * (wide ? 26u : 18u) + 64; }

    // Slots a chunk may have.  Chunk size hardly matters for the rate (2.4 GB of scratch: 1.80 G insertions/s, 23 GB: 1.95,
    // profiles/r06/build_chunk_sweep.txt) but decides how long a launch runs and how much memory a build takes next to a resident
    // index: 3 GB by default (the largest launch, a chunk's first round, stays under 2 ms), more only when ONE bin needs more,
    // never beyond half of what was free when the engine was opened; fewer than 2^32 slots (list positions are 32 bits).
    uint64_t slot_budget(bool wide) const { return std::min<uint64_t>(budget_bytes / (wide ? 26u : 18u), 0xFFFFFFF0ull); }
    void fit_bin(uint64_t slots_of_one_bin, bool wide)
    {
        budget_bytes = std::min<uint64_t>(free_half, std::max<uint64_t>(budget_bytes, bytes_for(slots_of_one_bin, wide)));
    }

    int ensure(uint64_t slots, uint64_t jobs, bool wide)
    {
        const double ta = now_s();
        struct Acc { double &s; double t0; ~Acc() { s += now_s() - t0; } } acc{stats.seconds_alloc, ta};
        const uint64_t wb = slots * (wide ? 8 : 4);
        if (wb > w_bytes) {
            if (d_w) (void)hipFree(d_w);
            d_w = nullptr;
            w_bytes = 0;
            if (hipMalloc(&d_w, wb) != hipSuccess) return bfail(TAXOR_E_NOMEM, "build: no device memory for the peeling state");
            w_bytes = wb;
        }
        if (slots > cap_slots) {
            for (void *p : {(void *)d_list, (void *)d_log, (void *)d_pushed})
                if (p) (void)hipFree(p);
            d_list = nullptr; d_log = nullptr; d_pushed = nullptr;
            cap_slots = 0;
            if (hipMalloc((void **)&d_list, slots * 8) != hipSuccess || hipMalloc((void **)&d_log, slots * 4) != hipSuccess ||
                hipMalloc((void **)&d_pushed, slots * 2) != hipSuccess)
                return bfail(TAXOR_E_NOMEM, "build: no device memory for the peeling work list");
            cap_slots = slots;
        }
        if (jobs > cap_jobs) {
            for (void *p : {(void *)d_jobs, (void *)d_job_peeled, (void *)d_skip})
                if (p) (void)hipFree(p);
            d_jobs = nullptr; d_job_peeled = nullptr; d_skip = nullptr;
            cap_jobs = 0;
            if (hipMalloc((void **)&d_jobs, jobs * sizeof(BinJob)) != hipSuccess || hipMalloc((void **)&d_job_peeled, jobs * 8) != hipSuccess ||
                hipMalloc((void **)&d_skip, jobs) != hipSuccess)
                return bfail(TAXOR_E_NOMEM, "build: no device memory for the bin table");
            cap_jobs = jobs;
        }
        stats.scratch_bytes = std::max<uint64_t>(stats.scratch_bytes, w_bytes + cap_slots * 14 + cap_jobs * (sizeof(BinJob) + 9) + sizeof(Ctl));
        return TAXOR_OK;
    }

    template <typename WT>
    int run_typed(std::vector<BinJob> &jobs, uint32_t n_groups, std::vector<uint8_t> &group_ok, uint64_t n_keys, uint64_t n_slots,
                  const std::vector<uint8_t> &group_full, const std::vector<uint64_t> &group_rows, uint32_t dbits);

    static bool lds_count_enabled()
    {
        const char *e = tune_env("TAXOR_BUILD_LDS_COUNT");        // A/B: 0 = every bin through k_count's global atomics
        return !(e && atoi(e) == 0);
    }
    // (sliced like the other sweeps: 2048 items, ~1 ms, per launch)
    void launch_count_lds(const Peel<uint32_t> &a, size_t n_items)
    {
        for (size_t i0 = 0; i0 < n_items; i0 += 2048)
            hipLaunchKernelGGL(k_count_lds, dim3((uint32_t)std::min<size_t>(2048, n_items - i0)), dim3(LDS_THREADS), LDS_ROWS * sizeof(uint32_t), st, a, d_items + i0);
    }
    void launch_count_lds(const Peel<uint64_t> &, size_t) {}      // (64-bit words: never)

    // construct the columns of `jobs` (key_base / slot_base are filled in here; every job has keys; jobs of one group are
    // adjacent).  group_ok[g] = 0: a bin of group g did not peel under its seed, its columns are untouched.
    // group_full[g] = 1: the jobs of group g are all the bins of its IXF (the whole array is cleared at once); 2: the caller has
    // cleared the array already (an IXF built in several chunks all of whose bins have keys); 0: column by column.
    int run(std::vector<BinJob> &jobs, uint32_t n_groups, std::vector<uint8_t> &group_ok, const std::vector<uint8_t> &group_full)
    {
        group_ok.assign(n_groups, 1);
        if (jobs.empty()) return TAXOR_OK;
        uint64_t nk = 0, ns = 0;
        bool wide = false;
        uint64_t max_keys = 0;
        std::vector<uint64_t> group_rows(n_groups, 0);
        for (auto &j : jobs) {
            j.key_base = nk;
            j.slot_base = ns;
            nk += j.n_keys;
            ns += 3ull * j.seg_len;
            wide |= j.n_keys >= WIDE_KEYS;
            max_keys = std::max<uint64_t>(max_keys, j.n_keys);
            group_rows[j.group] = 3ull * j.seg_len;
            if (j.n_kept > 3ull * j.seg_len) return bfail(TAXOR_E_ARG, "build: a bin holds more keys than its IXF has rows");
            if (j.n_keys >= 0xFFFFFFFFull) return bfail(TAXOR_E_ARG, "build: more than 2^32 - 2 keys in one bin");
        }
        if (ns >= 0xFFFFFFF8ull || jobs.size() >= (1ull << 31)) return bfail(TAXOR_E_INTERNAL, "build: chunk too large");
        E_TRY(hipSetDevice(device));
        const int rc = ensure(ns, jobs.size(), wide);
        if (rc != TAXOR_OK) return rc;
        ++stats.chunks;
        const uint32_t dbits = (!wide && max_keys >= (1ull << 24)) ? 6u : 8u;
        return wide ? run_typed<uint64_t>(jobs, n_groups, group_ok, nk, ns, group_full, group_rows, dbits)
                    : run_typed<uint32_t>(jobs, n_groups, group_ok, nk, ns, group_full, group_rows, dbits);
    }
};

template <typename WT>
int Engine::run_typed(std::vector<BinJob> &jobs, uint32_t n_groups, std::vector<uint8_t> &group_ok, uint64_t n_keys, uint64_t n_slots,
                      const std::vector<uint8_t> &group_full, const std::vector<uint64_t> &group_rows, uint32_t dbits)
{
    const double t0 = now_s();
    Peel<WT> a{};
    a.jobs = d_jobs;
    a.n_jobs = (uint32_t)jobs.size();
    a.n_keys = n_keys;
    a.n_slots = n_slots;
    a.w = (WT *)d_w;
    a.list = d_list;
    a.log = d_log;
    a.pushed = d_pushed;
    a.skip = nullptr;
    a.ctl = d_ctl;
    a.dbits = dbits;
    // which bins get their degree words built in LDS (k_count_lds): 32-bit words, enough keys to be worth a block per range, few enough
    // rows that a key is hashed a few dozen times at most
    std::vector<CountItem> items;
    if (sizeof(WT) == 4 && lds_count_enabled()) {
        for (size_t j = 0; j < jobs.size(); ++j) {
            BinJob &J = jobs[j];
            const uint32_t passes = (J.seg_len + LDS_ROWS - 1) / LDS_ROWS;
            J.lds_count = (J.n_keys >= 16384 && passes <= 12) ? 1u : 0u;
            if (J.lds_count)
                for (uint32_t sgm = 0; sgm < 3; ++sgm)
                    for (uint32_t r0 = 0; r0 < J.seg_len; r0 += LDS_ROWS) items.push_back(CountItem{(uint32_t)j, sgm, r0});
        }
    } else
        for (auto &J : jobs) J.lds_count = 0;
    if (!items.empty()) {
        if (!lds_ready) {
            E_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_count_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_ROWS * sizeof(uint32_t))));
            lds_ready = true;
        }
        if (items.size() > cap_items) {
            if (d_items) (void)hipFree(d_items);
            d_items = nullptr;
            cap_items = 0;
            if (hipMalloc((void **)&d_items, (items.size() + items.size() / 4) * sizeof(CountItem)) != hipSuccess) return bfail(TAXOR_E_NOMEM, "build: no device memory for the count items");
            cap_items = items.size() + items.size() / 4;
        }
        E_TRY(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(CountItem), hipMemcpyHostToDevice, st));
    }
    E_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(BinJob), hipMemcpyHostToDevice, st));
    E_TRY(hipMemsetAsync(d_w, 0, n_slots * sizeof(WT), st));
    E_TRY(hipMemsetAsync(d_pushed, 0xFF, n_slots * 2, st));
    E_TRY(hipMemsetAsync(d_ctl, 0, sizeof(Ctl), st));
    const uint32_t grid_keys = (uint32_t)((n_keys + KEYS_PER_BLOCK - 1) / KEYS_PER_BLOCK);
    E_TRY(hipEventRecord(ev_t[0], st));
    if (items.size() != 0) launch_count_lds(a, items.size());
    bool any_atomic = false;
    for (const auto &J : jobs) any_atomic |= !J.lds_count;
    if (any_atomic)
        for (uint32_t b = 0; b < grid_keys; b += SLICE_BLOCKS) hipLaunchKernelGGL((k_count<WT>), dim3(std::min(SLICE_BLOCKS, grid_keys - b)), dim3(BB), 0, st, a, b);
    E_TRY(hipEventRecord(ev_t[1], st));
    const uint32_t grid_slots = (uint32_t)((n_slots + SLOTS_PER_BLOCK - 1) / SLOTS_PER_BLOCK);
    for (uint32_t b = 0; b < grid_slots; b += SLICE_BLOCKS) hipLaunchKernelGGL((k_seed<WT>), dim3(std::min(SLICE_BLOCKS, grid_slots - b)), dim3(BB), 0, st, a, b, grid_slots);
    // rounds: two batches ahead of the host.  The grid follows the chunk's size (a small chunk's rounds are small)
    // (every block draws one ticket from ONE counter at the end of a round, ~13 ns each: 1024 blocks = 4 per CU is the knee)
    const uint32_t round_grid = (uint32_t)std::min<uint64_t>(1024, std::max<uint64_t>(8, n_keys / (2 * ROUND_ITER * BB)));
    uint32_t launched = 0, rounds = 0;      // rounds = index of the first empty round, once known
    bool done = false;
    int slot = 0;
    uint32_t pending_lo[2] = {0, 0}, pending_hi[2] = {0, 0};
    bool pending[2] = {false, false};
    auto check = [&](int s) -> int {        // wait for batch s's copy, look for the first empty round in it
        E_TRY(hipEventSynchronize(ev[s]));
        pending[s] = false;
        for (uint32_t t = pending_lo[s]; t < pending_hi[s] && !done; ++t)
            if (h_round_end[t + 1] == h_round_end[t]) { rounds = t; done = true; }
        return TAXOR_OK;
    };
    while (!done) {
        if (launched >= MAX_ROUNDS) return bfail(TAXOR_E_INTERNAL, "build: peeling did not end within 16000 rounds");
        const uint32_t lo = launched, hi = std::min(MAX_ROUNDS, launched + ROUND_BATCH);
        for (uint32_t t = lo; t < hi; ++t) hipLaunchKernelGGL((k_round<WT>), dim3(round_grid), dim3(BB), 0, st, a, t);
        launched = hi;
        E_TRY(hipEventRecord(ev_t[2], st));          // (re-recorded after every batch: the last one stands)
        // the ends of rounds lo .. hi (round t's own end is written by the launch before it, hi's by the last one of this batch)
        E_TRY(hipMemcpyAsync(h_round_end + lo, &d_ctl->round_end[lo], (hi - lo + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        E_TRY(hipEventRecord(ev[slot], st));
        pending_lo[slot] = lo;
        pending_hi[slot] = hi;
        pending[slot] = true;
        slot ^= 1;
        if (pending[slot]) { const int rc = check(slot); if (rc != TAXOR_OK) return rc; }
    }
    for (int s = 0; s < 2; ++s)
        if (pending[s]) E_TRY(hipEventSynchronize(ev[s]));                  // (launches behind the last round: empty, already queued)
    E_TRY(hipMemcpyAsync(h_round_end, &d_ctl->round_end[0], (rounds + 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    E_TRY(hipMemcpyAsync(&h_counts[0], &d_ctl->peeled, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    E_TRY(hipStreamSynchronize(st));
    E_TRY(hipGetLastError());
    stats.rounds_max = std::max(stats.rounds_max, rounds);
    {   // the two kernels that carry the read-modify-writes, by HIP events on the builder's stream
        float ms_count = 0.f, ms_rounds = 0.f;
        if (hipEventElapsedTime(&ms_count, ev_t[0], ev_t[1]) == hipSuccess) stats.seconds_count += ms_count * 1e-3;
        if (hipEventElapsedTime(&ms_rounds, ev_t[1], ev_t[2]) == hipSuccess) stats.seconds_rounds += ms_rounds * 1e-3;
    }
    const double t1 = now_s();
    stats.seconds_peel += t1 - t0;

    // which IXFs fell short?
    uint32_t n_failed = 0;
    uint64_t n_expected = 0;
    for (const auto &j : jobs) n_expected += j.n_kept;
    if (h_counts[0] != n_expected) {
        E_TRY(hipMemsetAsync(d_job_peeled, 0, jobs.size() * 8, st));
        hipLaunchKernelGGL(k_job_peeled, dim3(1024), dim3(BB), 0, st, d_list, d_log, h_round_end[rounds], (unsigned long long *)d_job_peeled);
        std::vector<uint64_t> got(jobs.size());
        E_TRY(hipMemcpyAsync(got.data(), d_job_peeled, jobs.size() * 8, hipMemcpyDeviceToHost, st));
        E_TRY(hipStreamSynchronize(st));
        for (size_t j = 0; j < jobs.size(); ++j)
            if (got[j] != jobs[j].n_kept) group_ok[jobs[j].group] = 0;
        for (uint32_t g = 0; g < n_groups; ++g) n_failed += !group_ok[g];
        if (!n_failed) return bfail(TAXOR_E_INTERNAL, "build: peeled keys fall short but every bin is complete");
        std::vector<uint8_t> skip(n_groups);
        for (uint32_t g = 0; g < n_groups; ++g) skip[g] = !group_ok[g];
        E_TRY(hipMemcpyAsync(d_skip, skip.data(), n_groups, hipMemcpyHostToDevice, st));
        E_TRY(hipStreamSynchronize(st));                                     // (skip is a stack vector)
        a.skip = d_skip;
        stats.reseeds += n_failed;
    }
    if (n_failed < n_groups) {
        // clear what is built: the whole array of an IXF all of whose bins are here, else column by column
        for (size_t j0 = 0; j0 < jobs.size();) {
            size_t j1 = j0;
            while (j1 < jobs.size() && jobs[j1].group == jobs[j0].group) ++j1;
            const uint32_t g = jobs[j0].group;
            if (group_ok[g]) {
                if (group_full[g] == 1) E_TRY(hipMemsetAsync(jobs[j0].data, 0, group_rows[g] * jobs[j0].stride, st));
                else if (group_full[g] == 0) {
                    const uint64_t total = group_rows[g] * (j1 - j0);
                    hipLaunchKernelGGL(k_zero_columns, dim3((uint32_t)std::min<uint64_t>(4096, (total + BB - 1) / BB)), dim3(BB), 0, st, d_jobs,
                                       (uint32_t)j0, (uint32_t)(j1 - j0), group_rows[g]);
                }
            }
            j0 = j1;
        }
        // rounds in reverse: runs of small rounds by the one-block walker, the others a launch each
        for (uint32_t t = rounds; t-- > 0;) {
            const uint32_t n = h_round_end[t + 1] - h_round_end[t];
            if (n <= TAIL_ENTRIES) {
                uint32_t t_lo = t;
                while (t_lo > 0 && h_round_end[t_lo] - h_round_end[t_lo - 1] <= TAIL_ENTRIES) --t_lo;
                hipLaunchKernelGGL((k_assign_tail<WT>), dim3(1), dim3(TAIL_THREADS), 0, st, a, t, t_lo);
                t = t_lo;
            } else
                hipLaunchKernelGGL((k_assign<WT>), dim3((uint32_t)std::min<uint32_t>(2048, (n + BB - 1) / BB)), dim3(BB), 0, st, a, t);
        }
        for (uint32_t b = 0; b < grid_keys; b += SLICE_BLOCKS) hipLaunchKernelGGL((k_verify<WT>), dim3(std::min(SLICE_BLOCKS, grid_keys - b)), dim3(BB), 0, st, a, b);
        E_TRY(hipMemcpyAsync(&h_counts[1], &d_ctl->mismatches, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        E_TRY(hipStreamSynchronize(st));
        E_TRY(hipGetLastError());
        if (h_counts[1] != 0)
            return bfail(TAXOR_E_INTERNAL, "build: " + std::to_string(h_counts[1]) + " keys do not match their own columns after construction");
    }
    for (const auto &j : jobs)
        if (group_ok[j.group]) {
            stats.keys_inserted += j.n_kept;
            if (j.lds_count) stats.keys_counted_in_lds += j.n_kept;
        }
    stats.seconds_assign += now_s() - t1;
    return TAXOR_OK;
}

// one IXF to construct: where its bins' keys are
struct IxfPlan {
    uint64_t ixf = 0;
    uint8_t *data = nullptr;
    uint64_t stride = 0, seg_len = 0, bins = 0;
    uint64_t seed = 0;
    std::vector<const uint64_t *> keys;      // per bin (device)
    std::vector<const uint8_t *> keep;       // per bin: marks of the keys that count (nullptr: all of them)
    std::vector<uint64_t> n, kept;           // per bin: length of the key range, keys that count
    std::vector<uint64_t> gen_first;         // per bin: NO_GEN, or the first index of a GENERATED key range (then keys[b] is nullptr)
    uint64_t gen_salt = 0;
    uint64_t total = 0, max_bin = 0, n_with_keys = 0;
    int attempts = 0;
};

constexpr uint64_t NO_GEN = ~0ull;

uint64_t next_seed(uint64_t s) { return s * 6364136223846793005ull + 1442695040888963407ull; }

// construct the IXFs of `plans` (one level of a hierarchy, or a single IXF): as many as fit go into one chunk; an IXF that
// does not peel is redone with a redrawn seed; an IXF larger than a chunk is built from several chunks of its bins
int build_plans(Engine &eng, taxor_gpu_index *idx, std::vector<IxfPlan> &plans)
{
    const uint32_t arith = taxor_index_arith(idx);
    std::vector<size_t> todo;
    for (size_t i = 0; i < plans.size(); ++i)
        if (plans[i].total) todo.push_back(i);
    auto add_jobs = [&](const IxfPlan &p, uint32_t group, uint64_t b0, uint64_t b1, std::vector<BinJob> &jobs) {
        for (uint64_t b = b0; b < b1; ++b) {
            if (!p.n[b]) continue;
            BinJob j{};
            j.keys = p.keys[b];
            j.gen_first = p.gen_first[b] == NO_GEN ? 0 : p.gen_first[b];
            j.gen_salt = p.gen_salt;
            j.keep = p.keep[b];
            j.n_keys = p.n[b];
            j.n_kept = p.kept[b];
            j.data = p.data;
            j.stride = p.stride;
            j.seed = p.seed;
            j.seg_len = (uint32_t)p.seg_len;
            j.arith = arith;
            j.bin = (uint32_t)b;
            j.group = group;
            jobs.push_back(j);
        }
    };
    for (size_t q : todo) eng.fit_bin(3 * plans[q].seg_len, plans[q].max_bin >= WIDE_KEYS);
    while (!todo.empty()) {
        std::vector<BinJob> jobs;
        std::vector<size_t> members;
        std::vector<uint8_t> full, ok;
        uint64_t slots = 0, keys = 0;
        bool wide = false;
        for (size_t q : todo) wide |= plans[q].max_bin >= WIDE_KEYS;
        const uint64_t budget = eng.slot_budget(wide);
        size_t taken = 0;
        for (; taken < todo.size(); ++taken) {
            const IxfPlan &p = plans[todo[taken]];
            const uint64_t s = p.n_with_keys * 3 * p.seg_len;
            if (!members.empty() && (slots + s > budget || keys + p.total >= (1ull << 32))) break;
            if (members.empty() && (s > budget || p.total >= (1ull << 32))) break;          // larger than a chunk: below
            add_jobs(p, (uint32_t)members.size(), 0, p.bins, jobs);
            full.push_back(p.n_with_keys == p.bins);
            members.push_back(todo[taken]);
            slots += s;
            keys += p.total;
        }
        if (members.empty()) {
            // one IXF in several chunks of its bins; a bin that does not peel restarts the IXF under a new seed
            IxfPlan &p = plans[todo[0]];
            const uint64_t per_bin = 3 * p.seg_len;
            if (per_bin > budget) return bfail(TAXOR_E_NOMEM, "build: one bin of IXF " + std::to_string(p.ixf) + " does not fit the peeling scratch");
            bool built = false;
            const bool all_bins = p.n_with_keys == p.bins;
            while (!built) {
                built = true;
                const uint64_t inserted_before = eng.stats.keys_inserted, lds_before = eng.stats.keys_counted_in_lds;      // (an attempt that is thrown away does not count)
                if (all_bins && hipMemsetAsync(p.data, 0, 3 * p.seg_len * p.stride, eng.st) != hipSuccess) return bfail(TAXOR_E_HIP, "build: clearing an IXF failed");
                for (uint64_t b0 = 0; b0 < p.bins && built;) {
                    uint64_t b1 = b0, s = 0, k = 0;
                    while (b1 < p.bins && (p.n[b1] == 0 || (s + per_bin <= budget && k + p.n[b1] < (1ull << 32)))) {
                        if (p.n[b1]) { s += per_bin; k += p.n[b1]; }
                        ++b1;
                    }
                    jobs.clear();
                    add_jobs(p, 0, b0, b1, jobs);
                    full.assign(1, all_bins ? 2 : 0);
                    const int rc = eng.run(jobs, 1, ok, full);
                    if (rc != TAXOR_OK) return rc;
                    if (!ok[0]) built = false;
                    b0 = b1;
                }
                if (!built) {
                    eng.stats.keys_inserted = inserted_before;
                    eng.stats.keys_counted_in_lds = lds_before;
                    if (++p.attempts >= 32) return bfail(TAXOR_E_INTERNAL, "build: no seed peeled every bin of IXF " + std::to_string(p.ixf) + " in 32 attempts (duplicate keys inside a bin?)");
                    p.seed = next_seed(p.seed);
                }
            }
            taxor_index_set_seed(idx, p.ixf, p.seed);
            todo.erase(todo.begin());
            continue;
        }
        const int rc = eng.run(jobs, (uint32_t)members.size(), ok, full);
        if (rc != TAXOR_OK) return rc;
        std::vector<size_t> again;
        for (size_t m = 0; m < members.size(); ++m) {
            IxfPlan &p = plans[members[m]];
            if (ok[m]) taxor_index_set_seed(idx, p.ixf, p.seed);
            else {
                if (++p.attempts >= 32) return bfail(TAXOR_E_INTERNAL, "build: no seed peeled every bin of IXF " + std::to_string(p.ixf) + " in 32 attempts (duplicate keys inside a bin?)");
                p.seed = next_seed(p.seed);         // re-seed and rebuild every bin of this IXF, like construct_ixf.cpp:100-108
                again.push_back(members[m]);
            }
        }
        todo.erase(todo.begin(), todo.begin() + (long)taken);
        todo.insert(todo.begin(), again.begin(), again.end());
    }
    return TAXOR_OK;
}

int plan_ixf(taxor_gpu_index *idx, uint64_t ixf, IxfPlan &p, int *device)
{
    if (taxor_index_ixf_info(idx, ixf, &p.data, &p.stride, &p.seg_len, &p.bins, device)) return bfail(TAXOR_E_ARG, "build: bad index / IXF id");
    p.ixf = ixf;
    p.keys.assign(p.bins, nullptr);
    p.keep.assign(p.bins, nullptr);
    p.n.assign(p.bins, 0);
    p.kept.assign(p.bins, 0);
    p.gen_first.assign(p.bins, NO_GEN);
    return TAXOR_OK;
}

void plan_totals(IxfPlan &p)
{
    p.total = p.max_bin = p.n_with_keys = 0;
    for (uint64_t b = 0; b < p.bins; ++b) {
        p.total += p.n[b];
        p.max_bin = std::max(p.max_bin, p.n[b]);
        p.n_with_keys += p.n[b] != 0;
    }
}

// What a build keeps with its index for the next one (released by taxor_gpu_index_destroy): the peeling scratch, the union table
// and two buffers of mark bytes (the marks of the level below are read while those of this level are written).
struct BuildCtx {
    std::mutex mu;             // one build of an index at a time
    Engine eng;
    KeyUnion unioner;
    uint8_t *mark[2] = {nullptr, nullptr};
    uint64_t mark_cap[2] = {0, 0};
    ~BuildCtx()
    {
        for (auto *m : mark)
            if (m) (void)hipFree(m);
    }
    // mark buffer `which` with room for n bytes
    uint8_t *marks(int which, uint64_t n)
    {
        if (n > mark_cap[which]) {
            const double ta = now_s();
            if (mark[which]) (void)hipFree(mark[which]);
            mark[which] = nullptr;
            mark_cap[which] = 0;
            if (hipMalloc((void **)&mark[which], n) != hipSuccess) return nullptr;
            mark_cap[which] = n;
            eng.stats.seconds_alloc += now_s() - ta;
        }
        return mark[which];
    }
};

void build_ctx_free(void *p) { delete static_cast<BuildCtx *>(p); }

BuildCtx *build_ctx_of(taxor_gpu_index *idx)
{
    auto *c = static_cast<BuildCtx *>(taxor_index_build_ctx(idx));
    if (!c) {
        c = new BuildCtx;
        taxor_index_set_build_ctx(idx, c, build_ctx_free);
    }
    return c;
}

} // namespace

extern "C" uint64_t taxor_synth_key(uint64_t i, uint64_t salt) { return synth_key(i, salt); }

extern "C" int taxor_gpu_synth_keys(int device, uint64_t *d_out, uint64_t first, uint64_t n, uint64_t salt)
{
    if (!d_out && n) return bfail(TAXOR_E_ARG, "synth_keys: null output");
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "synth_keys: hipSetDevice failed");
    for (uint64_t o = 0; o < n; o += (1ull << 30)) {          // (a billion keys per launch: ~1.5 ms)
        const uint64_t m = std::min<uint64_t>(1ull << 30, n - o);
        hipLaunchKernelGGL(k_synth_keys, dim3((uint32_t)std::min<uint64_t>(8192, (m + BB - 1) / BB)), dim3(BB), 0, nullptr, d_out + o, first + o, m, salt);
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return bfail(TAXOR_E_HIP, "synth_keys: kernel failed");
    return TAXOR_OK;
}

extern "C" int taxor_gpu_malloc(int device, uint64_t bytes, void **out)
{
    if (!out) return bfail(TAXOR_E_ARG, "taxor_gpu_malloc: null output");
    *out = nullptr;
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "taxor_gpu_malloc: hipSetDevice failed");
    if (hipMalloc(out, bytes ? bytes : 8) != hipSuccess) return bfail(TAXOR_E_NOMEM, "taxor_gpu_malloc: no device memory for " + std::to_string(bytes) + " bytes");
    return TAXOR_OK;
}

extern "C" void taxor_gpu_free(void *p)
{
    if (p) (void)hipFree(p);
}

extern "C" int taxor_gpu_memcpy_from_host(void *d_dst, const void *src, uint64_t bytes)
{
    return hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? TAXOR_OK : bfail(TAXOR_E_HIP, "taxor_gpu_memcpy_from_host failed");
}

extern "C" int taxor_gpu_memcpy_to_host(void *dst, const void *d_src, uint64_t bytes)
{
    return hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? TAXOR_OK : bfail(TAXOR_E_HIP, "taxor_gpu_memcpy_to_host failed");
}

// Keys from pageable host memory to the device.  One hipMemcpy of tens of GB goes through the runtime's own staging at ~10 GB/s; a
// binding has the hashes in host vectors (construct_ixf.cpp:82,123 reads them from temp files), so the upload is part of its build.
// Four threads, two page-locked 16-MB buffers and a stream each, pieces off a shared cursor -- the shape of the index upload
// (api.hip index_upload).  Small arrays take the plain copy.
static hipError_t upload_keys(int device, uint64_t *d_dst, const uint64_t *src, uint64_t n_keys)
{
    const uint64_t bytes = n_keys * 8, piece = 16ull << 20;
    if (bytes < (256ull << 20)) return hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice);
    const int n_threads = 4;
    std::atomic<uint64_t> cursor{0};
    std::atomic<int> err{(int)hipSuccess};
    auto worker = [&] {
        hipStream_t st = nullptr;
        void *buf[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipHostMalloc(&buf[k], piece, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
        }
        bool used[2] = {false, false};
        for (int k = 0; e == hipSuccess && err.load() == (int)hipSuccess; k ^= 1) {
            const uint64_t off = cursor.fetch_add(piece);
            if (off >= bytes) break;
            const uint64_t len = std::min(piece, bytes - off);
            if (used[k]) e = hipEventSynchronize(done[k]);           // the copy that last read this buffer has finished
            if (e != hipSuccess) break;
            memcpy(buf[k], (const char *)src + off, len);
            e = hipMemcpyAsync((char *)d_dst + off, buf[k], len, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipEventRecord(done[k], st);
            used[k] = true;
        }
        if (st && e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) err.store((int)e);
        for (int k = 0; k < 2; ++k) {
            if (done[k]) (void)hipEventDestroy(done[k]);
            if (buf[k]) (void)hipHostFree(buf[k]);
        }
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(worker);
    for (auto &t : th) t.join();
    return (hipError_t)err.load();
}

// keys: the bins' key lists concatenated, on the host or (keys_on_device) on the index's device
static int build_ixf_impl(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, int keys_on_device, const uint64_t *key_off, uint64_t seed0,
                          uint64_t *seed_out, uint32_t *rounds_out, taxor_build_stats *stats_out)
{
    std::vector<IxfPlan> plans(1);
    int device = 0;
    if (!idx || !key_off) return bfail(TAXOR_E_ARG, "build_ixf: bad index / IXF id");
    int rc = plan_ixf(idx, ixf, plans[0], &device);
    if (rc != TAXOR_OK) return rc;
    IxfPlan &p = plans[0];
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "build_ixf: hipSetDevice failed");
    for (uint64_t b = 0; b < p.bins; ++b)
        if (key_off[b + 1] < key_off[b]) return bfail(TAXOR_E_ARG, "build_ixf: key_off not monotone");
    const uint64_t total = key_off[p.bins] - key_off[0];
    if (seed_out) *seed_out = seed0;
    if (rounds_out) *rounds_out = 0;
    if (stats_out) *stats_out = taxor_build_stats{};
    if (!total) return TAXOR_OK;
    if (!keys) return bfail(TAXOR_E_ARG, "build_ixf: null keys");
    const double t0 = now_s();
    uint64_t *d_own = nullptr;
    const uint64_t *d_keys = keys + key_off[0];
    if (!keys_on_device) {
        if (hipMalloc((void **)&d_own, total * 8) != hipSuccess) return bfail(TAXOR_E_NOMEM, "build_ixf: no device memory for the keys");
        if (upload_keys(device, d_own, keys + key_off[0], total) != hipSuccess) {
            (void)hipFree(d_own);
            return bfail(TAXOR_E_HIP, "build_ixf: key upload failed");
        }
        d_keys = d_own;
    }
    for (uint64_t b = 0; b < p.bins; ++b) {
        p.keys[b] = d_keys + (key_off[b] - key_off[0]);
        p.n[b] = p.kept[b] = key_off[b + 1] - key_off[b];
    }
    plan_totals(p);
    p.seed = seed0;
    BuildCtx *ctx = build_ctx_of(idx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    Engine &eng = ctx->eng;
    eng.stats = taxor_build_stats{};
    rc = eng.open(device);
    if (rc == TAXOR_OK) rc = build_plans(eng, idx, plans);
    eng.stats.seconds_total = now_s() - t0;
    const double t_rel = now_s();
    if (d_own) (void)hipFree(d_own);
    eng.stats.seconds_release = now_s() - t_rel;
    if (rc != TAXOR_OK) return rc;
    if (seed_out) *seed_out = p.seed;
    if (rounds_out) *rounds_out = eng.stats.rounds_max;
    if (stats_out) *stats_out = eng.stats;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_index_build_ixf(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, const uint64_t *key_off,
                                         uint64_t seed0, uint64_t *seed_out, uint32_t *rounds_out)
{
    return build_ixf_impl(idx, ixf, keys, 0, key_off, seed0, seed_out, rounds_out, nullptr);
}

// gen_first / gen_count [total_bins] (both or neither): technical bin g holds gen_count[g] GENERATED keys, synth_key(gen_first[g] + k,
// gen_salt), instead of keys from `keys` (its key_off range must then be empty)
static int build_hixf_impl(taxor_gpu_index *idx, const uint64_t *keys, int keys_on_device, const uint64_t *key_off, uint64_t seed0,
                           uint32_t *rounds_out, taxor_build_stats *stats_out, const uint64_t *gen_first = nullptr, const uint64_t *gen_count = nullptr,
                           uint64_t gen_salt = 0)
{
    uint64_t n_ixf = 0;
    const uint32_t *bin_base = nullptr, *binfo = nullptr;
    if (!idx || !key_off || taxor_index_tree(idx, &n_ixf, &bin_base, &binfo)) return bfail(TAXOR_E_ARG, "build_hixf: bad index");
    std::vector<IxfPlan> plan(n_ixf);
    int device = 0;
    for (uint64_t i = 0; i < n_ixf; ++i) {
        const int rc = plan_ixf(idx, i, plan[i], &device);
        if (rc != TAXOR_OK) return rc;
        plan[i].seed = seed0 + 0x9E3779B97F4A7C15ull * i;
    }
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "build_hixf: hipSetDevice failed");
    const uint64_t total_bins = bin_base[n_ixf];
    for (uint64_t g = 0; g < total_bins; ++g) {
        if (key_off[g + 1] < key_off[g]) return bfail(TAXOR_E_ARG, "build_hixf: key_off not monotone");
        if ((binfo[g] & BINFO_MERGED) && (key_off[g + 1] != key_off[g] || (gen_count && gen_count[g])))
            return bfail(TAXOR_E_ARG, "build_hixf: a merged bin must not bring keys of its own (they come from its child)");
        if (gen_count && gen_count[g] && key_off[g + 1] != key_off[g])
            return bfail(TAXOR_E_ARG, "build_hixf: a bin brings its keys OR has them generated, not both");
        if (gen_count && gen_count[g] >= 0xFFFFFFFFull) return bfail(TAXOR_E_ARG, "build_hixf: more than 2^32 - 2 generated keys in one bin");
    }
    if ((gen_first == nullptr) != (gen_count == nullptr)) return bfail(TAXOR_E_ARG, "build_hixf: gen_first and gen_count come together");
    const uint64_t total = key_off[total_bins] - key_off[0];
    if (total && !keys) return bfail(TAXOR_E_ARG, "build_hixf: null keys");
    if (rounds_out) *rounds_out = 0;
    if (stats_out) *stats_out = taxor_build_stats{};
    const double t0 = now_s();
    // depth of every IXF (the root is IXF 0; children have larger ids than nothing in particular, so walk the tree)
    std::vector<int> depth(n_ixf, -1);
    std::vector<int64_t> parent(n_ixf, -1);
    int max_depth = 0;
    {
        std::vector<uint64_t> stack{0};
        depth[0] = 0;
        while (!stack.empty()) {
            const uint64_t i = stack.back();
            stack.pop_back();
            for (uint64_t b = 0; b < plan[i].bins; ++b) {
                const uint32_t bi = binfo[bin_base[i] + b];
                if (!(bi & BINFO_MERGED)) continue;
                const uint64_t c = bi & 0x3FFFFFFFu;
                if (c >= n_ixf || depth[c] >= 0) return bfail(TAXOR_E_ARG, "build_hixf: the hierarchy is not a tree");
                depth[c] = depth[i] + 1;
                parent[c] = (int64_t)i;
                max_depth = std::max(max_depth, depth[c]);
                stack.push_back(c);
            }
        }
    }
    uint64_t *d_leaf = nullptr;
    const uint64_t *d_keys = keys ? keys + key_off[0] : nullptr;
    std::vector<uint64_t *> arenas;            // unions of one level each; freed when the level above is built
    auto cleanup = [&] {
        if (d_leaf) (void)hipFree(d_leaf);
        for (auto *p : arenas)
            if (p) (void)hipFree(p);
    };
    double t_upload = 0.0;
    if (total && !keys_on_device) {
        const double tu0 = now_s();
        if (hipMalloc((void **)&d_leaf, total * 8) != hipSuccess) return bfail(TAXOR_E_NOMEM, "build_hixf: no device memory for the keys");
        if (upload_keys(device, d_leaf, keys + key_off[0], total) != hipSuccess) {
            cleanup();
            return bfail(TAXOR_E_HIP, "build_hixf: key upload failed");
        }
        d_keys = d_leaf;
        t_upload = now_s() - tu0;
    }
    BuildCtx *ctx = build_ctx_of(idx);
    std::lock_guard<std::mutex> lk(ctx->mu);
    Engine &eng = ctx->eng;
    eng.stats = taxor_build_stats{};
    eng.stats.seconds_upload = t_upload;
    int rc = eng.open(device);
    if (rc != TAXOR_OK) { cleanup(); return rc; }
    // What a merged bin of the level above holds: the keys of everything in this IXF, each once.  For an IXF whose bins are all
    // leaves that is its OWN key range in the caller's array plus one byte per key ("counts" / "a duplicate of an earlier one"):
    // nothing is copied, and the marks are an eighth of the keys (a class-scale level's unions were a 27-GB hipMalloc, half a second
    // of the driver's time on a good day).  An IXF with merged bins of its own (deeper hierarchies) gets its union materialised.
    struct Union { const uint64_t *p = nullptr; const uint8_t *keep = nullptr; uint64_t n = 0, kept = 0, gen_first = NO_GEN; };
    std::vector<Union> uni(n_ixf);
    KeyUnion &unioner = ctx->unioner;
    auto free_level = [&](size_t lvl) {        // (the mark bytes are the context's, two buffers taken in turn by the levels)
        if (lvl < arenas.size() && arenas[lvl]) { (void)hipFree(arenas[lvl]); arenas[lvl] = nullptr; }
    };
    for (int d = max_depth; d >= 0 && rc == TAXOR_OK; --d) {
        std::vector<IxfPlan> level;
        std::vector<uint64_t> ids;
        uint64_t mark_bytes = 0, arena_keys = 0, concat_max = 0;
        std::vector<uint8_t> leaf_only(n_ixf, 0);     // (kind of every IXF's union, see below)
        for (uint64_t i = 0; i < n_ixf; ++i) {
            if (depth[i] != d) continue;
            IxfPlan &p = plan[i];
            p.gen_salt = gen_salt;
            // kind of its union: 0 = its own key range with marks (all bins leaves with keys from `keys`), 1 = materialised (merged bins of its
            // own, or leaves of both kinds), 2 = a generated range (all bins leaves with generated keys whose index ranges follow one another)
            bool only_leaves = true, any_gen = false, any_ptr = false, gen_run = true;
            uint64_t run_next = NO_GEN;
            for (uint64_t b = 0; b < p.bins; ++b) {
                const uint64_t g = bin_base[i] + b;
                p.gen_first[b] = NO_GEN;
                if (binfo[g] & BINFO_MERGED) {
                    const Union &u = uni[binfo[g] & 0x3FFFFFFFu];
                    p.keys[b] = u.p;
                    p.keep[b] = u.keep;
                    p.n[b] = u.n;
                    p.kept[b] = u.kept;
                    p.gen_first[b] = u.gen_first;
                    only_leaves = false;
                } else if (gen_count && gen_count[g]) {
                    p.keys[b] = nullptr;
                    p.keep[b] = nullptr;
                    p.n[b] = p.kept[b] = gen_count[g];
                    p.gen_first[b] = gen_first[g];
                    any_gen = true;
                    if (run_next != NO_GEN && gen_first[g] != run_next) gen_run = false;
                    run_next = gen_first[g] + gen_count[g];
                } else {
                    p.keys[b] = d_keys ? d_keys + (key_off[g] - key_off[0]) : nullptr;
                    p.keep[b] = nullptr;
                    p.n[b] = p.kept[b] = key_off[g + 1] - key_off[g];
                    any_ptr |= p.n[b] != 0;
                }
            }
            plan_totals(p);
            leaf_only[i] = !only_leaves || (any_gen && (any_ptr || !gen_run)) ? 1 : any_gen ? 2 : 0;
            if (leaf_only[i] == 0) mark_bytes += p.total;
            else if (leaf_only[i] == 1) { arena_keys += p.total; concat_max = std::max(concat_max, p.total); }
            ids.push_back(i);
        }
        for (uint64_t i : ids) level.push_back(plan[i]);
        rc = build_plans(eng, idx, level);
        arenas.push_back(nullptr);
        if (rc != TAXOR_OK) break;
        for (size_t q = 0; q < ids.size(); ++q) plan[ids[q]].seed = level[q].seed;
        if (d == 0) break;
        // what the level above inserts into its merged bins
        const double tu = now_s();
        uint64_t *arena = nullptr, *concat = nullptr;
        uint8_t *mark = nullptr;
        if (mark_bytes && !(mark = ctx->marks(d & 1, mark_bytes))) { rc = bfail(TAXOR_E_NOMEM, "build_hixf: no device memory for the duplicate marks of one level"); break; }
        if (arena_keys && hipMalloc((void **)&arena, arena_keys * 8) != hipSuccess) { rc = bfail(TAXOR_E_NOMEM, "build_hixf: no device memory for the key unions of one level"); break; }
        arenas.back() = arena;
        uint64_t used = 0, marked = 0;
        for (uint64_t i : ids) {
            IxfPlan &p = plan[i];
            if (!p.total) continue;
            if (p.total >= 0xFFFFFFFFull) { rc = bfail(TAXOR_E_ARG, "build_hixf: more than 2^32 - 2 keys below one merged bin"); break; }
            if (leaf_only[i] == 2) {
                // generated keys of consecutive indices: their union is the range of indices itself (a bijection: no duplicates, no memory)
                uint64_t first = NO_GEN;
                for (uint64_t b = 0; b < p.bins && first == NO_GEN; ++b)
                    if (p.n[b]) first = p.gen_first[b];
                Union u;
                u.n = u.kept = p.total;
                u.gen_first = first;
                uni[i] = u;
                continue;
            }
            if (leaf_only[i] == 0) {
                // its leaf bins are adjacent in the caller's array: the range itself, with marks
                const uint64_t *src = d_keys + (key_off[bin_base[i]] - key_off[0]);
                uint64_t kept = 0;
                const hipError_t e = unioner.mark(src, p.total, mark + marked, &kept, eng.st);
                if (e != hipSuccess) { rc = bfail(TAXOR_E_HIP, std::string("build_hixf: key union failed: ") + hipGetErrorString(e)); break; }
                uni[i] = Union{src, kept == p.total ? nullptr : mark + marked, p.total, kept, NO_GEN};
                marked += p.total;
                continue;
            }
            // bins of several kinds: everything gathered (a child's range goes in WITH its duplicates: the set drops them again)
            if (!concat && hipMalloc((void **)&concat, concat_max * 8) != hipSuccess) { rc = bfail(TAXOR_E_NOMEM, "build_hixf: no device memory for the keys of one IXF"); break; }
            uint64_t o = 0;
            for (uint64_t b = 0; b < p.bins; ++b) {
                if (p.n[b] && p.gen_first[b] != NO_GEN)       // generated keys are written out for the set
                    hipLaunchKernelGGL(k_synth_keys, dim3((uint32_t)std::min<uint64_t>(8192, (p.n[b] + BB - 1) / BB)), dim3(BB), 0, eng.st, concat + o, p.gen_first[b], p.n[b], gen_salt);
                else if (p.n[b] && hipMemcpyAsync(concat + o, p.keys[b], p.n[b] * 8, hipMemcpyDeviceToDevice, eng.st) != hipSuccess) rc = bfail(TAXOR_E_HIP, "build_hixf: device copy failed");
                o += p.n[b];
            }
            if (rc != TAXOR_OK) break;
            uint64_t n_out = 0;
            const hipError_t e = unioner.unique(concat, p.total, arena + used, &n_out, eng.st);
            if (e != hipSuccess) { rc = bfail(TAXOR_E_HIP, std::string("build_hixf: key union failed: ") + hipGetErrorString(e)); break; }
            uni[i] = Union{arena + used, nullptr, n_out, n_out, NO_GEN};
            used += n_out;
        }
        if (concat) (void)hipFree(concat);
        // the unions of the level below are not needed any more
        if (arenas.size() >= 2) free_level(arenas.size() - 2);
        eng.stats.seconds_union += now_s() - tu;
    }
    // the job is done here; handing tens of GB of keys, unions and scratch back to the driver is timed apart (hipFree of that much
    // takes anything between milliseconds and over a second, whatever was done with the memory)
    eng.stats.seconds_total = now_s() - t0;
    const double t_rel = now_s();
    cleanup();
    eng.stats.seconds_release = now_s() - t_rel;
    if (rc != TAXOR_OK) return rc;
    if (rounds_out) *rounds_out = eng.stats.rounds_max;
    if (stats_out) *stats_out = eng.stats;
    return TAXOR_OK;
}

extern "C" int taxor_gpu_index_build_hixf(taxor_gpu_index *idx, const uint64_t *keys, const uint64_t *key_off, uint64_t seed0,
                                          uint32_t *rounds_out)
{
    return build_hixf_impl(idx, keys, 0, key_off, seed0, rounds_out, nullptr);
}

extern "C" int taxor_gpu_index_build_ixf_ex(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, int keys_on_device, const uint64_t *key_off,
                                            uint64_t seed0, uint64_t *seed_out, taxor_build_stats *stats)
{
    return build_ixf_impl(idx, ixf, keys, keys_on_device, key_off, seed0, seed_out, nullptr, stats);
}

extern "C" int taxor_gpu_index_build_hixf_ex(taxor_gpu_index *idx, const uint64_t *keys, int keys_on_device, const uint64_t *key_off, uint64_t seed0,
                                             taxor_build_stats *stats)
{
    return build_hixf_impl(idx, keys, keys_on_device, key_off, seed0, nullptr, stats);
}

extern "C" int taxor_gpu_index_build_hixf_gen(taxor_gpu_index *idx, const uint64_t *keys, int keys_on_device, const uint64_t *key_off,
                                              const uint64_t *gen_first, const uint64_t *gen_count, uint64_t gen_salt, uint64_t seed0, taxor_build_stats *stats)
{
    return build_hixf_impl(idx, keys, keys_on_device, key_off, seed0, nullptr, stats, gen_first, gen_count, gen_salt);
}
