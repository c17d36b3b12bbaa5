// keyset.hip -- device-side union of key sets for the hierarchical build: a merged bin of a parent IXF holds
// every key of its child's subtree, each once (src/hixf/build/hierarchical_build.cpp:27-236 collects the child's k-mers in a hash
// set and inserts them into the parent's merged bin; construct_ixf.cpp:62-66 keeps an ankerl::unordered_dense::set for the same
// purpose).  The same here, on the device: an open-addressing set in HBM -- one atomic compare-and-swap per key (the chip does
// 18-27 G random read-modify-writes a second, profiles/r06/atomics_bench.txt).  What comes out is either one MARK byte per input key
// (mark(): nothing is copied -- the form the builder uses for an IXF of leaf bins, whose keys already lie in one range) or the
// occupied slots written out as a new array (unique(): an IXF with merged bins of its own).
// A 64-bit radix sort + unique of the same keys (rocPRIM, rounds 3-5) took 16 ms for a GTDB-class child's 54 M keys, a fifth of
// the whole build; this takes 4.  The order of the output is the table's, not sorted: the builder does not care (the columns it
// constructs depend on the key SET only, builder.hip).
#include "keyset.h"

#include <algorithm>

namespace taxor {

namespace {

constexpr int KB = 256;
constexpr uint64_t EMPTY = ~0ull;
constexpr uint32_t ENTRIES_PER_BLOCK = 4096;
constexpr uint64_t LAUNCH_KEYS = 1ull << 24;     // keys per k_set_insert / k_set_mark launch

__device__ __forceinline__ uint64_t slot_hash(uint64_t k)
{
    k ^= k >> 32;
    k *= 0xD6E8FEB86659FD93ull;
    k ^= k >> 32;
    return k;
}

__global__ __launch_bounds__(KB) void k_set_insert(const uint64_t *__restrict__ in, uint64_t n, uint64_t *tab, uint64_t mask, unsigned long long *ctl)
{
    for (uint64_t i = (uint64_t)blockIdx.x * KB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * KB) {
        const uint64_t key = in[i];
        if (key == EMPTY) { ctl[1] = 1ull; continue; }            // the marker itself as a key: remembered aside
        uint64_t s = slot_hash(key) & mask;
        for (;;) {
            const uint64_t old = atomicCAS((unsigned long long *)&tab[s], (unsigned long long)EMPTY, (unsigned long long)key);
            if (old == EMPTY || old == key) break;
            s = (s + 1) & mask;
        }
    }
}

// the same insertion, but what comes out is one byte per INPUT position: 1 for the occurrence that took the slot, 0 for a duplicate
// (two lanes that insert one key at the same time: exactly one of them sees the slot empty)
__global__ __launch_bounds__(KB) void k_set_mark(const uint64_t *__restrict__ in, uint64_t n, uint64_t *tab, uint64_t mask, uint8_t *__restrict__ keep,
                                                 unsigned long long *ctl)
{
    __shared__ uint32_t kept_blk;
    if (threadIdx.x == 0) kept_blk = 0;
    __syncthreads();
    uint32_t kept = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * KB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * KB) {
        const uint64_t key = in[i];
        bool first;
        if (key == EMPTY) first = atomicCAS(&ctl[1], 0ull, 1ull) == 0ull;       // the marker itself as a key: one of them counts
        else {
            uint64_t s = slot_hash(key) & mask;
            for (;;) {
                const uint64_t old = atomicCAS((unsigned long long *)&tab[s], (unsigned long long)EMPTY, (unsigned long long)key);
                if (old == EMPTY) { first = true; break; }
                if (old == key) { first = false; break; }
                s = (s + 1) & mask;
            }
        }
        keep[i] = first ? 1 : 0;
        kept += first;
    }
    if (kept) atomicAdd(&kept_blk, kept);
    __syncthreads();
    if (threadIdx.x == 0 && kept_blk) atomicAdd(&ctl[0], (unsigned long long)kept_blk);
}

// occupied slots -> out, one returning atomic per block and 4096 slots (entries collected in LDS)
__global__ __launch_bounds__(KB) void k_set_compact(const uint64_t *__restrict__ tab, uint64_t entries, uint64_t *__restrict__ out, unsigned long long *ctl)
{
    __shared__ uint64_t stage[ENTRIES_PER_BLOCK];
    __shared__ uint32_t stage_n;
    __shared__ unsigned long long gbase;
    if (threadIdx.x == 0) stage_n = 0;
    __syncthreads();
    const uint64_t e0 = (uint64_t)blockIdx.x * ENTRIES_PER_BLOCK;
    for (uint64_t e = e0 + threadIdx.x; e < e0 + ENTRIES_PER_BLOCK; e += KB) {      // (uniform trip count: the ballot below is a wave operation)
        const uint64_t v = e < entries ? tab[e] : EMPTY;
        const bool have = v != EMPTY;
        const uint64_t m = __ballot(have);
        if (m) {
            const int lane = (int)__lane_id(), leader = __ffsll((unsigned long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&stage_n, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, leader);
            if (have) stage[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = v;
        }
    }
    __syncthreads();
    const uint32_t cnt = stage_n;
    if (threadIdx.x == 0 && cnt) gbase = atomicAdd(&ctl[0], (unsigned long long)cnt);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cnt; k += KB) out[gbase + k] = stage[k];
}

} // namespace

void KeyUnion::release()
{
    if (table) (void)hipFree(table);
    if (d_ctl) (void)hipFree(d_ctl);
    if (h_ctl) (void)hipHostFree(h_ctl);
    table = nullptr;
    d_ctl = nullptr;
    h_ctl = nullptr;
    table_entries = 0;
}

hipError_t KeyUnion::prepare(uint64_t n, uint64_t *entries_out, hipStream_t st)
{
    hipError_t e = hipSuccess;
    if (!d_ctl) e = hipMalloc((void **)&d_ctl, 2 * sizeof(unsigned long long));
    if (e == hipSuccess && !h_ctl) e = hipHostMalloc((void **)&h_ctl, 2 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    uint64_t entries = 1024;
    while (entries < n + n / 2) entries <<= 1;                   // load <= 2/3: short probe sequences
    if (entries > table_entries) {
        if (table) (void)hipFree(table);
        table = nullptr;
        table_entries = 0;
        e = hipMalloc((void **)&table, entries * sizeof(uint64_t));
        if (e != hipSuccess) return e;
        table_entries = entries;
    }
    e = hipMemsetAsync(table, 0xFF, entries * sizeof(uint64_t), st);
    if (e == hipSuccess) e = hipMemsetAsync(d_ctl, 0, 2 * sizeof(unsigned long long), st);
    *entries_out = entries;
    return e;
}

hipError_t KeyUnion::mark(const uint64_t *d_in, uint64_t n, uint8_t *d_keep, uint64_t *n_kept, hipStream_t st)
{
    *n_kept = 0;
    if (n == 0) return hipSuccess;
    uint64_t entries = 0;
    hipError_t e = prepare(n, &entries, st);
    if (e != hipSuccess) return e;
    for (uint64_t i0 = 0; i0 < n; i0 += LAUNCH_KEYS) {            // (launches of ~1.5 ms whatever the child's size)
        const uint64_t m = std::min<uint64_t>(LAUNCH_KEYS, n - i0);
        hipLaunchKernelGGL(k_set_mark, dim3((uint32_t)std::min<uint64_t>(8192, (m + KB - 1) / KB)), dim3(KB), 0, st, d_in + i0, m, table, entries - 1, d_keep + i0, d_ctl);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_ctl, d_ctl, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    *n_kept = h_ctl[0];
    return hipSuccess;
}

hipError_t KeyUnion::unique(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *n_out, hipStream_t st)
{
    *n_out = 0;
    if (n == 0) return hipSuccess;
    uint64_t entries = 0;
    hipError_t e = prepare(n, &entries, st);
    if (e != hipSuccess) return e;
    for (uint64_t i0 = 0; i0 < n; i0 += LAUNCH_KEYS) {
        const uint64_t m = std::min<uint64_t>(LAUNCH_KEYS, n - i0);
        hipLaunchKernelGGL(k_set_insert, dim3((uint32_t)std::min<uint64_t>(8192, (m + KB - 1) / KB)), dim3(KB), 0, st, d_in + i0, m, table, entries - 1, d_ctl);
    }
    hipLaunchKernelGGL(k_set_compact, dim3((uint32_t)((entries + ENTRIES_PER_BLOCK - 1) / ENTRIES_PER_BLOCK)), dim3(KB), 0, st, table, entries, d_out, d_ctl);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h_ctl, d_ctl, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    uint64_t cnt = h_ctl[0];
    if (h_ctl[1]) {                                               // the key that equals the marker, once
        const uint64_t marker = EMPTY;
        e = hipMemcpyAsync(d_out + cnt, &marker, sizeof marker, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        ++cnt;
    }
    *n_out = cnt;
    return hipSuccess;
}

} // namespace taxor
