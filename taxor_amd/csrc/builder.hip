// builder.hip -- GPU construction of the fingerprint columns of one IXF.
//
// The reference builds an IXF on the CPU, bin by bin: add_bin_elements(bin, hashes) peels the 3-uniform hypergraph
// of the bin's keys and assigns fingerprints in reverse peeling order; if any bin fails to peel the whole IXF is
// cleared, re-seeded and rebuilt (src/hixf/build/construct_ixf.cpp:50-165; XOR-filter algorithm family of
// src/main/xorfilter.hpp:142-334).  Bins are independent, so the GPU peels all bins of a chunk at once, in
// synchronous rounds over a work list of rows that just became singletons:
//
//   count   : per (bin,row) degree + XOR of incident keys                                   (k_build_count)
//   seed    : rows with degree 1 -> work list, flagged as this round's snapshot singletons   (k_build_seed)
//   round   : a key A reachable from a snapshot singleton is peeled by exactly one of them -- the smallest-index row
//             of A that is flagged -- which logs (key,row), and removes A from its three rows; rows whose degree
//             drops to 1 go to the next work list                                           (k_build_round)
//   flag    : flags of the next list are set at the round boundary, never inside a round, so the ownership
//             rule is evaluated on a stable snapshot                          (k_build_unflag, k_build_setflag)
//   assign  : rounds in reverse; keys peeled in the same round never touch each other's singleton row, so a
//             round is assigned in parallel: D[row] = fp ^ D[row'] ^ D[row'']                (k_build_assign)
//
// The fingerprints differ from a sequential peel (any peeling order yields a valid filter); every key of every bin
// matches, which is what the tests check through the query kernels and the CPU oracle.
//
// taxor_gpu_index_build_hixf builds a whole hierarchy bottom-up: the keys stay on the device, a merged bin's key set
// is the sorted, duplicate-free union of everything in its child IXF (keyset.hip), and every IXF goes through the
// same peeling.
#include "../../include/taxor_gpu_tools.h"
#include "ixf_arith.h"
#include "kernels.h"
#include "keyset.h"

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

using namespace taxor;

extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);

namespace {

constexpr int BB = 256;

struct BuildArgs {
    const uint64_t *keys;      // keys of the chunk's bins, concatenated
    const uint32_t *key_bin;   // chunk-local bin of every key
    uint64_t n_keys;
    uint64_t seed;
    uint32_t seg_len;
    uint32_t arith;            // arithmetic code of the index (ixf_arith.h), 0 = this library's reading
    uint64_t rows;             // 3 * seg_len
    uint32_t *cnt;             // [chunk_bins * rows]
    uint64_t *xr;              // [chunk_bins * rows]
    uint8_t *single;           // snapshot flags [chunk_bins * rows]
    uint64_t *wl[2];           // work lists of slots (bin * rows + row)
    uint32_t *wl_n;            // [2]
    uint64_t *st_key;          // peel log
    uint64_t *st_slot;
    uint32_t *st_n;            // entries logged so far
};

// chunk-local bin of every key: off[nb+1] are the chunk's bin boundaries inside its contiguous key range
__global__ __launch_bounds__(BB) void k_build_key_bin(const uint64_t *off, uint32_t nb, uint64_t n_keys, uint32_t *key_bin)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < n_keys; i += (uint64_t)gridDim.x * BB) {
        uint32_t lo = 0, hi = nb;                     // last bin whose start is <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= i) lo = mid; else hi = mid;
        }
        key_bin[i] = lo;
    }
}

__global__ __launch_bounds__(BB) void k_build_count(const BuildArgs a)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < a.n_keys; i += (uint64_t)gridDim.x * BB) {
        const uint64_t key = a.keys[i];
        const ixf_probe p = ixf_probe_key_arith(key, a.seed, a.seg_len, a.arith);
        const uint64_t base = (uint64_t)a.key_bin[i] * a.rows;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            atomicAdd(&a.cnt[base + p.row[j]], 1u);
            atomicXor((unsigned long long *)&a.xr[base + p.row[j]], (unsigned long long)key);
        }
    }
}

__global__ __launch_bounds__(BB) void k_build_seed(const BuildArgs a, uint64_t n_slots)
{
    for (uint64_t s = (uint64_t)blockIdx.x * BB + threadIdx.x; s < n_slots; s += (uint64_t)gridDim.x * BB) {
        if (a.cnt[s] == 1u) {
            a.single[s] = 1;
            a.wl[0][atomicAdd(&a.wl_n[0], 1u)] = s;
        }
    }
}

// one peeling round over work list `cur`; pushes rows that drop to degree 1 onto the other list
__global__ __launch_bounds__(BB) void k_build_round(const BuildArgs a, int cur)
{
    const uint32_t n = a.wl_n[cur];
    for (uint32_t i = blockIdx.x * BB + threadIdx.x; i < n; i += gridDim.x * BB) {
        const uint64_t slot = a.wl[cur][i];
        // a snapshot singleton holds exactly one key; only that key's owner may modify the slot during this round,
        // and it decrements the degree BEFORE it xors the key out, so degree==1 read after the key proves the key
        // read is intact (the key itself may legitimately be 0: wyhash(poly-A k-mer) = 0)
        if (__hip_atomic_load(&a.cnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) continue;
        __threadfence();
        const uint64_t key = __hip_atomic_load(&a.xr[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (__hip_atomic_load(&a.cnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) continue;
        const uint64_t bin = slot / a.rows;
        const uint32_t row = (uint32_t)(slot - bin * a.rows);
        const ixf_probe p = ixf_probe_key_arith(key, a.seed, a.seg_len, a.arith);
        const uint64_t base = bin * a.rows;
        // ownership: smallest-index row of this key that is flagged in the snapshot
        uint32_t owner = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (a.single[base + p.row[j]]) owner = min(owner, p.row[j]);
        if (owner != row) continue; // another singleton row of the same key peels it (or the slot is stale)
        const uint32_t pos = atomicAdd(a.st_n, 1u);
        a.st_key[pos] = key;
        a.st_slot[pos] = slot;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint64_t s2 = base + p.row[j];
            const uint32_t c = atomicSub(&a.cnt[s2], 1u) - 1u;
            __threadfence();
            atomicXor((unsigned long long *)&a.xr[s2], (unsigned long long)key);
            if (c == 1u) a.wl[cur ^ 1][atomicAdd(&a.wl_n[cur ^ 1], 1u)] = s2;
        }
    }
}

__global__ void k_build_reset(uint32_t *wl_n, int which) { wl_n[which] = 0; }

// round boundary, two launches so that retiring the old flags finishes grid-wide before the new ones are set:
// flags change only here, never inside a round, so the ownership rule sees a stable snapshot
__global__ __launch_bounds__(BB) void k_build_unflag(const BuildArgs a, int done)
{
    const uint32_t nd = a.wl_n[done];
    for (uint32_t i = blockIdx.x * BB + threadIdx.x; i < nd; i += gridDim.x * BB) a.single[a.wl[done][i]] = 0;
}

__global__ __launch_bounds__(BB) void k_build_setflag(const BuildArgs a, int next)
{
    const uint32_t nn = a.wl_n[next];
    for (uint32_t i = blockIdx.x * BB + threadIdx.x; i < nn; i += gridDim.x * BB) {
        const uint64_t s = a.wl[next][i];
        if (a.cnt[s] == 1u) a.single[s] = 1;
    }
}

// assign the log entries [lo, hi) of one round: D[row] = fp ^ D[other two rows]
__global__ __launch_bounds__(BB) void k_build_assign(const BuildArgs a, uint8_t *data, uint64_t stride, const uint32_t *bin_ids,
                                                     uint32_t lo, uint32_t hi)
{
    for (uint32_t i = lo + blockIdx.x * BB + threadIdx.x; i < hi; i += gridDim.x * BB) {
        const uint64_t key = a.st_key[i], slot = a.st_slot[i];
        const uint64_t cb = slot / a.rows;
        const uint32_t row = (uint32_t)(slot - cb * a.rows);
        const uint64_t bin = bin_ids[cb];
        const ixf_probe p = ixf_probe_key_arith(key, a.seed, a.seg_len, a.arith);
        uint8_t v = (uint8_t)(p.fp4 & 0xFFu);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (p.row[j] != row) v ^= data[(uint64_t)p.row[j] * stride + bin];
        data[(uint64_t)row * stride + bin] = v;
    }
}

__global__ __launch_bounds__(BB) void k_build_zero_columns(uint8_t *data, uint64_t stride, uint64_t rows, const uint32_t *bin_ids,
                                                           uint32_t n_bins)
{
    const uint64_t total = rows * n_bins;
    for (uint64_t i = (uint64_t)blockIdx.x * BB + threadIdx.x; i < total; i += (uint64_t)gridDim.x * BB) {
        const uint64_t r = i / n_bins;
        data[r * stride + bin_ids[i - r * n_bins]] = 0;
    }
}

int bfail(int code, const std::string &m)
{
    taxor_set_last_error(m.c_str());
    return code;
}

#define B_TRY(expr)                                                                                      \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { cleanup(); return bfail(TAXOR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

} // namespace

// library-internal accessors implemented in api.hip
extern "C" __attribute__((visibility("hidden"))) int taxor_index_ixf_info(taxor_gpu_index *idx, uint64_t ixf, uint8_t **data,
                                                                          uint64_t *stride, uint64_t *seg_len, uint64_t *bins,
                                                                          int *device);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_set_seed(taxor_gpu_index *idx, uint64_t ixf, uint64_t seed);
extern "C" __attribute__((visibility("hidden"))) uint32_t taxor_index_arith(const taxor_gpu_index *idx);

namespace {

// d_keys: the bins' key lists concatenated ON THE DEVICE (distinct within a bin); key_off[bins+1] on the host
int build_ixf_device(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *d_all_keys, const uint64_t *key_off, uint64_t seed0,
                     uint64_t *seed_out, uint32_t *rounds_out)
{
    const uint64_t *keys = d_all_keys;
    uint8_t *data = nullptr;
    uint64_t stride = 0, seg_len = 0, bins = 0;
    int device = 0;
    if (!idx || !key_off || taxor_index_ixf_info(idx, ixf, &data, &stride, &seg_len, &bins, &device))
        return bfail(TAXOR_E_ARG, "build_ixf: bad index / IXF id");
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "build_ixf: hipSetDevice failed");
    const uint64_t rows = 3 * seg_len;
    // bins that have keys, largest first; chunks bounded by scratch (count+xor+flag+lists = 29 B per (bin,row))
    std::vector<uint32_t> work;
    uint64_t max_keys = 0;
    for (uint64_t b = 0; b < bins; ++b) {
        const uint64_t n = key_off[b + 1] - key_off[b];
        if (key_off[b + 1] < key_off[b]) return bfail(TAXOR_E_ARG, "build_ixf: key_off not monotone");
        if (n) work.push_back((uint32_t)b);
        max_keys = std::max(max_keys, n);
    }
    if (seed_out) *seed_out = seed0;
    if (rounds_out) *rounds_out = 0;
    if (work.empty()) return TAXOR_OK;
    if (!keys) return bfail(TAXOR_E_ARG, "build_ixf: null keys");
    if (max_keys > rows) return bfail(TAXOR_E_ARG, "build_ixf: a bin holds more keys than the IXF has rows");
    const uint64_t scratch_budget = 24ull << 30;
    uint64_t chunk_bins = std::max<uint64_t>(1, std::min<uint64_t>(work.size(), scratch_budget / (29 * rows)));
    if (chunk_bins * rows >= (1ull << 32)) chunk_bins = std::max<uint64_t>(1, ((1ull << 32) - 1) / rows); // work-list counters are u32

    uint32_t *d_cnt = nullptr, *d_key_bin = nullptr, *d_ctr = nullptr, *d_bin_ids = nullptr;
    uint64_t *d_xr = nullptr, *d_wl0 = nullptr, *d_wl1 = nullptr, *d_st_key = nullptr, *d_st_slot = nullptr, *d_off = nullptr;
    uint8_t *d_single = nullptr;
    auto cleanup = [&] {
        for (void *p : {(void *)d_cnt, (void *)d_key_bin, (void *)d_ctr, (void *)d_bin_ids, (void *)d_xr, (void *)d_wl0, (void *)d_wl1,
                        (void *)d_st_key, (void *)d_st_slot, (void *)d_single, (void *)d_off})
            if (p) (void)hipFree(p);
    };
    const uint64_t n_slots_max = chunk_bins * rows;
    B_TRY(hipMalloc((void **)&d_cnt, n_slots_max * 4));
    B_TRY(hipMalloc((void **)&d_xr, n_slots_max * 8));
    B_TRY(hipMalloc((void **)&d_single, n_slots_max));
    B_TRY(hipMalloc((void **)&d_wl0, n_slots_max * 8));
    B_TRY(hipMalloc((void **)&d_wl1, n_slots_max * 8));
    B_TRY(hipMalloc((void **)&d_ctr, 64));
    B_TRY(hipMalloc((void **)&d_bin_ids, chunk_bins * 4));
    B_TRY(hipMalloc((void **)&d_off, (chunk_bins + 1) * 8));

    uint64_t seed = seed0;
    uint32_t max_rounds = 0;
    for (int attempt = 0; attempt < 32; ++attempt) {
        bool failed = false;
        for (size_t c0 = 0; c0 < work.size() && !failed; c0 += chunk_bins) {
            const size_t nb = std::min<size_t>(chunk_bins, work.size() - c0);
            // bins are taken in bin order, and bins without keys have empty ranges: the chunk's keys are one contiguous
            // range of the concatenated device array
            const uint64_t k0 = key_off[work[c0]], nk = key_off[work[c0 + nb - 1] + 1] - k0;
            const uint64_t *d_keys = keys + k0;
            std::vector<uint64_t> hoff(nb + 1);
            for (size_t i = 0; i < nb; ++i) hoff[i] = key_off[work[c0 + i]] - k0;
            hoff[nb] = nk;
            if (nk >= (1ull << 32)) { cleanup(); return bfail(TAXOR_E_ARG, "build_ixf: more than 2^32 keys in one chunk of bins"); }
            if (d_key_bin) { (void)hipFree(d_key_bin); (void)hipFree(d_st_key); (void)hipFree(d_st_slot); d_key_bin = nullptr; d_st_key = nullptr; d_st_slot = nullptr; }
            B_TRY(hipMalloc((void **)&d_key_bin, nk * 4));
            B_TRY(hipMalloc((void **)&d_st_key, nk * 8));
            B_TRY(hipMalloc((void **)&d_st_slot, nk * 8));
            B_TRY(hipMemcpy(d_off, hoff.data(), (nb + 1) * 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_build_key_bin, dim3(2048), dim3(BB), 0, nullptr, d_off, (uint32_t)nb, nk, d_key_bin);
            B_TRY(hipMemcpy(d_bin_ids, work.data() + c0, nb * 4, hipMemcpyHostToDevice));
            const uint64_t n_slots = nb * rows;
            B_TRY(hipMemset(d_cnt, 0, n_slots * 4));
            B_TRY(hipMemset(d_xr, 0, n_slots * 8));
            B_TRY(hipMemset(d_single, 0, n_slots));
            B_TRY(hipMemset(d_ctr, 0, 64));
            BuildArgs a{};
            a.keys = d_keys;
            a.key_bin = d_key_bin;
            a.n_keys = nk;
            a.seed = seed;
            a.arith = taxor_index_arith(idx);
            a.seg_len = (uint32_t)seg_len;
            a.rows = rows;
            a.cnt = d_cnt;
            a.xr = d_xr;
            a.single = d_single;
            a.wl[0] = d_wl0;
            a.wl[1] = d_wl1;
            a.wl_n = d_ctr;          // [0], [1]
            a.st_n = d_ctr + 2;
            a.st_key = d_st_key;
            a.st_slot = d_st_slot;
            const int grid = 2048;
            hipLaunchKernelGGL(k_build_count, dim3(grid), dim3(BB), 0, nullptr, a);
            hipLaunchKernelGGL(k_build_seed, dim3(grid), dim3(BB), 0, nullptr, a, n_slots);
            std::vector<uint32_t> round_end; // log size after each round
            uint32_t h[3] = {0, 0, 0};
            int cur = 0;
            for (uint32_t round = 0; round < 4096; ++round) {
                hipLaunchKernelGGL(k_build_round, dim3(grid), dim3(BB), 0, nullptr, a, cur);
                hipLaunchKernelGGL(k_build_unflag, dim3(grid), dim3(BB), 0, nullptr, a, cur);
                hipLaunchKernelGGL(k_build_setflag, dim3(grid), dim3(BB), 0, nullptr, a, cur ^ 1);
                hipLaunchKernelGGL(k_build_reset, dim3(1), dim3(1), 0, nullptr, d_ctr, cur);
                B_TRY(hipMemcpy(h, d_ctr, 12, hipMemcpyDeviceToHost)); // (the reset above zeroed h[cur])
                round_end.push_back(h[2]);
                cur ^= 1;
                if (h[2] == nk || h[cur] == 0) break;
            }
            max_rounds = std::max<uint32_t>(max_rounds, (uint32_t)round_end.size());
            if (h[2] != nk) { failed = true; break; } // not peelable under this seed (or duplicate keys in a bin)
            hipLaunchKernelGGL(k_build_zero_columns, dim3(grid), dim3(BB), 0, nullptr, data, stride, rows, d_bin_ids, (uint32_t)nb);
            for (size_t r = round_end.size(); r-- > 0;) {
                const uint32_t lo = r ? round_end[r - 1] : 0u, hi = round_end[r];
                if (hi > lo) {
                    const int g2 = (int)std::min<uint32_t>(2048u, (hi - lo + BB - 1) / BB);
                    hipLaunchKernelGGL(k_build_assign, dim3(g2), dim3(BB), 0, nullptr, a, data, stride, d_bin_ids, lo, hi);
                }
            }
            B_TRY(hipGetLastError());
            B_TRY(hipDeviceSynchronize());
        }
        if (!failed) {
            cleanup();
            taxor_index_set_seed(idx, ixf, seed);
            if (seed_out) *seed_out = seed;
            if (rounds_out) *rounds_out = max_rounds;
            return TAXOR_OK;
        }
        // re-seed and rebuild every bin of this IXF, like construct_ixf.cpp:100-108
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
    }
    cleanup();
    return bfail(TAXOR_E_INTERNAL, "build_ixf: no seed peeled every bin in 32 attempts (duplicate keys inside a bin?)");
}

} // namespace

extern "C" int taxor_gpu_index_build_ixf(taxor_gpu_index *idx, uint64_t ixf, const uint64_t *keys, const uint64_t *key_off,
                                         uint64_t seed0, uint64_t *seed_out, uint32_t *rounds_out)
{
    uint8_t *data = nullptr;
    uint64_t stride = 0, seg_len = 0, bins = 0;
    int device = 0;
    if (!idx || !key_off || taxor_index_ixf_info(idx, ixf, &data, &stride, &seg_len, &bins, &device))
        return bfail(TAXOR_E_ARG, "build_ixf: bad index / IXF id");
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "build_ixf: hipSetDevice failed");
    const uint64_t total = key_off[bins] - key_off[0];
    if (total && !keys) return bfail(TAXOR_E_ARG, "build_ixf: null keys");
    uint64_t *d_keys = nullptr;
    if (total) {
        if (hipMalloc((void **)&d_keys, total * 8) != hipSuccess) return bfail(TAXOR_E_NOMEM, "build_ixf: no device memory for the keys");
        if (hipMemcpy(d_keys, keys + key_off[0], total * 8, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_keys);
            return bfail(TAXOR_E_HIP, "build_ixf: key upload failed");
        }
    }
    std::vector<uint64_t> off(bins + 1);
    for (uint64_t b = 0; b <= bins; ++b) off[b] = key_off[b] - key_off[0];
    const int rc = build_ixf_device(idx, ixf, d_keys, off.data(), seed0, seed_out, rounds_out);
    if (d_keys) (void)hipFree(d_keys);
    return rc;
}

// library-internal: tree of the resident index (api.hip)
extern "C" __attribute__((visibility("hidden"))) int taxor_index_tree(taxor_gpu_index *idx, uint64_t *n_ixf, const uint32_t **bin_base,
                                                                      const uint32_t **binfo);

extern "C" int taxor_gpu_index_build_hixf(taxor_gpu_index *idx, const uint64_t *keys, const uint64_t *key_off, uint64_t seed0,
                                          uint32_t *rounds_out)
{
    uint64_t n_ixf = 0;
    const uint32_t *bin_base = nullptr, *binfo = nullptr;
    if (!idx || !key_off || taxor_index_tree(idx, &n_ixf, &bin_base, &binfo)) return bfail(TAXOR_E_ARG, "build_hixf: bad index");
    uint8_t *data = nullptr;
    uint64_t stride = 0, seg_len = 0, bins = 0;
    int device = 0;
    taxor_index_ixf_info(idx, 0, &data, &stride, &seg_len, &bins, &device);
    if (hipSetDevice(device) != hipSuccess) return bfail(TAXOR_E_HIP, "build_hixf: hipSetDevice failed");
    const uint64_t total_bins = bin_base[n_ixf];
    for (uint64_t g = 0; g < total_bins; ++g) {
        if (key_off[g + 1] < key_off[g]) return bfail(TAXOR_E_ARG, "build_hixf: key_off not monotone");
        if ((binfo[g] & BINFO_MERGED) && key_off[g + 1] != key_off[g])
            return bfail(TAXOR_E_ARG, "build_hixf: a merged bin must not bring keys of its own (they come from its child)");
    }
    const uint64_t total = key_off[total_bins] - key_off[0];
    if (total && !keys) return bfail(TAXOR_E_ARG, "build_hixf: null keys");
    uint64_t *d_leaf = nullptr;
    if (total) {
        if (hipMalloc((void **)&d_leaf, total * 8) != hipSuccess) return bfail(TAXOR_E_NOMEM, "build_hixf: no device memory for the keys");
        if (hipMemcpy(d_leaf, keys + key_off[0], total * 8, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d_leaf);
            return bfail(TAXOR_E_HIP, "build_hixf: key upload failed");
        }
    }
    uint32_t max_rounds = 0;
    std::string err;
    int err_code = TAXOR_OK;
    struct DevKeys { uint64_t *p = nullptr; uint64_t n = 0; };
    // post-order: children first; returns the union of everything below IXF i (not needed for the root)
    std::function<bool(uint64_t, bool, DevKeys *)> build = [&](uint64_t i, bool want_union, DevKeys *out) -> bool {
        taxor_index_ixf_info(idx, i, &data, &stride, &seg_len, &bins, &device);
        const uint64_t nb = bins, g0 = bin_base[i];
        std::vector<DevKeys> child(nb);
        std::vector<uint64_t> off(nb + 1, 0);
        bool ok = true;
        for (uint64_t b = 0; b < nb && ok; ++b) {
            if (binfo[g0 + b] & BINFO_MERGED) ok = build(binfo[g0 + b] & 0x3FFFFFFFu, true, &child[b]);
            off[b + 1] = off[b] + ((binfo[g0 + b] & BINFO_MERGED) ? child[b].n : key_off[g0 + b + 1] - key_off[g0 + b]);
        }
        uint64_t *d_all = nullptr;
        if (ok && off[nb]) {
            if (hipMalloc((void **)&d_all, off[nb] * 8) != hipSuccess) { err = "build_hixf: no device memory for the keys of one IXF"; err_code = TAXOR_E_NOMEM; ok = false; }
            for (uint64_t b = 0; b < nb && ok;) {              // runs of leaf bins are contiguous in the caller's array
                if (binfo[g0 + b] & BINFO_MERGED) {
                    if (child[b].n && hipMemcpyAsync(d_all + off[b], child[b].p, child[b].n * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) ok = false;
                    ++b;
                } else {
                    uint64_t e = b;
                    while (e < nb && !(binfo[g0 + e] & BINFO_MERGED)) ++e;
                    const uint64_t n = off[e] - off[b];
                    if (n && hipMemcpyAsync(d_all + off[b], d_leaf + (key_off[g0 + b] - key_off[0]), n * 8, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) ok = false;
                    b = e;
                }
            }
            if (!ok && err.empty()) { err = "build_hixf: device copy failed"; err_code = TAXOR_E_HIP; }
            if (ok && hipDeviceSynchronize() != hipSuccess) { err = "build_hixf: device copy failed"; err_code = TAXOR_E_HIP; ok = false; }
        }
        for (auto &c : child)
            if (c.p) (void)hipFree(c.p);
        if (ok) {
            uint64_t seed = 0;
            uint32_t rounds = 0;
            const int rc = build_ixf_device(idx, i, d_all, off.data(), seed0 + 0x9E3779B97F4A7C15ull * i, &seed, &rounds);
            if (rc != TAXOR_OK) { err_code = rc; ok = false; }     // message already set
            max_rounds = std::max(max_rounds, rounds);
        }
        if (ok && want_union && off[nb]) {
            const hipError_t e = sort_unique_u64(d_all, off[nb], &out->p, &out->n, nullptr);
            if (e != hipSuccess) { err = std::string("build_hixf: key union failed: ") + hipGetErrorString(e); err_code = TAXOR_E_HIP; ok = false; }
        }
        if (d_all) (void)hipFree(d_all);
        return ok;
    };
    const bool ok = build(0, false, nullptr);
    if (d_leaf) (void)hipFree(d_leaf);
    if (rounds_out) *rounds_out = max_rounds;
    if (!ok) return err.empty() ? err_code : bfail(err_code, err);
    return TAXOR_OK;
}
