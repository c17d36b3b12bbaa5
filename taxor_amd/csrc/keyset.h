// keyset.h -- sorted, duplicate-free copy of a device array of 64-bit keys (hierarchical build, see keyset.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace taxor {

// Scratch of the union step, kept from one IXF to the next (a hipMalloc / hipFree pair per IXF costs more than sorting a small one).
struct SortScratch {
    uint64_t *sorted = nullptr;
    size_t *d_count = nullptr;
    size_t *h_count = nullptr;     // page-locked
    void *tmp = nullptr;
    uint64_t cap = 0;
    size_t tmp_bytes = 0;
    ~SortScratch() { release(); }
    void release();
    // d_out (room for n keys) receives the sorted distinct keys of d_in[0, n), *n_out their number; n < 2^32.  Waits for `st`.
    hipError_t sort_unique(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *n_out, hipStream_t st);
};

}
