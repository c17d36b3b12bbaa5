// keyset.h -- duplicate-free copy of a device array of 64-bit keys (hierarchical build, see keyset.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace taxor {

// Scratch of the union step, kept from one IXF to the next (a hipMalloc / hipFree pair per IXF costs more than the union of a small one).
struct KeyUnion {
    uint64_t *table = nullptr;             // open-addressing set, 1.5 .. 3 slots per input key
    uint64_t table_entries = 0;
    unsigned long long *d_ctl = nullptr;   // [0] keys written, [1] the input held the empty marker itself
    unsigned long long *h_ctl = nullptr;   // page-locked
    ~KeyUnion() { release(); }
    void release();
    // d_out (room for n keys) receives the distinct keys of d_in[0, n) in no particular order, *n_out their number.  Waits for `st`.
    hipError_t unique(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *n_out, hipStream_t st);
    // nothing is copied: d_keep[i] = 1 for one occurrence of every distinct key of d_in[0, n), 0 for its duplicates; *n_kept = the ones.
    hipError_t mark(const uint64_t *d_in, uint64_t n, uint8_t *d_keep, uint64_t *n_kept, hipStream_t st);
private:
    hipError_t prepare(uint64_t n, uint64_t *entries, hipStream_t st);
public:
};

}
