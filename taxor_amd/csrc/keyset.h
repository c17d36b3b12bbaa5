// keyset.h -- sorted, duplicate-free copy of a device array of 64-bit keys (hierarchical build, see keyset.hip)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace taxor {
// *d_out: newly hipMalloc'ed array of *n_out keys (caller frees); n must be < 2^32
hipError_t sort_unique_u64(const uint64_t *d_in, uint64_t n, uint64_t **d_out, uint64_t *n_out, hipStream_t st);
}
