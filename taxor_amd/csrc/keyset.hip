// keyset.hip -- device-side union of key sets for the hierarchical build: a merged bin of a
// parent IXF holds every key of its child's subtree (src/hixf/build/hierarchical_build.cpp:27-236 inserts the child's
// k-mers into the parent's merged bin).  Union = sort + unique of the concatenated key lists; the sort and the
// compaction are rocPRIM device primitives (this is index construction, not the search path).
#include "keyset.h"

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>

namespace taxor {

hipError_t sort_unique_u64(const uint64_t *d_in, uint64_t n, uint64_t **d_out, uint64_t *n_out, hipStream_t st)
{
    *d_out = nullptr;
    *n_out = 0;
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 32)) return hipErrorInvalidValue;      // rocprim::unique counts in 32 bits
    uint64_t *sorted = nullptr, *uniq = nullptr;
    size_t *d_count = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0, need = 0;
    hipError_t e = hipMalloc((void **)&sorted, n * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&uniq, n * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&d_count, sizeof(size_t));
    if (e == hipSuccess) e = rocprim::radix_sort_keys(nullptr, tmp_bytes, d_in, sorted, (size_t)n, 0, 64, st);
    if (e == hipSuccess) e = rocprim::unique(nullptr, need, sorted, uniq, d_count, (size_t)n, rocprim::equal_to<uint64_t>(), st);
    if (e == hipSuccess) {
        tmp_bytes = std::max(tmp_bytes, need);
        e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 8);
    }
    size_t sz = tmp_bytes;
    if (e == hipSuccess) e = rocprim::radix_sort_keys(tmp, sz, d_in, sorted, (size_t)n, 0, 64, st);
    sz = tmp_bytes;
    if (e == hipSuccess) e = rocprim::unique(tmp, sz, sorted, uniq, d_count, (size_t)n, rocprim::equal_to<uint64_t>(), st);
    size_t cnt = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&cnt, d_count, sizeof(size_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (tmp) (void)hipFree(tmp);
    if (d_count) (void)hipFree(d_count);
    if (sorted) (void)hipFree(sorted);
    if (e != hipSuccess) {
        if (uniq) (void)hipFree(uniq);
        return e;
    }
    *d_out = uniq;
    *n_out = cnt;
    return hipSuccess;
}

} // namespace taxor
