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

void SortScratch::release()
{
    if (sorted) (void)hipFree(sorted);
    if (d_count) (void)hipFree(d_count);
    if (tmp) (void)hipFree(tmp);
    if (h_count) (void)hipHostFree(h_count);
    sorted = nullptr;
    d_count = nullptr;
    tmp = nullptr;
    h_count = nullptr;
    cap = 0;
    tmp_bytes = 0;
}

hipError_t SortScratch::sort_unique(const uint64_t *d_in, uint64_t n, uint64_t *d_out, uint64_t *n_out, hipStream_t st)
{
    *n_out = 0;
    if (n == 0) return hipSuccess;
    if (n >= (1ull << 32)) return hipErrorInvalidValue;      // rocprim::unique counts in 32 bits
    hipError_t e = hipSuccess;
    if (!d_count) e = hipMalloc((void **)&d_count, sizeof(size_t));
    if (e == hipSuccess && !h_count) e = hipHostMalloc((void **)&h_count, sizeof(size_t), hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (n > cap) {
        if (sorted) (void)hipFree(sorted);
        sorted = nullptr;
        cap = 0;
        const uint64_t want = n + n / 4;                     // (the next IXF is often a little larger)
        e = hipMalloc((void **)&sorted, want * sizeof(uint64_t));
        if (e != hipSuccess) return e;
        cap = want;
    }
    size_t need_sort = 0, need_uniq = 0;
    e = rocprim::radix_sort_keys(nullptr, need_sort, d_in, sorted, (size_t)n, 0, 64, st);
    if (e == hipSuccess) e = rocprim::unique(nullptr, need_uniq, sorted, d_out, d_count, (size_t)n, rocprim::equal_to<uint64_t>(), st);
    if (e != hipSuccess) return e;
    const size_t need = std::max<size_t>(std::max(need_sort, need_uniq), 8);
    if (need > tmp_bytes) {
        if (tmp) (void)hipFree(tmp);
        tmp = nullptr;
        tmp_bytes = 0;
        e = hipMalloc(&tmp, need + need / 4);
        if (e != hipSuccess) return e;
        tmp_bytes = need + need / 4;
    }
    size_t sz = tmp_bytes;
    e = rocprim::radix_sort_keys(tmp, sz, d_in, sorted, (size_t)n, 0, 64, st);
    sz = tmp_bytes;
    if (e == hipSuccess) e = rocprim::unique(tmp, sz, sorted, d_out, d_count, (size_t)n, rocprim::equal_to<uint64_t>(), st);
    if (e == hipSuccess) e = hipMemcpyAsync(h_count, d_count, sizeof(size_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    *n_out = *h_count;
    return hipSuccess;
}

} // namespace taxor
