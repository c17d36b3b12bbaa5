// ixf_arith.h -- the ONE place that defines the un-vendored arithmetic of the path, for host and device.
//
// Two pieces of the reference's hot path live in third-party code that is not in /root/reference:
//   * ankerl::unordered_dense::detail::wyhash::hash(uint64_t)  (martinus/unordered_dense v3.0.1;
//     call site src/hashing/syncmer.cpp:73-77)
//   * seqan3::interleaved_xor_filter<uint8_t>  (JensUweUlrich/seqan3@master; call sites
//     src/hixf/build/hierarchical_interleaved_xor_filter.hpp:96,307-309)
// They are restated here from the published wyhash mix and from the in-repo XOR-filter prototype
// (src/main/xorfilter.hpp:22-45,60-68,338-350, src/main/hashutil.hpp:50-61).  If a real .hixf shows the
// fork differs (hash, reduction, row stride), this header is the one-file change.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TAXOR_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TAXOR_HD inline
#endif

namespace taxor {

// wyhash::hash(x) = mix(x, 0x9E3779B97F4A7C15); mix = lo64 ^ hi64 of the 128-bit product
TAXOR_HD uint64_t wyhash_u64(uint64_t x)
{
    const uint64_t c = 0x9E3779B97F4A7C15ull;
#if defined(__HIP_DEVICE_COMPILE__)
    return (x * c) ^ __umul64hi(x, c);
#else
    __uint128_t r = (__uint128_t)x * c;
    return (uint64_t)r ^ (uint64_t)(r >> 64);
#endif
}

TAXOR_HD uint64_t murmur64(uint64_t h) // hashutil.hpp:50-57
{
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

TAXOR_HD uint64_t rotl64(uint64_t n, unsigned c) // xorfilter.hpp:22-28
{
    c &= 63u;
    return (n << c) | (n >> ((0u - c) & 63u));
}

struct ixf_probe {
    uint32_t row[3]; // absolute row index (segment offset included)
    uint32_t fp4;    // 8-bit fingerprint replicated into 4 bytes
};

// key -> three rows + fingerprint (hashutil.hpp:59-61, xorfilter.hpp:36-45,60-62,340-347)
TAXOR_HD ixf_probe ixf_probe_key(uint64_t key, uint64_t seed, uint32_t seg_len)
{
    const uint64_t h = murmur64(key + seed);
    ixf_probe p;
    p.fp4 = (uint32_t)((h ^ (h >> 32)) & 0xffu) * 0x01010101u;
    for (int i = 0; i < 3; ++i) {
        const uint32_t r = (uint32_t)rotl64(h, 21u * (unsigned)i);
        p.row[i] = (uint32_t)(((uint64_t)r * seg_len) >> 32) + (uint32_t)i * seg_len;
    }
    return p;
}

// rows per segment for a bin capacity of n keys: arrayLength = 32 + 1.23*n; blockLength = arrayLength/3
inline uint64_t ixf_seg_len(uint64_t max_bin_elements) // xorfilter.hpp:67-68
{
    const uint64_t array_len = (uint64_t)(32 + 1.23 * (double)max_bin_elements);
    return array_len / 3;
}

} // namespace taxor
