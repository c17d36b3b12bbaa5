// crc32_fold.cpp -- pgz.h's CRC-32 by carry-less multiplication against zlib's crc32() on random lengths, offsets and start values
// (built with and without sanitizers by tests/test_pgz_cpu.py).  Prints "bad 0" on success.
#include "pgz.h"

#include <cstdio>
#include <random>

int main()
{
    std::mt19937_64 rng(7);
    std::vector<uint8_t> buf((1u << 21) + 128);
    for (auto &b : buf) b = (uint8_t)rng();
    int bad = 0;
    for (int t = 0; t < 4000; ++t) {
        const size_t off = rng() % 64, n = t < 600 ? (size_t)t : rng() % (buf.size() - 64);
        const uint32_t init = t % 3 ? (uint32_t)rng() : 0;
        const uint32_t a = (uint32_t)crc32(init, buf.data() + off, (uInt)n), b = fastx::pgz_detail::crc32_bytes(init, buf.data() + off, n);
        if (a != b && bad++ < 5) printf("mismatch n=%zu off=%zu %08x %08x\n", n, off, a, b);
    }
    // continuation: the CRC of a buffer in pieces equals the CRC of the whole
    uint32_t whole = fastx::pgz_detail::crc32_bytes(0, buf.data(), buf.size()), parts = 0;
    for (size_t p = 0; p < buf.size();) {
        const size_t k = std::min<size_t>(buf.size() - p, 1 + rng() % 70000);
        parts = fastx::pgz_detail::crc32_bytes(parts, buf.data() + p, k);
        p += k;
    }
    if (whole != parts || whole != (uint32_t)crc32(0, buf.data(), (uInt)buf.size())) { ++bad; printf("pieces differ\n"); }
    printf("bad %d\n", bad);
    return bad != 0;
}
