// pgz_inflate.cpp -- the parallel single-member gzip reader (taxor_amd/csrc/pgz.h) under AddressSanitizer / UBSan / ThreadSanitizer:
//   pgz_inflate <file.gz> <threads> <chunk bytes> [out]   decompress, print "bytes <n> crc <hex>" or "error: <what>"
#include "pgz.h"

#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    fastx::ParallelGz g;
    try {
        if (!g.open(argv[1], (unsigned)atoi(argv[2]), (size_t)strtoull(argv[3], nullptr, 10), 0)) { printf("error: not gzip\n"); return 0; }
        std::vector<char> buf(1 << 20);
        uint64_t total = 0;
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        FILE *of = argc > 4 ? fopen(argv[4], "wb") : nullptr;
        for (;;) {
            const size_t n = g.read(buf.data(), buf.size());
            if (!n) break;
            total += n;
            crc = (uint32_t)crc32(crc, (const Bytef *)buf.data(), (uInt)n);
            if (of) fwrite(buf.data(), 1, n, of);
        }
        if (of) fclose(of);
        printf("bytes %llu crc %08x chunks %llu redecoded %llu members %llu\n", (unsigned long long)total, crc, (unsigned long long)g.chunks_total,
               (unsigned long long)g.chunks_redecoded, (unsigned long long)g.members);
    } catch (const std::exception &e) { printf("error: %s\n", e.what()); }
    return 0;
}
