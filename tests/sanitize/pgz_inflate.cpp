// pgz_inflate.cpp -- the parallel single-member gzip reader (taxor_amd/csrc/pgz.h) under AddressSanitizer / UBSan / ThreadSanitizer:
//   pgz_inflate <file.gz> <threads> <chunk bytes> [out|-] [memory budget MB]   decompress, print "bytes <n> crc <hex> ... maxrss_kb <n> inflight_peak_kb <n>
//   largest_chunk_kb <n>" or "error: <what>"
#include "pgz.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    fastx::ParallelGz g;
    try {
        if (argc > 5) g.set_memory_budget((size_t)strtoull(argv[5], nullptr, 10) << 20);
        if (!g.open(argv[1], (unsigned)atoi(argv[2]), (size_t)strtoull(argv[3], nullptr, 10), 0)) { printf("error: not gzip\n"); return 0; }
        std::vector<char> buf(1 << 20);
        uint64_t total = 0;
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        FILE *of = argc > 4 && strcmp(argv[4], "-") != 0 ? fopen(argv[4], "wb") : nullptr;
        for (;;) {
            const size_t n = g.read(buf.data(), buf.size());
            if (!n) break;
            total += n;
            crc = (uint32_t)crc32(crc, (const Bytef *)buf.data(), (uInt)n);
            if (of) fwrite(buf.data(), 1, n, of);
        }
        if (of) fclose(of);
        long hwm = 0;                       // peak resident set of THIS program (ru_maxrss survives fork + exec: it would report the parent's)
        if (FILE *st = fopen("/proc/self/status", "r")) {
            char line[256];
            while (fgets(line, sizeof line, st))
                if (strncmp(line, "VmHWM:", 6) == 0) hwm = atol(line + 6);
            fclose(st);
        }
        printf("bytes %llu crc %08x chunks %llu redecoded %llu members %llu trailing %llu maxrss_kb %ld inflight_peak_kb %zu largest_chunk_kb %zu\n", (unsigned long long)total, crc,
               (unsigned long long)g.chunks_total, (unsigned long long)g.chunks_redecoded, (unsigned long long)g.members, (unsigned long long)g.trailing_garbage, hwm,
               fastx::ParallelGz::memory_high_water() >> 10, g.largest_chunk_bytes() >> 10);
    } catch (const std::exception &e) { printf("error: %s\n", e.what()); }
    return 0;
}
