// AddressSanitizer / UBSan driver for the sequential and the ranged FASTA/FASTQ(.gz/.bz2) readers (fastx.h):
// usage: fastx_read <file> [ranged]   -> "<records> records <bases> bases" or "error: <what>"
#include "fastx.h"
#include <cstdio>
int main(int argc, char **argv) {
    std::string id, bases; size_t n = 0;
    try {
        if (argc > 2) {
            fastx::RangedFastx rf;
            if (!rf.open(argv[1])) { puts("not rangeable"); return 0; }
            rf.range_bytes = 4096;
            fastx::FastxReader rd; uint64_t b, e, seq;
            while (rf.next_range(b, e, seq)) { rd.open_range(rf.fd, b, e, rf.kind == '@'); while (rd.next(id, bases)) ++n; }
        } else {
            fastx::FastxReader rd;
            if (!rd.open(argv[1])) { puts("cannot open"); return 0; }
            while (rd.next(id, bases)) ++n;
        }
        printf("%zu records %zu bases\n", n, bases.size());
    } catch (const std::exception &ex) { printf("error: %s\n", ex.what()); }
}
