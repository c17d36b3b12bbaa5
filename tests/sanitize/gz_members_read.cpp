// AddressSanitizer / ThreadSanitizer driver for the multi-member gzip reader (fastx.h / gzmembers.h):
// usage: gz_members_read <file.gz>   -> prints "open=<0|1>" and the record / base counts
#include "fastx.h"
#include <cstdio>
int main(int argc, char **argv) {
    fastx::GzMembers m;
    bool ok = m.open(argv[1], 4);
    printf("open=%d\n", ok);
    if (!ok) return 0;
    fastx::FastxReader rd; rd.members = &m; rd.buf.resize(1 << 20);
    std::string id, bases; size_t n = 0;
    while (rd.next(id, bases)) ++n;
    printf("%zu records %zu bases\n", n, bases.size());
}
