// fastx.h -- FASTA / FASTQ record readers feeding `taxor search` (the reference reads its queries with
// seqan3::sequence_file_input, src/main/taxor_search.cpp:181-184,315-321: ids are the full header line, sequences
// are dna4).  Two sources produce the same numbered batches of records:
//   * FastxReader       : sequential, over zlib (plain or .gz, any stream); a multi-member gzip file is inflated
//                         by several threads underneath (gzmembers.h) and parsed by one;
//   * RangedFastx       : a plain file cut into byte ranges that start at record boundaries; each range is read
//                         (pread into a small per-thread buffer: no page-table traffic) and parsed by its own
//                         FastxReader, as many at a time as --threads allows.
// Errors are thrown as std::runtime_error.
#pragma once

#include "gzmembers.h"
#include "pgz.h"

#include <zlib.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <emmintrin.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <memory>
#include <string>
#include <vector>

namespace fastx {

// one chunk of records on its way through the pipeline: reader -> GPU -> formatter/writer
// the record ids of a chunk, back to back in one string: a chunk of short reads holds 10^5 of them, and a string object
// each is a heap allocation per record as soon as the id is longer than 15 characters (every real sequencer's is)
struct IdList {
    std::string data;
    std::vector<uint64_t> off{0};
    size_t size() const { return off.size() - 1; }
    bool empty() const { return off.size() == 1; }
    void clear() { data.clear(); off.assign(1, 0); }
    void reserve(size_t n, size_t bytes) { off.reserve(n + 1); data.reserve(bytes); }
    void push_back(const std::string &id) { data.append(id); off.push_back(data.size()); }
    const char *ptr(size_t i) const { return data.data() + off[i]; }
    size_t len(size_t i) const { return (size_t)(off[i + 1] - off[i]); }
    std::string str(size_t i) const { return std::string(ptr(i), len(i)); }
};

// The sequence bytes of a chunk: filled once by a parser, read once by the DMA engine -- never by this process again.  Unlike a
// std::string it (i) grows without zeroing what it adds, (ii) sits on 2-MiB boundaries and asks for huge pages, and (iii) copies long
// lines with non-temporal stores: a cached store first READS the line it is about to overwrite (three units of memory traffic per
// byte parsed instead of two) and evicts what the other threads of the core complex keep in their cache.  RefSeq-class CLI runs on
// three boxes (profiles/r04/cli_variants.txt, cli_variants2.txt, cli_ab_prev.txt): 0.81-0.85 x the library's rate with (seven runs,
// mean 0.83), 0.79-0.85 without (six runs, mean 0.82; TAXOR_CLI_NT=0 under TAXOR_TUNING), the round's earlier build with its
// std::string 0.73-0.77 beside them.  The subset of the string interface the readers and the CLI use.
inline bool &stream_stores()
{
    static bool on = true;
    return on;
}
// bytes a range reader asks pread() for at a time (its staging buffer between the page cache and the chunk)
inline size_t &range_buffer_bytes()
{
    static size_t n = 4u << 20;
    return n;
}
struct BaseBuf {
    char *p = nullptr;
    size_t n = 0, cap = 0;
    BaseBuf() = default;
    BaseBuf(const BaseBuf &) = delete;
    BaseBuf &operator=(const BaseBuf &) = delete;
    ~BaseBuf() { free(p); }
    const char *data() const { return p; }
    char *data() { return p; }
    size_t size() const { return n; }
    size_t capacity() const { return cap; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    char &operator[](size_t i) { return p[i]; }
    char &back() { return p[n - 1]; }
    void pop_back() { --n; }
    void resize(size_t m) { if (m > cap) reserve(m); n = m; }      // added bytes are NOT initialised
    void reserve(size_t want)
    {
        if (want <= cap) return;
        const size_t a = 2u << 20, c = (want + a - 1) & ~(a - 1);
        void *q = nullptr;
        if (posix_memalign(&q, a, c) != 0 || !q) throw std::bad_alloc();
        (void)madvise(q, c, MADV_HUGEPAGE);
        if (n) memcpy(q, p, n);
        free(p);
        p = (char *)q;
        cap = c;
    }
    void append(const char *s, size_t len)
    {
        if (n + len > cap) reserve(std::max(n + len, cap + cap / 2));
        char *d = p + n;
        n += len;
#if defined(__SSE2__)
        if (len >= 1024 && stream_stores()) {
            const size_t head = (64 - ((uintptr_t)d & 63)) & 63;
            memcpy(d, s, head);
            d += head; s += head; len -= head;
            for (; len >= 64; len -= 64, d += 64, s += 64) {
                const __m128i a = _mm_loadu_si128((const __m128i *)s), b = _mm_loadu_si128((const __m128i *)(s + 16));
                const __m128i c = _mm_loadu_si128((const __m128i *)(s + 32)), e = _mm_loadu_si128((const __m128i *)(s + 48));
                _mm_stream_si128((__m128i *)d, a);
                _mm_stream_si128((__m128i *)(d + 16), b);
                _mm_stream_si128((__m128i *)(d + 32), c);
                _mm_stream_si128((__m128i *)(d + 48), e);
            }
            _mm_sfence();
        }
#endif
        memcpy(d, s, len);
    }
};

struct Batch {
    uint64_t seq = 0;   // position of this chunk in its query file
    uint32_t file = 0;  // position of the query file in --query-file
    bool end_of_file = false;   // marker that follows the last chunk of a file (seq = number of chunks), carries no records
    IdList ids;
    BaseBuf bases;
    std::vector<uint64_t> offsets;
    // results copied out of the searcher (its buffers are reused by the next batch)
    std::vector<uint64_t> read_off;
    std::vector<int64_t> user_bin;
    std::vector<uint32_t> count, n_hashes;
    std::string text;   // the chunk's TSV lines (formatter threads), written by the writer in chunk order
    // bases.data() as pinned for DMA by the consumer (search_main.cpp); the producer keeps bases from reallocating
    void *pinned = nullptr;
    bool may_pin = false;
    uint64_t pool_bytes = 0;    // what the CLI's buffer pool has on its books for this batch (search_main.cpp, BatchPool)
    uint32_t fills = 0;         // how often this buffer has been filled by a parser: it is page-locked when it comes round again
};

// ---- bzip2 input (seqan3's sequence_file_input reads .bz2 when built with bzip2, which the reference's CMake fetches).
//      The image carries libbz2's shared object but not its header, so the three entry points of its stable stdio-like
//      interface are bound at run time; a host without the library gets a clear error instead of a silent mis-parse.
struct Bz2Api {
    // the library's low-level reading interface: a .bz2 may hold SEVERAL streams back to back (pbzip2 output,
    // `cat a.bz2 b.bz2`), and the stdio-like BZ2_bzread stops for good after the first one
    void *(*read_open)(int *, FILE *, int, int, void *, int) = nullptr;
    int (*read)(int *, void *, void *, int) = nullptr;
    void (*get_unused)(int *, void *, void **, int *) = nullptr;
    void (*read_close)(int *, void *) = nullptr;
    static const Bz2Api &get()
    {
        static const Bz2Api api = [] {
            Bz2Api a;
            void *h = nullptr;
            for (const char *name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"})
                if ((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (h) {
                a.read_open = reinterpret_cast<decltype(a.read_open)>(dlsym(h, "BZ2_bzReadOpen"));
                a.read = reinterpret_cast<decltype(a.read)>(dlsym(h, "BZ2_bzRead"));
                a.get_unused = reinterpret_cast<decltype(a.get_unused)>(dlsym(h, "BZ2_bzReadGetUnused"));
                a.read_close = reinterpret_cast<decltype(a.read_close)>(dlsym(h, "BZ2_bzReadClose"));
            }
            return a;
        }();
        return api;
    }
    bool ok() const { return read_open && read && get_unused && read_close; }
};

// all streams of a .bz2 file, one after the other (the loop of the bzip2 manual, "Reading a multi-stream file")
struct Bz2Reader {
    FILE *fp = nullptr;
    void *bz = nullptr;
    char unused[5000];       // BZ_MAX_UNUSED
    int n_unused = 0;
    bool done = false;
    bool open(const char *path)
    {
        fp = fopen(path, "rb");
        return fp && next_stream();
    }
    bool next_stream()
    {
        int err = 0;
        bz = Bz2Api::get().read_open(&err, fp, 0, 0, n_unused ? unused : nullptr, n_unused);
        if (err != 0 || !bz) { bz = nullptr; return false; }
        return true;
    }
    // bytes read, 0 at the end of the last stream, -1 on a corrupt stream
    long read(char *dst, int n)
    {
        const Bz2Api &api = Bz2Api::get();
        long got = 0;
        while (got < n && !done) {
            int err = 0;
            const int r = api.read(&err, bz, dst + got, n - (int)got);
            if (err != 0 && err != 4) return -1;            // neither BZ_OK nor BZ_STREAM_END
            got += r > 0 ? r : 0;
            if (err == 4) {                                  // end of one stream: is another one behind it?
                void *u = nullptr;
                int nu = 0, e2 = 0;
                api.get_unused(&e2, bz, &u, &nu);
                if (e2 != 0 || nu < 0 || nu > (int)sizeof unused) return -1;
                if (nu) memmove(unused, u, (size_t)nu);
                n_unused = nu;
                api.read_close(&e2, bz);
                bz = nullptr;
                if (n_unused == 0) {
                    const int c = fgetc(fp);
                    if (c == EOF) { done = true; break; }
                    ungetc(c, fp);
                }
                if (!next_stream()) return -1;
            }
        }
        return got;
    }
    ~Bz2Reader()
    {
        int e = 0;
        if (bz) Bz2Api::get().read_close(&e, bz);
        if (fp) fclose(fp);
    }
};

// ---- sequential reader over zlib.  Lines are located with memchr inside a large refill buffer; sequence lines are
//      appended straight into the batch and quality lines are skipped without being copied.
struct FastxReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool eof = false;
    std::string pending; // header line read ahead (FASTA)
    int fd = -1;         // range mode: bytes [rpos, rend) of a plain file, read with pread
    uint64_t rpos = 0, rend = 0;
    GzMembers *members = nullptr;   // multi-member gzip inflated in parallel (gzmembers.h)
    ParallelGz *pgz = nullptr;      // ONE gzip member inflated in parallel (pgz.h)
    std::deque<std::vector<char>> mem;   // memory mode: the input is these buffers, one after the other (open_mem)
    bool mem_mode = false;
    std::function<void(std::vector<char> &&)> mem_recycle;   // memory mode: a buffer that has been parsed goes back to its maker
    std::unique_ptr<Bz2Reader> bz;  // bzip2 input, all of its streams

    bool strict4 = false;           // range mode on FASTQ: records must be the four-line kind the ranges were cut for

    void open_range(int fd_, uint64_t b, uint64_t e, bool fastq_ranges = false)
    {
        strict4 = fastq_ranges;
        fd = fd_;
        rpos = b;
        rend = e;
        pos = len = 0;
        eof = false;
        pending.clear();
        if (buf.size() != range_buffer_bytes()) buf.resize(range_buffer_bytes());
    }
    // the input is a list of buffers in memory (whole records from the first byte to the last; the parallel gzip reader's chunks)
    void open_mem(std::deque<std::vector<char>> &&parts, bool fastq_ranges)
    {
        strict4 = fastq_ranges;
        mem = std::move(parts);
        mem_mode = true;
        pos = len = 0;
        eof = false;
        pending.clear();
    }
    bool open(const std::string &path)
    {
        buf.resize(8u << 20);
        unsigned char magic[3] = {0, 0, 0};
        if (FILE *probe = fopen(path.c_str(), "rb")) {
            const size_t got = fread(magic, 1, 3, probe);
            fclose(probe);
            if (got == 3 && magic[0] == 'B' && magic[1] == 'Z' && magic[2] == 'h') {
                const Bz2Api &api = Bz2Api::get();
                if (!api.ok()) throw std::runtime_error("query file " + path + " is bzip2-compressed and libbz2 is not available on this host");
                bz.reset(new Bz2Reader());
                if (!bz->open(path.c_str())) { bz.reset(); return false; }
                return true;
            }
        }
        f = gzopen(path.c_str(), "rb");
        if (f) gzbuffer(f, 1 << 20);
        return f != nullptr;
    }
    ~FastxReader()
    {
        if (f) gzclose(f);
    }
    bool refill()
    {
        if (eof) return false;
        if (mem_mode) {
            while (!mem.empty() && mem.front().empty()) mem.pop_front();
            if (mem.empty()) { eof = true; return false; }
            buf.swap(mem.front());
            if (mem_recycle && mem.front().capacity()) mem_recycle(std::move(mem.front()));
            mem.pop_front();
            pos = 0;
            len = buf.size();
            return true;
        }
        long n;
        if (members) {
            n = (long)members->read(buf.data(), buf.size());
        } else if (pgz) {
            n = (long)pgz->read(buf.data(), buf.size());
        } else if (fd >= 0) {
            const uint64_t want = std::min<uint64_t>(buf.size(), rend - rpos);
            n = want ? (long)pread(fd, buf.data(), want, (off_t)rpos) : 0;
            if (n > 0) rpos += (uint64_t)n;
        } else if (bz) {
            n = bz->read(buf.data(), (int)std::min<size_t>(buf.size(), 1u << 30));
            if (n < 0) throw std::runtime_error("bzip2 stream is corrupt");
        } else {
            n = gzread(f, buf.data(), (unsigned)buf.size());
        }
        if (n <= 0) { eof = true; return false; }
        pos = 0;
        len = (size_t)n;
        return true;
    }
    // next line -> appended to `out` (if non-null); returns false at end of input with nothing read.  *count (if
    // non-null) receives the line's length without its line terminator.
    bool line_to(std::nullptr_t, size_t *count = nullptr) { return line_to((std::string *)nullptr, count); }
    template <class Out> bool line_to(Out *out, size_t *count = nullptr)
    {
        bool any = false;
        size_t total = 0;
        char last = 0;
        for (;;) {
            if (pos == len && !refill()) {
                if (any && out && !out->empty() && out->back() == '\r') out->pop_back();
                if (count) *count = total - (last == '\r' ? 1 : 0);
                return any;
            }
            const char *s = buf.data() + pos;
            const char *e = (const char *)memchr(s, '\n', len - pos);
            const size_t n = e ? (size_t)(e - s) : len - pos;
            if (out) out->append(s, n);
            if (n) last = s[n - 1];
            total += n;
            any = any || n || e;
            pos += n + (e ? 1 : 0);
            if (e) {
                if (out && !out->empty() && out->back() == '\r') out->pop_back();
                if (count) *count = total - (last == '\r' ? 1 : 0);
                return true;
            }
        }
    }
    // step over n bytes of input without looking at them; false if the input ends first
    bool skip_bytes(size_t n)
    {
        while (n) {
            if (pos == len && !refill()) return false;
            const size_t take = std::min(n, len - pos);
            pos += take;
            n -= take;
        }
        return true;
    }
    bool getline(std::string &line)
    {
        line.clear();
        return line_to(&line);
    }
    // appends the record's sequence to `bases`; returns false at end of file
    std::string line;    // the header line of the record being read (a member: its buffer is reused from record to record)
    template <class Bases> bool next(std::string &id, Bases &bases)
    {
        if (pending.empty()) {
            do {
                if (!getline(line)) return false;
            } while (line.empty());
        } else {
            line.swap(pending);
            pending.clear();
        }
        if (line[0] == '>') {
            id.assign(line, 1, std::string::npos);
            for (;;) {
                // peek the first character of the next line
                if (pos == len && !refill()) break;
                if (buf[pos] == '>') { getline(pending); break; }
                line_to(&bases);
            }
            return true;
        }
        if (line[0] == '@') {
            id.assign(line, 1, std::string::npos);
            // like seqan3's format_fastq: the sequence runs until the line that starts with '+' (usually one line),
            // the quality string is as long as the sequence (so a quality line may start with '@' or '+')
            const size_t seq_begin = bases.size();
            if (!line_to(&bases)) throw std::runtime_error("truncated FASTQ record: " + id);
            for (;;) {
                if (pos == len && !refill()) throw std::runtime_error("truncated FASTQ record: " + id);
                if (buf[pos] == '+') break;
                // A byte range of the parallel reader was cut on the assumption of four-line records (resync below; the
                // file's first records were): a record that wraps its sequence means the cuts of this file cannot be
                // trusted -- stop, do not mis-parse
                if (strict4) throw std::runtime_error("FASTQ record '" + id + "' wraps its sequence over several lines; the parallel range reader "
                                                      "needs four-line records -- rerun with --sequential");
                line_to(&bases);
            }
            line_to(nullptr);                                                                    // the '+' line
            const size_t want = bases.size() - seq_begin;
            size_t got = 0, n = 0;
            if (strict4 && want) {
                // four-line records (range reader): the quality line is exactly as long as the sequence, so it is stepped
                // over without looking at its bytes -- only its end is checked (half of a FASTQ file is quality characters)
                if (!skip_bytes(want)) throw std::runtime_error("truncated FASTQ record: " + id);
                got = want;
                if (pos == len && !refill()) return true;                                        // file ends without a final newline
                if (buf[pos] == '\r') { ++pos; if (pos == len && !refill()) return true; }
                if (buf[pos] != '\n')     // a shorter quality line puts this position inside the next record, a longer one inside itself
                    throw std::runtime_error("malformed FASTQ record '" + id + "': quality line and sequence (" + std::to_string(want) +
                                             " bases) differ in length (parallel range reader: rerun with --sequential if the file is valid)");
                ++pos;
            }
            while (got < want) {                                                                 // quality: skipped, never copied
                if (!line_to(nullptr, &n)) throw std::runtime_error("truncated FASTQ record: " + id);
                got += n;
                if (strict4 && got != want) break;
            }
            if (want == 0) line_to(nullptr, &n);                                                 // an empty read still has its (empty) quality line
            // the quality string is exactly as long as the sequence and the next record (if any) starts with '@': anything
            // else means the record boundaries are not where this parser thinks they are
            if (got != want) throw std::runtime_error("malformed FASTQ record '" + id + "': " + std::to_string(want) + " bases, " + std::to_string(got) +
                                                      " quality characters" + (strict4 ? " (parallel range reader: rerun with --sequential if the file is valid)" : ""));
            for (;;) {
                if (pos == len && !refill()) break;                                              // end of input
                if (buf[pos] == '\n' || buf[pos] == '\r') { ++pos; continue; }                   // blank lines between records
                if (buf[pos] != '@') throw std::runtime_error("malformed FASTQ: the line after record '" + id + "' does not start with '@'");
                break;
            }
            return true;
        }
        throw std::runtime_error("query file is neither FASTA nor FASTQ");
    }
};

// ---- plain file in byte ranges ---------------------------------------------------------------------------------------
inline const char *next_line(const char *p, const char *e)
{
    const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
    return q ? q + 1 : e;
}

// first record start at or after line start `p`, or `e` if the window [p, e) shows none.  FASTA: a line beginning
// with '>'.  Four-line FASTQ: a line beginning with '@' whose second successor begins with '+' -- a quality line may
// begin with '@' too, but then its second successor is a sequence line, which cannot begin with '+'.
inline const char *resync(const char *p, const char *e, char kind)
{
    while (p < e) {
        if (kind == '>') {
            if (*p == '>') return p;
        } else if (*p == '@') {
            const char *l2 = next_line(next_line(p, e), e);
            if (l2 < e && *l2 == '+') return p;
        }
        p = next_line(p, e);
    }
    return e;
}

// upper bound of the sequence bytes in `len` bytes of file
inline size_t bases_bound(uint64_t len, char kind) { return kind == '@' ? (size_t)len / 2 + 64 : (size_t)len; }

struct RangedFastx {
    int fd = -1;
    uint64_t size = 0, cur = 0;
    char kind = 0;              // '>' FASTA, '@' FASTQ
    uint64_t range_bytes = 0;   // target size of one range
    uint64_t seq = 0;
    std::mutex mu;
    std::vector<char> win;

    ~RangedFastx() { if (fd >= 0) ::close(fd); }

    bool read_at(uint64_t off, size_t n)
    {
        win.resize(n);
        size_t got = 0;
        while (got < n) {
            const ssize_t r = pread(fd, win.data() + got, n - got, (off_t)(off + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        return true;
    }
    // false: not a regular plain-text file (gzip, pipe, empty) -- use the sequential FastxReader
    bool open(const std::string &path)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 2) return false;
        size = (uint64_t)sb.st_size;
        const size_t n = (size_t)std::min<uint64_t>(size, 1u << 16);
        if (!read_at(0, n)) return false;
        if ((unsigned char)win[0] == 0x1f && (unsigned char)win[1] == 0x8b) return false;      // gzip
        if (n >= 3 && win[0] == 'B' && win[1] == 'Z' && win[2] == 'h') return false;           // bzip2
        size_t i = 0;
        while (i < n && (win[i] == '\n' || win[i] == '\r')) ++i;
        if (i == n) return false;
        cur = i;
        kind = win[i];
        if (kind != '>' && kind != '@') throw std::runtime_error("query file is neither FASTA nor FASTQ");
        if (kind == '@') {
            // the range cutter (resync below) relies on four-line records; a file that wraps its sequences over several
            // lines (legal FASTQ, seqan3 reads it) goes through the sequential reader instead
            const char *p = win.data() + i, *e = win.data() + n;
            for (int rec = 0; rec < 16 && p < e; ++rec) {
                const char *l3 = next_line(next_line(p, e), e);
                if (l3 >= e) break;
                if (*l3 != '+') return false;
                const char *nx = next_line(next_line(l3, e), e);
                if (nx < e && *nx != '@' && *nx != '\n' && *nx != '\r') return false;
                p = nx;
            }
        }
        return true;
    }
    // file offset of the first record that starts after the line containing offset `from`
    uint64_t record_start_after(uint64_t from)
    {
        for (size_t w = 1u << 20;; w *= 4) {
            const size_t n = (size_t)std::min<uint64_t>(w, size - from);
            if (!read_at(from, n)) throw std::runtime_error("query file: read error");
            const char *b = win.data(), *e = b + n;
            const char *r = resync(next_line(b, e), e, kind);
            if (r < e) return from + (uint64_t)(r - b);
            if (from + n == size) return size;          // no further record: the rest belongs to the current range
        }
    }
    // size ranges to hold about `batch_reads` records (calibrated on the first records) and at most `max_bytes`
    void plan(uint64_t batch_reads, uint64_t max_bytes)
    {
        uint64_t p = cur;
        unsigned n = 0;
        while (n < 64 && p < size) {
            p = record_start_after(p);
            ++n;
        }
        const double per_rec = n ? (double)(p - cur) / n : 1.0;
        double want = per_rec * (double)batch_reads;
        if (want > (double)max_bytes) want = (double)max_bytes;
        range_bytes = want < 1.0 ? 1 : (uint64_t)want;
    }
    // next byte range [b, e) of whole records; thread-safe; false when the file is exhausted
    bool next_range(uint64_t &b, uint64_t &e, uint64_t &s)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (cur >= size) return false;
        b = cur;
        e = cur = (size - cur > range_bytes) ? record_start_after(cur + range_bytes) : size;
        s = seq++;
        return true;
    }
};

} // namespace fastx
