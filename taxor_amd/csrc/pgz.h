// pgz.h -- parallel inflate of ONE gzip member (the commonest real input: reads.fastq.gz as the sequencer or `gzip` wrote it).
//
// The reference reads .gz through seqan3's stream layer, i.e. one zlib stream on one thread (taxor_search.cpp:181-184): about
// 0.55 GB/s of text, a twentieth of what the GPU path classifies.  A deflate stream has no index and every byte may refer to the
// 32 KiB before it, so it cannot simply be cut -- but it can be decoded SPECULATIVELY from the middle (the scheme of pugz and
// rapidgzip, restated here on a deflate decoder of this file's own; the image carries zlib only, which cannot start inside a
// stream):
//   1. the compressed bytes are cut into chunks; for every chunk but the first a worker looks for the first position at which a
//      dynamic-Huffman block header parses under strict checks (code-length code and both alphabets complete, end-of-block
//      present) and whose whole block decodes to text -- deflate blocks start at arbitrary BIT offsets, so every bit is tried;
//   2. from there the chunk is decoded with its 32 KiB history UNKNOWN: the output is 16 bits per symbol, a literal byte or a
//      marker "byte w of the window before this chunk"; back-references copy markers like anything else.  A chunk ends at the
//      first block boundary at or behind the next chunk's nominal start;
//   3. in order, the consumer checks that chunk i+1 started exactly where chunk i ended (a start that does not match -- a false
//      positive of the search, a stored or fixed block at the boundary, a chunk that failed -- is decoded again from the right
//      position, with nothing speculative left), resolves the last 32 KiB of chunk i against its window to get chunk i+1's
//      window, and hands the full resolution (markers -> bytes, CRC-32) back to the workers;
//   4. resolved chunks are delivered in order; at the end the member's CRC-32 and length (combined from the chunks') must equal
//      the trailer, as zlib checks them.  Further members, if any, are taken the same way one after the other.
// Nothing speculative is ever delivered: a chunk's bytes leave only after its start has been tied to its predecessor's end, so
// what comes out is the one decoding of the stream -- or an exception naming the corruption, as from zlib.
#pragma once

#include <zlib.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#include <smmintrin.h>
#include <wmmintrin.h>
#endif

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace fastx {

namespace pgz_detail {

constexpr uint32_t WIN = 32768;

struct BitIn {
    const uint8_t *base = nullptr, *p = nullptr, *end = nullptr;
    uint64_t bb = 0;
    int bc = 0;
    void seek(const uint8_t *b, const uint8_t *e, uint64_t bit)
    {
        base = b;
        end = e;
        p = b + (bit >> 3);
        bb = 0;
        bc = 0;
        refill();
        const int s = (int)(bit & 7);
        bb >>= s;
        bc -= s;
    }
    inline void refill()
    {
        if (p + 8 <= end) {
            uint64_t v;
            memcpy(&v, p, 8);
            bb |= v << bc;
            p += (63 - bc) >> 3;
            bc |= 56;
        } else {
            while (bc <= 56 && p < end) { bb |= (uint64_t)*p++ << bc; bc += 8; }
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(bb & ((1ull << n) - 1)); }
    inline void drop(int n) { bb >>= n; bc -= n; }
    uint64_t bitpos() const { return (uint64_t)(p - base) * 8 - (uint64_t)bc; }   // negative bc (ran past the end) shows up as pos > size
    bool overrun() const { return bc < 0; }
};

inline uint32_t rev_bits(uint32_t c, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1); c >>= 1; }
    return r;
}

// two-level decoding table of a canonical prefix code: entry = symbol << 16 | bits to drop; a primary entry with bit 15 set
// points at a sub-table (offset << 16 | 0x8000 | index bits)
template <int PB, int MAXSYM> struct Huff {
    uint32_t tab[1 << PB];
    std::vector<uint32_t> sub;                     // sub-tables of the prefixes that have codes longer than PB bits
    // returns 0 complete code, 1 incomplete but usable (a single code, or no code at all), -1 invalid (over-subscribed / incomplete)
    int build(const uint8_t *len, int n)
    {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) count[len[i]]++;
        if (count[0] == n) { for (auto &e : primary()) e = 0; sub.clear(); return 1; }
        int left = 1;
        for (int l = 1; l <= 15; ++l) {
            left <<= 1;
            left -= count[l];
            if (left < 0) return -1;
        }
        const int used = n - count[0];
        if (left > 0 && !(used == 1 && count[1] == 1)) return -1;                // incomplete: only the one-code case is legal (zlib: inflate_table)
        uint32_t next[16];
        {   // first canonical code of every length (RFC 1951, 3.2.2)
            uint32_t code = 0;
            for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)(l > 1 ? count[l - 1] : 0)) << 1; next[l] = code; }
        }
        for (auto &e : primary()) e = 0;
        sub.clear();
        // sub-tables: for every PB-bit prefix that has longer codes, one table indexed by the next (maxlen - PB) bits
        int sub_bits_of[1 << PB];
        for (int i = 0; i < (1 << PB); ++i) sub_bits_of[i] = 0;
        {   // first pass: longest code under each prefix
            uint32_t nx[16];
            memcpy(nx, next, sizeof nx);
            for (int s = 0; s < n; ++s) {
                const int l = len[s];
                if (l <= PB) { if (l) nx[l]++; continue; }
                const uint32_t r = rev_bits(nx[l]++, l);
                const int pre = (int)(r & ((1u << PB) - 1));
                sub_bits_of[pre] = std::max(sub_bits_of[pre], l - PB);
            }
        }
        uint32_t off = 0;
        uint32_t *P = primary().data();
        for (int i = 0; i < (1 << PB); ++i)
            if (sub_bits_of[i]) {
                P[i] = (off << 16) | 0x8000u | (uint32_t)sub_bits_of[i];
                off += 1u << sub_bits_of[i];
            }
        sub.assign(off, 0);
        for (int s = 0; s < n; ++s) {
            const int l = len[s];
            if (!l) continue;
            const uint32_t r = rev_bits(next[l]++, l);
            if (l <= PB) {
                for (uint32_t i = r; i < (1u << PB); i += 1u << l) P[i] = ((uint32_t)s << 16) | (uint32_t)l;
            } else {
                const uint32_t pre = r & ((1u << PB) - 1), e = P[pre];
                const int sb = (int)(e & 0xFF);
                const uint32_t o = e >> 16, hi = r >> PB;
                for (uint32_t i = hi; i < (1u << sb); i += 1u << (l - PB)) sub[o + i] = ((uint32_t)s << 16) | (uint32_t)l;
            }
        }
        return left > 0 ? 1 : 0;
    }
    struct Span { uint32_t *b, *e; uint32_t *data() { return b; } uint32_t *begin() { return b; } uint32_t *end() { return e; } };
    Span primary() { return Span{tab, tab + (1 << PB)}; }
    // every entry's symbol field rewritten by f (the decoder wants a length code's base and extra bits, not its number, in the entry:
    // two dependent table reads less per match)
    template <class F> void remap(F f)
    {
        for (uint32_t &e : tab)
            if (e && !(e & 0x8000u)) e = (f(e >> 16) << 16) | (e & 0xFFFFu);
        for (uint32_t &e : sub)
            if (e) e = (f(e >> 16) << 16) | (e & 0xFFFFu);
    }
    // entry for the bits at the bottom of `bb` (0 = no such code)
    inline uint32_t lookup(uint64_t bb) const
    {
        uint32_t e = tab[bb & ((1u << PB) - 1)];
        if (e & 0x8000u) e = sub[(e >> 16) + (uint32_t)((bb >> PB) & ((1u << (e & 0xFF)) - 1))];
        return e;
    }
};

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Codes {
    Huff<10, 288> lit;      // symbol field: a literal (< 256), 256 = end of block, LEN_FLAG | extra bits << 9 | base for a length code, BAD_SYM for 286 / 287
    Huff<8, 32> dist;       // symbol field: extra bits << 12 ... see pack_dist: base in the high bits; 0 = codes 30 / 31
    bool have_dist = true;
    void pack();
};
constexpr uint32_t LEN_FLAG = 0x8000u, BAD_SYM = 0x4000u;
inline void Codes::pack()
{
    lit.remap([](uint32_t s) -> uint32_t {
        if (s <= 256) return s;
        if (s >= 286) return BAD_SYM;
        return LEN_FLAG | ((uint32_t)LEXT[s - 257] << 9) | LBASE[s - 257];
    });
    // distance: base 1 .. 24577 does not fit beside four bits of extra count in 16 bits; the entry's bits 8..14 are free (a code
    // length is at most 15): extra count there, the base in the symbol field
    for (int pass = 0; pass < 2; ++pass) {
        auto fix = [](uint32_t &e) {
            if (!e || (e & 0x8000u)) return;
            const uint32_t s = e >> 16, l = e & 0xFFu;
            e = s >= 30 ? l : (((uint32_t)DBASE[s]) << 16) | ((uint32_t)DEXT[s] << 8) | l;
        };
        if (pass == 0) for (uint32_t &e : dist.tab) fix(e);
        else for (uint32_t &e : dist.sub) fix(e);
    }
}

// the header of a dynamic block at the reader's position; strict = what a block START SEARCH demands beyond validity (complete
// alphabets); false on anything a decoder must reject
inline bool read_dynamic_header(BitIn &in, Codes &c, bool strict)
{
    in.refill();
    const uint32_t hlit = in.peek(5) + 257;
    in.drop(5);
    const uint32_t hdist = in.peek(5) + 1;
    in.drop(5);
    const uint32_t hclen = in.peek(4) + 4;
    in.drop(4);
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    in.refill();
    for (uint32_t i = 0; i < hclen; ++i) {
        if (in.bc < 3) in.refill();
        cl[CLORD[i]] = (uint8_t)in.peek(3);
        in.drop(3);
    }
    Huff<7, 19> clh;
    const int clr = clh.build(cl, 19);
    if (clr < 0 || (strict && clr != 0)) return false;
    uint8_t lens[286 + 30 + 140];
    uint32_t n = 0;
    const uint32_t total = hlit + hdist;
    while (n < total) {
        in.refill();
        if (in.overrun()) return false;
        const uint32_t e = clh.lookup(in.bb);
        const int l = (int)(e & 0xFF);
        if (!l) return false;
        in.drop(l);
        const uint32_t sym = e >> 16;
        if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
        uint32_t rep, val = 0;
        if (sym == 16) {
            if (n == 0) return false;
            val = lens[n - 1];
            rep = 3 + in.peek(2);
            in.drop(2);
        } else if (sym == 17) {
            rep = 3 + in.peek(3);
            in.drop(3);
        } else {
            rep = 11 + in.peek(7);
            in.drop(7);
        }
        if (n + rep > total) return false;
        while (rep--) lens[n++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;                       // no end-of-block code: the block could never end
    const int lr = c.lit.build(lens, (int)hlit);
    if (lr < 0 || (strict && lr != 0)) return false;
    const int dr = c.dist.build(lens + hlit, (int)hdist);
    if (dr < 0) return false;
    c.have_dist = true;
    c.pack();
    return !in.overrun();
}

inline void fixed_codes(Codes &c)
{
    uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    c.lit.build(l, 288);
    uint8_t d[32];                       // 32 five-bit codes; 30 and 31 never occur in valid data (rejected where they are decoded)
    for (int i = 0; i < 32; ++i) d[i] = 5;
    c.dist.build(d, 32);
    c.pack();
}

// Bytes the reader holds: symbol buffers (chunks being decoded, decoded, pooled, the per-thread trial buffers of the block search)
// and the resolved output it still owns (chunks not yet consumed, the recycle pool).  `live` is what the memory budget is checked
// against BEFORE another chunk is started (ParallelGz::may_start); `peak` is its high-water mark -- the number a test of the budget
// asserts, the process's RSS being the allocator's business (sixteen threads' arenas keep what they freed).  One meter per
// process: the diagnostic counts every reader that is open.
struct MemMeter {
    std::atomic<size_t> live{0}, peak{0};
    void add(size_t n)
    {
        const size_t v = live.fetch_add(n, std::memory_order_relaxed) + n;
        size_t p = peak.load(std::memory_order_relaxed);
        while (v > p && !peak.compare_exchange_weak(p, v, std::memory_order_relaxed)) {}
    }
    void sub(size_t n) { live.fetch_sub(n, std::memory_order_relaxed); }
};
inline MemMeter &mem_meter()
{
    static MemMeter m;
    return m;
}

// symbols of a chunk: grown by realloc, never value-initialised (a std::vector would zero 32 MB per chunk before it is written),
// and handed back to a pool when the chunk has been resolved
struct SymBuf {
    uint16_t *p = nullptr;
    size_t cap = 0;
    SymBuf() = default;
    SymBuf(const SymBuf &) = delete;
    SymBuf &operator=(const SymBuf &) = delete;
    SymBuf(SymBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    SymBuf &operator=(SymBuf &&o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~SymBuf() { release(); }
    size_t size() const { return cap; }
    uint16_t *data() { return p; }
    const uint16_t *data() const { return p; }
    uint16_t &operator[](size_t i) { return p[i]; }
    const uint16_t &operator[](size_t i) const { return p[i]; }
    void resize(size_t n)
    {
        if (n <= cap) return;
        uint16_t *q = (uint16_t *)realloc(p, n * sizeof(uint16_t));
        if (!q) throw std::bad_alloc();
        mem_meter().add((n - cap) * sizeof(uint16_t));
        p = q;
        cap = n;
    }
    void release()
    {
        if (p) mem_meter().sub(cap * sizeof(uint16_t));
        free(p);
        p = nullptr;
        cap = 0;
    }
};

// how a symbol buffer grows: by half (doubling would let a chunk hold 4 bytes per output byte where 2 are needed), at least 1 M
// symbols, never beyond the caller's bound
inline size_t sym_grow(size_t have, size_t need, size_t max_out)
{
    return std::min(max_out + 320, std::max<size_t>(std::max<size_t>(have + have / 2, 1u << 20), need));
}

struct TextSet {
    bool ok[256];
    TextSet()
    {
        for (int i = 0; i < 256; ++i) ok[i] = (i >= 0x20 && i <= 0x7E) || i == '\n' || i == '\r' || i == '\t';
    }
};

// One block's symbols -> out (16 bits each: a byte, or 256 + window index).  out holds WIN window slots in front of the chunk's
// own symbols, so a reference behind the chunk's start copies markers like anything else.  Returns false on invalid data.
// max_out bounds the chunk (a trial decode of a candidate block start must not run away).  TEXT: every literal must be text (the
// trial decode of a candidate block start); the check is a flag OR-ed per literal and looked at once per block, not a branch.
// STORE = false: the block is only VALIDATED -- same accept / reject, same n, nothing written and `out` not touched (what a symbol
// is never decides whether a block is valid: literals are checked as they are decoded, a match only has to stay inside what
// has been produced).  The block search's trial decodes run this way: they used to keep up to 8 M symbols per thread.
template <bool TEXT, bool STORE = true>
inline bool decode_block_symbols(BitIn &in, const Codes &c, SymBuf &out, size_t &n_io, size_t max_out, const bool *text_ok, size_t floor = 0)
{
    size_t n = n_io;
    uint32_t bad = 0;
    for (;;) {
        if (STORE && n + 320 > out.size()) {
            if (n + 320 > max_out) return false;
            out.resize(sym_grow(out.size(), n + 320, max_out));
        }
        if (!STORE && n + 320 > max_out) return false;
        uint16_t *const o0 = STORE ? out.data() : nullptr;
        // a stretch that cannot run out of room or of input: up to 64 rounds of (refill, up to three literals or one match)
        size_t room = STORE ? (out.size() - n - 320) / 264 : 64;
        if (room > 64) room = 64;
        if (in.p + 8 * (room + 1) > in.end) room = 0;           // near the end of the input: one careful round at a time
        size_t rounds = room ? room : 1;
        while (rounds--) {
            in.refill();
            if (in.bc < 0) return false;                          // ran past the end of the input
            uint32_t e = c.lit.lookup(in.bb);
            int l = (int)(e & 0xFF);
            uint32_t sym = e >> 16;
            if (sym < 256 && l) {
                if (TEXT) bad |= (uint32_t)!text_ok[sym];
                in.drop(l);
                if (STORE) o0[n] = (uint16_t)sym;
                ++n;
                e = c.lit.lookup(in.bb);
                l = (int)(e & 0xFF);
                sym = e >> 16;
                if (sym < 256 && l) {
                    if (TEXT) bad |= (uint32_t)!text_ok[sym];
                    in.drop(l);
                    if (STORE) o0[n] = (uint16_t)sym;
                    ++n;
                    e = c.lit.lookup(in.bb);
                    l = (int)(e & 0xFF);
                    sym = e >> 16;
                    if (sym < 256 && l) {
                        if (TEXT) bad |= (uint32_t)!text_ok[sym];
                        in.drop(l);
                        if (STORE) o0[n] = (uint16_t)sym;
                        ++n;
                        continue;
                    }
                }
                // not a literal: it needs up to 48 bits of its own
                in.refill();
                if (in.bc < 0) return false;
            }
            if (!l) return false;
            in.drop(l);
            if (sym == 256) { n_io = n; return !in.overrun() && !bad; }
            if (sym & BAD_SYM) return false;
            const int lx = (int)((sym >> 9) & 7);
            const uint32_t len = (sym & 0x1FFu) + in.peek(lx);
            in.drop(lx);
            const uint32_t de = c.dist.lookup(in.bb);
            const int dl = (int)(de & 0xFF);
            if (!dl) return false;
            in.drop(dl);
            const uint32_t dbase = de >> 16;
            if (!dbase) return false;                          // codes 30 and 31
            const int dx = (int)((de >> 8) & 15);
            const uint32_t d = dbase + in.peek(dx);
            in.drop(dx);
            if (d > n - floor) return false;                    // (n counts the WIN window slots too: farther back than 32 KiB + chunk is invalid;
                                                                //  floor = WIN: the stream STARTS here, nothing lies before it)
            if (in.overrun()) return false;
            if (!STORE) { n += len; continue; }
            uint16_t *o = o0 + n;
            const uint16_t *s = o - d;
            if (d >= 8) {
                // eight symbols (16 bytes) at a time, the last step running over the match's end into the slack every stretch keeps
                // free (a call of memcpy with a run-time length costs more than the typical match of a sequence line is long)
                uint16_t *const e = o + len;
                do {
                    __m128i v = _mm_loadu_si128((const __m128i *)s);
                    _mm_storeu_si128((__m128i *)o, v);
                    o += 8;
                    s += 8;
                } while (o < e);
            } else
                for (uint32_t i = 0; i < len; ++i) o[i] = s[i];
            n += len;
        }
        if (TEXT && bad) return false;
    }
}

struct ChunkOut {
    SymBuf sym;                    // WIN window slots (markers 256 + w) followed by the chunk's symbols
    size_t n = WIN;                 // symbols used, window slots included
    uint64_t start_bit = 0, end_bit = 0;
    bool final_block = false;       // the member's last block ends at end_bit
    bool ok = false;
};

// blocks from start_bit on until a block boundary at or behind stop_bit (or the member's final block); text_ok != nullptr: every
// literal must be text.  max_out bounds the output.
inline bool decode_from(const uint8_t *base, const uint8_t *end, uint64_t start_bit, uint64_t stop_bit, ChunkOut &co, size_t max_out, const bool *text_ok,
                        int max_blocks = 1 << 30, int stream_start = 0, bool store = true)
{
    // store = false: validate only (decode_block_symbols<.., false>): co.sym is not touched, co.n / end_bit / final_block are set
    // stream_start != 0: the deflate stream begins at start_bit (a gzip member): a reference behind it is invalid ("distance too far
    // back", like zlib).  1: the window slots are left as they are (64 KB of markers written per 64-KB bgzip member would double
    // the decoder's stores); 2: they are marked like any chunk's (the first chunk of a member that goes on: its successor's window
    // is cut from this output, window slots included if the chunk is short)
    BitIn in;
    in.seek(base, end, start_bit);
    if (store && co.sym.size() < WIN + (1u << 16)) co.sym.resize(WIN + (1u << 16));
    const size_t floor = stream_start ? WIN : 0;
    if (store && stream_start != 1)
        for (uint32_t w = 0; w < WIN; ++w) co.sym[w] = (uint16_t)(256 + w);
    co.n = WIN;
    co.start_bit = start_bit;
    co.final_block = false;
    co.ok = false;
    static thread_local Codes codes;
    const uint64_t size_bits = (uint64_t)(end - base) * 8;
    for (int blocks = 0;; ++blocks) {
        const uint64_t pos = in.bitpos();
        if (pos >= stop_bit || blocks >= max_blocks) { co.end_bit = pos; co.ok = true; return true; }
        if (pos + 3 > size_bits) return false;
        in.refill();
        const uint32_t bfinal = in.peek(1);
        in.drop(1);
        const uint32_t btype = in.peek(2);
        in.drop(2);
        if (btype == 3) return false;
        if (btype == 0) {
            in.drop(in.bc & 7);                              // to the byte boundary
            in.refill();
            if (in.bc < 32) return false;
            const uint32_t len = in.peek(16);
            in.drop(16);
            const uint32_t nlen = in.peek(16);
            in.drop(16);
            if ((len ^ nlen) != 0xFFFFu) return false;
            const uint8_t *src = base + (in.bitpos() >> 3);
            if (src + len > end) return false;
            if (co.n + len > max_out) return false;
            if (store && co.n + len + 320 > co.sym.size()) co.sym.resize(sym_grow(co.sym.size(), co.n + len + 320, max_out));
            for (uint32_t i = 0; i < len; ++i) {
                if (text_ok && !text_ok[src[i]]) return false;
                if (store) co.sym[co.n + i] = src[i];
            }
            co.n += len;
            in.seek(base, end, (uint64_t)(src + len - base) * 8);
        } else {
            if (btype == 1) fixed_codes(codes);
            else if (!read_dynamic_header(in, codes, false)) return false;
            const bool good = store ? (text_ok ? decode_block_symbols<true>(in, codes, co.sym, co.n, max_out, text_ok, floor)
                                               : decode_block_symbols<false>(in, codes, co.sym, co.n, max_out, text_ok, floor))
                                    : (text_ok ? decode_block_symbols<true, false>(in, codes, co.sym, co.n, max_out, text_ok, floor)
                                               : decode_block_symbols<false, false>(in, codes, co.sym, co.n, max_out, text_ok, floor));
            if (!good) return false;
        }
        if (bfinal) { co.final_block = true; co.end_bit = in.bitpos(); co.ok = true; return true; }
    }
}

// the first bit position in [from_bit, to_bit) at which a non-final dynamic block starts whose header passes the strict checks and
// whose whole block decodes (to text, if text_ok); ~0 if there is none
inline uint64_t find_block(const uint8_t *base, const uint8_t *end, uint64_t from_bit, uint64_t to_bit, const bool *text_ok)
{
    static thread_local Codes codes;
    ChunkOut trial;                                        // (validated only: no symbols are kept)
    const uint64_t size_bits = (uint64_t)(end - base) * 8;
    if (to_bit + 64 > size_bits) to_bit = size_bits > 64 ? size_bits - 64 : 0;
    for (uint64_t b = from_bit; b < to_bit; ++b) {
        uint64_t v;
        memcpy(&v, base + (b >> 3), 8);
        v >>= (b & 7);
        if ((v & 7) != 4) continue;                          // BFINAL = 0, BTYPE = 2 (bits: final, then type LSB first -> 0b100)
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;      // HLIT <= 286 symbols, HDIST <= 30
        BitIn in;
        in.seek(base, end, b + 3);
        if (!read_dynamic_header(in, codes, true)) continue;
        // the whole block must decode (at most 8 MiB of output) and be followed by something that can be a block header
        if (!decode_from(base, end, b, ~0ull, trial, WIN + (8u << 20), text_ok, 1, 0, false)) continue;
        if (trial.n == WIN) continue;
        const uint64_t e = trial.end_bit;
        if (e + 3 <= size_bits) {
            uint64_t w;
            memcpy(&w, base + (e >> 3), std::min<size_t>(8, (size_t)(end - (base + (e >> 3)))));
            if ((((w >> (e & 7)) >> 1) & 3) == 3) continue;
        }
        return b;
    }
    return ~0ull;
}

// CRC-32 (the gzip polynomial) of n bytes, continuing `crc` like zlib's crc32().  The image's zlib does ~2 GB/s per thread -- a sixth of
// the reader's CPU time once the chunks are resolved; with carry-less multiplication (four 128-bit lanes folded per 64 bytes, then
// 128 -> 64 -> 32 bits by Barrett reduction: the published folding scheme, constants for the reflected polynomial 0x1DB710641) it is a
// memory pass.  Hosts without PCLMULQDQ, and what the folding leaves (under 64 bytes, and the bytes behind the last 16), go to zlib.
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) inline __m128i crc32_fold_step(__m128i x, __m128i k, __m128i next)
{
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), next);
}
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(uint32_t crc, const uint8_t *p, size_t n)       // n >= 64, a multiple of 16; crc without the inversions
{
    const __m128i k12 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);        // x^(512+64) mod P and x^512 mod P, bit-reflected
    const __m128i k34 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);        // x^(128+64), x^128
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll);                       // x^64
    const __m128i pm = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);          // mu, P
    const __m128i lo32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i a = _mm_loadu_si128((const __m128i *)p), b = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i c = _mm_loadu_si128((const __m128i *)(p + 32)), d = _mm_loadu_si128((const __m128i *)(p + 48));
    a = _mm_xor_si128(a, _mm_cvtsi32_si128((int)crc));
    p += 64;
    n -= 64;
    for (; n >= 64; n -= 64, p += 64) {
        a = crc32_fold_step(a, k12, _mm_loadu_si128((const __m128i *)p));
        b = crc32_fold_step(b, k12, _mm_loadu_si128((const __m128i *)(p + 16)));
        c = crc32_fold_step(c, k12, _mm_loadu_si128((const __m128i *)(p + 32)));
        d = crc32_fold_step(d, k12, _mm_loadu_si128((const __m128i *)(p + 48)));
    }
    a = crc32_fold_step(a, k34, b);
    a = crc32_fold_step(a, k34, c);
    a = crc32_fold_step(a, k34, d);
    for (; n >= 16; n -= 16, p += 16) a = crc32_fold_step(a, k34, _mm_loadu_si128((const __m128i *)p));
    // 128 -> 64 bits
    __m128i t = _mm_clmulepi64_si128(a, k34, 0x10);
    a = _mm_xor_si128(_mm_srli_si128(a, 8), t);
    t = _mm_srli_si128(a, 4);
    a = _mm_and_si128(a, lo32);
    a = _mm_xor_si128(_mm_clmulepi64_si128(a, k5, 0x00), t);
    // Barrett: 64 -> 32 bits
    t = _mm_and_si128(a, lo32);
    t = _mm_clmulepi64_si128(t, pm, 0x10);
    t = _mm_and_si128(t, lo32);
    t = _mm_clmulepi64_si128(t, pm, 0x00);
    a = _mm_xor_si128(a, t);
    return (uint32_t)_mm_extract_epi32(a, 1);
}
#endif

inline uint32_t crc32_bytes(uint32_t crc, const uint8_t *p, size_t n)
{
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (fast && n >= 64) {
        const size_t m = n & ~(size_t)15;
        crc = ~crc32_fold(~crc, p, m);
        p += m;
        n -= m;
    }
#endif
    for (; n; ) {
        const size_t k = std::min<size_t>(n, 1u << 30);
        crc = (uint32_t)crc32(crc, p, (uInt)k);
        p += k;
        n -= k;
    }
    return crc;
}

// gzip member header at p: returns its length, 0 if there is none
inline size_t gzip_header_len(const uint8_t *p, size_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const uint8_t flg = p[3];
    size_t o = 10;
    if (flg & 4) { if (o + 2 > n) return 0; o += 2 + (size_t)(p[o] | (p[o + 1] << 8)); }
    if (flg & 8) { while (o < n && p[o]) ++o; ++o; }
    if (flg & 16) { while (o < n && p[o]) ++o; ++o; }
    if (flg & 2) o += 2;
    return o < n ? o : 0;
}

} // namespace pgz_detail

class ParallelGz {
public:
    ~ParallelGz()
    {
        shutdown();
        drop_outputs();
        for (auto &v : out_pool_) pgz_detail::mem_meter().sub(v.capacity());
        if (map_) munmap(const_cast<uint8_t *>(map_), size_);
    }

    // true: `path` is a gzip file large enough to be worth the threads; inflating starts at once.  chunk_bytes = compressed bytes
    // per chunk (0: 4 MiB).
    bool open(const std::string &path, unsigned threads, size_t chunk_bytes = 0, size_t min_size = 8u << 20)
    {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat sb;
        if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || (size_t)sb.st_size < std::max<size_t>(min_size, 18)) { ::close(fd); return false; }
        size_ = (size_t)sb.st_size;
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return false;
        map_ = (const uint8_t *)m;
        if (!pgz_detail::gzip_header_len(map_, size_)) return false;
        threads_ = std::max(1u, threads);
        chunk_bytes_ = chunk_bytes ? chunk_bytes : (4u << 20);
        lookahead_ = 2 * threads_ + 2;
        start_member(0);
        for (unsigned t = 0; t < threads_; ++t) th_.emplace_back([this] { worker(); });
        return true;
    }


    // next bytes of the decompressed stream; 0 at the end.  Throws std::runtime_error on a corrupt stream.
    size_t read(char *dst, size_t n)
    {
        size_t got = 0;
        while (got < n) {
            if (cur_out_ && cur_pos_ < cur_out_->size()) {
                const size_t take = std::min(n - got, cur_out_->size() - cur_pos_);
                memcpy(dst + got, cur_out_->data() + cur_pos_, take);
                cur_pos_ += take;
                got += take;
                continue;
            }
            if (!next_chunk()) break;
        }
        return got;
    }

    // the next chunk of the decompressed stream as a whole (its buffer changes hands, nothing is copied); false at the end.  Not to
    // be mixed with read() on one object.
    bool take(std::vector<char> &dst)
    {
        if (!next_chunk()) return false;
        pgz_detail::mem_meter().sub(cur_out_->capacity());
        dst.swap(*cur_out_);
        pgz_detail::mem_meter().add(cur_out_->capacity());      // (the caller's previous buffer: recycled or dropped when this chunk is retired)
        cur_pos_ = 0;
        cur_out_->clear();
        return true;
    }

    // a buffer that take() handed out, when the caller is done with it: the next chunk is resolved into it instead of into fresh
    // pages (a chunk's 16 MB from the allocator are mapped, faulted in and zeroed before the first byte is written -- on 32 threads
    // that is the allocator's lock and the kernel's, not memory bandwidth).  Any thread.
    void recycle(std::vector<char> &&v) { pool_out(std::move(v), false); }

    // bytes this process's readers hold right now / at most so far (symbols, resolved output not yet handed out, both pools):
    // what set_memory_budget() bounds, up to one chunk (the one the consumer is waiting for is always decoded)
    static size_t memory_in_flight() { return pgz_detail::mem_meter().live.load(); }
    static size_t memory_high_water() { return pgz_detail::mem_meter().peak.load(); }
    size_t largest_chunk_bytes() const { return largest_chunk_.load(); }      // symbols + output of the largest chunk so far

    // after take(): go on with read() (the caller has consumed what it took)
    void switch_to_read() { cur_pos_ = cur_out_ ? cur_out_->size() : 0; }

    // statistics for the diagnostic command
    uint64_t chunks_total = 0, chunks_redecoded = 0, bytes_out = 0, members = 0, trailing_garbage = 0;
    // bytes of decoded-but-unread output the reader may hold (symbols and resolved bytes); set before open()
    void set_memory_budget(size_t bytes) { budget_ = std::max<size_t>(bytes, 64u << 20); }
    std::atomic<uint64_t> ns_find{0}, ns_decode{0}, ns_resolve{0};     // summed over the worker threads

private:
    using ChunkOut = pgz_detail::ChunkOut;
    struct Chunk {
        uint64_t nominal_start = 0, nominal_stop = 0;   // bits
        ChunkOut co;
        int state = 0;                                  // 0 waiting, 1 decoding, 2 decoded, 3 resolving, 4 ready, 5 consumed
        std::vector<uint8_t> window;                    // the 32 KiB before this chunk (set when it is tied to its predecessor)
        std::vector<char> out;
        size_t out_len = 0;                             // out.size() when it was resolved (the vector itself may have been taken)
        uint32_t crc = 0;
    };

    // output buffers of chunks that will not be consumed (behind a member's final block, or at destruction) leave the meter
    void drop_outputs()
    {
        for (auto &c : chunks_) {
            pgz_detail::mem_meter().sub(c.out.capacity());
            std::vector<char>().swap(c.out);
        }
    }

    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        cv_done_.notify_all();
        for (auto &t : th_) t.join();
        th_.clear();
    }

    // lay the chunks of the member whose header starts at byte `off`
    void start_member(size_t off)
    {
        using namespace pgz_detail;
        const size_t h = gzip_header_len(map_ + off, size_ - off);
        if (!h) throw std::runtime_error("gzip: bad member header");
        std::unique_lock<std::mutex> lk(mu_);
        n_live_ = 0;                                           // no new decodes of the old member ...
        cv_done_.wait(lk, [&] { return busy_ == 0; });         // ... and none of them still running on its chunks
        drop_outputs();
        chunks_.clear();
        resolve_q_.clear();
        member_data_ = off + h;
        const uint64_t first = (uint64_t)member_data_ * 8, endb = (uint64_t)size_ * 8;
        for (uint64_t b = first; b < endb; b += (uint64_t)chunk_bytes_ * 8) {
            chunks_.emplace_back();
            chunks_.back().nominal_start = b;
            chunks_.back().nominal_stop = std::min(endb, b + (uint64_t)chunk_bytes_ * 8);
        }
        next_decode_ = 0;
        tied_ = 0;
        cur_ = 0;
        ratio_seen_ = 1.0;
        n_live_ = chunks_.size();
        member_crc_ = crc32(0L, Z_NULL, 0);
        member_len_ = 0;
        text_ = true;
        ++members;
        cv_work_.notify_all();
        cv_done_.notify_all();
    }

    void worker()
    {
        using namespace pgz_detail;
        static const TextSet ts;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || !resolve_q_.empty() || (next_decode_ < n_live_ && next_decode_ < cur_ + eff_lookahead() && may_start()); });
            if (stop_) return;
            if (!resolve_q_.empty()) {
                const size_t i = resolve_q_.front();
                resolve_q_.pop_front();
                Chunk &c = chunks_[i];
                ++busy_;
                lk.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                std::string what;               // an exception (std::bad_alloc of a huge chunk) must not leave a std::thread: next_chunk() rethrows it
                try {
                    resolve(c);
                } catch (const std::exception &ex) { what = std::string("gzip reader: ") + ex.what(); } catch (...) { what = "gzip reader: unknown exception"; }
                ns_resolve += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                lk.lock();
                if (!what.empty() && dev_error_.empty()) dev_error_ = what;
                --busy_;
                c.state = 4;
                cv_done_.notify_all();
                continue;
            }
            const size_t i = next_decode_++;
            Chunk &c = chunks_[i];
            c.state = 1;
            ++busy_;
            ++decoding_;
            {
                std::lock_guard<std::mutex> pl(pool_mu_);
                if (!pool_.empty()) { c.co.sym = std::move(pool_.back()); pool_.pop_back(); }
            }
            const bool text = text_;
            const size_t spec_out = speculative_out();
            lk.unlock();
            const uint8_t *base = map_, *end = map_ + size_;
            uint64_t s = c.nominal_start;
            const auto t0 = std::chrono::steady_clock::now();
            auto t1 = t0;
            bool ok = false;
            std::string what;
            try {
                if (i > 0) s = find_block(base, end, c.nominal_start, c.nominal_stop, text ? ts.ok : nullptr);
                t1 = std::chrono::steady_clock::now();
                // (the start's first block was held to text by the search.)  A member's first chunk starts the stream: a distance that
                // reaches before it is invalid there, as in zlib.  The output is capped at what this member's ratio so far makes
                // plausible: a chunk beyond it is decoded again, alone, when it is tied
                if (s != ~0ull) ok = decode_from(base, end, s, c.nominal_stop, c.co, WIN + spec_out, nullptr, 1 << 30, i == 0 ? 2 : 0);
            } catch (const std::exception &ex) { what = std::string("gzip reader: ") + ex.what(); } catch (...) { what = "gzip reader: unknown exception"; }
            ns_find += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            ns_decode += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
            if (!ok) { c.co.ok = false; give_back(c.co.sym); }
            lk.lock();
            if (!what.empty() && dev_error_.empty()) dev_error_ = what;
            --busy_;
            --decoding_;
            c.state = 2;
            cv_done_.notify_all();
        }
    }

    size_t max_chunk_out() const { return (size_t)1032 * chunk_bytes_ + (1u << 20); }    // deflate expands at most ~1032:1

    // Memory in flight is bounded by the member's own compression ratio (mu_ held).  A decoded chunk holds two bytes per output byte
    // (symbols) until it is resolved, and deflate may expand 1032:1 -- thirty-odd chunks of a very repetitive (but valid) member
    // would be tens of gigabytes.  So: chunks decoded ahead of the reader are limited to what fits `budget_` at the largest
    // ratio seen so far in this member (at least one; before the first chunk is tied, one per thread), and a speculative decode
    // is cut off at 8x that ratio (at least 32:1) -- such a chunk is decoded again, alone, when it is tied (tie(), up to 1032:1).
    size_t eff_lookahead() const
    {
        if (tied_ == 0) return std::min<size_t>(lookahead_, (size_t)threads_ + 1);
        const double per_chunk = 3.0 * ratio_seen_ * (double)chunk_bytes_;      // symbols + resolved bytes
        return (size_t)std::max(1.0, std::min((double)lookahead_, (double)budget_ / std::max(1.0, per_chunk)));
    }
    size_t speculative_out() const
    {
        const double r = std::min(1032.0, std::max(32.0, 8.0 * ratio_seen_));
        return (size_t)(r * (double)chunk_bytes_) + (1u << 20);
    }

    // The budget is checked against what is HELD (pgz_detail::mem_meter: every symbol buffer, every output buffer the reader owns,
    // both pools), before a chunk is started: it may start if the held bytes plus what the chunks now being decoded may still
    // grow to plus its own expected need stay inside the budget -- or if nothing else is in flight (the chunk the consumer waits
    // for is always decoded, so the bound is budget + one chunk).  mu_ held.
    // (a chunk's need: 2 B of symbols per output byte, grown by halves, + the resolved bytes; before the member's first chunk is tied
    // its ratio is taken as 8:1 -- a speculative decode is cut off at 32:1 then, speculative_out)
    // and never less than the largest chunk met so far: a chunk ends at the first block boundary behind its nominal end, and the
    // blocks of very repetitive data are long)
    size_t chunk_need() const
    {
        const size_t est = (size_t)(3.2 * (tied_ == 0 ? 8.0 : ratio_seen_) * (double)chunk_bytes_) + 2 * pgz_detail::WIN + (1u << 20);
        return std::max(est, largest_chunk_.load(std::memory_order_relaxed));
    }
    bool may_start() const
    {
        if (next_decode_ <= cur_) return true;
        const size_t need = chunk_need();
        return pgz_detail::mem_meter().live.load(std::memory_order_relaxed) + (decoding_ + 1) * need <= budget_ + pooled_bytes();
    }
    size_t pooled_bytes() const          // buffers a starting chunk takes over instead of allocating (at most one of each kind)
    {
        std::lock_guard<std::mutex> lk(pool_mu_);
        size_t a = 0, b = 0;
        for (const auto &q : pool_) a = std::max(a, q.cap * sizeof(uint16_t));
        for (const auto &q : out_pool_) b = std::max(b, q.capacity());
        return a + b;
    }

    void give_back(pgz_detail::SymBuf &b)
    {
        if (!b.p) return;
        std::lock_guard<std::mutex> lk(pool_mu_);
        size_t held = b.cap * sizeof(uint16_t);
        for (const auto &q : pool_) held += q.cap * sizeof(uint16_t);
        if (pool_.size() < 4 * (size_t)threads_ && held <= budget_ / 8) pool_.push_back(std::move(b));     // (the pools are part of the budget)
        else b.release();
    }

    // an output buffer comes (back) to the pool or is dropped; counted: it is on the meter already (a chunk's own buffer)
    void pool_out(std::vector<char> &&v, bool counted)
    {
        const size_t cap = v.capacity();
        bool kept = false;
        if (cap >= (1u << 20)) {
            std::lock_guard<std::mutex> lk(pool_mu_);
            size_t held = cap;
            for (const auto &q : out_pool_) held += q.capacity();
            if (out_pool_.size() < 2 * (size_t)threads_ + 8 && held <= budget_ / 8) { out_pool_.push_back(std::move(v)); kept = true; }
        }
        if (kept && !counted) pgz_detail::mem_meter().add(cap);
        if (!kept && counted) pgz_detail::mem_meter().sub(cap);
        if (!kept) std::vector<char>().swap(v);
    }

    void resolve(Chunk &c)
    {
        using namespace pgz_detail;
        const size_t n = c.co.n - WIN;
        {
            // the largest recycled buffer (they are not cleared: resize() then touches only what it has to add)
            std::lock_guard<std::mutex> lk(pool_mu_);
            if (!out_pool_.empty()) {
                size_t best = 0;
                for (size_t k = 1; k < out_pool_.size(); ++k)
                    if (out_pool_[k].capacity() > out_pool_[best].capacity()) best = k;
                c.out.swap(out_pool_[best]);
                out_pool_[best].swap(out_pool_.back());
                out_pool_.pop_back();
            }
        }
        if (c.out.capacity() < n) {
            pgz_detail::mem_meter().sub(c.out.capacity());
            std::vector<char>().swap(c.out);
            c.out.reserve(n + n / 16);
            pgz_detail::mem_meter().add(c.out.capacity());
        }
        c.out.resize(n);
        c.out_len = n;
        {
            const size_t mine = c.co.sym.cap * sizeof(uint16_t) + c.out.capacity();
            size_t seen = largest_chunk_.load(std::memory_order_relaxed);
            while (mine > seen && !largest_chunk_.compare_exchange_weak(seen, mine, std::memory_order_relaxed)) {}
        }
        const uint16_t *s = c.co.sym.data() + WIN;
        const uint8_t *w = c.window.data();
        uint8_t *o = reinterpret_cast<uint8_t *>(c.out.data());
        size_t i = 0;
#if defined(__SSE2__)
        // sixteen symbols at a time while none of them is a marker (behind the first stretch of a chunk hardly any is)
        const __m128i zero = _mm_setzero_si128();
        for (; i + 16 <= n; i += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 8));
            if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_srli_epi16(_mm_or_si128(a, b), 8), zero)) == 0xFFFF)
                _mm_storeu_si128((__m128i *)(o + i), _mm_packus_epi16(a, b));
            else
                for (size_t k = i; k < i + 16; ++k) { const uint16_t v = s[k]; o[k] = v < 256 ? (uint8_t)v : w[v - 256]; }
        }
#endif
        for (; i < n; ++i) {
            const uint16_t v = s[i];
            o[i] = v < 256 ? (uint8_t)v : w[v - 256];
        }
        give_back(c.co.sym);
        c.crc = crc32_bytes((uint32_t)crc32(0L, Z_NULL, 0), o, n);
    }

    // the window of the next chunk: the last WIN bytes of (window ++ this chunk), markers resolved
    static void next_window_of(const std::vector<uint8_t> &window, const ChunkOut &co, std::vector<uint8_t> &nw)
    {
        using namespace pgz_detail;
        const size_t n = co.n - WIN;
        nw.resize(WIN);
        for (size_t k = 0; k < WIN; ++k) {
            const size_t pos = n + k;                          // index into (window ++ chunk), counted from the start of the window
            if (pos < WIN) nw[k] = window[pos];
            else {
                const uint16_t v = co.sym[pos];
                nw[k] = v < 256 ? (uint8_t)v : window[v - 256];
            }
        }
    }

    // tie chunk i to its predecessor (mu_ held on entry and exit; released while a chunk is decoded again): its start must be the
    // predecessor's end.  Sets its window, computes the next one, queues the resolution.  Returns false if chunk i is not decoded yet.
    bool tie(std::unique_lock<std::mutex> &lk, size_t i)
    {
        using namespace pgz_detail;
        Chunk &c = chunks_[i];
        if (c.state < 2) return false;
        const uint64_t want = i == 0 ? (uint64_t)member_data_ * 8 : prev_end_;
        if (!c.co.ok || c.co.start_bit != want) {
            // not where the stream says the chunk starts (or it failed): decode it from the right place -- nothing speculative
            lk.unlock();
            ChunkOut co;
            co.sym = std::move(c.co.sym);                      // (what the speculative decode left is of no use: its buffer is)
            const bool ok = decode_from(map_, map_ + size_, want, std::max(want, c.nominal_stop), co, WIN + max_chunk_out() * 2, nullptr, 1 << 30, i == 0 ? 2 : 0);
            lk.lock();
            if (!ok) throw std::runtime_error("gzip: invalid deflate data near byte " + std::to_string(want / 8));
            c.co = std::move(co);
            ++chunks_redecoded;
        }
        {   // this member's compression ratio so far (eff_lookahead, speculative_out)
            const uint64_t in_bits = c.co.end_bit > c.co.start_bit ? c.co.end_bit - c.co.start_bit : 1;
            ratio_seen_ = std::max(ratio_seen_, (double)(c.co.n - WIN) * 8.0 / (double)std::max<uint64_t>(in_bits, 8 * 1024));
        }
        if (i == 0) {
            c.window.assign(WIN, 0);
            // is it text?  (then candidate block starts of later chunks must decode to text as well)
            bool text = true;
            static const TextSet ts;
            for (size_t k = WIN; k < std::min<size_t>(c.co.n, WIN + (1u << 16)); ++k)
                if (c.co.sym[k] >= 256 || !ts.ok[c.co.sym[k]]) { text = false; break; }
            text_ = text;
        } else
            c.window = next_window_;
        std::vector<uint8_t> nw;
        next_window_of(c.window, c.co, nw);
        next_window_.swap(nw);
        prev_end_ = c.co.end_bit;
        ++chunks_total;
        if (c.co.final_block) n_live_ = i + 1;                 // nothing behind the final block belongs to this member
        c.state = 3;
        resolve_q_.push_back(i);
        cv_work_.notify_all();
        return true;
    }


    // make the next chunk's bytes current; false at the end of the file
    bool next_chunk()
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (cur_out_) {                                       // the chunk just finished
            Chunk &c = chunks_[cur_];
            member_crc_ = (uint32_t)crc32_combine(member_crc_, c.crc, (z_off_t)c.out_len);
            member_len_ += c.out_len;
            bytes_out += c.out_len;
            pool_out(std::move(c.out), true);                   // (read(): consumed; take(): the caller's previous buffer)
            std::vector<char>().swap(c.out);
            c.state = 5;
            cur_out_ = nullptr;
            ++cur_;
            cv_work_.notify_all();
        }
        for (;;) {
            while (tied_ < n_live_ && tied_ < cur_ + std::max<size_t>(eff_lookahead(), 1) && chunks_[tied_].state >= 2 && chunks_[tied_].state < 3) {
                if (!tie(lk, tied_)) break;
                ++tied_;
            }
            if (!dev_error_.empty()) throw std::runtime_error(dev_error_);
            if (cur_ >= n_live_) {
                // member complete: trailer (CRC-32, ISIZE) at the next byte boundary behind the final block
                if (tied_ == 0 || !chunks_[n_live_ - 1].co.final_block) throw std::runtime_error("gzip: stream ends before its final block");
                const size_t t = (size_t)((prev_end_ + 7) / 8);
                if (t + 8 > size_) throw std::runtime_error("gzip: truncated trailer");
                uint32_t crc, isize;
                memcpy(&crc, map_ + t, 4);
                memcpy(&isize, map_ + t + 4, 4);
                if (crc != member_crc_ || isize != (uint32_t)member_len_) throw std::runtime_error("gzip: CRC or length mismatch at the end of the member");
                size_t nx = t + 8;
                while (nx < size_ && map_[nx] == 0) ++nx;        // zero padding behind a member is legal
                if (nx >= size_) return false;
                // bytes behind the last member that are no gzip header: gzip(1) says "trailing garbage ignored" and keeps what it
                // decoded; every byte of the members has been delivered and checked by now, so this reader does the same
                if (!pgz_detail::gzip_header_len(map_ + nx, size_ - nx)) {
                    trailing_garbage = size_ - nx;
                    fprintf(stderr, "[TAXOR SEARCH WARNING] gzip: %zu bytes of trailing garbage behind the last member ignored\n", size_ - nx);
                    return false;
                }
                lk.unlock();
                start_member(nx);                                // another member: the same way
                lk.lock();
                continue;
            }
            if (chunks_[cur_].state == 4) {
                cur_out_ = &chunks_[cur_].out;
                cur_pos_ = 0;
                if (cur_out_->empty()) { lk.unlock(); return next_chunk(); }
                return true;
            }
            cv_done_.wait(lk);
        }
    }

    const uint8_t *map_ = nullptr;
    size_t size_ = 0, chunk_bytes_ = 4u << 20, member_data_ = 0;
    unsigned threads_ = 1;
    size_t lookahead_ = 4;
    size_t budget_ = (size_t)4 << 30;
    double ratio_seen_ = 1.0;              // largest output : input ratio of a tied chunk of the current member
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::deque<Chunk> chunks_;
    std::deque<size_t> resolve_q_;
    std::vector<pgz_detail::SymBuf> pool_;
    std::vector<std::vector<char>> out_pool_;
    mutable std::mutex pool_mu_;
    std::atomic<size_t> largest_chunk_{0};
    size_t next_decode_ = 0, tied_ = 0, cur_ = 0, n_live_ = 0, busy_ = 0, decoding_ = 0;
    uint64_t prev_end_ = 0;
    std::vector<uint8_t> next_window_;
    uint32_t member_crc_ = 0;
    uint64_t member_len_ = 0;
    bool text_ = true, stop_ = false;
    std::vector<char> *cur_out_ = nullptr;
    size_t cur_pos_ = 0;
    std::string dev_error_;
};

} // namespace fastx
