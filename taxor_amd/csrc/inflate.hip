// inflate.hip -- deflate chunks decoded on the device, for the reader of single-member .gz query files (pgz.h; the reference reads
// .gz through seqan3's stream layer, one zlib stream on one thread, taxor_search.cpp:181-184).
//
// pgz.h cuts a gzip member's deflate stream into chunks, finds a block start inside every chunk on the host and -- without this
// file -- decodes every chunk on a host thread into 16-bit symbols (a byte, or a marker "byte w of the 32 KiB before this chunk"),
// ties the chunks together and resolves the markers.  The decoding is ~77 % of that CPU time and the pool's GPU box gives a
// container 16 CPUs (profiles/r04/cpu_quota_probe.txt); a deflate stream is serial, but a 10-GB file is thousands of chunks.
// Here: one WAVE per chunk.  The decoder's state is wave-uniform (every lane computes the same bit buffer, table index, symbol;
// values that come out of memory are made uniform with readfirstlane, so the compiler keeps the state in scalar registers and
// branches on the scalar unit); the lanes differ only where there is something to do in parallel: literals are collected 64 to a
// wave and stored in one coalesced write, a match is copied by as many lanes as it is long, tables are filled by all lanes.  The
// code tables of the current block live in LDS (12 KB per wave).  Symbols go to HBM in the host decoder's format, so a chunk the
// device gives up on (a start that was not one, an output beyond its share of the arena) is decoded on the host and put in its place.
// Second kernel pair: the windows are chained from chunk to chunk (one block, chunk after chunk -- 32 Ki symbols each), then every
// symbol of every chunk is resolved to its byte in parallel.
#include "../../include/taxor_gpu_tools.h"
#include "tuning.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);

namespace {

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    taxor_set_last_error(buf);
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(TAXOR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                              \
    } while (0)

constexpr uint32_t WIN = 32768;
constexpr int LIT_PB = 10, DIST_PB = 8;

struct ChunkIn { uint64_t start_bit, stop_bit, sym_off; uint64_t sym_cap; };              // sym_cap counts the WIN window slots too
struct ChunkRes { uint64_t end_bit; uint64_t n_sym; uint32_t status, final_block; };       // n_sym counts the WIN window slots too

enum : uint32_t { ST_OK = 0, ST_INVALID = 2, ST_OVERFLOW = 3, ST_INPUT_END = 4 };

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// Behind every statement that only some lanes execute.  The compiler's uniformity analysis marks the values merged at the join of a
// lane-dependent branch as divergent, and a one-armed `if` joins in whatever block follows it -- often a loop latch where the
// decoder's whole state is merged, which then leaves the scalar registers for good.  The (empty) wave barrier gives the `if` a join
// block of its own.
#define LANES_DONE __builtin_amdgcn_wave_barrier()

__device__ __constant__ uint8_t c_clord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the input as 32-bit words (the buffer is 16-byte aligned and followed by zero padding); wave-uniform.  The words come through a
// window of IN_WORDS words in LDS that all lanes fill together (eight words each, one coalesced read of 2 KB): a word fetched from
// memory by itself costs the wave a memory round trip for every four or five symbols
constexpr uint32_t IN_WORDS = 512;
__shared__ __attribute__((aligned(16))) uint32_t s_in_ring[IN_WORDS];
// (out of line, and with nothing but values for arguments: the decoder's loops are a few hundred instructions that many waves run
// at different places -- what is rare must not sit in between -- and its state must stay in scalar registers, which a reference
// handed to a function would end)
__device__ __noinline__ void fill_ring(const uint32_t *w, uint64_t ring_base, uint64_t n_words)
{
    const uint64_t mine = ring_base + threadIdx.x * 8u;
    uint4 a = make_uint4(0, 0, 0, 0), b = a;
    if (mine < n_words) a = *reinterpret_cast<const uint4 *>(w + mine);
    if (mine + 4 < n_words) b = *reinterpret_cast<const uint4 *>(w + mine + 4);
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<uint4 *>(s_in_ring + threadIdx.x * 8u) = a;
    *reinterpret_cast<uint4 *>(s_in_ring + threadIdx.x * 8u + 4) = b;
    __builtin_amdgcn_wave_barrier();
}
struct Bits {
    const uint32_t *w;
    uint32_t *ring;        // LDS, IN_WORDS words
    uint64_t n_words;      // words that hold input (a multiple of 4, the padding included); beyond them zeros are read
    uint64_t ring_base, ring_end;
    uint64_t wp;           // next word to take
    uint64_t bb;
    int bc;
    uint32_t ahead;        // word wp, loaded ahead of its use
    __device__ __forceinline__ uint32_t word(uint64_t i)
    {
        if (i >= n_words) return 0u;
        if (i < ring_base || i >= ring_end) {
            ring_base = i & ~3ull;
            ring_end = ring_base + IN_WORDS < n_words ? ring_base + IN_WORDS : n_words;
            fill_ring(w, ring_base, n_words);
        }
        return uni(ring[i - ring_base]);
    }
    __device__ __forceinline__ void seek(uint64_t bit)
    {
        ring_base = ring_end = 0;
        wp = bit >> 5;
        bb = 0;
        bc = 0;
        ahead = word(wp);
        refill();
        const int s = (int)(bit & 31);
        bb >>= s;
        bc -= s;
        refill();
    }
    __device__ __forceinline__ void refill()               // afterwards at least 33 bits are in bb
    {
        if (bc <= 32) {
            bb |= (uint64_t)ahead << bc;
            bc += 32;
            ++wp;
            ahead = word(wp);
        }
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(bb & ((1ull << n) - 1)); }
    __device__ __forceinline__ void drop(int n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint64_t bitpos() const { return wp * 32 - (uint64_t)bc; }
};

__device__ __forceinline__ uint32_t rev_bits(uint32_t c, int n) { return __builtin_bitreverse32(c) >> (32 - n); }

// One canonical prefix code in LDS: a primary table of 2^PB entries (symbol << 8 | length; 0 = the code is longer than PB bits, or
// no code) and, for the long codes, the symbols in canonical order with the count of codes per length (decoded bit by bit: in
// deflate's alphabets the long codes are the rare symbols).
template <int PB, int MAXSYM> struct Code {
    uint32_t tab[1 << PB];
    uint16_t sorted[MAXSYM];
    uint16_t count[16];
};

// lens[0..n) in LDS -> the code; wave-uniform control flow, the table filled by all lanes.  Returns 0 complete, 1 the legal
// incomplete cases (one code of one bit, or no code at all), -1 invalid -- as the host decoder's Huff::build (pgz.h).
template <int PB, int MAXSYM> __device__ __noinline__ int build_code(Code<PB, MAXSYM> &c, const uint8_t *lens, int n, uint16_t *scratch /* 2 * MAXSYM */)
{
    const uint32_t lane = threadIdx.x;
    if (lane < 16) c.count[lane] = 0;
    for (uint32_t i = lane; i < (1u << PB); i += 64) c.tab[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t l = uni(lens[i]);
#pragma unroll
        for (int k = 0; k < 16; ++k) cnt[k] += (l == (uint32_t)k);
    }
    if (cnt[0] == (uint32_t)n) return 1;
    int left = 1;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= (int)cnt[l];
        if (left < 0) return -1;
    }
    const uint32_t used = (uint32_t)n - cnt[0];
    if (left > 0 && !(used == 1 && cnt[1] == 1)) return -1;
    uint32_t next[16], offs[16];
    {
        uint32_t code = 0, o = 0;
        next[0] = 0;
        offs[0] = 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) {
            code = (code + (l > 1 ? cnt[l - 1] : 0u)) << 1;
            next[l] = code;
            offs[l] = o;
            o += cnt[l];
        }
    }
#pragma unroll
    for (int l = 0; l < 16; ++l)
        if (lane == (uint32_t)l) c.count[l] = (uint16_t)cnt[l];
    // every symbol's canonical code: symbol order within a length.  scratch[i] = code of symbol i, scratch[MAXSYM + i] = its place
    // among the sorted symbols (one lane walks the symbols: a few hundred steps per block of ~10^5 symbols)
    if (lane == 0) {
        uint32_t nx[16], of[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) { nx[l] = next[l]; of[l] = offs[l]; }
        for (int i = 0; i < n; ++i) {
            const uint32_t l = lens[i];
            if (!l) continue;
            uint32_t code = 0, place = 0;
#pragma unroll
            for (int k = 1; k < 16; ++k)
                if (l == (uint32_t)k) { code = nx[k]++; place = of[k]++; }
            scratch[i] = (uint16_t)code;
            c.sorted[place] = (uint16_t)i;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = (int)lane; i < n; i += 64) {
        const uint32_t l = lens[i];
        if (!l || l > (uint32_t)PB) continue;
        const uint32_t r = rev_bits(scratch[i], (int)l), e = ((uint32_t)i << 8) | l;
        for (uint32_t k = r; k < (1u << PB); k += 1u << l) c.tab[k] = e;
    }
    __builtin_amdgcn_wave_barrier();
    return left > 0 ? 1 : 0;
}

// a code longer than the primary table's bits (or none): canonical decoding, bit by bit (puff's loop)
__device__ __noinline__ uint32_t decode_long(const uint16_t *count_, const uint16_t *sorted_, uint64_t bb)
{
    uint32_t code = 0, first = 0, index = 0;
    uint64_t b = bb;
    for (int len = 1; len <= 15; ++len) {
        code |= (uint32_t)(b & 1);
        b >>= 1;
        const uint32_t count = uni(count_[len]);
        if (code < first + count) return ((uint32_t)uni(sorted_[index + (code - first)]) << 8) | (uint32_t)len;
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return 0;
}

// the symbol at the bottom of `bb` -> (symbol << 8 | length), 0 if there is no such code; wave-uniform
template <int PB, int MAXSYM> __device__ __forceinline__ uint32_t decode_sym(const Code<PB, MAXSYM> &c, uint64_t bb)
{
    const uint32_t e = uni(c.tab[(uint32_t)bb & ((1u << PB) - 1)]);
    if (e) return e;
    return uni(decode_long(c.count, c.sorted, bb));
}

struct WaveLds {
    Code<LIT_PB, 288> lit;
    Code<DIST_PB, 32> dist;
    Code<7, 19> cl;
    uint8_t lens[32 + 320];
    uint16_t scratch[2 * 288];
};

// One wave per chunk: blocks from start_bit on until a block boundary at or behind stop_bit, or the member's final block
// (pgz.h: decode_from).  Output: the chunk's symbols behind WIN window slots that hold the markers 256 + w.
__global__ __launch_bounds__(64) void k_inflate(const uint32_t *__restrict__ in, uint64_t in_bytes, const ChunkIn *__restrict__ chunks,
                                                ChunkRes *__restrict__ res, uint16_t *__restrict__ sym_arena, uint32_t n_chunks)
{
    __shared__ WaveLds S;
    const uint32_t lane = threadIdx.x;
    const uint32_t ci = blockIdx.x;
    if (ci >= n_chunks) return;
    const ChunkIn C = chunks[ci];
    uint16_t *const out = sym_arena + C.sym_off;
    const uint64_t cap = C.sym_cap - 64;        // (up to 63 collected literals are put out without another look at the bound)
    // (loops over lanes' elements count in wave-uniform steps: a loop whose exit depends on the lane makes the compiler treat the
    // decoder's whole state as divergent -- vector registers and exec-mask branches instead of the scalar unit)
    for (uint32_t w0 = 0; w0 < WIN; w0 += 64) out[w0 + lane] = (uint16_t)(256 + w0 + lane);
    uint64_t n = WIN;                   // symbols in memory
    uint32_t pend = 0;                  // literals collected in `lit_buf` (lane k holds the k-th), not yet in memory
    uint32_t lit_buf = 0;
    const uint64_t size_bits = in_bytes * 8;
    Bits B;
    B.w = in;
    B.ring = s_in_ring;
    B.n_words = ((in_bytes + 15) / 16) * 4;          // (whole 16-byte pieces: the buffer is padded with zeros beyond in_bytes)
    B.seek(C.start_bit);
    uint32_t status = ST_OK, final_block = 0;
    uint64_t end_bit = C.start_bit;

    auto flush = [&]() {
        if (pend) {
            if (lane < pend) out[n + lane] = (uint16_t)lit_buf;
            LANES_DONE;
            n += pend;
            pend = 0;
        }
    };
    // A match of up to 64 symbols is not stored when it is decoded: its sources are loaded (one per lane), the wave goes on
    // decoding, and the values are stored when the NEXT match has issued its loads -- a load from the 64 KB behind the write
    // position is a memory round trip of a microsecond, and in sequence data nearly every symbol belongs to a short match
    // (any six letters of four have occurred in the last 32 KiB).  A match whose source reaches into the pending one's
    // destination makes it land first.
    uint32_t p_len = 0, p_val = 0;
    uint64_t p_dst = 0;
    auto commit = [&]() {
        if (p_len) {
            if (lane < p_len) out[p_dst + lane] = (uint16_t)p_val;
            LANES_DONE;
            p_len = 0;
        }
    };

    for (;;) {
        const uint64_t pos = B.bitpos();
        if (pos >= C.stop_bit) { end_bit = pos; break; }
        if (pos + 3 > size_bits) { status = ST_INPUT_END; break; }
        B.refill();
        const uint32_t bfinal = B.peek(1);
        B.drop(1);
        const uint32_t btype = B.peek(2);
        B.drop(2);
        if (btype == 3) { status = ST_INVALID; break; }
        if (btype == 0) {
            B.drop(B.bc & 7);
            B.refill();
            const uint32_t len = B.peek(16);
            B.drop(16);
            B.refill();
            const uint32_t nlen = B.peek(16);
            B.drop(16);
            if ((len ^ nlen) != 0xFFFFu) { status = ST_INVALID; break; }
            const uint64_t src = B.bitpos() >> 3;
            if (src + len > in_bytes) { status = ST_INPUT_END; break; }
            flush();
            commit();
            if (n + len > cap) { status = ST_OVERFLOW; break; }
            const uint8_t *sb = reinterpret_cast<const uint8_t *>(in) + src;
            for (uint32_t i0 = 0; i0 < len; i0 += 64)
                if (i0 + lane < len) out[n + i0 + lane] = sb[i0 + lane];
                LANES_DONE;
            n += len;
            B.seek((src + len) * 8);
        } else {
            // ---- the block's two codes
            if (btype == 1) {
                for (uint32_t i0 = 0; i0 < 288; i0 += 64) {
                    const uint32_t i = i0 + lane;
                    if (i < 288) S.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8);
                    LANES_DONE;
                }
                __builtin_amdgcn_wave_barrier();
                (void)uni((uint32_t)build_code(S.lit, S.lens, 288, S.scratch));
                if (lane < 32) S.lens[lane] = 5;
                LANES_DONE;
                __builtin_amdgcn_wave_barrier();
                (void)uni((uint32_t)build_code(S.dist, S.lens, 32, S.scratch));
            } else {
                B.refill();
                const uint32_t hlit = B.peek(5) + 257;
                B.drop(5);
                const uint32_t hdist = B.peek(5) + 1;
                B.drop(5);
                const uint32_t hclen = B.peek(4) + 4;
                B.drop(4);
                if (hlit > 286 || hdist > 30) { status = ST_INVALID; break; }
                if (lane < 19) S.lens[lane] = 0;
                LANES_DONE;
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = 0; i < hclen; ++i) {
                    B.refill();
                    const uint32_t v = B.peek(3);
                    B.drop(3);
                    if (lane == 0) S.lens[c_clord[i]] = (uint8_t)v;
                    LANES_DONE;
                }
                __builtin_amdgcn_wave_barrier();
                if ((int)uni((uint32_t)build_code(S.cl, S.lens, 19, S.scratch)) < 0) { status = ST_INVALID; break; }
                __builtin_amdgcn_wave_barrier();
                const uint32_t total = hlit + hdist;
                uint32_t k = 0, prev = 0;
                bool bad = false;
                while (k < total) {
                    B.refill();
                    const uint32_t e = decode_sym(S.cl, B.bb);
                    const int l = (int)(e & 0xFF);
                    if (!l) { bad = true; break; }
                    B.drop(l);
                    const uint32_t s = e >> 8;
                    if (s < 16) {
                        if (lane == 0) S.lens[32 + k] = (uint8_t)s;
                        LANES_DONE;
                        prev = s;
                        ++k;
                        continue;
                    }
                    uint32_t rep, val = 0;
                    if (s == 16) {
                        if (k == 0) { bad = true; break; }
                        val = prev;
                        rep = 3 + B.peek(2);
                        B.drop(2);
                    } else if (s == 17) {
                        rep = 3 + B.peek(3);
                        B.drop(3);
                    } else {
                        rep = 11 + B.peek(7);
                        B.drop(7);
                    }
                    if (k + rep > total) { bad = true; break; }
                    if (lane < rep) S.lens[32 + k + lane] = (uint8_t)val;
                    LANES_DONE;
                    if (lane + 64 < rep) S.lens[32 + k + lane + 64] = (uint8_t)val;
                    LANES_DONE;
                    if (lane + 128 < rep) S.lens[32 + k + lane + 128] = (uint8_t)val;
                    LANES_DONE;
                    prev = val;
                    k += rep;
                }
                if (bad || B.bitpos() > size_bits) { status = bad ? ST_INVALID : ST_INPUT_END; break; }
                __builtin_amdgcn_wave_barrier();
                if (uni(S.lens[32 + 256]) == 0) { status = ST_INVALID; break; }
                if ((int)uni((uint32_t)build_code(S.lit, S.lens + 32, (int)hlit, S.scratch)) < 0) { status = ST_INVALID; break; }
                if ((int)uni((uint32_t)build_code(S.dist, S.lens + 32 + hlit, (int)hdist, S.scratch)) < 0) { status = ST_INVALID; break; }
            }
            // ---- the block's symbols
            bool done = false;
            while (!done) {
                B.refill();
                uint32_t e = decode_sym(S.lit, B.bb);
                int l = (int)(e & 0xFF);
                uint32_t s = e >> 8;
                if (!l) { status = ST_INVALID; break; }
                B.drop(l);
                if (s < 256) {
                    if (lane == pend) lit_buf = s;
                    if (++pend == 64) {
                        if (n + 64 > cap) { status = ST_OVERFLOW; break; }
                        out[n + lane] = (uint16_t)lit_buf;
                        n += 64;
                        pend = 0;
                    }
                    continue;
                }
                if (s == 256) { done = true; break; }
                s -= 257;
                if (s >= 29) { status = ST_INVALID; break; }
                B.refill();
                // (RFC 1951's length and distance tables, computed: a table in memory is a scalar load's latency per look)
                const int lx = s < 8 || s == 28 ? 0 : (int)((s - 4) >> 2);
                const uint32_t len = (s == 28 ? 258u : s < 8 ? 3u + s : 3u + ((4u + (s & 3u)) << lx)) + B.peek(lx);
                B.drop(lx);
                const uint32_t de = decode_sym(S.dist, B.bb);
                const int dl = (int)(de & 0xFF);
                if (!dl) { status = ST_INVALID; break; }
                B.drop(dl);
                const uint32_t ds = de >> 8;
                if (ds >= 30) { status = ST_INVALID; break; }
                B.refill();
                const int dx = ds < 4 ? 0 : (int)((ds - 2) >> 1);
                const uint32_t d = (ds < 4 ? 1u + ds : 1u + ((2u + (ds & 1u)) << dx)) + B.peek(dx);
                B.drop(dx);
                flush();
                if (d > n) { status = ST_INVALID; break; }
                if (n + len > cap) { status = ST_OVERFLOW; break; }
                if (B.bitpos() > size_bits) { status = ST_INPUT_END; break; }
                // out[n + i] = out[n - d + (i mod d)]: every source lies in what is written already, so the lanes copy side by side
                const uint64_t src0 = n - d, src1 = src0 + (d < len ? d : len);
                if (p_len && src0 < p_dst + p_len && src1 > p_dst) commit();
                const uint16_t *srcp = out + src0;
                if (len <= 64) {
                    uint32_t v = 0;
                    if (lane < len) v = srcp[d >= len ? lane : lane % d];
                    LANES_DONE;
                    commit();                                     // (the previous match: its loads are long back)
                    p_val = v;
                    p_dst = n;
                    p_len = len;
                } else {
                    commit();
                    for (uint32_t i0 = 0; i0 < len; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        if (i < len) out[n + i] = srcp[d >= len ? i : i % d];
                        LANES_DONE;
                    }
                }
                n += len;
            }
            if (status != ST_OK) break;
            if (B.bitpos() > size_bits) { status = ST_INPUT_END; break; }
        }
        if (bfinal) { final_block = 1; end_bit = B.bitpos(); break; }
    }
    if (status == ST_OK) {
        commit();
        if (n + pend > cap) status = ST_OVERFLOW;
        else flush();
    }
    if (lane == 0) {
        ChunkRes r;
        r.end_bit = end_bit;
        r.n_sym = n;
        r.status = status;
        r.final_block = final_block;
        res[ci] = r;
    }
}

// the windows, chunk after chunk: win[c + 1] = the last WIN bytes of (win[c] ++ chunk c resolved).  One block.
__global__ __launch_bounds__(1024) void k_chain_windows(const ChunkIn *__restrict__ chunks, const ChunkRes *__restrict__ res,
                                                       const uint16_t *__restrict__ sym_arena, uint8_t *__restrict__ win, uint32_t first, uint32_t count)
{
    for (uint32_t c = first; c < first + count; ++c) {
        const uint16_t *s = sym_arena + chunks[c].sym_off;
        const uint64_t n = res[c].n_sym - WIN;          // the chunk's own symbols
        const uint8_t *w = win + (size_t)(c - first) * WIN;
        uint8_t *nw = win + (size_t)(c - first + 1) * WIN;
        for (uint32_t k = threadIdx.x; k < WIN; k += blockDim.x) {
            const uint64_t pos = n + k;                  // index into (window ++ chunk)
            uint8_t v;
            if (pos < WIN) v = w[pos];
            else {
                const uint16_t x = s[pos];
                v = x < 256 ? (uint8_t)x : w[x - 256];
            }
            nw[k] = v;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// every symbol of every chunk -> its byte, packed chunk after chunk at byte_off[c]
__global__ __launch_bounds__(256) void k_resolve(const ChunkIn *__restrict__ chunks, const ChunkRes *__restrict__ res, const uint16_t *__restrict__ sym_arena,
                                                 const uint8_t *__restrict__ win, const uint64_t *__restrict__ byte_off, uint8_t *__restrict__ out,
                                                 uint32_t first, uint32_t count, uint32_t blocks_per_chunk)
{
    const uint32_t c = first + blockIdx.x / blocks_per_chunk, part = blockIdx.x % blocks_per_chunk;
    if (c >= first + count) return;
    const uint16_t *s = sym_arena + chunks[c].sym_off + WIN;
    const uint64_t n = res[c].n_sym - WIN;
    const uint8_t *w = win + (size_t)(c - first) * WIN;
    uint8_t *o = out + byte_off[c - first];
    // four symbols per thread and step: 8 bytes in, 4 bytes out
    const uint64_t quads = n / 4;
    for (uint64_t q = (uint64_t)part * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)blocks_per_chunk * blockDim.x) {
        ushort4 v;
        memcpy(&v, s + q * 4, 8);                       // (chunk starts are 8-byte aligned in the arena)
        uchar4 b;
        b.x = v.x < 256 ? (uint8_t)v.x : w[v.x - 256];
        b.y = v.y < 256 ? (uint8_t)v.y : w[v.y - 256];
        b.z = v.z < 256 ? (uint8_t)v.z : w[v.z - 256];
        b.w = v.w < 256 ? (uint8_t)v.w : w[v.w - 256];
        memcpy(o + q * 4, &b, 4);                       // (byte_off is a multiple of 4)
    }
    if (part == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = quads * 4 + threadIdx.x;
        const uint16_t x = s[i];
        o[i] = x < 256 ? (uint8_t)x : w[x - 256];
    }
}

} // namespace

struct taxor_gpu_inflater {
    int device = 0;
    hipStream_t st = nullptr;
    uint8_t *d_in = nullptr;
    uint64_t in_cap = 0;
    uint16_t *d_sym = nullptr;
    uint64_t sym_cap = 0;           // symbols
    ChunkIn *d_chunks = nullptr;
    ChunkRes *d_res = nullptr;
    uint32_t max_chunks = 0;
    uint8_t *d_win = nullptr;       // (max_chunks + 1) windows
    uint64_t *d_boff = nullptr;
    uint8_t *d_out = nullptr;
    uint64_t out_cap = 0;
    // the batch in flight
    std::vector<ChunkIn> chunks;
    std::vector<ChunkRes> res;
    ChunkRes *h_res = nullptr;      // page-locked: the results come back without the host waiting for them
    bool in_flight = false;
    std::chrono::steady_clock::time_point t_begin, t_launch;
    uint32_t n = 0;
    uint64_t in_bytes = 0;
};

extern "C" int taxor_gpu_inflater_create(int device, uint64_t max_in_bytes, uint32_t max_chunks, uint64_t max_symbols, taxor_gpu_inflater **out)
{
    if (!out || !max_chunks || !max_in_bytes || max_symbols < (uint64_t)max_chunks * (WIN + 256)) return fail(TAXOR_E_ARG, "taxor_gpu_inflater_create: bad sizes");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    auto *h = new taxor_gpu_inflater;
    h->device = device;
    h->max_chunks = max_chunks;
    h->in_cap = (max_in_bytes + 64 + 3) & ~3ull;
    h->sym_cap = max_symbols;
    h->out_cap = max_symbols - (uint64_t)max_chunks * WIN + 8ull * max_chunks;
    hipError_t e = hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_in, h->in_cap);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_sym, h->sym_cap * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_chunks, sizeof(ChunkIn) * max_chunks);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_res, sizeof(ChunkRes) * max_chunks);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_win, (size_t)WIN * (max_chunks + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_boff, sizeof(uint64_t) * (max_chunks + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_out, h->out_cap);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_res, sizeof(ChunkRes) * max_chunks, hipHostMallocDefault);
    if (e != hipSuccess) {
        const std::string msg = hipGetErrorString(e);
        taxor_gpu_inflater_destroy(h);
        return fail(TAXOR_E_HIP, "taxor_gpu_inflater_create: %s (%.1f GB of symbols asked for)", msg.c_str(), max_symbols * 2 / 1e9);
    }
    *out = h;
    return TAXOR_OK;
}

extern "C" void taxor_gpu_inflater_destroy(taxor_gpu_inflater *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->st) { (void)hipStreamSynchronize(h->st); (void)hipStreamDestroy(h->st); }
    for (void *p : {(void *)h->d_in, (void *)h->d_sym, (void *)h->d_chunks, (void *)h->d_res, (void *)h->d_win, (void *)h->d_boff, (void *)h->d_out})
        if (p) (void)hipFree(p);
    if (h->h_res) (void)hipHostFree(h->h_res);
    delete h;
}

extern "C" int taxor_gpu_inflate_decode(taxor_gpu_inflater *h, const uint8_t *in, uint64_t in_bytes, const taxor_inflate_chunk *chunks, uint32_t n,
                                        taxor_inflate_result *results)
{
    const int rc = taxor_gpu_inflate_decode_begin(h, in, in_bytes, chunks, n);
    return rc != TAXOR_OK ? rc : taxor_gpu_inflate_decode_end(h, results);
}

extern "C" int taxor_gpu_inflate_decode_begin(taxor_gpu_inflater *h, const uint8_t *in, uint64_t in_bytes, const taxor_inflate_chunk *chunks, uint32_t n)
{
    if (!h || !in || !chunks) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_decode: null argument");
    // (a batch that was begun and never ended is abandoned: this one is queued behind it on the same stream)
    if (n > h->max_chunks || in_bytes + 64 > h->in_cap) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_decode: the batch is larger than the inflater was created for");
    HIP_TRY(hipSetDevice(h->device));
    h->n = n;
    h->in_bytes = in_bytes;
    h->chunks.resize(n);
    h->res.resize(n);
    h->in_flight = true;
    if (!n) return TAXOR_OK;
    // the arena is shared out by the chunks' compressed lengths (a chunk's output is, to first order, its input times the file's ratio)
    uint64_t total_in = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (chunks[i].stop_bit < chunks[i].start_bit || chunks[i].start_bit > in_bytes * 8) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_decode: chunk %u lies outside the input", i);
        total_in += chunks[i].weight + 8;
    }
    // (an eighth stays free behind the last chunk: where taxor_gpu_inflate_replace puts a chunk that outgrew its share)
    const uint64_t spare = (h->sym_cap - (uint64_t)n * (WIN + 256)) / 8 * 7;
    uint64_t off = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t share = (uint64_t)((double)(chunks[i].weight + 8) / (double)total_in * (double)spare);
        h->chunks[i].start_bit = chunks[i].start_bit;
        h->chunks[i].stop_bit = chunks[i].stop_bit;
        h->chunks[i].sym_off = off;
        h->chunks[i].sym_cap = (WIN + 128 + share) & ~3ull;
        off += h->chunks[i].sym_cap;
    }
    h->t_begin = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(h->d_in, in, in_bytes, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemsetAsync(h->d_in + in_bytes, 0, h->in_cap - in_bytes, h->st));
    HIP_TRY(hipMemcpyAsync(h->d_chunks, h->chunks.data(), sizeof(ChunkIn) * n, hipMemcpyHostToDevice, h->st));
    h->t_launch = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_inflate, dim3(n), dim3(64), 0, h->st, reinterpret_cast<const uint32_t *>(h->d_in), in_bytes, h->d_chunks, h->d_res, h->d_sym, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h->h_res, h->d_res, sizeof(ChunkRes) * n, hipMemcpyDeviceToHost, h->st));
    return TAXOR_OK;
}

extern "C" int taxor_gpu_inflate_decode_end(taxor_gpu_inflater *h, taxor_inflate_result *results)
{
    if (!h || !results) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_decode_end: null argument");
    if (!h->in_flight) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_decode_end: no batch was begun");
    h->in_flight = false;
    const uint32_t n = h->n;
    if (!n) return TAXOR_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->st));
    memcpy(h->res.data(), h->h_res, sizeof(ChunkRes) * n);
    static const bool trace = taxor::tune_env("TAXOR_INFLATE_TRACE") != nullptr;
    if (trace) {
        const auto t2 = std::chrono::steady_clock::now();
        uint64_t syms = 0;
        for (uint32_t i = 0; i < n; ++i) syms += h->res[i].n_sym - WIN;
        fprintf(stderr, "[inflate] batch of %u chunks, %.1f MB in: upload %.1f ms (host), launch to end %.1f ms for %.1f M symbols = %.2f G symbols/s\n", n, h->in_bytes / 1e6,
                std::chrono::duration<double, std::milli>(h->t_launch - h->t_begin).count(), std::chrono::duration<double, std::milli>(t2 - h->t_launch).count(), syms / 1e6,
                syms / 1e9 / std::chrono::duration<double>(t2 - h->t_launch).count());
    }
    for (uint32_t i = 0; i < n; ++i) {
        results[i].end_bit = h->res[i].end_bit;
        results[i].n_out = h->res[i].n_sym - WIN;
        results[i].status = h->res[i].status;
        results[i].final_block = h->res[i].final_block;
    }
    return TAXOR_OK;
}

extern "C" int taxor_gpu_inflate_replace(taxor_gpu_inflater *h, uint32_t chunk, const uint16_t *symbols, uint64_t n_out, uint64_t end_bit, uint32_t final_block)
{
    if (!h || chunk >= h->n || (!symbols && n_out)) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_replace: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    // the chunk's place in the arena may be too small for what the host decoded: the free space behind the last chunk is used then
    ChunkIn &c = h->chunks[chunk];
    if (WIN + n_out > c.sym_cap) {
        uint64_t end = 0;
        for (uint32_t i = 0; i < h->n; ++i) end = std::max(end, h->chunks[i].sym_off + h->chunks[i].sym_cap);
        end = (end + 3) & ~3ull;
        if (end + WIN + n_out + 68 > h->sym_cap) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_replace: no room for a chunk of %llu symbols", (unsigned long long)n_out);
        c.sym_off = end;
        c.sym_cap = (WIN + n_out + 64 + 3) & ~3ull;
        HIP_TRY(hipMemcpyAsync(h->d_chunks + chunk, &c, sizeof c, hipMemcpyHostToDevice, h->st));
    }
    HIP_TRY(hipMemcpyAsync(h->d_sym + c.sym_off + WIN, symbols, n_out * 2, hipMemcpyHostToDevice, h->st));
    ChunkRes r;
    r.end_bit = end_bit;
    r.n_sym = WIN + n_out;
    r.status = ST_OK;
    r.final_block = final_block;
    h->res[chunk] = r;
    HIP_TRY(hipMemcpyAsync(h->d_res + chunk, &h->res[chunk], sizeof r, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    return TAXOR_OK;
}

extern "C" int taxor_gpu_inflate_resolve(taxor_gpu_inflater *h, const uint8_t *window_in, uint32_t first, uint32_t count, uint8_t *const *out,
                                         uint8_t *window_out)
{
    if (!h || !window_in || !out || first + (uint64_t)count > h->n) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_resolve: bad argument");
    if (!count) return TAXOR_OK;
    HIP_TRY(hipSetDevice(h->device));
    std::vector<uint64_t> boff(count + 1, 0);
    for (uint32_t i = 0; i < count; ++i) {
        if (h->res[first + i].status != ST_OK) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_resolve: chunk %u was not decoded", first + i);
        boff[i + 1] = boff[i] + ((h->res[first + i].n_sym - WIN + 7) & ~7ull);
    }
    if (boff[count] > h->out_cap) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_resolve: output larger than the inflater's buffer");
    static const bool trace = taxor::tune_env("TAXOR_INFLATE_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(h->d_win, window_in, WIN, hipMemcpyHostToDevice, h->st));
    HIP_TRY(hipMemcpyAsync(h->d_boff, boff.data(), sizeof(uint64_t) * (count + 1), hipMemcpyHostToDevice, h->st));
    hipLaunchKernelGGL(k_chain_windows, dim3(1), dim3(1024), 0, h->st, h->d_chunks, h->d_res, h->d_sym, h->d_win, first, count);
    const uint32_t bpc = 32;
    hipLaunchKernelGGL(k_resolve, dim3(count * bpc), dim3(256), 0, h->st, h->d_chunks, h->d_res, h->d_sym, h->d_win, h->d_boff, h->d_out, first, count, bpc);
    HIP_TRY(hipGetLastError());
    if (trace) HIP_TRY(hipStreamSynchronize(h->st));
    const auto t1 = std::chrono::steady_clock::now();
    for (uint32_t i = 0; i < count; ++i) {
        const uint64_t nb = h->res[first + i].n_sym - WIN;
        if (nb) HIP_TRY(hipMemcpyAsync(out[i], h->d_out + boff[i], nb, hipMemcpyDeviceToHost, h->st));
    }
    if (window_out) HIP_TRY(hipMemcpyAsync(window_out, h->d_win + (size_t)count * WIN, WIN, hipMemcpyDeviceToHost, h->st));
    HIP_TRY(hipStreamSynchronize(h->st));
    if (trace)
        fprintf(stderr, "[inflate] %u chunks resolved: windows + symbols %.1f ms, %.1f MB to the host %.1f ms\n", count,
                std::chrono::duration<double, std::milli>(t1 - t0).count(), boff[count] / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return TAXOR_OK;
}

// test hook: a decoded chunk's symbols (without the window slots)
extern "C" int taxor_gpu_inflate_symbols(taxor_gpu_inflater *h, uint32_t chunk, uint16_t *out)
{
    if (!h || chunk >= h->n || !out) return fail(TAXOR_E_ARG, "taxor_gpu_inflate_symbols: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(out, h->d_sym + h->chunks[chunk].sym_off + WIN, (h->res[chunk].n_sym - WIN) * 2, hipMemcpyDeviceToHost));
    return TAXOR_OK;
}
