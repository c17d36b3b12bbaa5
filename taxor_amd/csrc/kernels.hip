// kernels.hip -- hand-written gfx950 kernels of the `taxor search` hot path.
//
//   k_pack_dna4     ASCII -> dna4 mapping -> 2-bit packing              (dna4_traits.hpp:15-18)
//   k_syncmers      open canonical syncmer selection + wyhash + per-read dedup + threshold
//                   (src/hashing/syncmer.cpp:80-165, taxor_search.cpp:221-237,261-263)
//   k_query_level   one level of the HIXF traversal: IXF bulk_count for every (read, IXF) work item of the
//                   level, split-bin tally, threshold test, emission of child work items and tuples
//                   (hierarchical_interleaved_xor_filter.hpp:303-340)
//   k_scan/k_scatter/k_sort_*   CSR assembly in the reference's DFS emission order
//
// The path is integer hashing + row gathers + byte compares: HBM-bound, no MFMA.  Lanes run across the
// bins of a fingerprint row (16 bins = one aligned 16-B unit per lane), so every probe is three coalesced
// row segments; hashes run in the loop; per-bin counters live in registers as packed bytes.
#include "kernels.h"
#include "ixf_arith.h"
#include "tuning.h"

#include <algorithm>
#include <cstdlib>

namespace taxor {

static constexpr int BLK = 256;

// ------------------------------------------------------------------------------------------------------
// small wave / block primitives (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ int wave_incl_max(int v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d);
        if ((int)lane_id() >= d) v = max(v, t);
    }
    return v;
}

// exclusive prefix sum over the 256 threads of a block; *total = block sum.  scratch: >= 4 words.
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t *scratch, uint32_t *total)
{
    const uint32_t incl = wave_incl_add(v);
    const uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63) scratch[w] = incl;
    __syncthreads();
    uint32_t off = 0, tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < BLK / 64; ++i) {
        const uint32_t x = scratch[i];
        if (i < w) off += x;
        tot += x;
    }
    *total = tot;
    return off + incl - v;
}

// exclusive prefix max over the 256 threads (identity -1)
__device__ __forceinline__ int block_excl_max(int v, int *scratch)
{
    const int incl = wave_incl_max(v);
    const uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63) scratch[w] = incl;
    __syncthreads();
    int off = -1;
#pragma unroll
    for (uint32_t i = 0; i < BLK / 64; ++i)
        if (i < w) off = max(off, scratch[i]);
    int prev = __shfl_up(incl, 1);
    if (lane_id() == 0) prev = -1;
    return max(off, prev);
}

// append one record per participating lane with a single atomic per wave (ballot + popcount).
// Returns the slot for lanes with pred, undefined otherwise.
__device__ __forceinline__ uint32_t wave_append(bool pred, uint32_t *counter)
{
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return 0;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane_id()) - 1ull));
}

// ------------------------------------------------------------------------------------------------------
// k_pack_dna4 : one block per read (grid-stride), one thread per 16-base word
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dna4_code(uint8_t c)
{
    // seqan3 dna4 char_to_rank: IUPAC codes -> first base, U -> T, N -> A; both cases. 0xFF = not dna15.
    switch (c | 0x20) {
    case 'a': case 'r': case 'w': case 'm': case 'd': case 'h': case 'v': case 'n': return 0;
    case 'c': case 'y': case 's': case 'b': return 1;
    case 'g': case 'k': return 2;
    case 't': case 'u': return 3;
    default: return 0xFFu;
    }
}

__global__ __launch_bounds__(BLK) void k_pack_dna4(const uint8_t *__restrict__ ascii,
                                                   const uint64_t *__restrict__ aoff,
                                                   const uint64_t *__restrict__ poff,
                                                   uint32_t *__restrict__ packed, uint32_t n_reads,
                                                   Counters *ctr)
{
    __shared__ uint8_t sLut[256]; // char -> 2-bit code, 0xFF = not a dna15 letter
    sLut[threadIdx.x] = (uint8_t)(((uint8_t)((threadIdx.x | 0x20u) - 'a') < 26u) ? dna4_code((uint8_t)threadIdx.x) : 0xFFu);
    __syncthreads();
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint64_t a0 = aoff[r];
        const uint32_t len = (uint32_t)(aoff[r + 1] - a0);
        const uint32_t nw = ((len + 15u) >> 4);
        const uint32_t nw_pad = (nw + 3u) & ~3u; // reads start on a 4-word (64-base) boundary
        uint32_t *dst = packed + poff[r];
        bool bad = false;
        for (uint32_t w = threadIdx.x; w < nw_pad; w += BLK) {
            uint32_t word = 0;
            if (w < nw) {
                // 16 characters = five aligned dwords funnel-shifted by the read's byte misalignment (the ASCII
                // buffer has slack past its end, so the fifth dword is always readable)
                const uint64_t A = a0 + ((uint64_t)w << 4);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(ascii + (A & ~3ull));
                const uint32_t sh = (uint32_t)(A & 3ull) * 8u;
                uint32_t q[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) q[j] = src[j];
                const uint32_t b0 = w << 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t x = sh ? ((q[j] >> sh) | (q[j + 1] << (32u - sh))) : q[j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t pos = b0 + 4u * (uint32_t)j + (uint32_t)c;
                        if (pos < len) {
                            uint32_t code = sLut[(x >> (8 * c)) & 0xFFu];
                            if (code == 0xFFu) { bad = true; code = 0; }
                            word |= code << (30u - 2u * (4u * (uint32_t)j + (uint32_t)c));
                        }
                    }
                }
            }
            dst[w] = word;
        }
        if (__any(bad) && lane_id() == 0) atomicOr(&ctr->flags, FLAG_ALPHABET);
    }
}

void launch_pack_dna4(const uint8_t *ascii, const uint64_t *aoff, const uint64_t *poff, uint32_t *packed,
                      uint32_t n_reads, Counters *ctr, hipStream_t st, int max_grid)
{
    if (!n_reads) return;
    int grid = (int)(n_reads < 8192u ? n_reads : 8192u);
    if (max_grid > 0 && grid > max_grid) grid = max_grid;
    hipLaunchKernelGGL(k_pack_dna4, dim3(grid), dim3(BLK), 0, st, ascii, aoff, poff, packed, n_reads, ctr);
}

// ------------------------------------------------------------------------------------------------------
// k_syncmers
//
// The reference walks the read base by base with a deque and a stateful tie rule (syncmer.cpp:116-140):
// first window -> leftmost minimum; when the tracked minimum leaves the window -> rescan from the right
// (rightmost minimum); otherwise a new s-mer replaces it only if strictly smaller.  Parallel form used
// here (exact, validated against the sequential code on tie-heavy reads): for window x over s-mer starts
// [x, x+w-1] let Lm/Rm be the leftmost/rightmost argmin.  If the minimum is unique the tracked position
// p_x is that argmin regardless of history.  Across a run of tied windows p only changes when it drops
// out of the window, and then becomes Rm of the first window that no longer contains it, i.e. the chain
// q -> Rm(q+1).  So p_x = walk that chain from the nearest earlier anchor (a unique-minimum window, or
// window 0 with p = Lm) until q >= x.  A window is an open syncmer iff p_x == x + t - 1 (syncmer.cpp:142).
// ------------------------------------------------------------------------------------------------------
static constexpr int SY_C = 8;                 // consecutive windows per thread in the resolve pass
static constexpr int SY_T = BLK * SY_C;        // windows per tile
static constexpr int SY_WORDS = SY_T / 16 + 8; // packed words staged per tile
static constexpr int SY_LDS_TAB = 4096;        // dedup slots held in LDS
static constexpr int SY_LDS_CAND = 2048;       // candidate hashes held in LDS (the rest spill to global)
static constexpr int SY_CHUNK_MAX = 8;         // reads taken per cursor atomic, at most

__device__ __forceinline__ uint32_t revcomp32(uint32_t x, int nb)
{
    x = __brev(~x);
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    return x >> (32 - 2 * nb);
}

__device__ __forceinline__ uint64_t revcomp64(uint64_t x, int nb)
{
    x = __brevll(~x);
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    return x >> (64 - 2 * nb);
}

// n-base value starting at base `pos` (tile-local word array W, wbase = first staged word), n <= 32
__device__ __forceinline__ uint64_t extract_bases(const uint32_t *W, uint32_t pos_word, uint32_t o, int n)
{
    const uint64_t hi = ((uint64_t)W[pos_word] << 32) | W[pos_word + 1];
    const int end = 2 * (int)o + 2 * n;
    const uint64_t mask = (n < 32) ? ((1ull << (2 * n)) - 1ull) : ~0ull;
    if (end <= 64) return (hi >> (64 - end)) & mask;
    const int sh = end - 64; // 1..30
    return ((hi << sh) | (uint64_t)(W[pos_word + 2] >> (32 - sh))) & mask;
}

__device__ __forceinline__ uint32_t dedup_slot(uint64_t h, uint32_t mask)
{
    h ^= h >> 31;
    h *= 0x9E3779B97F4A7C15ull;
    return (uint32_t)(h >> 32) & mask;
}

// FW > 0: fast path for a window of exactly FW s-mers (FW = 11 is taxor's k22/s12) with 2s+5 <= 32: each thread
// owns SY_C consecutive windows, pulls their FW+SY_C-1 s-mer values out of a transposed (bank-conflict-free) LDS
// tile once, and gets leftmost/rightmost argmin from two sparse-table min trees over keys (value<<5 | offset) and
// (value<<5 | 31-offset).  FW == 0: generic window length, per-window scan in LDS.
static constexpr int SY_RS = SY_T / SY_C + 5; // row stride (words) of the transposed s-mer tile; odd -> no bank conflicts

// PROF: the same kernel with s_memtime marks at its phase boundaries (block-uniform, scalar), summed per launch into
// a.prof -- a measurement aid compiled as a separate instantiation, the production kernel carries none of it.
#define PMARK(i)                                                                                        \
    if constexpr (PROF) {                                                                               \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();                                             \
        pacc[i] += now_ - plast;                                                                        \
        plast = now_;                                                                                   \
    }

template <int FW, bool PROF = false> __global__ __launch_bounds__(BLK) void k_syncmers(const SyncmerArgs a)
{
    uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    __shared__ uint32_t sW[SY_WORDS];
    __shared__ uint32_t sV[SY_C * SY_RS];
    __shared__ __attribute__((aligned(16))) uint8_t sLm[SY_T];
    __shared__ __attribute__((aligned(16))) uint8_t sRm[SY_T];
    __shared__ uint32_t sTab[SY_LDS_TAB];
    __shared__ uint64_t sCand[SY_LDS_CAND];
    __shared__ uint32_t sScr[8];
    __shared__ int sCarry;
    __shared__ uint32_t sRead;
    // Per-read metadata of a cursor chunk, fetched by one lane per read: the chain  cursor -> order -> (length, offsets,
    // capacity) -> packed words  is four dependent global round trips, which is most of what a SHORT read costs
    // (10 us per 1-kb read at two blocks per CU).  A chunk pays the first three once; the packed words of the next tile
    // -- of this read, or the first tile of the next read of the chunk -- are loaded into a register while the current
    // tile is processed, so staging them finds them there.
    struct ReadMeta { uint64_t poff, hoff; uint32_t r, L, cap, pad; };
    __shared__ ReadMeta sMeta[SY_CHUNK_MAX];
    static_assert(SY_WORDS <= BLK, "one staged word per thread");

    const int k = a.k, s = a.s, t = a.t;
    const int w = k - s + 1;
    const uint32_t smask = (s < 16) ? ((1u << (2 * s)) - 1u) : 0xFFFFFFFFu;
    const uint32_t tid = threadIdx.x;
    const uint32_t chunk = min(max(a.chunk, 1u), (uint32_t)SY_CHUNK_MAX);
    uint32_t ri = 0, chunk_n = 0;      // position in / reads of the current chunk (block-uniform)
    uint32_t nextW = 0;                // prefetched word tid of the next staging event
    bool have_next = false;            // block-uniform

    for (;;) {
        __syncthreads();
        if (ri == chunk_n) {
            if (tid == 0) sRead = atomicAdd(a.cursor, chunk);
            __syncthreads();
            const uint32_t first = sRead;
            if (first >= a.n_reads) break;
            chunk_n = min(chunk, a.n_reads - first);
            ri = 0;
            have_next = false;
            if (tid < chunk_n) {
                const uint32_t r_ = a.order ? a.order[first + tid] : first + tid;
                ReadMeta m;
                m.r = r_;
                m.L = a.rlen[r_];
                m.poff = a.poff[r_];
                m.hoff = a.hoff[r_];
                m.cap = a.hcap[r_];
                m.pad = 0;
                sMeta[tid] = m;
            }
            __syncthreads();
        }
        const uint32_t r = sMeta[ri].r;
        PMARK(0)                                                     // 0: work cursor + chunk metadata

        const uint32_t L = sMeta[ri].L;
        const uint32_t *__restrict__ pk = a.packed + sMeta[ri].poff;
        const uint32_t nwords = (((L + 15u) >> 4) + 3u) & ~3u;
        uint64_t *__restrict__ cand = a.cand + sMeta[ri].hoff;
        uint64_t *__restrict__ outh = a.hashes + sMeta[ri].hoff;
        const uint32_t cap = sMeta[ri].cap;
        const int nwin = (int)L - k + 1; // number of k-mer windows (<= 0: none)
        uint32_t n_sel = 0;              // block-uniform
        // first tile of the next read of this chunk (prefetched during this read's last tile)
        const bool nx_ok = ri + 1u < chunk_n && (int)sMeta[ri + 1u < chunk_n ? ri + 1u : ri].L - k + 1 > 0;
        const uint32_t *__restrict__ nx_pk = a.packed + sMeta[ri + 1u < chunk_n ? ri + 1u : ri].poff;
        const uint32_t nx_words = (((sMeta[ri + 1u < chunk_n ? ri + 1u : ri].L + 15u) >> 4) + 3u) & ~3u;
        ++ri;

        if (tid == 0) sCarry = 0;
        for (int x0 = 0; x0 < nwin; x0 += SY_T) {
            __syncthreads();
            // ---- stage packed words of the tile ---------------------------------------------------
            const uint32_t wbase = (uint32_t)x0 >> 4;
            if (tid < (uint32_t)SY_WORDS) {
                uint32_t wv = nextW;
                if (!have_next) {
                    const uint32_t wi = wbase + tid;
                    wv = wi < nwords ? pk[wi] : 0u;
                }
                sW[tid] = wv;
            }
            __syncthreads();
            // the words of the next staging event: in flight while this tile is processed
            have_next = false;
            if (x0 + SY_T < nwin) {
                const uint32_t wi = (((uint32_t)x0 + (uint32_t)SY_T) >> 4) + tid;
                if (tid < (uint32_t)SY_WORDS) nextW = wi < nwords ? pk[wi] : 0u;
                have_next = true;
            } else if (nx_ok) {
                if (tid < (uint32_t)SY_WORDS) nextW = tid < nx_words ? nx_pk[tid] : 0u;
                have_next = true;
            }
            PMARK(1)                                                 // 1: staging the packed words
            // ---- canonical s-mer values (syncmer.cpp:103-110; the s-mer "hash" is the raw 2-bit value)
            const int nv = min(SY_T + w - 1, (int)L - s + 1 - x0); // valid s-mer starts in this tile
            const int nw_tile = min(SY_T, nwin - x0);
            const int xs = (int)tid * SY_C;
            int last_anchor = -1;
            int lmr[SY_C], rmr[SY_C]; // fast path: argmin offsets of this thread's windows
            if constexpr (FW > 0) {
                for (int i = (int)tid; i < SY_T + FW - 1; i += BLK) {
                    uint32_t v = 0x07FFFFFFu; // above every canonical value of <= 13 bases, and (v << 5) still fits
                    if (i < nv) {
                        const uint32_t pos = (uint32_t)(x0 + i);
                        const uint32_t f = (uint32_t)extract_bases(sW, (pos >> 4) - wbase, pos & 15u, s) & smask;
                        const uint32_t rc = revcomp32(f, s);
                        v = min(f, rc);
                    }
                    sV[(i & (SY_C - 1)) * SY_RS + (i >> 3)] = v;
                }
                __syncthreads();
                PMARK(2)                                             // 2: canonical s-mer values -> LDS
                constexpr int NV = FW + SY_C - 1;
                uint32_t kl[NV], kr[NV];
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    const uint32_t v = sV[(c & (SY_C - 1)) * SY_RS + (int)tid + (c >> 3)];
                    kl[c] = (v << 5) | (uint32_t)c;
                    kr[c] = (v << 5) | (uint32_t)(31 - c);
                }
                // sparse-table minima: m2 over 2, m4 over 4, m8 over 8 consecutive keys; FW = 8 + (FW - 8)
                static_assert(FW >= 8 && FW <= 16, "fast path covers window lengths 8..16");
                uint32_t l2[NV - 1], r2[NV - 1], l4[NV - 3], r4[NV - 3], l8[SY_C + FW - 8], r8[SY_C + FW - 8];
#pragma unroll
                for (int c = 0; c < NV - 1; ++c) { l2[c] = min(kl[c], kl[c + 1]); r2[c] = min(kr[c], kr[c + 1]); }
#pragma unroll
                for (int c = 0; c < NV - 3; ++c) { l4[c] = min(l2[c], l2[c + 2]); r4[c] = min(r2[c], r2[c + 2]); }
#pragma unroll
                for (int c = 0; c < SY_C + FW - 8; ++c) { l8[c] = min(l4[c], l4[c + 4]); r8[c] = min(r4[c], r4[c + 4]); }
                uint64_t packL = 0, packR = 0;
#pragma unroll
                for (int c = 0; c < SY_C; ++c) {
                    // window c covers keys c .. c+FW-1 = [c, c+8) U [c+FW-8, c+FW)
                    const uint32_t ml = min(l8[c], l8[c + FW - 8]);
                    const uint32_t mr = min(r8[c], r8[c + FW - 8]);
                    lmr[c] = (int)(ml & 31u) - c;
                    rmr[c] = 31 - (int)(mr & 31u) - c;
                    packL |= (uint64_t)(uint32_t)lmr[c] << (8 * c);
                    packR |= (uint64_t)(uint32_t)rmr[c] << (8 * c);
                    const int xl = xs + c;
                    if (xl < nw_tile && (lmr[c] == rmr[c] || x0 + xl == 0)) last_anchor = xl;
                }
                *reinterpret_cast<uint64_t *>(&sLm[xs]) = packL;
                *reinterpret_cast<uint64_t *>(&sRm[xs]) = packR;
            } else {
                for (int i = (int)tid; i < SY_T + w - 1; i += BLK) {
                    uint32_t v = 0xFFFFFFFFu;
                    if (i < nv) {
                        const uint32_t pos = (uint32_t)(x0 + i);
                        const uint32_t f = (uint32_t)extract_bases(sW, (pos >> 4) - wbase, pos & 15u, s) & smask;
                        const uint32_t rc = revcomp32(f, s);
                        v = min(f, rc);
                    }
                    sV[i] = v;
                }
                __syncthreads();
                // ---- per window: leftmost / rightmost argmin (interleaved assignment: conflict-free LDS)
#pragma unroll 2
                for (int c = 0; c < SY_C; ++c) {
                    const int xl = c * BLK + (int)tid;
                    if (xl < nw_tile) {
                        uint32_t m = sV[xl];
                        int lm = 0, rm = 0;
                        for (int j = 1; j < w; ++j) {
                            const uint32_t v = sV[xl + j];
                            if (v < m) { m = v; lm = j; rm = j; }
                            else if (v == m) rm = j;
                        }
                        sLm[xl] = (uint8_t)lm;
                        sRm[xl] = (uint8_t)rm;
                    }
                }
                __syncthreads();
                // ---- resolve the tracked position per window (chunked assignment: SY_C consecutive windows)
                for (int c = 0; c < SY_C; ++c) {
                    const int xl = xs + c;
                    if (xl < nw_tile) {
                        lmr[c] = sLm[xl];
                        rmr[c] = sRm[xl];
                        if (lmr[c] == rmr[c] || x0 + xl == 0) last_anchor = xl;
                    }
                }
            }
            PMARK(3)                                                 // 3: window argmins (min trees / scan)
            const int anchor = block_excl_max(last_anchor, (int *)sScr);
            uint32_t selmask = 0;
            int p; // tile-local s-mer start currently tracked (may be -1: last position of the previous tile)
            if (xs < nw_tile) {
                p = anchor >= 0 ? anchor + (int)sLm[anchor] : sCarry - x0;
                while (p < xs) p = (p + 1) + (int)sRm[p + 1];
#pragma unroll
                for (int c = 0; c < SY_C; ++c) {
                    const int xl = xs + c;
                    if (xl < nw_tile) {
                        const int lm = lmr[c], rm = rmr[c];
                        if (lm == rm || x0 + xl == 0) p = xl + lm;
                        else if (p < xl) p = xl + rm;
                        if (p == xl + t - 1) selmask |= 1u << c;
                    }
                }
            }
            __syncthreads();
            if (xs < nw_tile && xs + SY_C >= nw_tile) sCarry = p + x0; // owner of the tile's last window
            PMARK(4)                                                 // 4: anchor scan + chain walk + selection
            // ---- emit wyhash(canonical k-mer) of the selected windows, in window order --------------
            uint32_t tot;
            uint32_t pos = n_sel + block_excl_add((uint32_t)__popc(selmask), sScr, &tot);
            while (selmask) {
                const int c = __ffs((int)selmask) - 1;
                selmask &= selmask - 1;
                const uint32_t x = (uint32_t)(x0 + xs + c);
                const uint64_t f = extract_bases(sW, (x >> 4) - wbase, x & 15u, k);
                const uint64_t rc = revcomp64(f, k);
                const uint64_t h = wyhash_u64(f < rc ? f : rc); // syncmer.cpp:144-145
                if (pos < cap) {
                    if (pos < (uint32_t)SY_LDS_CAND) sCand[pos] = h; // reads up to ~22 kb never touch global here
                    else cand[pos] = h;
                }
                ++pos;
            }
            n_sel += tot;
            PMARK(5)                                                 // 5: emit wyhash(canonical k-mer)
        }
        __syncthreads();
        if (n_sel > cap) { // capacity bound (nwin / min(t, w-t+1) + 2) violated: internal invariant
            if (tid == 0) atomicOr(&a.ctr->flags, FLAG_CAND_OVERFLOW);
            n_sel = cap;
        }

        // ---- per-read dedup (ankerl::unordered_dense::set semantics, syncmer.cpp:157-165): keep the first
        //      occurrence of every hash, preserve first-insertion order.
        //      The 4096-slot LDS table takes the candidates in P = ceil(n_sel / 1536) passes, pass p holding the
        //      hashes of partition p (by their high bits; equal hashes share a partition), so reads of any length up
        //      to ~750 kb dedup in LDS -- a per-read table in global memory costs one L2 atomic per candidate, and a
        //      sub-batch of long reads is then bound by the chip's atomic rate.  Duplicates are rare: a pass that
        //      meets none skips its lookup sweep, and a read without any is copied straight out. ---------------
        uint32_t n_dist = 0;
        if (n_sel > 0 && n_sel <= (uint32_t)BLK) {
            // short read (up to ~2.8 kb at k22/s12): one candidate per thread, compared with every earlier one.  The
            // loop is uniform and its LDS reads are broadcasts -- no table to clear, no atomics, one barrier; the hash
            // table below costs a short read several times its selection work.
            const uint64_t h = tid < n_sel ? sCand[tid] : 0ull;
            bool dupf = false;
            for (uint32_t j0 = 0; j0 + 1u < n_sel; j0 += 8u) {        // eight independent LDS reads per trip (a loop of
                uint64_t o[8];                                       // single dependent reads is all LDS latency);
#pragma unroll
                for (int u = 0; u < 8; ++u) o[u] = sCand[j0 + (uint32_t)u];   // j0 + 7 < 264 <= SY_LDS_CAND: in bounds
#pragma unroll
                for (int u = 0; u < 8; ++u) dupf |= (j0 + (uint32_t)u < tid) && (o[u] == h);   // j < tid < n_sel
            }
            bool first = tid < n_sel && !dupf;
            if (first && a.scaling_limit > 0.0 && !((double)wyhash_u64(h) <= a.scaling_limit)) first = false;   // taxor_search.cpp:223-233
            PMARK(6)
            if (!__syncthreads_or((tid < n_sel && !first) ? 1 : 0)) {
                if (tid < n_sel) outh[tid] = h;
                n_dist = n_sel;
            } else {
                uint32_t tot;
                const uint32_t rank = block_excl_add(first ? 1u : 0u, sScr, &tot);
                if (first) outh[rank] = h;
                n_dist = tot;
            }
        } else if (n_sel > 0) {
            // (two loads in two address spaces, each under its own branch: `c ? sCand[i] : cand[i]` lets the compiler select between
            //  the ADDRESSES, and a pointer that may be LDS or global is a flat one -- flat_load ties up lgkmcnt as well as vmcnt)
            typedef const uint64_t __attribute__((address_space(1))) *gcand_t;
            const gcand_t gcand = (gcand_t)(uintptr_t)cand;
            auto cand_at = [&](uint32_t i) -> uint64_t {
                uint64_t v;
                if (i < (uint32_t)SY_LDS_CAND) v = sCand[i];
                else v = gcand[i];
                return v;
            };
            uint32_t *const sDupBits = sV;                               // the s-mer tile is dead by now: 1 bit per candidate
            static_assert((uint32_t)(SY_C * SY_RS) * 32u == SYNC_LDS_DEDUP_MAX, "dup-bit capacity");
            const bool in_lds = n_sel <= SYNC_LDS_DEDUP_MAX;
            bool any_dup = false;                                        // block-uniform
            if (in_lds) {
                const uint32_t P = (n_sel + 1535u) / 1536u;
                uint32_t ts = 64;                                        // a short read clears and probes a short table
                while (ts < 2u * n_sel && ts < (uint32_t)SY_LDS_TAB) ts <<= 1;
                const uint32_t mask = ts - 1u;
                for (uint32_t i = tid; i < (n_sel + 31u) / 32u; i += BLK) sDupBits[i] = 0u;
                for (uint32_t p = 0; p < P; ++p) {
                    for (uint32_t i = tid; i < ts; i += BLK) sTab[i] = 0xFFFFFFFFu;
                    if (tid == 0) { sScr[6] = 0u; sScr[7] = 0u; }
                    __syncthreads();
                    bool dup = false;
                    uint32_t fresh = 0;
                    for (uint32_t i = tid; i < n_sel; i += BLK) {
                        const uint64_t h = cand_at(i);
                        if (P > 1u && __umulhi((uint32_t)(h >> 32), P) != p) continue;
                        uint32_t q = dedup_slot(h, mask);
                        for (;;) {
                            const uint32_t cur = atomicCAS(&sTab[q], 0xFFFFFFFFu, i);
                            if (cur == 0xFFFFFFFFu) { ++fresh; break; }
                            if (cand_at(cur) == h) { atomicMin(&sTab[q], i); dup = true; break; }
                            q = (q + 1u) & mask;
                        }
                    }
                    if (dup) sScr[6] = 1u;
                    {   // one LDS atomic per wave, not per lane (64 lanes on one address serialise)
                        uint32_t f = fresh;
#pragma unroll
                        for (int d = 32; d > 0; d >>= 1) f += __shfl_xor(f, d);
                        if (lane_id() == 0 && f) atomicAdd(&sScr[7], f);
                    }
                    __syncthreads();
                    const bool pass_dup = sScr[6] != 0u;
                    if (sScr[7] > 3584u && tid == 0) atomicOr(&a.ctr->flags, FLAG_DEDUP_OVERFLOW); // cannot happen for mixed hashes
                    if (pass_dup) {
                        any_dup = true;
                        for (uint32_t i = tid; i < n_sel; i += BLK) {
                            const uint64_t h = cand_at(i);
                            if (P > 1u && __umulhi((uint32_t)(h >> 32), P) != p) continue;
                            uint32_t q = dedup_slot(h, mask);
                            for (;;) {
                                const uint32_t cur = sTab[q];
                                if (cur == 0xFFFFFFFFu) break; // cannot happen for an inserted key
                                if (cand_at(cur) == h) {
                                    if (cur != i) atomicOr(&sDupBits[i >> 5], 1u << (i & 31u));
                                    break;
                                }
                                q = (q + 1u) & mask;
                            }
                        }
                    }
                    __syncthreads();
                }
            } else {
                // longer than ~750 kb: per-block table in global memory (one pass)
                uint32_t ts = 64;
                while (ts < 2u * n_sel) ts <<= 1;
                // (always the global table: an LDS stand-in for the overflow case would make `tab` a flat pointer)
                typedef uint32_t __attribute__((address_space(1))) *gtab_t;
                uint32_t *const tab = (uint32_t *)(gtab_t)(uintptr_t)(a.gtab + (size_t)blockIdx.x * a.gtab_stride);
                if (ts > a.gtab_stride) {
                    if (tid == 0) atomicOr(&a.ctr->flags, FLAG_DEDUP_OVERFLOW);
                    n_sel = 0;
                    ts = 0;
                }
                const uint32_t mask = ts - 1u;
                for (uint32_t i = tid; i < ts; i += BLK) tab[i] = 0xFFFFFFFFu;
                __syncthreads();
                for (uint32_t i = tid; i < n_sel; i += BLK) {
                    const uint64_t h = cand_at(i);
                    uint32_t q = dedup_slot(h, mask);
                    for (;;) {
                        const uint32_t cur = atomicCAS(&tab[q], 0xFFFFFFFFu, i);
                        if (cur == 0xFFFFFFFFu) break;
                        if (cand_at(cur) == h) { atomicMin(&tab[q], i); break; }
                        q = (q + 1u) & mask;
                    }
                }
                __syncthreads();
                any_dup = true;                                          // flags are evaluated from the table below
            }
            PMARK(6)                                                 // 6: dedup table passes
            if (!any_dup && !(a.scaling_limit > 0.0)) {
                for (uint32_t i = tid; i < n_sel; i += BLK) outh[i] = cand_at(i);
                n_dist = n_sel;
            } else {
                const uint32_t gmask = in_lds ? 0u : [&] { uint32_t ts = 64; while (ts < 2u * n_sel) ts <<= 1; return ts - 1u; }();
                const uint32_t *gtab = a.gtab + (size_t)blockIdx.x * a.gtab_stride;
                for (uint32_t base = 0; base < n_sel; base += BLK) {
                    const uint32_t i = base + tid;
                    uint64_t h = 0;
                    uint32_t first = 0;
                    if (i < n_sel) {
                        h = cand_at(i);
                        if (in_lds) first = ((sDupBits[i >> 5] >> (i & 31u)) & 1u) ^ 1u;
                        else {
                            uint32_t q = dedup_slot(h, gmask);
                            for (;;) {
                                const uint32_t cur = gtab[q];
                                if (cur == 0xFFFFFFFFu) break;
                                if (cand_at(cur) == h) { first = (cur == i); break; }
                                q = (q + 1u) & gmask;
                            }
                        }
                        // FracMinHash down-sampling of a scaled index: a pure function of the hash, so filtering the
                        // first occurrences equals filtering the reference's set (taxor_search.cpp:223-233)
                        if (first && a.scaling_limit > 0.0 && !((double)wyhash_u64(h) <= a.scaling_limit)) first = 0;
                    }
                    uint32_t tot;
                    const uint32_t rank = block_excl_add(first, sScr, &tot);
                    if (first) outh[n_dist + rank] = h;
                    n_dist += tot;
                }
            }
        }
        if (tid == 0) {
            a.nh[r] = n_dist;                                                   // taxor_search.cpp:261
            a.thr[r] = (uint64_t)((double)n_dist * a.ratio);                    // threshold.hpp:60,76-79
            atomicAdd(&a.ctr->n_hashes, (unsigned long long)n_dist);
        }
        PMARK(7)                                                     // 7: copy out / ordered compaction
    }
    if constexpr (PROF) {
        if (tid == 0 && a.prof)
            for (int i = 0; i < 8; ++i) atomicAdd(&a.prof[i], (unsigned long long)pacc[i]);
    }
}


// ------------------------------------------------------------------------------------------------------
// k_syncmers_wave -- the same selection for SHORT reads (at most WV_CAND candidate slots: ~2.5 kb at k22/s12), one
// wavefront per read, four reads per block at a time.  A 1-kb read fills less than half of the block kernel's
// 2048-window tile and pays ~ten block barriers and a four-deep chain of dependent global loads per read; here the
// tile is 512 windows (64 lanes x 8), scans are wave scans, the tracked position is carried in a register, dedup is the
// all-pairs comparison, and nothing ever waits for another wave.  LDS is private per wave; a wave's LDS operations
// complete in program order, so the only "barrier" is a compiler-level one.  Results are identical to k_syncmers
// (tests/test_gpu_parity.py runs both on the same reads).
// ------------------------------------------------------------------------------------------------------
static constexpr int WV_C = 8;                      // windows per lane
static constexpr int WV_T = 64 * WV_C;              // windows per tile
static constexpr int WV_WORDS = WV_T / 16 + 8;      // packed words per tile (<= 64: one per lane)
static constexpr int WV_RS = 64 + 5;                // row stride of the transposed s-mer tile (odd)
static constexpr int WV_PER_BLOCK = BLK / 64;

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int FW> __global__ __launch_bounds__(BLK) void k_syncmers_wave(const SyncmerArgs a)
{
    static_assert(FW >= 8 && FW <= 16 && WV_WORDS <= 64, "fast path only");
    __shared__ uint32_t sWall[WV_PER_BLOCK][WV_WORDS];
    __shared__ uint32_t sVall[WV_PER_BLOCK][WV_C * WV_RS];
    __shared__ __attribute__((aligned(16))) uint8_t sLmAll[WV_PER_BLOCK][WV_T];
    __shared__ __attribute__((aligned(16))) uint8_t sRmAll[WV_PER_BLOCK][WV_T];
    __shared__ uint64_t sCandAll[WV_PER_BLOCK][SYNC_WAVE_CAND];
    const uint32_t wid = threadIdx.x >> 6, lane = lane_id();
    uint32_t *sW = sWall[wid];
    uint32_t *sV = sVall[wid];
    uint8_t *sLm = sLmAll[wid];
    uint8_t *sRm = sRmAll[wid];
    uint64_t *sCand = sCandAll[wid];

    const int k = a.k, s = a.s, t = a.t;
    const uint32_t smask = (s < 16) ? ((1u << (2 * s)) - 1u) : 0xFFFFFFFFu;
    const uint32_t chunk = min(max(a.chunk, 1u), 16u);
    unsigned long long hashes_total = 0;

    for (;;) {
        uint32_t first = 0;
        if (lane == 0) first = atomicAdd(a.cursor, chunk);
        first = __shfl(first, 0);
        if (first >= a.n_reads) break;
        const uint32_t chunk_n = min(chunk, a.n_reads - first);
        // one lane per read of the chunk fetches its metadata; the others get it by shuffle
        uint32_t m_r = 0, m_L = 0, m_cap = 0, m_plo = 0, m_phi = 0, m_hlo = 0, m_hhi = 0;
        if (lane < chunk_n) {
            m_r = a.order ? a.order[first + lane] : first + lane;
            m_L = a.rlen[m_r];
            m_cap = a.hcap[m_r];
            const uint64_t po = a.poff[m_r], ho = a.hoff[m_r];
            m_plo = (uint32_t)po; m_phi = (uint32_t)(po >> 32);
            m_hlo = (uint32_t)ho; m_hhi = (uint32_t)(ho >> 32);
        }
        for (uint32_t ri = 0; ri < chunk_n; ++ri) {
            const uint32_t r = __shfl(m_r, (int)ri), L = __shfl(m_L, (int)ri), cap = min(__shfl(m_cap, (int)ri), (uint32_t)SYNC_WAVE_CAND);
            const uint64_t po = ((uint64_t)__shfl(m_phi, (int)ri) << 32) | __shfl(m_plo, (int)ri);
            const uint64_t ho = ((uint64_t)__shfl(m_hhi, (int)ri) << 32) | __shfl(m_hlo, (int)ri);
            const uint32_t *__restrict__ pk = a.packed + po;
            uint64_t *__restrict__ outh = a.hashes + ho;
            const uint32_t nwords = (((L + 15u) >> 4) + 3u) & ~3u;
            const int nwin = (int)L - k + 1;
            uint32_t n_sel = 0;     // wave-uniform
            int carry = 0;          // tracked position (absolute) of the previous tile's last window, wave-uniform

            for (int x0 = 0; x0 < nwin; x0 += WV_T) {
                wave_lds_sync();
                const uint32_t wbase = (uint32_t)x0 >> 4;
                if (lane < (uint32_t)WV_WORDS) {
                    const uint32_t wi = wbase + lane;
                    sW[lane] = wi < nwords ? pk[wi] : 0u;
                }
                wave_lds_sync();
                const int nv = min(WV_T + FW - 1, (int)L - s + 1 - x0);
                const int nw_tile = min(WV_T, nwin - x0);
                for (int i = (int)lane; i < WV_T + FW - 1; i += 64) {
                    uint32_t v = 0x07FFFFFFu;
                    if (i < nv) {
                        const uint32_t pos = (uint32_t)(x0 + i);
                        const uint32_t f = (uint32_t)extract_bases(sW, (pos >> 4) - wbase, pos & 15u, s) & smask;
                        v = min(f, revcomp32(f, s));
                    }
                    sV[(i & (WV_C - 1)) * WV_RS + (i >> 3)] = v;
                }
                wave_lds_sync();
                // leftmost / rightmost argmin of this lane's eight windows (same sparse-table minima as k_syncmers)
                constexpr int NV = FW + WV_C - 1;
                const int xs = (int)lane * WV_C;
                int lmr[WV_C], rmr[WV_C], last_anchor = -1;
                {
                    uint32_t kl[NV], kr[NV];
#pragma unroll
                    for (int c = 0; c < NV; ++c) {
                        const uint32_t v = sV[(c & (WV_C - 1)) * WV_RS + (int)lane + (c >> 3)];
                        kl[c] = (v << 5) | (uint32_t)c;
                        kr[c] = (v << 5) | (uint32_t)(31 - c);
                    }
                    uint32_t l2[NV - 1], r2[NV - 1], l4[NV - 3], r4[NV - 3], l8[WV_C + FW - 8], r8[WV_C + FW - 8];
#pragma unroll
                    for (int c = 0; c < NV - 1; ++c) { l2[c] = min(kl[c], kl[c + 1]); r2[c] = min(kr[c], kr[c + 1]); }
#pragma unroll
                    for (int c = 0; c < NV - 3; ++c) { l4[c] = min(l2[c], l2[c + 2]); r4[c] = min(r2[c], r2[c + 2]); }
#pragma unroll
                    for (int c = 0; c < WV_C + FW - 8; ++c) { l8[c] = min(l4[c], l4[c + 4]); r8[c] = min(r4[c], r4[c + 4]); }
                    uint64_t packL = 0, packR = 0;
#pragma unroll
                    for (int c = 0; c < WV_C; ++c) {
                        const uint32_t ml = min(l8[c], l8[c + FW - 8]);
                        const uint32_t mr = min(r8[c], r8[c + FW - 8]);
                        lmr[c] = (int)(ml & 31u) - c;
                        rmr[c] = 31 - (int)(mr & 31u) - c;
                        packL |= (uint64_t)(uint32_t)lmr[c] << (8 * c);
                        packR |= (uint64_t)(uint32_t)rmr[c] << (8 * c);
                        const int xl = xs + c;
                        if (xl < nw_tile && (lmr[c] == rmr[c] || x0 + xl == 0)) last_anchor = xl;
                    }
                    *reinterpret_cast<uint64_t *>(&sLm[xs]) = packL;
                    *reinterpret_cast<uint64_t *>(&sRm[xs]) = packR;
                }
                // nearest earlier anchor: exclusive prefix max over the lanes
                int anchor = wave_incl_max(last_anchor);
                anchor = __shfl_up(anchor, 1);
                if (lane == 0) anchor = -1;
                wave_lds_sync();
                uint32_t selmask = 0;
                int p = 0;
                if (xs < nw_tile) {
                    p = anchor >= 0 ? anchor + (int)sLm[anchor] : carry - x0;
                    while (p < xs) p = (p + 1) + (int)sRm[p + 1];
#pragma unroll
                    for (int c = 0; c < WV_C; ++c) {
                        const int xl = xs + c;
                        if (xl < nw_tile) {
                            const int lm = lmr[c], rm = rmr[c];
                            if (lm == rm || x0 + xl == 0) p = xl + lm;
                            else if (p < xl) p = xl + rm;
                            if (p == xl + t - 1) selmask |= 1u << c;
                        }
                    }
                }
                carry = __shfl(p, (nw_tile - 1) / WV_C) + x0;     // the owner of the tile's last window
                // emit wyhash(canonical k-mer) of the selected windows, in window order
                const uint32_t cnt = (uint32_t)__popc(selmask);
                const uint32_t incl = wave_incl_add(cnt);
                uint32_t pos = n_sel + incl - cnt;
                while (selmask) {
                    const int c = __ffs((int)selmask) - 1;
                    selmask &= selmask - 1;
                    const uint32_t x = (uint32_t)(x0 + xs + c);
                    const uint64_t f = extract_bases(sW, (x >> 4) - wbase, x & 15u, k);
                    const uint64_t rc = revcomp64(f, k);
                    if (pos < cap) sCand[pos] = wyhash_u64(f < rc ? f : rc);       // syncmer.cpp:144-145
                    ++pos;
                }
                n_sel += __shfl(incl, 63);
            }
            if (n_sel > cap) {     // capacity bound (nwin / min(t, w-t+1) + 2) violated: internal invariant
                if (lane == 0) atomicOr(&a.ctr->flags, FLAG_CAND_OVERFLOW);
                n_sel = cap;
            }
            wave_lds_sync();
            // ---- dedup: candidate i = lane + 64 q against every earlier one (ankerl set semantics, syncmer.cpp:157-165)
            constexpr int QMAX = SYNC_WAVE_CAND / 64;
            const uint32_t Q = (n_sel + 63u) >> 6;
            uint64_t h[QMAX];
            uint32_t firstmask = 0;
            bool removed = false;
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                h[q] = 0;
                if ((uint32_t)q < Q) {                          // wave-uniform
                    const uint32_t i = lane + 64u * (uint32_t)q;
                    h[q] = sCand[min(i, (uint32_t)SYNC_WAVE_CAND - 1u)];
                    bool dupf = false;
                    const uint32_t jend = min(n_sel, 64u * (uint32_t)q + 63u);   // candidates before the last lane's
                    for (uint32_t j0 = 0; j0 < jend; j0 += 8u) {
                        uint64_t o[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) o[u] = sCand[min(j0 + (uint32_t)u, (uint32_t)SYNC_WAVE_CAND - 1u)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) dupf |= (j0 + (uint32_t)u < i) && (o[u] == h[q]);
                    }
                    bool first = i < n_sel && !dupf;
                    if (first && a.scaling_limit > 0.0 && !((double)wyhash_u64(h[q]) <= a.scaling_limit)) first = false;   // taxor_search.cpp:223-233
                    if (first) firstmask |= 1u << q;
                    removed |= (i < n_sel && !first);
                }
            }
            uint32_t n_dist = 0;
            if (!__any(removed ? 1 : 0)) {
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    const uint32_t i = lane + 64u * (uint32_t)q;
                    if ((uint32_t)q < Q && i < n_sel) outh[i] = h[q];
                }
                n_dist = n_sel;
            } else {
#pragma unroll
                for (int q = 0; q < QMAX; ++q) {
                    if ((uint32_t)q < Q) {
                        const uint32_t f = (firstmask >> q) & 1u;
                        const uint32_t incl = wave_incl_add(f);
                        if (f) outh[n_dist + incl - 1u] = h[q];
                        n_dist += __shfl(incl, 63);
                    }
                }
            }
            if (lane == 0) {
                a.nh[r] = n_dist;                                                   // taxor_search.cpp:261
                a.thr[r] = (uint64_t)((double)n_dist * a.ratio);                    // threshold.hpp:60,76-79
            }
            hashes_total += n_dist;
        }
    }
    if (lane == 0 && hashes_total) atomicAdd(&a.ctr->n_hashes, hashes_total);
}

int syncmers_wave_grid(int device, int want_per_cu)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 1024;
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_syncmers_wave<11>, BLK, 0) != hipSuccess || per < 1) per = 2;
    if (per > want_per_cu) per = want_per_cu;
    return p.multiProcessorCount * per;
}

void launch_syncmers_wave(const SyncmerArgs &a, int grid, hipStream_t st)
{
    if (!a.n_reads) return;
    hipLaunchKernelGGL(k_syncmers_wave<11>, dim3(grid), dim3(BLK), 0, st, a);
}

bool syncmers_wave_applies(int k, int s)
{
    static const bool off = [] { const char *e = tune_env("TAXOR_SYNC_WAVE"); return e && atoi(e) == 0; }();
    static const bool generic_only = [] { const char *e = tune_env("TAXOR_SYNC_GENERIC"); return e && atoi(e) != 0; }();
    return !off && !generic_only && k - s + 1 == 11 && s <= 13;
}


// ------------------------------------------------------------------------------------------------------
// k_minimisers -- hashing for indexes built WITHOUT --use-syncmer (taxor_search.cpp:210-212,239-260):
// seqan3::views::minimiser_hash(ungapped{k}, window_size{w}, seed{adjust_seed(k)}).  Value of the k-mer at
// position i = min(fwd ^ seed, revcomp ^ seed); a window is W = w-k+1 consecutive values (all of them when the read
// has fewer); the view emits the minimiser of the first window and then one value whenever the minimiser changes.
// Which of several equal minima is the minimiser is stateful (first window and every re-scan after the minimiser
// left the window: the RIGHTMOST minimum; a newcomer replaces it only when strictly smaller).  Per tile of MN_T
// windows every thread finds the rightmost / leftmost minimum of its windows; a tile in which every window has a
// unique minimum (any tile of ordinary sequence; every tile when w == k) needs no state at all, otherwise one
// thread replays the reference's rule over the tile (low-complexity sequence only).  No dedup: every emitted
// value counts (the reference pushes them into a vector), FracMinHash down-sampling as in the syncmer path.
// ------------------------------------------------------------------------------------------------------
static constexpr int MN_C = 4;                    // windows per thread
static constexpr int MN_T = BLK * MN_C;           // windows per tile
static constexpr int MN_WMAX = 512;               // k-mers per window supported (w - k + 1)
static constexpr int MN_WORDS = (MN_T + MN_WMAX + 32) / 16 + 4;

__global__ __launch_bounds__(BLK) void k_minimisers(const SyncmerArgs a)
{
    __shared__ uint32_t sW[MN_WORDS];
    __shared__ uint64_t sV[MN_T + MN_WMAX];
    __shared__ uint16_t sP[MN_T];                 // minimiser position of window xl, relative to the tile's first value
    __shared__ uint32_t sScr[8];
    __shared__ uint32_t sRead, sTie;
    __shared__ int sCarry;                        // minimiser position (absolute) of the last window of the previous tile

    const int k = a.k;
    const uint32_t tid = threadIdx.x;
    const uint64_t seed = 0x8F3F73B5CF1C9ADEull >> (64 - 2 * k);        // hixf::adjust_seed, adjust_seed.hpp:40-44

    for (;;) {
        __syncthreads();
        if (tid == 0) sRead = atomicAdd(a.cursor, 1u);
        __syncthreads();
        if (sRead >= a.n_reads) break;
        const uint32_t r = a.order ? a.order[sRead] : sRead;
        const uint32_t L = a.rlen[r];
        const uint32_t *__restrict__ pk = a.packed + a.poff[r];
        const uint32_t nwords = (((L + 15u) >> 4) + 3u) & ~3u;
        uint64_t *__restrict__ outh = a.hashes + a.hoff[r];
        const uint32_t cap = a.hcap[r];
        const int nk = (int)L - k + 1;                                   // k-mer values of the read
        const int W = nk > 0 ? min(a.w_min - k + 1, nk) : 0;             // the view shrinks the window to the text
        const int nwin = nk > 0 ? nk - W + 1 : 0;
        uint32_t n_out = 0;

        for (int j0 = 0; j0 < nwin; j0 += MN_T) {
            __syncthreads();
            const uint32_t wbase = (uint32_t)j0 >> 4;
            for (uint32_t i = tid; i < (uint32_t)MN_WORDS; i += BLK) {
                const uint32_t wi = wbase + i;
                sW[i] = wi < nwords ? pk[wi] : 0u;
            }
            if (tid == 0) sTie = 0;
            __syncthreads();
            const int nw_tile = min(MN_T, nwin - j0);
            const int nval = nw_tile + W - 1;
            for (int i = (int)tid; i < nval; i += BLK) {
                const uint32_t pos = (uint32_t)(j0 + i);
                const uint64_t f = extract_bases(sW, (pos >> 4) - wbase, pos & 15u, k);
                const uint64_t rc = revcomp64(f, k);
                sV[i] = min(f ^ seed, rc ^ seed);
            }
            __syncthreads();
            int rm[MN_C];
            bool tie = false;
#pragma unroll
            for (int c = 0; c < MN_C; ++c) {
                const int xl = (int)tid * MN_C + c;
                rm[c] = 0;
                if (xl < nw_tile) {
                    uint64_t m = sV[xl];
                    int lm = 0;
                    for (int x = 1; x < W; ++x) {
                        const uint64_t v = sV[xl + x];
                        if (v < m) { m = v; lm = x; rm[c] = x; }
                        else if (v == m) rm[c] = x;
                    }
                    tie |= lm != rm[c];
                    sP[xl] = (uint16_t)(xl + rm[c]);                      // unique minimum: the minimiser, whatever the history
                }
            }
            if (tie) sTie = 1;
            __syncthreads();
            if (sTie && tid == 0) {
                // replay of minimiser_view::next_minimiser over the tile; sP holds xl + Rm(xl) on entry
                int p = sCarry;
                for (int xl = 0; xl < nw_tile; ++xl) {
                    const int j = j0 + xl, newest = xl + W - 1;
                    if (j == 0 || p < j) p = j0 + (int)sP[xl];
                    else if (sV[newest] < sV[p - j0]) p = j0 + newest;
                    sP[xl] = (uint16_t)(p - j0);
                }
            }
            __syncthreads();
            const int carry_in = sCarry;
            uint32_t cnt = 0;
            uint64_t val[MN_C];
            bool em[MN_C];
#pragma unroll
            for (int c = 0; c < MN_C; ++c) {
                const int xl = (int)tid * MN_C + c;
                em[c] = false;
                if (xl < nw_tile) {
                    const int p = j0 + (int)sP[xl];
                    const int prev = xl ? j0 + (int)sP[xl - 1] : (j0 ? carry_in : -1);
                    em[c] = p != prev;
                    val[c] = sV[p - j0];
                    if (em[c] && a.scaling_limit > 0.0 && !((double)wyhash_u64(val[c]) <= a.scaling_limit)) em[c] = false; // :243-249
                    cnt += em[c] ? 1u : 0u;
                }
            }
            uint32_t tot;
            uint32_t off = n_out + block_excl_add(cnt, sScr, &tot);
#pragma unroll
            for (int c = 0; c < MN_C; ++c)
                if (em[c]) {
                    if (off < cap) outh[off] = val[c];
                    else atomicOr(&a.ctr->flags, FLAG_CAND_OVERFLOW);
                    ++off;
                }
            n_out += tot;
            __syncthreads();
            if (tid == 0) sCarry = j0 + (int)sP[nw_tile - 1];
        }
        if (tid == 0) {
            a.nh[r] = n_out;                                                    // taxor_search.cpp:261
            if (a.thr_on_device) a.thr[r] = (uint64_t)((double)n_out * a.ratio); // percentage model, threshold.hpp:76-79
            atomicAdd(&a.ctr->n_hashes, (unsigned long long)n_out);
        }
    }
}

int syncmers_grid(int device)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 1024;
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_syncmers<11, false>, BLK, 0) != hipSuccess || per < 1) per = 2;
    if (const char *e = tune_env("TAXOR_SYNC_BPC")) { const int v = atoi(e); if (v >= 1 && v <= 8) per = v; }
    return p.multiProcessorCount * per;
}

void launch_syncmers(const SyncmerArgs &a, int grid, hipStream_t st)
{
    if (!a.n_reads) return;
    if (a.w_min > 0) { // index built without --use-syncmer
        hipLaunchKernelGGL(k_minimisers, dim3(grid), dim3(BLK), 0, st, a);
        return;
    }
    static const bool generic_only = [] { const char *e = tune_env("TAXOR_SYNC_GENERIC"); return e && atoi(e) != 0; }();
    if (!generic_only && a.k - a.s + 1 == 11 && a.s <= 13) {
        if (a.prof) hipLaunchKernelGGL((k_syncmers<11, true>), dim3(grid), dim3(BLK), 0, st, a);
        else hipLaunchKernelGGL((k_syncmers<11, false>), dim3(grid), dim3(BLK), 0, st, a);
    } else hipLaunchKernelGGL((k_syncmers<0, false>), dim3(grid), dim3(BLK), 0, st, a);
}

// ------------------------------------------------------------------------------------------------------
// k_query_level
//
// One work item = (read, IXF).  The block stages the read's probes (rows + fingerprint, per this IXF's seed
// and segment length) in LDS -- all of them at once when n_h <= Q_CAP, otherwise a tile at a time; thread (u, g)
// owns the 16-B unit u of every row and the hash subset g, g+G, ...; each hash costs the thread three 16-B
// global loads (whole block: three contiguous row segments), an XOR, and an exact zero-byte test whose 0/1
// bytes accumulate in packed byte counters (a thread adds at most 240 per tile before they are widened).
// After n_h - thr + margin hashes, bin runs that can no longer reach the threshold are dropped and the remaining
// hashes probe only the surviving 16-bin units (threshold-aware pruning, see the kernel).  Counters are merged
// through LDS, then the bins are walked exactly like bulk_contains_impl (hixf.hpp:313-338).
// ------------------------------------------------------------------------------------------------------
static constexpr int Q_HT = 240;   // hashes per probe tile when one thread sees every hash (byte counters stay < 256)
static constexpr int Q_HT2 = 480;  // probe tile when hashes are split over G >= 2 thread groups
static constexpr int Q_OB = 64;    // child pushes / hit records buffered in LDS before they are appended globally
static constexpr int Q_CAP = 1024; // LDS probe capacity: reads with n_h <= Q_CAP stage all their probes once per work item
static constexpr int Q_CAP_SMALL = 256; // ... of the single-wave instantiation for tiny items
static constexpr int Q_BLK_SMALL = 64;

__device__ __forceinline__ uint32_t zero_bytes01(uint32_t y)
{
    uint32_t t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    t = ~(t | y | 0x7F7F7F7Fu); // 0x80 in every byte of y that is zero
    return t >> 7;
}

// The row pointer comes out of an IxfDesc in memory, so the compiler cannot infer its address space and would
// emit flat_load (which also ties up lgkmcnt and so serialises against the LDS probe reads): cast to global.
typedef const uint32_t __attribute__((address_space(1))) *gptr32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *gptr128;

template <bool NT> __device__ __forceinline__ uint4 ld16(const uint8_t *p)
{
    if constexpr (NT) {
        // rows are random in a table far larger than any cache: stream them (no reuse to protect)
        gptr32 q = (gptr32)(uintptr_t)p;
        return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1),
                          __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
    } else {
        const u32x4 v = *(gptr128)(uintptr_t)p;
        return make_uint4(v.x, v.y, v.z, v.w);
    }
}

__device__ __forceinline__ void probe_accumulate(uint4 &acc8, const uint4 &x, const uint4 &y, const uint4 &z,
                                                 uint32_t fp4)
{
    acc8.x += zero_bytes01(x.x ^ y.x ^ z.x ^ fp4);
    acc8.y += zero_bytes01(x.y ^ y.y ^ z.y ^ fp4);
    acc8.z += zero_bytes01(x.z ^ y.z ^ z.z ^ fp4);
    acc8.w += zero_bytes01(x.w ^ y.w ^ z.w ^ fp4);
}

// LDS layout of k_query_level (all dynamic, base 16-B aligned):
//   [0, Q_CAP*16)                     probes (all of the read's if n_h <= Q_CAP, else the current hash tile)
//   [.., +64)                          scalars: work item, alive-unit count
//   [.., +Q_MAXU*4)                    list of alive 16-bin units
//   [.., +map_words*4)                 bitmap of alive units (map_words = max_units/32 rounded up to 4 words)
//   [.., +max_stride*4)                per-bin counts
//   [.., +max_stride*4)                per-bin info words (binfo) of the current IXF
static constexpr int Q_MAXU = 32;   // more alive units than this -> finish the item densely
static constexpr int Q_MAXC = 64;   // candidate units remembered for the tally; more -> the tally walks every bin
static constexpr int Q_CHUNK_MAX = 16; // work items taken per cursor atomic, at most

// what a block needs to know about a work item, fetched for a whole cursor chunk at once (one lane per item) so that
// the dependent loads  cursor -> (read, IXF) -> descriptor / hash count / threshold / hash offset  are paid once per
// chunk and not once per item by every thread
struct ItemMeta {
    IxfDesc D;
    uint64_t thr, hoff;
    uint32_t r, n;
};

__host__ __device__ inline size_t query_lds_map_words(uint32_t max_stride) { return (((size_t)max_stride / 16 + 31) / 32 + 3) & ~(size_t)3; }

uint32_t query_map_words(uint32_t max_stride) { return (uint32_t)query_lds_map_words(max_stride); }

size_t query_lds_bytes(uint32_t max_stride, bool small)
{
    return (size_t)(small ? Q_CAP_SMALL : Q_CAP) * 16 + 64 + (size_t)Q_MAXU * 4 + 2 * query_lds_map_words(max_stride) * 4 + (size_t)max_stride * 8 +
           (size_t)Q_MAXC * 4;
}

// dense pass over hashes [h0, h1): every thread (u, g) reads its 16-B unit of the three rows of every hash of
// its subset and counts byte matches; counters are flushed into the LDS counts at the end.
template <bool NT, int U, int BS, int QC>
__device__ __forceinline__ void query_dense_range(const IxfDesc &D, const uint64_t *__restrict__ hp, uint32_t h0,
                                                  uint32_t h1, uint4 *sProbe, uint32_t *sC, bool staged)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t units = D.units, stride = D.stride;
    // column passes: 256 units (4096 bins) per pass; almost always exactly one
    for (uint32_t u0 = 0; u0 < units; u0 += BS) {
        const uint32_t upass = min(units - u0, (uint32_t)BS);
        const uint32_t G = BS / upass;           // hashes processed concurrently by the block
        const uint32_t g = tid / upass;
        const uint32_t u = u0 + (tid - g * upass);
        const bool active = g < G;
        const uint8_t *__restrict__ base = D.data + (size_t)u * 16u;

        // a tile bounds the per-thread increments of the packed byte counters (<= 240); when the probes are not
        // pre-staged it is also what fits the staging loop
        const uint32_t HT = staged ? 240u * G : min((G == 1u) ? (uint32_t)Q_HT : (uint32_t)Q_HT2, (uint32_t)QC);
        for (uint32_t t0 = h0; t0 < h1; t0 += HT) {
            const uint32_t nt = min(HT, h1 - t0);
            const uint4 *pr = sProbe + (staged ? t0 : 0u);
            if (!staged) {
                __syncthreads();
                for (uint32_t i = tid; i < nt; i += BS) {
                    const ixf_probe p = ixf_probe_key_arith(hp[t0 + i], D.seed, D.seg_len, D.arith);
                    sProbe[i] = make_uint4(p.row[0], p.row[1], p.row[2], p.fp4);
                }
                __syncthreads();
            }
            if (active) {
                uint4 acc8 = make_uint4(0, 0, 0, 0);
                uint32_t i = g;
                for (; i + (uint32_t)(U - 1) * G < nt; i += (uint32_t)U * G) { // U hashes = 3U row loads in flight per lane
                    uint4 p[U], ra[U], rb[U], rc[U];
#pragma unroll
                    for (int j = 0; j < U; ++j) p[j] = pr[i + (uint32_t)j * G];
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        ra[j] = ld16<NT>(base + (size_t)p[j].x * stride);
                        rb[j] = ld16<NT>(base + (size_t)p[j].y * stride);
                        rc[j] = ld16<NT>(base + (size_t)p[j].z * stride);
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j) probe_accumulate(acc8, ra[j], rb[j], rc[j], p[j].w);
                }
                if (i < nt) { // up to U-1 left: issue their loads together (one memory round trip)
                    uint4 p[U - 1], ra[U - 1], rb[U - 1], rc[U - 1];
                    bool ok[U - 1];
#pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        ok[j] = i + (uint32_t)j * G < nt;
                        p[j] = pr[ok[j] ? i + (uint32_t)j * G : i];
                    }
#pragma unroll
                    for (int j = 0; j < U - 1; ++j) {
                        ra[j] = rb[j] = rc[j] = make_uint4(0, 0, 0, 0);
                        if (ok[j]) {
                            ra[j] = ld16<NT>(base + (size_t)p[j].x * stride);
                            rb[j] = ld16<NT>(base + (size_t)p[j].y * stride);
                            rc[j] = ld16<NT>(base + (size_t)p[j].z * stride);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < U - 1; ++j)
                        if (ok[j]) probe_accumulate(acc8, ra[j], rb[j], rc[j], p[j].w);
                }
                // the packed byte counters (<= 240 per byte) of this tile go straight into the LDS tally: almost always one
                // tile per item, and sixteen 32-bit accumulators held across the loop are sixteen registers the loads need
                const uint32_t wv[4] = {acc8.x, acc8.y, acc8.z, acc8.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (wv[q] == 0u) continue;
                    const uint32_t b0 = wv[q] & 0xFFu, b1 = (wv[q] >> 8) & 0xFFu, b2 = (wv[q] >> 16) & 0xFFu, b3 = wv[q] >> 24;
                    uint32_t *c = &sC[u * 16u + 4u * (uint32_t)q];
                    if (b0) atomicAdd(c + 0, b0);
                    if (b1) atomicAdd(c + 1, b1);
                    if (b2) atomicAdd(c + 2, b2);
                    if (b3) atomicAdd(c + 3, b3);
                }
            }
        }
    }
}

// BS threads per block; QC probes fit the LDS staging area.  (256, 1024) is the general instantiation; (64, 256) serves
// launches of TINY items -- levels of narrow IXFs (<= 512 bins) under short reads -- where an item is a handful of memory
// round trips and fixed cost: sixteen single-wave blocks per CU keep four times as many items in flight as four
// four-wave blocks, and a single-wave block's barriers cost nothing.
//
// TREE = true: ONE launch for the whole traversal of a small batch (api.hip, "small batches").  The root's items are implicit as
// at level 0; every child a block pushes goes into the one queue a.q_out and is taken from there by whichever block asks next --
// no level ends when its slowest item ends, and a read's child starts when ITS parent is done, not when every parent is.  The
// queue's 64-bit entries are published with single agent-scope atomic stores into slots that hold ~0 (invalid) until then, and
// read the same way: the entry IS the message (read, IXF), everything else an item needs was written by earlier launches, so
// relaxed ordering suffices -- acquire / release at agent scope would write back and invalidate the XCD's L2 on every poll and
// every push (measured: the launch then takes milliseconds).  Work is handed out by TICKET (one
// atomic add on one counter: root items first, then queue slots in order); a block whose ticket has no entry yet first puts out
// its own pending pushes (nobody ever waits on a block that waits) and then polls ITS slot -- a word of its own, no shared hot
// spot -- until the entry arrives or the launch is complete.  Complete means completed == roots + reserved: an unfinished item is
// the only thing that can reserve more, so the block whose flush makes the two equal knows it is final and says so through 64
// copies of one word, each polled by a sixty-fourth of the waiting blocks.  (Tried first: claiming entries by compare-and-swap
// while cursor < reserved, idle blocks leaving -- a thousand blocks retrying a CAS on one word serialise into O(n^2) atomics,
// milliseconds per launch.)  Consumers put ~0 back, so the queue is clean for the next launch.  A wait that outlives any
// plausible run raises FLAG_TREE_STALL, releases the other waiting blocks and leaves -- never a hung GPU, and the host classifies
// the piece again level by level (api.hip, small_harvest_one).
template <bool NT, int U, bool PROF = false, int BS = BLK, int QC = Q_CAP, bool TREE = false>
__global__ __launch_bounds__(BS) void k_query_level(const QueryArgs a)
{
    uint64_t pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t plast = PROF ? __builtin_amdgcn_s_memtime() : 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *sProbe = reinterpret_cast<uint4 *>(smem);
    uint32_t *sScal = reinterpret_cast<uint32_t *>(smem + QC * sizeof(uint4));       // [0] item, [1] n alive units
    uint32_t *sUnits = sScal + 16;
    uint32_t *sMap = sUnits + Q_MAXU;
    uint32_t *sC = sMap + a.map_words;
    uint32_t *sInfo = sC + a.max_stride;
    uint32_t *sCandMap = sInfo + a.max_stride;    // units holding a run that could still pass when the dense phase ended
    uint32_t *sCandUnits = sCandMap + a.map_words;
    __shared__ ItemMeta sItems[Q_CHUNK_MAX];

    const uint32_t tid = threadIdx.x;
    const uint32_t lvl = a.level;
    // This kernel is HBM-bound and needs few issue slots, but it needs them promptly: when the (VALU/LDS heavy)
    // syncmer kernel of the next sub-batch shares the CU, age-based arbitration would starve these waves.
    __builtin_amdgcn_s_setprio(3);
    const uint32_t n_parts = (!a.q_in && a.parts > 1u) ? a.parts : 1u;
    const uint32_t n_items = a.q_in ? min(a.ctr->q_n[lvl].v, a.q_cap) : a.n_level0 * n_parts;
    unsigned long long st_bytes = 0, st_touched = 0, st_work = 0, st_rows = 0, st_sparse = 0;

    // Returning atomics on one word are served serially by one L2 channel, ~13 ns each.  A launch of small work items
    // (deeper levels, short reads) issues one per item on the work cursor and one per item on the hit / queue append
    // word, and either alone caps it at ~76 M items/s however many blocks are resident (every block then spends its
    // time queueing for the word).  So: `cursor_chunk` items per cursor atomic, and child pushes / hit records are
    // collected in LDS across items and appended a few dozen at a time.
    __shared__ uint4 sOutH[Q_OB];
    __shared__ uint2 sOutQ[Q_OB];
    if (tid == 0) { sScal[2] = 0u; sScal[3] = 0u; }
    unsigned long long *q64 = reinterpret_cast<unsigned long long *>(a.q_out);      // TREE: the one queue, entries (ixf << 32 | read), ~0 = empty slot
    const uint32_t tree_n0 = a.n_level0 * n_parts;                                   // TREE: the root's items; everything else comes out of the queue
    if (tid == 0) sScal[8] = 0u;                                                     // TREE: items finished since the last flush
    auto flush_out = [&](bool force) {          // block-uniform; the caller has just passed a barrier
        const uint32_t nq = min(sScal[2], (uint32_t)Q_OB), nh = min(sScal[3], (uint32_t)Q_OB);
        const uint32_t n_done = TREE ? sScal[8] : 0u;
        if (!(force ? (nq | nh | n_done) != 0u : (nq >= (uint32_t)Q_OB / 2u || nh >= (uint32_t)Q_OB / 2u))) return;
        if (tid == 0) {
            sScal[4] = nq ? atomicAdd(&a.ctr->q_n[lvl + 1].v, nq) : 0u;       // TREE: q_n[1] = children reserved so far, all levels
            sScal[5] = nh ? atomicAdd(&a.ctr->n_hits.v, nh) : 0u;
        }
        __syncthreads();
        const uint32_t bq = sScal[4], bh = sScal[5];
        for (uint32_t i = tid; i < nq; i += BS) {
            if (bq + i < a.q_cap) {
                if constexpr (TREE) __hip_atomic_store(&q64[bq + i], (unsigned long long)sOutQ[i].y << 32 | sOutQ[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else a.q_out[bq + i] = sOutQ[i];
            } else atomicOr(&a.ctr->flags, FLAG_QUEUE_OVERFLOW);
        }
        for (uint32_t i = tid; i < nh; i += BS) {
            if (bh + i < a.hit_cap) a.hits[bh + i] = sOutH[i];
            else atomicOr(&a.ctr->flags, FLAG_HITS_OVERFLOW);
        }
        __syncthreads();
        if (tid == 0) {
            sScal[2] = 0u; sScal[3] = 0u;
            if constexpr (TREE) {
                sScal[8] = 0u;
                if (n_done) {
                    // the items this block finished count as completed now that their pushes are reserved and written.  Whoever
                    // makes completed == roots + reserved knows that nothing is running and nothing more can be pushed: it tells
                    // the waiting blocks through 64 copies of one word (each polled by a sixty-fourth of them)
                    const uint32_t c = atomicAdd(&a.ctr->q_n[2].v, n_done) + n_done;
                    if (c == tree_n0 + __hip_atomic_load(&a.ctr->q_n[1].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        for (uint32_t g = 0; g < 64u; ++g) __hip_atomic_store(&a.ctr->q_xcur[g >> 3][g & 7u].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };

    const uint32_t chunk = min(max(a.cursor_chunk, 1u), (uint32_t)Q_CHUNK_MAX);
    uint32_t item = 0, item_end = 0, item0 = 0;
    uint32_t info_of = 0xFFFFFFFFu;     // bin_base of the IXF whose info words sit in sInfo (block-uniform): at the root every
                                        // item is IXF 0, below it the items arrive grouped by IXF -- no need to fetch them again
    for (;;) {
        __syncthreads();
        if (item == item_end) {
            flush_out(false);
            if constexpr (TREE) {
                if (tid == 0) sScal[0] = atomicAdd(&a.ctr->q_cursor[0].v, 1u);          // a ticket: root items first, then queue slots in order
                __syncthreads();
                const uint32_t idx = sScal[0];
                uint32_t got = 1u;                                           // 1 = an item to work on, 0 = all done, 2 = a slot beyond the queue's capacity
                if (idx >= tree_n0) {
                    flush_out(true);                                         // nobody may wait for pushes (or completions) this block still holds
                    if (tid == 0) {
                        const uint32_t k = idx - tree_n0;
                        const uint32_t *done = &a.ctr->q_xcur[(blockIdx.x >> 3) & 7u][blockIdx.x & 7u].v;      // this block's copy of the "all done" word
                        unsigned long long e = ~0ull;
                        got = 3u;
                        for (uint32_t spin = 0; spin <= a.tree_polls; ++spin) {
                            if (k < a.q_cap) {
                                e = __hip_atomic_load(&q64[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                if (e != ~0ull) { got = 1u; break; }
                            } else if (k < __hip_atomic_load(&a.ctr->q_n[1].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = 2u; break; }
                            if (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = 0u; break; }
                            __builtin_amdgcn_s_sleep(16);
                        }
                        if (got == 3u) {
                            // the watchdog: the launch is abandoned as a whole (an entry that arrives in this slot later would never be
                            // worked on), so everybody who waits is told to leave as well; the host reruns the piece level by level
                            atomicOr(&a.ctr->flags, FLAG_TREE_STALL);
                            for (uint32_t g = 0; g < 64u; ++g) __hip_atomic_store(&a.ctr->q_xcur[g >> 3][g & 7u].v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            got = 0u;
                        }
                        if (got == 1u) {
                            q64[k] = ~0ull;                                  // the slot is empty again for the next launch
                            ItemMeta m;
                            m.r = (uint32_t)e;
                            m.D = a.ixf[(uint32_t)(e >> 32)];
                            m.thr = a.thr[m.r];
                            m.hoff = a.hoff[m.r];
                            m.n = a.nh[m.r];
                            sItems[0] = m;
                        }
                        if (got == 2u) sScal[8] += 1u;                       // pushed past the queue's end (FLAG_QUEUE_OVERFLOW is up): completed without work
                        sScal[7] = got;
                    }
                    __syncthreads();
                    got = sScal[7];
                    if (got == 0u) break;
                    if (got == 2u) continue;
                } else if (tid == 0) {
                    uint32_t part = 0, ri = idx;
                    if (n_parts > 1u) { part = ri / a.n_level0; ri -= part * a.n_level0; }
                    ItemMeta m;
                    m.r = a.order0 ? a.order0[ri] : ri;
                    m.D = a.ixf[0];
                    if (n_parts > 1u) {
                        const uint32_t u0 = a.part_cut[part], u1 = a.part_cut[part + 1u];
                        m.D.data += (size_t)u0 * 16u;
                        m.D.bin_base += u0 * 16u;
                        m.D.bins = min(m.D.bins, u1 * 16u) - u0 * 16u;
                        m.D.units = u1 - u0;
                    }
                    m.thr = a.thr[m.r];
                    m.hoff = a.hoff[m.r];
                    m.n = a.nh[m.r];
                    sItems[0] = m;
                }
                item = item0 = 0u;            // (indices are only used relative to item0 from here on)
                item_end = 1u;
                __syncthreads();
            } else
            if (a.xcd_slices) {
                // Eight slices of the grouped queue, eight cursors.  A block starts in the slice of its XCD (the dispatcher is
                // observed to place block b on XCD b % 8; a wrong guess is slower, not wrong) and moves on to the next slice when
                // its own is exhausted, so the launch ends balanced.  The items of one IXF are neighbours in the queue: they are
                // now read through ONE L2 (4 MiB per XCD) instead of being fetched into all eight.
                if (tid == 0) {
                    const uint32_t S = a.xcd_slices;                       // 8, or fewer (power of two): neighbouring XCDs share a slice
                    const uint32_t per = (n_items + S - 1u) / S;
                    uint32_t got = n_items, got_end = n_items;
                    for (uint32_t k = 0; k < S; ++k) {
                        const uint32_t sl = (((blockIdx.x & 7u) * S >> 3) + k) & (S - 1u);
                        const uint32_t lo = min(sl * per, n_items), hi = min(lo + per, n_items);
                        if (lo >= hi) continue;
                        const uint32_t i = atomicAdd(&a.ctr->q_xcur[min(lvl, 15u)][sl].v, chunk);
                        if (i < hi - lo) { got = lo + i; got_end = hi; break; }
                    }
                    sScal[0] = got;
                    sScal[7] = got_end;
                }
                __syncthreads();
                item = item0 = sScal[0];
                item_end = min(item + chunk, sScal[7]);
                if (item >= n_items) break;
            } else {
                if (tid == 0) sScal[0] = atomicAdd(&a.ctr->q_cursor[lvl].v, chunk);
                __syncthreads();
                item = item0 = sScal[0];
                item_end = min(item + chunk, n_items);
                if (item >= n_items) break;
            }
            if (!TREE && tid < item_end - item) {                    // one lane per item of the chunk
                uint32_t r_, v_, part = 0;
                if (a.q_in) { const uint2 it = a.q_in[item + tid]; r_ = it.x; v_ = it.y; }
                else {
                    uint32_t ri = item + tid;
                    if (n_parts > 1u) { part = ri / a.n_level0; ri -= part * a.n_level0; }      // part-major: a block's consecutive items share their bin info
                    r_ = a.order0 ? a.order0[ri] : ri;
                    v_ = 0;
                }
                ItemMeta m;
                m.D = a.ixf[v_];
                if (n_parts > 1u) {       // this item is one column range of the row: a narrower IXF of its own as far as the rest of the kernel is concerned
                    const uint32_t u0 = a.part_cut[part], u1 = a.part_cut[part + 1u];
                    m.D.data += (size_t)u0 * 16u;
                    m.D.bin_base += u0 * 16u;
                    m.D.bins = min(m.D.bins, u1 * 16u) - u0 * 16u;
                    m.D.units = u1 - u0;
                }
                m.thr = a.thr[r_];
                m.hoff = a.hoff[r_];
                m.r = r_;
                m.n = a.nh[r_];
                sItems[tid] = m;
            }
            __syncthreads();
        }
        PMARK(0)                                                     // 0: work cursor, chunk metadata, output flushes
        const IxfDesc D = sItems[item - item0].D;
        const uint32_t r = sItems[item - item0].r, n = sItems[item - item0].n;
        const uint64_t thr = sItems[item - item0].thr;
        const uint64_t *__restrict__ hp = a.hashes + sItems[item - item0].hoff;
        const uint32_t stride = D.stride;
        const uint32_t *__restrict__ bi = a.binfo + D.bin_base;
        const uint32_t nb_round = (D.bins + 63u) & ~63u;

        // The per-bin info words are read twice per item (pruning check, tally): out of LDS, not out of L2 -- a global
        // load per phase is a round trip of its own, which is most of what a small item costs.  Rows of up to 1024 bins
        // take them through registers so that these loads and the hash loads of the probe staging below fly together.
        const bool info_cached = info_of == D.bin_base && !(a.tally_mode & 2u);
        const bool info_regs = nb_round <= 4u * BS;
        uint32_t infoReg[4] = {0u, 0u, 0u, 0u};
        if (info_regs && !info_cached) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t b = tid + (uint32_t)j * BS;
                if (b < D.bins) infoReg[j] = bi[b];
            }
        }
        for (uint32_t i = tid; i < stride; i += BS) sC[i] = 0;
        for (uint32_t i = tid; i < a.map_words; i += BS) { sMap[i] = 0; sCandMap[i] = 0; }
        if (tid == 0) { sScal[1] = 0; sScal[6] = 0; }

        // ---- threshold-aware pruning ---------------------------------------------------------------------------
        // After `dense_end` hashes a run of `len` technical bins whose partial sum satisfies
        // sum + (n - dense_end) * len < thr can never reach the threshold: the reference would neither report
        // nor descend it, so its remaining counts are irrelevant.  dense_end = n - thr + margin leaves random bins
        // (expected count dense_end/256) below the bound, so normally only true matches stay alive and the
        // remaining hashes probe just their 16-bin units.  Runs that stay alive are counted exactly.
        uint32_t dense_end = n;
        if (a.prune && thr > 0 && stride > a.dense_max_stride) {
            // margin above the minimum n - thr + 1: a random bin collects ~Poisson((n-thr)/256) matches in the dense
            // phase; mean + 4 sigma + 3.5 keeps the expected number of falsely surviving units per item below ~0.1 (a
            // constant of 4.5 until round 3: 3.5 is 2 % faster on 1-kb reads, whose margin is mostly this constant, and
            // within noise on 10-kb reads; 1.5 lets too many random units survive -- profiles/r03/margin.txt)
            // for reads of any length (a fixed margin either wastes dense traffic on short reads or lets every bin
            // of a 100-kb read survive).  The margin only trades dense against sparse work, never exactness.
            const float mu = thr < (uint64_t)n ? (float)((uint64_t)n - thr) * (1.0f / 256.0f) : 0.0f;
            const uint32_t margin = (uint32_t)(mu + 4.0f * sqrtf(mu) + (a.prune_margin > 0.f ? a.prune_margin : 3.5f));
            dense_end = (thr >= (uint64_t)n + margin) ? 0u : min(n, (uint32_t)((uint64_t)n + margin - thr));
        }
        uint64_t touched = 0, rows_read = 0, sparse_loads = 0;

        const bool staged = n <= (uint32_t)QC; // all probes of this read fit: stage them once for both phases
        if (staged)
            for (uint32_t i = tid; i < n; i += BS) {
                const ixf_probe p = ixf_probe_key_arith(hp[i], D.seed, D.seg_len, D.arith);
                sProbe[i] = make_uint4(p.row[0], p.row[1], p.row[2], p.fp4);
            }
        if (info_cached) {
            // sInfo already holds this IXF's words
        } else if (info_regs) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t b = tid + (uint32_t)j * BS;
                if (b < nb_round) sInfo[b] = infoReg[j];
            }
        } else {
            for (uint32_t b = tid; b < nb_round; b += BS) sInfo[b] = b < D.bins ? bi[b] : 0u;
        }
        info_of = D.bins ? D.bin_base : 0xFFFFFFFFu;
        __syncthreads();
        PMARK(1)                                                     // 1: clearing the tally, bin info + probe staging
        query_dense_range<NT, U, BS, QC>(D, hp, 0, dense_end, sProbe, sC, staged);
        touched += (uint64_t)dense_end * 3ull * (D.units * 16u);      // (= stride for a whole row; a column part reads its own units)
        rows_read += (uint64_t)dense_end * 3ull;
        __syncthreads();
        PMARK(2)                                                     // 2: dense phase (row gathers)

        bool cand_valid = false;
        if (dense_end < n) {
            // The remaining hashes probe only the 16-bin units that still matter, in up to three stages; between stages the
            // set is re-evaluated with the counts so far.  A unit matters while it holds
            //   * a LEAF run that can still reach the threshold (or has reached it): its count is reported (:328-331), so it
            //     is counted to the last hash;
            //   * a MERGED bin whose fate is open: only `sum >= thr` is ever used of it (:321) -- once it has reached the
            //     threshold it needs no more counting (the child is entered either way), once it cannot reach it it is dead.
            // A matching read's merged bins reach the threshold after ~thr / (fraction of its hashes that match) hashes,
            // i.e. well before the last one, and every later stage also drops the random survivors of the stage before: each
            // unit dropped saves a 128-B line per hash and row (what HBM moves for a 16-B load).
            uint32_t done = dense_end;
            uint32_t chunk = 0;
            bool first_eval = true;
            cand_valid = !a.counts_out && !(a.tally_mode & 1u);
            for (;;) {
                const uint64_t rem = n - done;
                if (!first_eval) {
                    for (uint32_t i = tid; i < a.map_words; i += BS) sMap[i] = 0;
                    if (tid == 0) sScal[1] = 0;
                    __syncthreads();
                }
                for (uint32_t b = tid; b < nb_round; b += BS) {
                    if (b < D.bins) {
                        const uint32_t info = sInfo[b];
                        if (info & BINFO_END) { // merged bins are runs of length one and carry BINFO_END too
                            int bb = (int)b;
                            uint64_t sum = sC[bb];
                            bool alive, cand;
                            if (info & BINFO_MERGED) {
                                cand = sum + rem >= thr;
                                alive = cand && sum < thr;
                            } else {
                                while (bb > 0 && (sInfo[bb - 1] >> 30) == 0u) sum += sC[--bb];
                                const uint64_t len = (uint64_t)b - (uint64_t)bb + 1u;
                                alive = cand = sum + rem * len >= thr;
                            }
                            if (first_eval && cand) {      // whatever passes at the end was a candidate now: the tally walks only these units
                                for (uint32_t x = (uint32_t)bb >> 4; x <= (b >> 4); ++x) {
                                    const uint32_t bit = 1u << (x & 31u);
                                    if (!(atomicOr(&sCandMap[x >> 5], bit) & bit)) {
                                        const uint32_t k = atomicAdd(&sScal[6], 1u);
                                        if (k < (uint32_t)Q_MAXC) sCandUnits[k] = x;
                                    }
                                }
                            }
                            if (alive) {
                                for (uint32_t x = (uint32_t)bb >> 4; x <= (b >> 4); ++x) {
                                    const uint32_t bit = 1u << (x & 31u);
                                    if (!(atomicOr(&sMap[x >> 5], bit) & bit)) {
                                        const uint32_t k = atomicAdd(&sScal[1], 1u);
                                        if (k < (uint32_t)Q_MAXU) sUnits[k] = x;
                                    }
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (a.sort_units && sScal[1] > 1u && sScal[1] <= (uint32_t)Q_MAXU) {
                    // ascending units: the lanes of one load instruction that work on neighbouring units of one row share its 128-B
                    // lines (the units arrive here in the order of the LDS atomics above).  Q_MAXU <= 64: the first wave ranks them.
                    const uint32_t na = sScal[1];
                    uint32_t x = 0xFFFFFFFFu, rank = 0;
                    if (tid < na) x = sUnits[tid];
                    if (tid < 64u) {
                        for (uint32_t j = 0; j < na; ++j) rank += (__shfl(x, (int)j) < x) ? 1u : 0u;
                    }
                    __syncthreads();
                    if (tid < na) sUnits[rank] = x;
                    __syncthreads();
                }
                if (first_eval) { PMARK(3) }                             // 3: which runs can still reach the threshold
                const uint32_t n_alive = sScal[1];
                if (n_alive == 0) break;
                if (n_alive > (uint32_t)Q_MAXU) { // too many survivors (long split runs, tiny thresholds): stay dense
                    query_dense_range<NT, U, BS, QC>(D, hp, done, n, sProbe, sC, staged);
                    touched += rem * 3ull * (D.units * 16u);
                    rows_read += rem * 3ull;
                    break;
                }
                if (first_eval) {
                    // staged probes: thirds (halves for medium reads, one stage for short ones -- a stage costs a pass over
                    // the bins and two barriers); otherwise a stage is what fits the probe staging area
                    const uint32_t want = a.sparse_stages ? a.sparse_stages : 3u;
                    const uint32_t stages = rem >= 128u * want ? want : max(1u, (uint32_t)(rem / 128u));
                    chunk = staged ? (uint32_t)((rem + stages - 1u) / stages) : min((uint32_t)Q_HT2, (uint32_t)QC);
                    first_eval = false;
                }
                const uint32_t nt = min(chunk, n - done);
                const uint4 *pr = sProbe + (staged ? done : 0u);
                if (!staged) {
                    for (uint32_t i = tid; i < nt; i += BS) {
                        const ixf_probe p = ixf_probe_key_arith(hp[done + i], D.seed, D.seg_len, D.arith);
                        sProbe[i] = make_uint4(p.row[0], p.row[1], p.row[2], p.fp4);
                    }
                    __syncthreads();
                }
                // one task = (hash, alive unit): three 16-B loads from three different rows.  A thread has only a few
                // tasks (nt * n_alive / 256) and each is a full memory round trip, so four are issued together (twelve
                // loads in flight per lane); one at a time this phase is a chain of dependent latencies.
                const uint32_t tasks = nt * n_alive;
                constexpr int SU = 4;
                for (uint32_t task0 = tid; task0 < tasks; task0 += BS * SU) {
                    uint4 r0[SU], r1[SU], r2[SU];
                    uint32_t xs[SU], fp4[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const uint32_t task = task0 + (uint32_t)u * BS;
                        xs[u] = 0xFFFFFFFFu;
                        r0[u] = r1[u] = r2[u] = make_uint4(0, 0, 0, 0);
                        fp4[u] = 0;
                        if (task < tasks) {
                            const uint32_t i = task / n_alive, j = task - i * n_alive;
                            xs[u] = sUnits[j];
                            const uint4 p = pr[i];
                            fp4[u] = p.w;
                            const uint8_t *base = D.data + (size_t)xs[u] * 16u;
                            r0[u] = ld16<NT>(base + (size_t)p.x * stride);
                            r1[u] = ld16<NT>(base + (size_t)p.y * stride);
                            r2[u] = ld16<NT>(base + (size_t)p.z * stride);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        if (xs[u] == 0xFFFFFFFFu) continue;
                        const uint32_t z[4] = {zero_bytes01(r0[u].x ^ r1[u].x ^ r2[u].x ^ fp4[u]), zero_bytes01(r0[u].y ^ r1[u].y ^ r2[u].y ^ fp4[u]),
                                               zero_bytes01(r0[u].z ^ r1[u].z ^ r2[u].z ^ fp4[u]), zero_bytes01(r0[u].w ^ r1[u].w ^ r2[u].w ^ fp4[u])};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            uint32_t m = z[q];
                            while (m) {
                                const int bit = __ffs((int)m) - 1; // 0, 8, 16 or 24
                                m &= m - 1u;
                                atomicAdd(&sC[xs[u] * 16u + 4u * (uint32_t)q + ((uint32_t)bit >> 3)], 1u);
                            }
                        }
                    }
                }
                touched += (uint64_t)nt * (uint64_t)n_alive * 3ull * 64ull; // one 64-B sector per 16-B unit load (sector64 accounting; HBM moves a 128-B line)
                rows_read += (uint64_t)nt * (uint64_t)n_alive * 3ull;
                sparse_loads += (uint64_t)nt * (uint64_t)n_alive * 3ull;
                done += nt;
                __syncthreads();                                         // the stage's counts are in the tally; sUnits / sProbe may be rewritten
                if (done >= n) break;
            }
        }
        __syncthreads();
        PMARK(4)                                                     // 4: sparse phase (surviving units only)

        if (a.counts_out)
            for (uint32_t b = tid; b < D.bins; b += BS) a.counts_out[b] = sC[b];

        // ---- tally: hierarchical_interleaved_xor_filter.hpp:313-338 --------------------------------------
        // A run that passes had `sum + rem * len >= thr` when the dense phase ended, so when pruning ran the walk covers just
        // the units recorded then (a handful) instead of every bin of the row: for a small item this pass over the bins
        // is a tenth of its time.  The loop bounds stay wave-uniform (wave_append ballots).
        const uint32_t n_cand = sScal[6];
        const bool cand_tally = cand_valid && n_cand <= (uint32_t)Q_MAXC;
        const uint32_t t_end = cand_tally ? ((n_cand * 16u + 63u) & ~63u) : nb_round;
        for (uint32_t t = tid; t < t_end; t += BS) {
            bool push_child = false, push_hit = false;
            uint32_t sum = 0, info = 0;
            uint32_t b = t;
            if (cand_tally) b = t < n_cand * 16u ? sCandUnits[t >> 4] * 16u + (t & 15u) : 0xFFFFFFFFu;
            if (b < D.bins) {
                info = sInfo[b];
                if (info & BINFO_MERGED) {
                    sum = sC[b];                                   // merged bins are runs of their own
                    push_child = (uint64_t)sum >= thr;             // :321
                } else if (info & BINFO_END) {
                    int bb = (int)b;
                    sum = sC[bb];
                    while (bb > 0 && (sInfo[bb - 1] >> 30) == 0u) sum += sC[--bb]; // split bin: :315,325-326
                    push_hit = (uint64_t)sum >= thr;               // :328
                }
            }
            const uint32_t qs = wave_append(push_child, &sScal[2]);             // LDS slot; a full buffer spills
            if (push_child) {
                const uint2 rec = make_uint2(r, info & 0x3FFFFFFFu);
                if (qs < (uint32_t)Q_OB) sOutQ[qs] = rec;
                else {
                    const uint32_t g = atomicAdd(&a.ctr->q_n[lvl + 1].v, 1u);
                    if (g < a.q_cap) {
                        if constexpr (TREE) __hip_atomic_store(&q64[g], (unsigned long long)rec.y << 32 | rec.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else a.q_out[g] = rec;
                    } else atomicOr(&a.ctr->flags, FLAG_QUEUE_OVERFLOW);
                }
            }
            const uint32_t hs = wave_append(push_hit, &sScal[3]);
            if (push_hit) {
                const uint4 rec = make_uint4(r, D.bin_base + b, sum, 0u);
                atomicAdd(&a.read_hits[r], 1u);
                if (hs < (uint32_t)Q_OB) sOutH[hs] = rec;
                else {                                                           // e.g. a threshold-0 read reporting every leaf run
                    const uint32_t g = atomicAdd(&a.ctr->n_hits.v, 1u);
                    if (g < a.hit_cap) a.hits[g] = rec;
                    else atomicOr(&a.ctr->flags, FLAG_HITS_OVERFLOW);
                }
            }
        }
        st_bytes += (unsigned long long)n * 3ull * D.bins;
        st_touched += touched;
        st_rows += rows_read;
        st_sparse += sparse_loads;
        st_work += 1ull;
        ++item;
        if (TREE && tid == 0) sScal[8] += 1u;                         // counted as completed when this block's pushes are out (flush_out)
        PMARK(5)                                                     // 5: run tally, child pushes, hit records
    }
    flush_out(true);           // the break above is taken by the whole block right after a barrier
    PMARK(6)
    if constexpr (PROF) {
        if (tid == 0 && a.prof)
            for (int i = 0; i < 8; ++i) atomicAdd(&a.prof[8 + i], (unsigned long long)pacc[i]);
    }
    if (tid == 0 && st_work) { // one set of statistics atomics per block, not per work item
        atomicAdd(&a.ctr->query_bytes, st_bytes);
        atomicAdd(&a.ctr->touched_bytes, st_touched);
        atomicAdd(&a.ctr->n_work, st_work);
        atomicAdd(&a.ctr->lvl_touched[min(lvl, 7u)], st_touched);
        atomicAdd(&a.ctr->lvl_rows[min(lvl, 7u)], st_rows);
        if (st_sparse) atomicAdd(&a.ctr->lvl_sparse[min(lvl, 7u)], st_sparse);
    }
}

void launch_query_tree(const QueryArgs &a, int grid, size_t lds_bytes, hipStream_t st, int unroll)
{
    // a small batch leaves the chip mostly idle: its items are latency chains, and four hashes (twelve row loads) in flight per
    // lane halve the round trips of the dense phase; the registers that costs (134: three waves per SIMD) are not missed here
    if (unroll == 4) hipLaunchKernelGGL((k_query_level<false, 4, false, BLK, Q_CAP, true>), dim3(grid), dim3(BLK), lds_bytes, st, a);
    else hipLaunchKernelGGL((k_query_level<false, 2, false, BLK, Q_CAP, true>), dim3(grid), dim3(BLK), lds_bytes, st, a);
}

int query_grid(int device, size_t lds_bytes, int want_per_cu)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 1024;
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_query_level<true, 2, false>, BLK, lds_bytes) != hipSuccess || per < 1)
        per = 2;
    // Measured: HBM streaming saturates from two resident blocks per CU (1 -> -4 %, 2 = 3 = 4), but the per-item
    // latency-bound phases (metadata fetch, probe staging, pruning check, tally) hide better with more: three for long
    // reads (+14 % viral-class over two), four for short reads, whose items spend half their time outside the gather
    // loop (1-kb reads +4 % unrelated / +11 % family workload over three; 10-kb reads unchanged).  The caller asks.
    if (per > want_per_cu) per = want_per_cu;
    if (const char *e = tune_env("TAXOR_QUERY_BPC")) { const int v = atoi(e); if (v >= 1 && v <= 8) per = v; }
    return p.multiProcessorCount * per;
}

// single-wave blocks for tiny items: as many per CU as registers and LDS allow (sixteen at 108 VGPRs)
int query_grid_small(int device, size_t lds_bytes)
{
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 4096;
    int per = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_query_level<true, 2, false, Q_BLK_SMALL, Q_CAP_SMALL>, Q_BLK_SMALL, lds_bytes) != hipSuccess || per < 1)
        per = 8;
    if (per > 16) per = 16;
    if (const char *e = tune_env("TAXOR_QUERY_BPC_SMALL")) { const int v = atoi(e); if (v >= 1 && v <= 32) per = v; }
    return p.multiProcessorCount * per;
}

void launch_query_level(const QueryArgs &a, int grid, size_t lds_bytes, hipStream_t st, bool small, bool root_streams)
{
    // the root's rows: non-temporal when its table is far larger than the caches (random rows in tens of gigabytes: no
    // reuse to protect), plain when it is small enough for the 256 MB memory-side cache to matter (viral-class root,
    // 150 MB: +8 %; RefSeq-class, 4 GB: +3 %; GTDB-class, 45 GB: -1 %)
    static const int nt_env = [] { const char *e = tune_env("TAXOR_QUERY_NT"); return e ? atoi(e) : -1; }();
    const bool nt0 = nt_env >= 0 ? nt_env != 0 : root_streams;
    static const int unroll0 = [] { const char *e = tune_env("TAXOR_QUERY_UNROLL"); return e ? atoi(e) : 2; }();
    static const int unroll1 = [] { const char *e = tune_env("TAXOR_QUERY_UNROLL_L1"); return e ? atoi(e) : 0; }();   // levels below the root
    const int unroll = (a.level >= 1 && unroll1) ? unroll1 : unroll0;
    // levels below the root: their items arrive grouped by IXF (launch_queue_group_by_ixf), so consecutive items re-read
    // the rows of the same child -- plain loads let them stay in L2 / the memory-side cache; the root's rows are random
    // in tens of gigabytes and stream (non-temporal)
    static const int nt1 = [] { const char *e = tune_env("TAXOR_QUERY_NT_L1"); return e ? atoi(e) : 0; }();
    const bool nt = a.level >= 1 ? nt1 != 0 : nt0;
    if (small) {
        if (nt) hipLaunchKernelGGL((k_query_level<true, 2, false, Q_BLK_SMALL, Q_CAP_SMALL>), dim3(grid), dim3(Q_BLK_SMALL), lds_bytes, st, a);
        else hipLaunchKernelGGL((k_query_level<false, 2, false, Q_BLK_SMALL, Q_CAP_SMALL>), dim3(grid), dim3(Q_BLK_SMALL), lds_bytes, st, a);
        return;
    }
    if (a.prof) {
        hipLaunchKernelGGL((k_query_level<true, 2, true>), dim3(grid), dim3(BLK), lds_bytes, st, a);
        return;
    }
    if (unroll == 2) {
        if (nt) hipLaunchKernelGGL((k_query_level<true, 2>), dim3(grid), dim3(BLK), lds_bytes, st, a);
        else hipLaunchKernelGGL((k_query_level<false, 2>), dim3(grid), dim3(BLK), lds_bytes, st, a);
    } else {
        if (nt) hipLaunchKernelGGL((k_query_level<true, 4>), dim3(grid), dim3(BLK), lds_bytes, st, a);
        else hipLaunchKernelGGL((k_query_level<false, 4>), dim3(grid), dim3(BLK), lds_bytes, st, a);
    }
}

// ------------------------------------------------------------------------------------------------------
// Work-queue ordering between levels.  The (read, IXF) items a level pushes arrive in no particular order; the next
// level takes them in queue order, so grouping them by IXF makes the blocks that run at one time read the SAME few
// child IXFs -- a child of tens of megabytes then stays in the memory-side cache while its items are processed (reads of
// an abundant organism all descend into the same children), instead of every fingerprint row being a DRAM row
// activation.  A counting sort by IXF id: histogram, scan, scatter.  The order inside a group does not matter -- hits
// are ordered per read by DFS key at the end -- and results do not change.
// ------------------------------------------------------------------------------------------------------
// Block-aggregated: a block takes a contiguous chunk of the queue, counts its IXF ids in LDS and touches the global
// counters once per (block, id) -- the items of a sub-batch concentrate on a few dozen child IXFs, and one global atomic per
// ITEM on those few words costs more than the level it orders (1-kb reads: 12 ms per step).
static constexpr uint32_t QG_LDS_IDS = 8192;    // IXF ids counted in LDS; larger ids (huge hierarchies) go to the global words directly

__device__ __forceinline__ void queue_chunk(uint32_t n, uint32_t &lo, uint32_t &hi)
{
    const uint32_t per = (n + gridDim.x - 1u) / gridDim.x;
    lo = min(blockIdx.x * per, n);
    hi = min(lo + per, n);
}

__global__ __launch_bounds__(BLK) void k_queue_hist(const uint2 *__restrict__ q, const Counters *__restrict__ ctr, uint32_t lvl,
                                                    uint32_t q_cap, uint32_t *__restrict__ hist, uint32_t n_ixf)
{
    __shared__ uint32_t sH[QG_LDS_IDS];
    const uint32_t n = min(ctr->q_n[lvl].v, q_cap), nl = min(n_ixf, QG_LDS_IDS);
    uint32_t lo, hi;
    queue_chunk(n, lo, hi);
    if (lo >= hi) return;
    for (uint32_t i = threadIdx.x; i < nl; i += BLK) sH[i] = 0u;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += BLK) {
        const uint32_t id = q[i].y;
        if (id < nl) atomicAdd(&sH[id], 1u);
        else atomicAdd(&hist[id], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nl; i += BLK)
        if (sH[i]) atomicAdd(&hist[i], sH[i]);
}

__global__ __launch_bounds__(1024) void k_queue_scan(uint32_t *__restrict__ hist, uint32_t n_ixf)
{
    __shared__ uint32_t sW[16];
    __shared__ uint32_t sCarry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) sCarry = 0;
    for (uint32_t b0 = 0; b0 < n_ixf; b0 += 1024) {
        __syncthreads();
        const uint32_t i = b0 + tid;
        const uint32_t v = i < n_ixf ? hist[i] : 0u;
        const uint32_t incl = wave_incl_add(v);
        if (lane_id() == 63) sW[tid >> 6] = incl;
        __syncthreads();
        uint32_t off = sCarry, tot = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t x = sW[w];
            if (w < (tid >> 6)) off += x;
            tot += x;
        }
        if (i < n_ixf) hist[i] = off + incl - v;          // exclusive start of this IXF's group
        __syncthreads();
        if (tid == 0) sCarry += tot;
    }
}

__global__ __launch_bounds__(BLK) void k_queue_scatter(const uint2 *__restrict__ q, const Counters *__restrict__ ctr, uint32_t lvl,
                                                       uint32_t q_cap, uint32_t *__restrict__ hist, uint2 *__restrict__ out, uint32_t n_ixf)
{
    __shared__ uint32_t sH[QG_LDS_IDS];       // count of this block's chunk per id, then the block's next slot for that id
    const uint32_t n = min(ctr->q_n[lvl].v, q_cap), nl = min(n_ixf, QG_LDS_IDS);
    uint32_t lo, hi;
    queue_chunk(n, lo, hi);
    if (lo >= hi) return;
    for (uint32_t i = threadIdx.x; i < nl; i += BLK) sH[i] = 0u;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += BLK) {
        const uint32_t id = q[i].y;
        if (id < nl) atomicAdd(&sH[id], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nl; i += BLK)
        if (sH[i]) sH[i] = atomicAdd(&hist[i], sH[i]);      // reserve this block's run inside the id's group
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += BLK) {
        const uint2 it = q[i];
        const uint32_t pos = it.y < nl ? atomicAdd(&sH[it.y], 1u) : atomicAdd(&hist[it.y], 1u);
        out[pos] = it;
    }
}

void launch_queue_group_by_ixf(const uint2 *q, const Counters *ctr, uint32_t lvl, uint32_t q_cap, uint32_t *hist, uint32_t n_ixf,
                               uint2 *out, hipStream_t st)
{
    (void)hipMemsetAsync(hist, 0, (size_t)n_ixf * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_queue_hist, dim3(256), dim3(BLK), 0, st, q, ctr, lvl, q_cap, hist, n_ixf);
    hipLaunchKernelGGL(k_queue_scan, dim3(1), dim3(1024), 0, st, hist, n_ixf);
    hipLaunchKernelGGL(k_queue_scatter, dim3(256), dim3(BLK), 0, st, q, ctr, lvl, q_cap, hist, out, n_ixf);
}

// ------------------------------------------------------------------------------------------------------
// finalize: per-read tuple counts -> CSR offsets -> scatter -> sort each read's tuples by DFS key
// ------------------------------------------------------------------------------------------------------
// CSR offsets of a sub-batch = exclusive scan of its per-read tuple counts.  Three small launches (block totals, scan of
// the totals + bookkeeping, local scan + block offset): a single 1024-thread block walking a sub-batch of 524288 short reads
// took 0.6 ms by itself.
static constexpr uint32_t SCAN_PER_BLOCK = 4096;    // reads per block: 1024 threads x 4

__global__ __launch_bounds__(1024) void k_scan_block_totals(const FinalizeArgs a, uint32_t *__restrict__ block_sums)
{
    __shared__ uint32_t sW[16];
    const uint32_t tid = threadIdx.x, base = blockIdx.x * SCAN_PER_BLOCK + tid * 4u;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        if (base + j < a.n_reads) v += a.read_hits[base + j];
    const uint32_t incl = wave_incl_add(v);
    if (lane_id() == 63) sW[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < 16; ++w) t += sW[w];
        block_sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(1024) void k_scan_totals(const FinalizeArgs a, uint32_t *__restrict__ block_sums, uint32_t n_blocks)
{
    __shared__ uint32_t sW[16];
    __shared__ uint32_t sCarry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) sCarry = 0;
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += 1024) {
        __syncthreads();
        const uint32_t i = b0 + tid;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        const uint32_t incl = wave_incl_add(v);
        if (lane_id() == 63) sW[tid >> 6] = incl;
        __syncthreads();
        uint32_t off = sCarry, tot = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            const uint32_t x = sW[w];
            if (w < (tid >> 6)) off += x;
            tot += x;
        }
        if (i < n_blocks) block_sums[i] = off + incl - v;          // exclusive start of block i
        __syncthreads();
        if (tid == 0) sCarry += tot;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long base = a.ctr->tuple_total;
        const unsigned long long total = base + sCarry;
        block_sums[n_blocks] = (uint32_t)(base & 0xFFFFFFFFull);  // the batch-wide base of this sub-batch, for the next kernel
        block_sums[n_blocks + 1] = (uint32_t)(base >> 32);
        a.roff[a.n_reads] = sCarry;
        a.ctr->tuple_total = total;
        if (total > a.tuple_cap) atomicOr(&a.ctr->flags, FLAG_TUPLE_OVERFLOW);
        if (a.is_last) a.read_off[a.n_reads] = total;
    }
}

__global__ __launch_bounds__(1024) void k_scan_offsets(const FinalizeArgs a, const uint32_t *__restrict__ block_sums, uint32_t n_blocks)
{
    __shared__ uint32_t sW[16];
    const uint32_t tid = threadIdx.x, first = blockIdx.x * SCAN_PER_BLOCK + tid * 4u;
    const unsigned long long base = (unsigned long long)block_sums[n_blocks] | ((unsigned long long)block_sums[n_blocks + 1] << 32);
    uint32_t c[4], v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        c[j] = first + j < a.n_reads ? a.read_hits[first + j] : 0u;
        v += c[j];
    }
    const uint32_t incl = wave_incl_add(v);
    if (lane_id() == 63) sW[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = block_sums[blockIdx.x];
    for (uint32_t w = 0; w < (tid >> 6); ++w) off += sW[w];
    off += incl - v;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        if (first + j < a.n_reads) {
            a.roff[first + j] = off;
            a.read_off[first + j] = base + off;
        }
        off += c[j];
    }
}

__global__ __launch_bounds__(BLK) void k_scatter_hits(const FinalizeArgs a)
{
    // after a queue/hit overflow the host grows the buffers and reruns; nothing here may index past hit_cap
    if (a.ctr->flags & (FLAG_QUEUE_OVERFLOW | FLAG_HITS_OVERFLOW)) return;
    const uint32_t n = min(a.ctr->n_hits.v, a.hit_cap);
    const unsigned long long base = a.ctr->tuple_total - a.roff[a.n_reads]; // first tuple of this sub-batch
    for (uint32_t i = blockIdx.x * BLK + threadIdx.x; i < n; i += gridDim.x * BLK) {
        const uint4 h = a.hits[i];
        const uint32_t slot = atomicAdd(&a.cursor[h.x], 1u);
        const unsigned long long pos = base + a.roff[h.x] + slot;
        if (pos < a.tuple_cap) {
            a.out_key[pos] = a.dfs_key[h.y];
            a.out_ub[pos] = a.ubin[h.y];
            a.out_cnt[pos] = h.z;
        }
    }
}

// one wave per read; reads with more than 64 tuples go to the block-wide sorter
__global__ __launch_bounds__(BLK) void k_sort_small(const FinalizeArgs a)
{
    const uint32_t wave = (blockIdx.x * BLK + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * BLK) >> 6;
    const unsigned long long base0 = a.ctr->tuple_total - a.roff[a.n_reads];
    if (a.ctr->tuple_total > a.tuple_cap || (a.ctr->flags & (FLAG_QUEUE_OVERFLOW | FLAG_HITS_OVERFLOW))) return;
    for (uint32_t r = wave; r < a.n_reads; r += nwaves) {
        const uint32_t n = a.read_hits[r];
        if (n < 2) continue;
        if (n > 64) {
            if (lane_id() == 0) a.biglist[atomicAdd(&a.ctr->n_big, 1u)] = r;
            continue;
        }
        const unsigned long long base = base0 + a.roff[r];
        const uint32_t l = lane_id();
        uint32_t key = 0xFFFFFFFFu, cnt = 0;
        int64_t ub = 0;
        if (l < n) { key = a.out_key[base + l]; ub = a.out_ub[base + l]; cnt = a.out_cnt[base + l]; }
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += (__shfl(key, (int)j) < key) ? 1u : 0u;
        if (l < n) { a.out_key[base + rank] = key; a.out_ub[base + rank] = ub; a.out_cnt[base + rank] = cnt; }
    }
}

// block per big read: sorting network with ascending comparators only (flip stage + half-cleaners);
// elements past n act as +inf and are never touched
__device__ __forceinline__ void sort_stage(const FinalizeArgs &a, unsigned long long base, uint32_t n, uint32_t N,
                                           uint32_t x)
{
    for (uint32_t i = threadIdx.x; i < N; i += BLK) {
        const uint32_t l = i ^ x;
        if (l > i && l < n) {
            const uint32_t ki = a.out_key[base + i], kl = a.out_key[base + l];
            if (kl < ki) {
                a.out_key[base + i] = kl;
                a.out_key[base + l] = ki;
                const int64_t ui = a.out_ub[base + i];
                a.out_ub[base + i] = a.out_ub[base + l];
                a.out_ub[base + l] = ui;
                const uint32_t ci = a.out_cnt[base + i];
                a.out_cnt[base + i] = a.out_cnt[base + l];
                a.out_cnt[base + l] = ci;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
}

__global__ __launch_bounds__(BLK) void k_sort_big(const FinalizeArgs a)
{
    const uint32_t nbig = a.ctr->n_big;
    const unsigned long long base0 = a.ctr->tuple_total - a.roff[a.n_reads];
    if (a.ctr->tuple_total > a.tuple_cap || (a.ctr->flags & (FLAG_QUEUE_OVERFLOW | FLAG_HITS_OVERFLOW))) return;
    for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const uint32_t r = a.biglist[bi];
        const uint32_t n = a.read_hits[r];
        const unsigned long long base = base0 + a.roff[r];
        uint32_t N = 1;
        while (N < n) N <<= 1;
        __syncthreads();
        for (uint32_t kk = 2; kk <= N; kk <<= 1) {
            sort_stage(a, base, n, N, kk - 1u);                       // mirror within blocks of kk
            for (uint32_t j = kk >> 2; j > 0; j >>= 1) sort_stage(a, base, n, N, j);
        }
    }
}

void launch_finalize(const FinalizeArgs &a, hipStream_t st)
{
    const uint32_t n_blocks = (a.n_reads + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK;
    if (n_blocks) hipLaunchKernelGGL(k_scan_block_totals, dim3(n_blocks), dim3(1024), 0, st, a, a.block_sums);
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, st, a, a.block_sums, n_blocks);
    if (n_blocks) hipLaunchKernelGGL(k_scan_offsets, dim3(n_blocks), dim3(1024), 0, st, a, a.block_sums, n_blocks);
    hipLaunchKernelGGL(k_scatter_hits, dim3(1024), dim3(BLK), 0, st, a);
    const uint32_t waves_needed = a.n_reads;
    uint32_t grid = (waves_needed + 3u) / 4u;
    if (grid > 4096u) grid = 4096u;
    if (grid == 0) grid = 1;
    hipLaunchKernelGGL(k_sort_small, dim3(grid), dim3(BLK), 0, st, a);
    hipLaunchKernelGGL(k_sort_big, dim3(256), dim3(BLK), 0, st, a);
}

// ------------------------------------------------------------------------------------------------------
// k_finalize_small: the six launches above as one block, for batches of up to SMALL_FIN_MAX reads (the reference's chunk is
// 1024 records, taxor_search.cpp:315) whose results leave through host memory the device writes directly -- a call of that size
// is a millisecond, and six launches, five result copies and three memsets were a fifth of it.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_finalize_small(const SmallFinalizeArgs a)
{
    __shared__ uint32_t sRoff[SMALL_FIN_MAX + 1];
    __shared__ uint32_t sCur[SMALL_FIN_MAX];
    __shared__ uint32_t sW[16];
    __shared__ uint32_t sBig[64];
    __shared__ uint32_t sNBig;
    const uint32_t tid = threadIdx.x, n = a.n_reads;
    const uint32_t flags = a.ctr->flags;
    const uint32_t n_hits = min(a.ctr->n_hits.v, a.hit_cap);
    // exclusive scan of the per-read tuple counts (four reads per thread)
    uint32_t c[4], v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t r = tid * 4u + j;
        c[j] = r < n ? a.read_hits[r] : 0u;
        v += c[j];
        if (r < SMALL_FIN_MAX) sCur[r] = 0u;
    }
    if (tid == 0) sNBig = 0u;
    const uint32_t incl = wave_incl_add(v);
    if (lane_id() == 63) sW[tid >> 6] = incl;
    __syncthreads();
    uint32_t off = incl - v, total = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < (tid >> 6)) off += sW[w];
        total += sW[w];
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t r = tid * 4u + j;
        if (r < n) sRoff[r] = off;
        off += c[j];
    }
    if (tid == 0) sRoff[n] = total;
    const bool overflow = (flags & (FLAG_QUEUE_OVERFLOW | FLAG_HITS_OVERFLOW | FLAG_TREE_STALL)) != 0u || total > a.tuple_cap;     // (a stalled traversal's hits are partial)
    if (tid == 0) {
        a.h_status[0] = (uint64_t)flags | (total > a.tuple_cap ? (uint64_t)FLAG_TUPLE_OVERFLOW : 0ull);
        a.h_status[1] = total;
        a.h_status[2] = a.ctr->n_hashes;
        a.h_status[3] = a.ctr->n_work;
        a.h_status[4] = a.ctr->query_bytes;
        a.h_status[5] = a.ctr->touched_bytes;
    }
    __syncthreads();
    if (!overflow) {
        for (uint32_t r = tid; r <= n; r += 1024u) a.h_read_off[r] = sRoff[r];
        for (uint32_t r = tid; r < n; r += 1024u) a.h_nh[r] = a.nh[r];
        // hit records -> their read's slice, in arrival order
        for (uint32_t i = tid; i < n_hits; i += 1024u) {
            const uint4 h = a.hits[i];
            const uint32_t pos = sRoff[h.x] + atomicAdd(&sCur[h.x], 1u);
            a.key[pos] = a.dfs_key[h.y];
            a.ub[pos] = a.ubin[h.y];
            a.cnt[pos] = h.z;
        }
        __threadfence_block();
        __syncthreads();
        // DFS order within a read (hierarchical_interleaved_xor_filter.hpp:313-338): one wave per read, ranks by comparison;
        // the sorted tuples go straight to the host arrays
        const uint32_t wave = tid >> 6, l = lane_id();
        for (uint32_t r = wave; r < n; r += 16u) {
            const uint32_t base = sRoff[r], m = sRoff[r + 1] - base;
            if (m == 0u) continue;
            if (m > 64u) {
                if (l == 0) { const uint32_t k = atomicAdd(&sNBig, 1u); if (k < 64u) sBig[k] = r; }
                continue;
            }
            uint32_t key = 0xFFFFFFFFu, cnt = 0;
            int64_t ub = 0;
            if (l < m) { key = a.key[base + l]; ub = a.ub[base + l]; cnt = a.cnt[base + l]; }
            uint32_t rank = 0;
            for (uint32_t j = 0; j < m; ++j) rank += (__shfl(key, (int)j) < key) ? 1u : 0u;
            if (l < m) { a.ub[base + rank] = ub; a.cnt[base + rank] = cnt; }       // in place: the wave has read all m before it writes
        }
        __threadfence_block();
        __syncthreads();
        // reads with more than 64 tuples (a threshold-0 read reports every leaf run): block-wide sorting network in the device
        // scratch, then out; more than 64 such reads in one small batch are found by a second sweep
        const uint32_t n_big = sNBig;
        for (uint32_t bi = 0; bi < n_big; ++bi) {
            uint32_t r;
            if (bi < 64u) r = sBig[bi];
            else {            // beyond the list: the bi-th read with more than 64 tuples, found by counting (block-uniform)
                uint32_t seen = 0;
                r = 0;
                for (uint32_t q = 0; q < n; ++q)
                    if (sRoff[q + 1] - sRoff[q] > 64u && seen++ == bi) { r = q; break; }
            }
            const uint32_t base = sRoff[r], m = sRoff[r + 1] - base;
            uint32_t N = 1;
            while (N < m) N <<= 1;
            auto stage = [&](uint32_t x) {                                     // ascending compare-exchange with partner i ^ x
                for (uint32_t i = tid; i < N; i += 1024u) {
                    const uint32_t p = i ^ x;
                    if (p > i && p < m) {
                        const uint32_t ki = a.key[base + i], kp = a.key[base + p];
                        if (kp < ki) {
                            a.key[base + i] = kp; a.key[base + p] = ki;
                            const int64_t ui = a.ub[base + i]; a.ub[base + i] = a.ub[base + p]; a.ub[base + p] = ui;
                            const uint32_t ci = a.cnt[base + i]; a.cnt[base + i] = a.cnt[base + p]; a.cnt[base + p] = ci;
                        }
                    }
                }
                __threadfence_block();
                __syncthreads();
            };
            for (uint32_t kk = 2; kk <= N; kk <<= 1) {
                stage(kk - 1u);                                                // mirror within blocks of kk
                for (uint32_t j = kk >> 2; j > 0; j >>= 1) stage(j);           // half-cleaners
            }
        }
        // out to the host in whole lines (scattered 8-byte stores over PCIe were most of this kernel's time)
        for (uint32_t i = tid; i < total; i += 1024u) { a.h_ub[i] = a.ub[i]; a.h_cnt[i] = a.cnt[i]; }
    }
    // leave the lane's counters as a fresh searcher has them: the next batch on this lane starts without memset launches
    __syncthreads();
    for (uint32_t r = tid; r < n; r += 1024u) a.read_hits[r] = 0u;
    uint32_t *cw = reinterpret_cast<uint32_t *>(a.ctr);
    for (uint32_t i = tid; i < (uint32_t)(sizeof(Counters) / 4u); i += 1024u) cw[i] = 0u;
    if (tid < 2u) a.sync_cursor[tid] = 0u;
}

void launch_finalize_small(const SmallFinalizeArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_finalize_small, dim3(1), dim3(1024), 0, st, a);
}

// ------------------------------------------------------------------------------------------------------
// gather ceiling: the same access pattern as the dense phase of k_query_level -- whole rows of one
// IXF at random row indices, 16 B per lane, neighbouring lanes on neighbouring units of one row -- with nothing else:
// no probes, no counting, no queues.  What this reaches is what the memory system gives a random-row reader of that
// row size; the query kernel's requested-bytes rate is judged against it.
// ------------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(BLK) void k_gather_ceiling(const uint8_t *data, uint64_t rows, uint32_t stride, uint32_t units,
                                                        uint32_t passes, uint64_t seed, uint32_t *sink, uint32_t n_ixf,
                                                        uint64_t spacing)
{
    const uint32_t per_pass = BLK / units;                  // rows one block reads per pass
    const uint32_t slot = threadIdx.x / units, u = threadIdx.x - slot * units;
    if (slot >= per_pass) return;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint64_t d = ((uint64_t)blockIdx.x * passes) * per_pass + slot;
    constexpr int U = 8;
    for (uint32_t p = 0; p < passes; p += U) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint64_t h = murmur64(seed + d + (uint64_t)j * per_pass);
            const uint64_t row = (uint64_t)(((unsigned __int128)h * rows) >> 64);
            // n_ixf equally shaped IXFs `spacing` bytes apart (the children of a synthetic index): a random one of them
            const uint64_t which = n_ixf > 1u ? (uint64_t)__umulhi((uint32_t)(h * 0x9E3779B97F4A7C15ull >> 32), n_ixf) : 0ull;
            v[j] = ld16<NT>(data + which * spacing + row * stride + (uint64_t)u * 16);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        d += (uint64_t)U * per_pass;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) atomicAdd(sink, 1u);   // keeps the loads alive
}

// The sparse phase's access shape with nothing else: every lane one 16-B load from a row and unit of its own (in the
// query kernel: one (hash, alive unit) task per lane, three rows each).  Used to calibrate the traffic counter for that
// shape and to know what the memory system gives it.
template <bool NT>
__global__ __launch_bounds__(BLK) void k_gather_sparse(const uint8_t *data, uint64_t rows, uint32_t stride, uint32_t units, uint32_t passes,
                                                       uint64_t seed, uint32_t *sink)
{
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint64_t d = ((uint64_t)blockIdx.x * BLK + threadIdx.x) * passes;
    constexpr int U = 12;                                   // twelve loads in flight per lane, like four sparse tasks
    for (uint32_t p = 0; p < passes; p += U) {
        uint4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const uint64_t h = murmur64(seed + d + (uint64_t)j);
            const uint64_t row = (uint64_t)(((unsigned __int128)h * rows) >> 64);
            const uint32_t u = __umulhi((uint32_t)(h * 0x9E3779B97F4A7C15ull >> 32), units);
            v[j] = ld16<NT>(data + row * stride + (uint64_t)u * 16);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        d += U;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) atomicAdd(sink, 1u);
}

uint64_t launch_gather_sparse(const uint8_t *data, uint64_t rows, uint32_t stride, uint32_t bins, uint64_t want_loads, uint64_t seed,
                              uint32_t *sink, bool nt, hipStream_t st)
{
    const uint32_t units = (bins + 15) / 16;
    const uint32_t passes = 96;                             // a multiple of the unroll
    const uint64_t per_block = (uint64_t)BLK * passes;
    uint64_t grid = std::max<uint64_t>(1, want_loads / per_block);
    if (grid > (1u << 30)) grid = 1u << 30;
    if (nt) hipLaunchKernelGGL(k_gather_sparse<true>, dim3((uint32_t)grid), dim3(BLK), 0, st, data, rows, stride, units, passes, seed, sink);
    else hipLaunchKernelGGL(k_gather_sparse<false>, dim3((uint32_t)grid), dim3(BLK), 0, st, data, rows, stride, units, passes, seed, sink);
    return grid * per_block;
}

uint64_t launch_gather_ceiling(const uint8_t *data, uint64_t rows, uint32_t stride, uint32_t bins, uint64_t want_bytes,
                               uint64_t seed, uint32_t *sink, bool nt, hipStream_t st, uint32_t n_ixf, uint64_t spacing)
{
    const uint32_t units = (bins + 15) / 16;                // like the query kernel: only units that hold bins are read
    const uint32_t per_pass = BLK / units;
    const uint32_t passes = 256;
    const uint64_t per_block = (uint64_t)passes * per_pass * units * 16;
    uint64_t grid = std::max<uint64_t>(1, want_bytes / per_block);
    if (grid > (1u << 30)) grid = 1u << 30;
    if (nt) hipLaunchKernelGGL(k_gather_ceiling<true>, dim3((uint32_t)grid), dim3(BLK), 0, st, data, rows, stride, units, passes, seed, sink, n_ixf, spacing);
    else hipLaunchKernelGGL(k_gather_ceiling<false>, dim3((uint32_t)grid), dim3(BLK), 0, st, data, rows, stride, units, passes, seed, sink, n_ixf, spacing);
    return grid * per_block;
}

// ------------------------------------------------------------------------------------------------------
// index construction helpers
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void k_fill_random(uint4 *data, uint64_t n16, uint64_t seed)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * BLK) {
        uint64_t z0 = murmur64(seed + 2ull * i + 1ull), z1 = murmur64(seed ^ (0x9E3779B97F4A7C15ull * (2ull * i + 2ull)));
        data[i] = make_uint4((uint32_t)z0, (uint32_t)(z0 >> 32), (uint32_t)z1, (uint32_t)(z1 >> 32));
    }
}

void launch_fill_random(uint8_t *data, uint64_t n_bytes, uint64_t seed, hipStream_t st)
{
    if (!n_bytes) return;
    hipLaunchKernelGGL(k_fill_random, dim3(8192), dim3(BLK), 0, st, reinterpret_cast<uint4 *>(data), n_bytes / 16, seed);
}

__global__ __launch_bounds__(BLK) void k_scatter_column(uint8_t *data, uint64_t stride, uint64_t bin,
                                                        const uint8_t *col, uint64_t rows)
{
    for (uint64_t i = (uint64_t)blockIdx.x * BLK + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * BLK)
        data[i * stride + bin] = col[i];
}

void launch_scatter_column(uint8_t *data, uint64_t stride, uint64_t bin, const uint8_t *col, uint64_t rows,
                           hipStream_t st)
{
    if (!rows) return;
    uint64_t grid = (rows + BLK - 1) / BLK;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_scatter_column, dim3((uint32_t)grid), dim3(BLK), 0, st, data, stride, bin, col, rows);
}

} // namespace taxor
