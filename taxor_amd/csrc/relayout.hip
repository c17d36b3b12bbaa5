// relayout.hip -- fingerprint bytes that are NOT in the search layout -> the slab, transposed on the device while they are uploaded.
//
// The kernels read one layout (data[row * stride + bin], stride a multiple of 64, rows in segment order).  What the un-vendored
// serialiser of seqan3::interleaved_xor_filter writes (hierarchical_interleaved_xor_filter.hpp:152-158 hands ixf_vector to it) is
// only known once a published file is at hand; ixf_layout.h lists the layouts `taxor verify --variants` / `taxor pin` can name,
// and this file makes every one of them searchable: index creation pulls the source's bytes chunk by chunk into page-locked
// staging (several threads, each with two buffers and a stream of its own, like the plain upload in api.hip), copies a chunk to a
// device staging buffer and runs the kernel that writes it into its place in the slab:
//   k_rows_repitch   row-interleaved with another pitch (e.g. exactly `bins`), and / or rows in position-major order
//   k_bin_major      data[bin * rows + row]: 128-bin x 128-row tiles through LDS, 128-B lines read and 128-B lines written
//   k_bit_sliced     eight 64-bit plane words per (row, 64 bins): 8 x 8 bit transposes in registers
// One launch per 8-MiB chunk: 40-57 us each, 140-210 GB/s per stream -- far above the PCIe rate that feeds them and 13-19 % of the
// load's wall time (profiles/r05/relayout_kernel_stats.txt; the load stays bound by pread + PCIe: 41 GB/s bin-major against 44 GB/s
// for the search layout at 64 GB).  The search layout itself never comes through here (api.hip index_upload copies it as it is).
#include "../../include/taxor_gpu_tools.h"
#include "ixf_layout.h"
#include "tuning.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);
extern "C" __attribute__((visibility("hidden"))) int taxor_index_ixf_info(taxor_gpu_index *idx, uint64_t ixf, uint8_t **data, uint64_t *stride, uint64_t *seg_len,
                                                                          uint64_t *bins, int *device);

namespace {

using namespace taxor;

// search-layout row of source row rs
__device__ inline uint64_t dst_row(uint32_t code, uint64_t rs, uint64_t seg_len)
{
    if (!(code & IXF_ROWS_POSITION_MAJOR)) return rs;
    const uint64_t pos = rs / 3u;
    return (rs - pos * 3u) * seg_len + pos;
}

// stage = nr source rows of `pitch` bytes each, source rows [r0, r0 + nr).  One thread per destination dword (4 bins); source
// bytes are fetched as two aligned dwords and funnel-shifted (a pitch that is not a multiple of 4 leaves rows unaligned); bins
// beyond `bins` are written as zeros.  The staging buffer has 8 bytes of slack behind the chunk.
__global__ __launch_bounds__(256) void k_rows_repitch(const uint8_t *__restrict__ stage, uint8_t *__restrict__ dst, uint32_t code, uint64_t r0, uint64_t nr,
                                                      uint64_t pitch, uint64_t stride, uint64_t bins, uint64_t seg_len)
{
    const uint64_t dw_per_row = stride / 4u;
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= nr * dw_per_row) return;
    const uint64_t rl = i / dw_per_row, j = i - rl * dw_per_row, b = 4u * j;
    uint32_t v = 0;
    if (b < bins) {
        const uint64_t o = rl * pitch + b;
        const uint32_t *w = reinterpret_cast<const uint32_t *>(stage + (o & ~(uint64_t)3));
        const uint32_t sh = (uint32_t)(o & 3u) * 8u;
        const uint64_t two = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        v = (uint32_t)(two >> sh);
        if (bins - b < 4u) v &= (1u << ((uint32_t)(bins - b) * 8u)) - 1u;
    }
    *reinterpret_cast<uint32_t *>(dst + dst_row(code, r0 + rl, seg_len) * stride + b) = v;
}

// stage = nb columns (bins b0 .. b0 + nb) of nr source rows each, column c at stage + c * col_pitch (col_pitch a multiple of 16).
// Block (tile_r, tile_c) transposes 128 rows x 128 columns: column strips in as dwords (128 B of one bin per 32 lanes), rows out
// as dwords (128 B of one row per 32 lanes).
__global__ __launch_bounds__(256) void k_bin_major(const uint8_t *__restrict__ stage, uint8_t *__restrict__ dst, uint32_t code, uint64_t r0, uint64_t nr,
                                                   uint64_t col_pitch, uint64_t b0, uint64_t nb, uint64_t stride, uint64_t seg_len)
{
    __shared__ uint32_t tile[128][33];            // [column][row / 4], padded: the read-out walks columns
    const uint32_t t = threadIdx.x;
    const uint64_t tr = (uint64_t)blockIdx.x * 128u, tc = (uint64_t)blockIdx.y * 128u;
    for (uint32_t c = t >> 5; c < 128u; c += 8u) {
        const uint32_t q = t & 31u;
        uint32_t v = 0;
        if (tc + c < nb && tr + 4u * q < nr) v = *reinterpret_cast<const uint32_t *>(stage + (tc + c) * col_pitch + tr + 4u * q);   // (rows past nr inside the dword: unused below)
        tile[c][q] = v;
    }
    __syncthreads();
    const uint64_t width = min((uint64_t)128u, stride - (b0 + tc));          // bytes of this tile's rows inside the row (the last tile ends at the stride)
    for (uint32_t r = t >> 5; r < 128u; r += 8u) {
        if (tr + r >= nr) break;
        const uint32_t q = t & 31u;                                         // columns 4q .. 4q + 3
        if (4u * q >= width) continue;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t c = 4u * q + k;
            const uint32_t byte = (tile[c][r >> 2] >> ((r & 3u) * 8u)) & 0xFFu;
            v |= (tc + c < nb ? byte : 0u) << (8u * k);
        }
        *reinterpret_cast<uint32_t *>(dst + dst_row(code, r0 + tr + r, seg_len) * stride + b0 + tc + 4u * q) = v;
    }
}

// stage = nr source rows of groups * 64 bytes: per group of 64 bins eight u64 plane words (bit j of word p = bit p of bin 64 g + j).
// Eight lanes share one group: lane q loads word q, the eight exchange so that lane q holds byte q of every word (bins 8q .. 8q+7,
// one bit plane per byte), transposes the 8 x 8 bit matrix and stores the eight fingerprints as one u64.
__global__ __launch_bounds__(256) void k_bit_sliced(const uint8_t *__restrict__ stage, uint8_t *__restrict__ dst, uint32_t code, uint64_t r0, uint64_t nr,
                                                    uint64_t groups, uint64_t stride, uint64_t bins, uint64_t seg_len)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;          // (row, group, q)
    const uint64_t total = nr * groups * 8u;
    const uint64_t ic = i < total ? i : total - 1u;                          // every lane takes part in the exchange
    const uint64_t blk = ic >> 3;
    const uint32_t q = (uint32_t)(ic & 7u);
    const uint64_t w = *reinterpret_cast<const uint64_t *>(stage + blk * 64u + q * 8u);
    const int lane0 = (int)(threadIdx.x & 63u & ~7u);
    uint64_t x = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)w, lane0 + p), hi = (uint32_t)__shfl((int)(uint32_t)(w >> 32), lane0 + p);
        const uint64_t wp = (uint64_t)lo | ((uint64_t)hi << 32);
        x |= ((wp >> (8u * q)) & 0xFFull) << (8u * (uint32_t)p);             // byte p of x = plane p of bins 8q .. 8q+7
    }
    // 8 x 8 bit transpose (Hacker's Delight 7-3): afterwards byte j of x = fingerprint of bin 8q + j (bit p from plane p)
    uint64_t tt;
    tt = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x = x ^ tt ^ (tt << 7);
    tt = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ tt ^ (tt << 14);
    tt = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ tt ^ (tt << 28);
    if (i >= total) return;
    const uint64_t rl = blk / groups, g = blk - rl * groups, b = g * 64u + 8u * q;
    if (b >= stride) return;
    if (b >= bins) x = 0;
    else if (bins - b < 8u) x &= (1ull << ((uint32_t)(bins - b) * 8u)) - 1ull;
    *reinterpret_cast<uint64_t *>(dst + dst_row(code, r0 + rl, seg_len) * stride + b) = x;
}

struct Chunk {
    uint64_t ixf;
    uint64_t r0, nr;      // source rows
    uint64_t b0, nb;      // bin-major only: columns
};

int rfail(int code, const std::string &m)
{
    taxor_set_last_error(m.c_str());
    return code;
}

} // namespace

// library-internal (api.hip index_upload): every IXF of `v` -- source bytes under v->ixf_layout / ixf[i].src_stride -- into the
// slab of `idx` in the search layout.  Rows / columns beyond what a chunk covers are not touched; padding columns get zeros.
extern "C" __attribute__((visibility("hidden"))) int taxor_index_upload_relayout(taxor_gpu_index *idx, const taxor_hixf_view *v)
{
    const uint32_t code = v->ixf_layout, kind = ixf_layout_kind(code);
    if (!ixf_layout_valid(code)) return rfail(TAXOR_E_ARG, "index upload: unknown fingerprint layout code " + std::to_string(code));
    static const uint64_t piece_bytes = [] { const char *e = tune_env("TAXOR_UPLOAD_PIECE_MB"); const long m = e ? atol(e) : 0; return (uint64_t)(m > 0 ? m : 8) << 20; }();
    struct Ixf { uint8_t *dst; uint64_t stride, seg_len, bins, rows, pitch, groups; };
    std::vector<Ixf> X(v->n_ixf);
    std::vector<Chunk> chunks;
    int device = 0;
    uint64_t stage_bytes = 0;
    for (uint64_t i = 0; i < v->n_ixf; ++i) {
        Ixf &x = X[i];
        if (taxor_index_ixf_info(idx, i, &x.dst, &x.stride, &x.seg_len, &x.bins, &device) != 0) return rfail(TAXOR_E_ARG, "index upload: view does not match the index");
        x.rows = 3 * x.seg_len;
        x.groups = (x.bins + 63) / 64;
        x.pitch = kind == IXF_KIND_BIT_SLICED ? x.groups * 64 : ixf_src_pitch(v->ixf_layout, v->ixf[i].src_stride, x.stride, x.bins);
        if (x.pitch < x.bins) return rfail(TAXOR_E_ARG, "index upload: IXF " + std::to_string(i) + ": source pitch " + std::to_string(x.pitch) + " below its " + std::to_string(x.bins) + " bins");
        if (!v->source && !v->ixf[i].data) continue;
        if (kind == IXF_KIND_BIN_MAJOR) {
            const uint64_t nr_max = std::max<uint64_t>(128, (piece_bytes / 128) & ~(uint64_t)127);       // rows per strip: 128 columns of them fill a piece
            for (uint64_t b0 = 0; b0 < x.bins; b0 += 128)
                for (uint64_t r0 = 0; r0 < x.rows; r0 += nr_max) {
                    const Chunk c{i, r0, std::min(nr_max, x.rows - r0), b0, std::min<uint64_t>(128, x.bins - b0)};
                    stage_bytes = std::max(stage_bytes, c.nb * ((c.nr + 15) & ~(uint64_t)15));
                    chunks.push_back(c);
                }
        } else {
            const uint64_t nr_max = std::max<uint64_t>(1, piece_bytes / x.pitch);
            for (uint64_t r0 = 0; r0 < x.rows; r0 += nr_max) {
                const Chunk c{i, r0, std::min(nr_max, x.rows - r0), 0, 0};
                stage_bytes = std::max(stage_bytes, c.nr * x.pitch);
                chunks.push_back(c);
            }
        }
    }
    if (chunks.empty()) return 0;
    const auto t_begin = std::chrono::steady_clock::now();
    stage_bytes = (stage_bytes + 64 + 255) & ~(uint64_t)255;                 // slack: the re-pitch kernel reads whole dwords
    static const int n_threads = [] { const char *e = tune_env("TAXOR_UPLOAD_THREADS"); const int t = e ? atoi(e) : 0; return t >= 1 && t <= 64 ? t : 8; }();
    const int T = (int)std::min<size_t>((size_t)n_threads, chunks.size());
    std::atomic<size_t> cursor{0};
    std::atomic<int> failed{0};
    std::mutex mu;
    std::string err;
    int err_code = TAXOR_E_HIP;
    auto set_err = [&](int c, const std::string &m) {
        std::lock_guard<std::mutex> lk(mu);
        if (err.empty()) { err = m; err_code = c; }
        failed.store(1);
    };
    auto fetch = [&](uint64_t ixf, uint64_t off, uint64_t len, void *dstp) -> bool {
        if (v->source) return v->source->read(v->source->ctx, ixf, off, len, dstp) == 0;
        std::memcpy(dstp, v->ixf[ixf].data + off, len);
        return true;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&] {
            hipStream_t st = nullptr;
            void *hbuf[2] = {nullptr, nullptr};
            uint8_t *dbuf[2] = {nullptr, nullptr};
            hipEvent_t ev[2] = {nullptr, nullptr};
            bool busy[2] = {false, false};
            hipError_t e = hipSetDevice(device);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            for (int b = 0; b < 2 && e == hipSuccess; ++b) {
                e = hipHostMalloc(&hbuf[b], stage_bytes, hipHostMallocDefault);
                if (e == hipSuccess) e = hipMalloc((void **)&dbuf[b], stage_bytes);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
            }
            if (e != hipSuccess) set_err(TAXOR_E_HIP, std::string("index upload (re-layout): staging buffers: ") + hipGetErrorString(e));
            int b = 0;
            while (!failed.load()) {
                const size_t ci = cursor.fetch_add(1);
                if (ci >= chunks.size()) break;
                const Chunk &c = chunks[ci];
                const Ixf &x = X[c.ixf];
                if (busy[b]) {                                   // the kernel that read this buffer pair last must be done
                    if ((e = hipEventSynchronize(ev[b])) != hipSuccess) { set_err(TAXOR_E_HIP, std::string("index upload (re-layout): ") + hipGetErrorString(e)); break; }
                    busy[b] = false;
                }
                uint64_t bytes;
                bool ok = true;
                if (kind == IXF_KIND_BIN_MAJOR) {
                    const uint64_t cp = (c.nr + 15) & ~(uint64_t)15;
                    for (uint64_t k = 0; k < c.nb && ok; ++k) ok = fetch(c.ixf, (c.b0 + k) * x.rows + c.r0, c.nr, (uint8_t *)hbuf[b] + k * cp);
                    bytes = c.nb * cp;
                } else {
                    bytes = c.nr * x.pitch;
                    ok = fetch(c.ixf, c.r0 * x.pitch, bytes, hbuf[b]);
                }
                if (!ok) { set_err(TAXOR_E_IO, "index upload: reading IXF " + std::to_string(c.ixf) + " from its source failed"); break; }
                e = hipMemcpyAsync(dbuf[b], hbuf[b], bytes, hipMemcpyHostToDevice, st);
                if (e == hipSuccess) {
                    if (kind == IXF_KIND_BIN_MAJOR)
                        hipLaunchKernelGGL(k_bin_major, dim3((uint32_t)((c.nr + 127) / 128), (uint32_t)((std::min<uint64_t>(x.stride - c.b0, 128) + 127) / 128)), dim3(256), 0, st,
                                           dbuf[b], x.dst, code, c.r0, c.nr, (c.nr + 15) & ~(uint64_t)15, c.b0, c.nb, x.stride, x.seg_len);
                    else if (kind == IXF_KIND_BIT_SLICED)
                        hipLaunchKernelGGL(k_bit_sliced, dim3((uint32_t)((c.nr * x.groups * 8 + 255) / 256)), dim3(256), 0, st, dbuf[b], x.dst, code, c.r0, c.nr, x.groups, x.stride,
                                           x.bins, x.seg_len);
                    else
                        hipLaunchKernelGGL(k_rows_repitch, dim3((uint32_t)((c.nr * (x.stride / 4) + 255) / 256)), dim3(256), 0, st, dbuf[b], x.dst, code, c.r0, c.nr, x.pitch, x.stride,
                                           x.bins, x.seg_len);
                    e = hipGetLastError();
                }
                if (e == hipSuccess) e = hipEventRecord(ev[b], st);
                if (e != hipSuccess) { set_err(TAXOR_E_HIP, std::string("index upload (re-layout): ") + hipGetErrorString(e)); break; }
                busy[b] = true;
                b ^= 1;
            }
            if (st) {
                const hipError_t es = hipStreamSynchronize(st);
                if (es != hipSuccess && !failed.load()) set_err(TAXOR_E_HIP, std::string("index upload (re-layout): ") + hipGetErrorString(es));
            }
            for (int k = 0; k < 2; ++k) {
                if (ev[k]) (void)hipEventDestroy(ev[k]);
                if (hbuf[k]) (void)hipHostFree(hbuf[k]);
                if (dbuf[k]) (void)hipFree(dbuf[k]);
            }
            if (st) (void)hipStreamDestroy(st);
        });
    for (auto &t : th) t.join();
    if (failed.load()) return rfail(err_code, err);
    if (tune_env("TAXOR_TRACE_UPLOAD")) {
        uint64_t b = 0;
        for (uint64_t i = 0; i < v->n_ixf; ++i)
            if (v->source || v->ixf[i].data) b += ixf_src_bytes(code, X[i].rows, X[i].pitch, X[i].bins);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        fprintf(stderr, "[upload] %.2f GB in %.3f s = %.1f GB/s (%s, layout code %u transposed on the device, %zu chunks)\n", b / 1e9, dt, b / 1e9 / dt,
                v->source ? "source reader" : "host pointers", code, chunks.size());
    }
    return 0;
}
