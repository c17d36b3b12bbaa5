// verify.hip -- diagnosis aid for indexes this library did not write.
//
// Both the arithmetic of seqan3::interleaved_xor_filter<uint8_t> and the layout its serialiser gives the fingerprint vector are
// un-vendored in the reference; the product's reading of the former lives in ixf_arith.h, the layouts a
// file may follow in ixf_layout.h.  This scan tells WHICH reading a file follows: the RAW fingerprint bytes of one IXF, as the
// file holds them, are probed under a family of variants (how the seed enters, which mixer, rotation step, range reduction,
// fingerprint fold, segment length; row-interleaved at one of several pitches, bin-major, bit-sliced words, rows in segment- or
// position-major order), each scored by the best-bin match ratio of hash lists taken from sequences that ARE in the index.  The
// right variant scores ~1.0, every other one sits at the 2^-8 false-positive floor.  Not on the search path; a plain kernel
// (one block per (hash list, variant), a thread per bin).
#include "../../include/taxor_gpu_tools.h"
#include "ixf_arith.h"
#include "ixf_layout.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);

namespace {

using namespace taxor;

// the scan decodes a variant through the same functions the search kernels use for a non-default arithmetic code
// (ixf_arith.h): what scores ~1.0 here is exactly what `taxor search --ixf-arithmetic` will compute
__host__ __device__ inline uint32_t variant_code(const taxor_ixf_variant &v)
{
    return ixf_arith_pack(v.key_hash, v.seed_mode, v.rot, v.reduce, v.fp_mode);
}

__global__ __launch_bounds__(256) void k_variant_scan(const uint8_t *__restrict__ data, uint64_t data_len, uint32_t bins,
                                                      const taxor_ixf_variant *__restrict__ variants,
                                                      const uint64_t *__restrict__ hashes, const uint64_t *__restrict__ hoff,
                                                      float *__restrict__ best_ratio, uint32_t n_lists)
{
    __shared__ uint32_t sBest;
    const uint32_t list = blockIdx.x, vi = blockIdx.y;
    const taxor_ixf_variant v = variants[vi];
    const uint64_t h0 = hoff[list], n = hoff[list + 1] - h0;
    if (threadIdx.x == 0) sBest = 0;
    __syncthreads();
    const uint64_t rows = 3 * v.seg_len;
    const uint32_t lay = v.layout;
    // every byte a probe may touch lies inside the raw array?  (a variant whose shape does not fit the length scores 0)
    const bool fits = ixf_src_bytes(lay, rows, v.stride, bins) <= data_len && (ixf_layout_kind(lay) == IXF_KIND_BIT_SLICED || v.stride >= bins);
    uint32_t best = 0;
    for (uint32_t b = threadIdx.x; fits && b < bins; b += 256) {
        uint32_t cnt = 0;
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t code = variant_code(v);
            const uint64_t h = ixf_key_hash_arith(hashes[h0 + i], v.seed, code);
            uint8_t x = ixf_fingerprint_arith(h, code);
#pragma unroll
            for (int j = 0; j < 3; ++j) x ^= ixf_src_fingerprint(data, lay, ixf_row_arith(h, j, v.seg_len, code), b, v.seg_len, v.stride, bins);
            cnt += x == 0 ? 1u : 0u;
        }
        best = max(best, cnt);
    }
    atomicMax(&sBest, best);
    __syncthreads();
    if (threadIdx.x == 0) best_ratio[(size_t)vi * n_lists + list] = n ? (float)sBest / (float)n : 0.f;
}

int vfail(int code, const std::string &msg)
{
    taxor_set_last_error(msg.c_str());
    return code;
}

} // namespace

extern "C" int taxor_gpu_ixf_variant_scan(int device, const uint8_t *raw, uint64_t raw_len, uint64_t bins, const taxor_ixf_variant *variants, uint32_t n_variants,
                                          const uint64_t *hashes, const uint64_t *hash_off, uint64_t n_lists, float *best_ratio)
{
    if (!raw || !raw_len || !bins || !variants || !n_variants || !hashes || !hash_off || !n_lists || !best_ratio)
        return vfail(TAXOR_E_ARG, "ixf_variant_scan: null or empty argument");
    for (uint32_t i = 0; i < n_variants; ++i)
        if (variants[i].seg_len == 0 || variants[i].stride == 0 || !ixf_layout_valid(variants[i].layout))
            return vfail(TAXOR_E_ARG, "ixf_variant_scan: variant with zero segment length or stride, or an unknown layout");
    // the kernel's "does this shape fit the raw bytes" test multiplies rows by the pitch in 64 bits: bound both here so that it cannot wrap
    // (an index cannot hold more than 2^32 rows or rows wider than 2^20 bytes either, api.hip index_create)
    for (uint32_t i = 0; i < n_variants; ++i)
        if (variants[i].seg_len > (1ull << 31) || variants[i].stride > (1ull << 20))
            return vfail(TAXOR_E_ARG, "ixf_variant_scan: variant " + std::to_string(i) + " with a segment length above 2^31 or a pitch above 2^20");
    if (n_lists > 65535 || n_variants > 65535 || bins >= (1ull << 32)) return vfail(TAXOR_E_ARG, "ixf_variant_scan: more than 65535 lists or variants");
    const uint64_t nh = hash_off[n_lists];
    uint8_t *d_raw = nullptr;
    taxor_ixf_variant *d_v = nullptr;
    uint64_t *d_h = nullptr, *d_off = nullptr;
    float *d_out = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&d_raw, raw_len);
    if (e == hipSuccess) e = hipMalloc((void **)&d_v, n_variants * sizeof(taxor_ixf_variant));
    if (e == hipSuccess) e = hipMalloc((void **)&d_h, std::max<uint64_t>(nh, 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_off, (n_lists + 1) * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, (size_t)n_variants * n_lists * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_raw, raw, raw_len, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_v, variants, n_variants * sizeof(taxor_ixf_variant), hipMemcpyHostToDevice);
    if (e == hipSuccess && nh) e = hipMemcpy(d_h, hashes, nh * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off, hash_off, (n_lists + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_variant_scan, dim3((uint32_t)n_lists, n_variants), dim3(256), 0, nullptr, d_raw, raw_len, (uint32_t)bins, d_v,
                           d_h, d_off, d_out, (uint32_t)n_lists);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(best_ratio, d_out, (size_t)n_variants * n_lists * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_raw);
    (void)hipFree(d_v);
    (void)hipFree(d_h);
    (void)hipFree(d_off);
    (void)hipFree(d_out);
    if (e != hipSuccess) return vfail(TAXOR_E_HIP, std::string("ixf_variant_scan: ") + hipGetErrorString(e));
    return TAXOR_OK;
}

extern "C" uint32_t taxor_ixf_arith_code(const taxor_ixf_variant *v) { return v ? variant_code(*v) : 0u; }

extern "C" void taxor_ixf_arith_decode(uint32_t code, taxor_ixf_variant *out)
{
    if (!out) return;
    out->key_hash = (uint8_t)(code & 3u);
    out->seed_mode = (uint8_t)((code >> 2) & 3u);
    out->reduce = (uint8_t)((code >> 4) & 3u);
    out->fp_mode = (uint8_t)((code >> 6) & 3u);
    out->rot = (uint8_t)(((code >> 8) & 0xFFu) ^ 21u);
}

extern "C" void taxor_ixf_variant_default(taxor_ixf_variant *out, uint64_t seed, uint64_t seg_len, uint64_t stride)
{
    if (!out) return;
    *out = taxor_ixf_variant{};
    out->seed = seed;
    out->seg_len = seg_len;
    out->stride = stride;
    out->rot = 21;
}

extern "C" uint64_t taxor_ixf_variant_describe(const taxor_ixf_variant *v, char *buf, uint64_t cap)
{
    if (!v || !buf || !cap) return 0;
    static const char *hash_name[] = {"murmur64", "identity", "wyhash-mix", "splitmix64"};
    static const char *seed_name[] = {"h(key + seed)", "h(key ^ seed)", "h(key) + seed", "h(key), seed unused"};
    static const char *red_name[] = {"(u32)rot * seg >> 32", "(u32)rot % seg", "mulhi64(rot, seg)"};
    static const char *fp_name[] = {"(u8)(h ^ h>>32)", "(u8)h", "(u8)(h>>56)", "(u8)(h>>32)"};
    static const char *lay_name[] = {"data[row*pitch + bin]", "data[bin*rows + row]", "bit-sliced 64-bin words"};
    const uint32_t kind = ixf_layout_kind(v->layout);
    const int n = snprintf(buf, cap, "%s as %s, seed %llu, row_i = %s + i*seg with rot = rotl(h, %u*i), fingerprint %s, seg_len %llu, %s %llu, %s%s",
                           hash_name[v->key_hash & 3], seed_name[v->seed_mode & 3], (unsigned long long)v->seed, red_name[v->reduce % 3],
                           (unsigned)v->rot, fp_name[v->fp_mode & 3], (unsigned long long)v->seg_len, kind == IXF_KIND_BIN_MAJOR ? "columns" : "row pitch",
                           (unsigned long long)v->stride, lay_name[kind <= 2 ? kind : 0], (v->layout & IXF_ROWS_POSITION_MAJOR) ? ", rows position-major" : "");
    return n < 0 ? 0 : (uint64_t)n;
}
