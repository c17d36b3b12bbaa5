// ixf_layout.h -- how a SOURCE (a .hixf this library did not write) may store the fingerprint bytes of one IXF, for host and
// device.  The search kernels read ONE layout: row-interleaved, data[row * stride + bin] with stride a multiple of 64 and rows
// in segment order (row = segment * seg_len + position).  seqan3::interleaved_xor_filter's serialiser is un-vendored
// (hierarchical_interleaved_xor_filter.hpp:152-158 hands ixf_vector to it; src/main/index.hpp:208-244 pins only the envelope), so
// what a published file holds can only be learnt from the file: `taxor verify --variants` / `taxor pin` probe the raw bytes
// under every layout below, and index creation transposes whatever they find into the search layout on the device while
// uploading (relayout.hip).  A layout CODE (taxor_hixf_view::ixf_layout, taxor_ixf_schema::layout):
//   bits 0-7   kind   0 row-interleaved     byte of (row, bin) at row * pitch + bin
//                     1 bin-major           byte of (row, bin) at bin * rows + row       (each bin's XOR filter on its own)
//                     2 bit-sliced words    per row and group of 64 bins eight little-endian u64 words; bit j of word p = bit p of
//                                           the fingerprint of bin 64 g + j -- the shape of an interleaved Bloom filter's rows, one
//                                           bit plane per fingerprint bit; byte address (row * groups + g) * 64 + p * 8
//   bit  8     rows   0 segment-major (row = segment * seg_len + position), 1 position-major (row = position * 3 + segment)
//   bits 9-10  pitch  row pitch of kind 0 / number of bin columns stored of kind 1:
//                     0 bins padded to a multiple of 64, 1 exactly `bins` (unpadded), 2 the record's stored scalar
// Code 0 is the search layout itself and is uploaded as it is.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TAXOR_LHD __host__ __device__ inline __attribute__((always_inline))
#else
#define TAXOR_LHD inline
#endif

namespace taxor {

enum : uint32_t {
    IXF_KIND_ROWS = 0u, IXF_KIND_BIN_MAJOR = 1u, IXF_KIND_BIT_SLICED = 2u, IXF_KIND_MASK = 0xFFu,
    IXF_ROWS_POSITION_MAJOR = 0x100u,
    IXF_PITCH_PADDED = 0u, IXF_PITCH_BINS = 0x200u, IXF_PITCH_STORED = 0x400u, IXF_PITCH_MASK = 0x600u
};

TAXOR_LHD uint32_t ixf_layout_kind(uint32_t code) { return code & IXF_KIND_MASK; }
TAXOR_LHD bool ixf_layout_valid(uint32_t code)
{
    return (code & ~(IXF_KIND_MASK | IXF_ROWS_POSITION_MAJOR | IXF_PITCH_MASK)) == 0u && ixf_layout_kind(code) <= 2u &&
           (code & IXF_PITCH_MASK) != IXF_PITCH_MASK && !(ixf_layout_kind(code) == IXF_KIND_BIT_SLICED && (code & IXF_PITCH_MASK) != 0u);
}

// row pitch (kind 0) / columns stored (kind 1) of an IXF's SOURCE bytes: the view's explicit src_stride, else what the code's pitch
// rule says -- `bins` (unpadded), the index's own stride (padded; the meaning of src_stride = 0 in a plain view), or 0 when the
// rule names the record's stored scalar and none was handed over (the caller must refuse that)
TAXOR_LHD uint64_t ixf_src_pitch(uint32_t code, uint64_t src_stride, uint64_t stride, uint64_t bins)
{
    if (src_stride) return src_stride;
    if ((code & IXF_PITCH_MASK) == IXF_PITCH_BINS) return bins;
    if ((code & IXF_PITCH_MASK) == IXF_PITCH_STORED) return 0;
    return stride;
}

// the source's row index of search-layout row r (segment * seg_len + position)
TAXOR_LHD uint64_t ixf_src_row(uint32_t code, uint64_t r, uint64_t seg_len)
{
    if (!(code & IXF_ROWS_POSITION_MAJOR)) return r;
    const uint64_t seg = r / seg_len, pos = r - seg * seg_len;
    return pos * 3u + seg;
}

// bytes the source holds for one IXF: rows = 3 * seg_len, pitch = row pitch (kind 0) / columns stored (kind 1), bins for kind 2
TAXOR_LHD uint64_t ixf_src_bytes(uint32_t code, uint64_t rows, uint64_t pitch, uint64_t bins)
{
    switch (ixf_layout_kind(code)) {
    case IXF_KIND_BIN_MAJOR: return pitch * rows;
    case IXF_KIND_BIT_SLICED: return rows * ((bins + 63u) / 64u) * 64u;
    default: return rows * pitch;
    }
}

// the fingerprint of (search-layout row r, bin b) read from source bytes -- the definition every transposing kernel is tested
// against, and what the variant scan and the fixture writer use directly
TAXOR_LHD uint8_t ixf_src_fingerprint(const uint8_t *src, uint32_t code, uint64_t r, uint64_t b, uint64_t seg_len, uint64_t pitch, uint64_t bins)
{
    const uint64_t rs = ixf_src_row(code, r, seg_len);
    switch (ixf_layout_kind(code)) {
    case IXF_KIND_BIN_MAJOR: return src[b * (3u * seg_len) + rs];
    case IXF_KIND_BIT_SLICED: {
        const uint64_t groups = (bins + 63u) / 64u, base = (rs * groups + b / 64u) * 64u, j = b & 63u;
        uint8_t v = 0;
        for (uint32_t p = 0; p < 8u; ++p) v = (uint8_t)(v | (((src[base + p * 8u + (j >> 3)] >> (j & 7u)) & 1u) << p));
        return v;
    }
    default: return src[rs * pitch + b];
    }
}

} // namespace taxor
