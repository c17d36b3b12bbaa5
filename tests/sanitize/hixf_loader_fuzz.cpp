// AddressSanitizer / UBSan driver for the .hixf loader, the probe and the TSV formatter (host code only, no GPU):
// usage: hixf_loader_fuzz <valid.hixf> <data_lo> <data_hi> <trials> <scratch file>
// mutates the file (bit flips, huge integers, truncation, insertions, deletions outside [data_lo, data_hi)) and loads it.
#include <taxor_gpu_tools.h>
#include <ixf_layout.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <string>
extern "C" void taxor_set_last_error(const char *m) { (void)m; }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> raw(n); if (fread(raw.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
    long data_lo = atol(argv[2]), data_hi = atol(argv[3]);
    int trials = atoi(argv[4]);
    std::mt19937_64 rng(7);
    int ok = 0, err = 0;
    {   // the writer's re-layout (threaded host transform) and the vector framings, round trip on the unmutated file: every fingerprint of
        // every IXF read back through ixf_src_fingerprint must be the original's
        taxor_hixf *h0 = nullptr;
        if (taxor_hixf_load(argv[1], &h0) != TAXOR_OK) return 7;
        const taxor_hixf_view *v0 = taxor_hixf_get_view(h0);
        struct Case { uint32_t layout, unit, before, after; };
        for (const Case &cs : {Case{0x201, 1, 0, 0}, Case{0x002, 64, 0, 1}, Case{0x300, 1, 1, 0}, Case{0x102, 8, 0, 0}, Case{0x001, 1, 0, 0}}) {
            taxor_ixf_schema sc;
            taxor_ixf_schema_default(&sc);
            sc.layout = cs.layout; sc.len_unit = cs.unit; sc.skip_before_len = cs.before; sc.skip_after_len = cs.after;
            taxor_hixf_view plain = *v0;
            plain.source = nullptr;                       // bytes through the mapping
            if (taxor_hixf_store_schema(argv[5], &plain, taxor_hixf_get_meta(h0), &sc) != TAXOR_OK) return 8;
            taxor_hixf *h1 = nullptr;
            if (taxor_hixf_load_schema(argv[5], &sc, &h1) != TAXOR_OK) return 9;
            const taxor_hixf_view *v1 = taxor_hixf_get_view(h1);
            if (v1->n_ixf != v0->n_ixf || (v1->ixf_layout & 0x1FF) != (cs.layout & 0x1FF)) return 10;
            for (size_t i = 0; i < v0->n_ixf; ++i) {
                const taxor_ixf_view &a = v0->ixf[i], &b = v1->ixf[i];
                if (a.bins != b.bins || a.seg_len != b.seg_len || a.seed != b.seed) return 11;
                for (uint64_t r = 0; r < 3 * a.seg_len; r += 1 + r % 7)
                    for (uint64_t bin = 0; bin < a.bins; ++bin)
                        if (a.data[r * a.stride + bin] != taxor::ixf_src_fingerprint(b.data, v1->ixf_layout, r, bin, b.seg_len, b.src_stride ? b.src_stride : b.stride, b.bins)) return 12;
            }
            taxor_hixf_free(h1);
        }
        taxor_hixf_free(h0);
    }
    std::vector<long> meta; for (long i = 0; i < n; ++i) if (i < data_lo || i >= data_hi) meta.push_back(i);
    for (int t = 0; t < trials; ++t) {
        std::vector<unsigned char> b = raw;
        switch (t % 5) {
        case 0: for (int j = 0; j < 1 + (int)(rng() % 3); ++j) b[meta[rng() % meta.size()]] ^= (unsigned char)(1 + rng() % 255); break;
        case 1: { long p = meta[rng() % meta.size()] & ~7L; unsigned long long v[] = {0, 1, 1ull << 63, ~0ull, 1ull << 40, 1ull << 31, 65};
                  unsigned long long x = v[rng() % 7]; if (p + 8 <= (long)b.size()) memcpy(&b[p], &x, 8); break; }
        case 2: b.resize(1 + rng() % b.size()); break;
        case 3: { long p = rng() % b.size(); std::vector<unsigned char> g(1 + rng() % 64); for (auto &c : g) c = (unsigned char)rng(); b.insert(b.begin() + p, g.begin(), g.end()); break; }
        case 4: { long p = meta[rng() % meta.size()]; long len = 1 + rng() % 32; if (p + len < (long)b.size()) b.erase(b.begin() + p, b.begin() + p + len); break; }
        }
        FILE *o = fopen(argv[5], "wb"); fwrite(b.data(), 1, b.size(), o); fclose(o);
        taxor_hixf *h = nullptr;
        int rc = taxor_hixf_load(argv[5], &h);
        if (rc == TAXOR_OK) {
            const taxor_hixf_view *v = taxor_hixf_get_view(h);
            unsigned long long sum = 0;
            for (size_t i = 0; i < v->n_ixf; ++i) {   // touch every byte the view claims
                const taxor_ixf_view &x = v->ixf[i];
                size_t sz = taxor_hixf_ixf_raw_bytes(h, i);
                if (sz != taxor::ixf_src_bytes(v->ixf_layout, 3 * x.seg_len, x.src_stride ? x.src_stride : x.stride, x.bins)) return 5;
                sum += x.data[0] + x.data[sz - 1];
                for (size_t j = 0; j < x.bins; ++j) sum += x.next_ixf[j] + x.fname_idx[j];
            }
            const taxor_hixf_meta *m = taxor_hixf_get_meta(h);
            for (size_t i = 0; i < m->n_species; ++i) sum += strlen(m->species[i].organism_name) + strlen(m->species[i].taxid_string);
            // format a line with arbitrary user bins
            int64_t ub[2] = {0, (int64_t)v->n_user_bins - 1}; unsigned ct[2] = {5, 4}; char buf[8192];
            taxor_format_read(h, "r", 1, 100, 10, ub, ct, 2, buf, sizeof buf);
            {   // the chunk formatter with user bins the index does not have (species lookup falls back), and a buffer that is too small
                const char *ids[3] = {"read_a", "b", ""};
                uint64_t idl[3] = {6, 1, 0}, rl[3] = {5000, 30, 0}, ro[4] = {0, 2, 2, 4};
                uint32_t nh[3] = {400, 0, 7}, c4[4] = {9, 8, 0, 3};
                int64_t u4[4] = {0, (int64_t)v->n_user_bins + 5, -1, (int64_t)v->n_user_bins - 1};
                const uint64_t need = taxor_format_reads(h, 3, ids, idl, rl, nh, ro, u4, c4, nullptr, 0);
                std::vector<char> big(need + 1);
                if (taxor_format_reads(h, 3, ids, idl, rl, nh, ro, u4, c4, big.data(), need) != need) return 3;
                if (need > 4 && taxor_format_reads(h, 3, ids, idl, rl, nh, ro, u4, c4, big.data(), need - 1) != need) return 3;
            }
            if (v->source) {   // the pread() reader the index upload uses: first and last bytes of every IXF, and a request past its end
                for (size_t i = 0; i < v->n_ixf; ++i) {
                    const taxor_ixf_view &x = v->ixf[i];
                    const size_t sz = taxor_hixf_ixf_raw_bytes(h, i);
                    unsigned char two[2] = {0, 0};
                    if (v->source->read(v->source->ctx, i, 0, 1, two) != 0 || v->source->read(v->source->ctx, i, sz - 1, 1, two + 1) != 0) return 4;
                    if (two[0] != x.data[0] || two[1] != x.data[sz - 1]) return 4;
                }
                unsigned char one;
                (void)v->source->read(v->source->ctx, v->n_ixf, 0, 1, &one);
            }
            // every layout the scan can name: either refused, or every fingerprint it addresses lies inside the raw array
            for (uint32_t code : {0x000u, 0x200u, 0x400u, 0x001u, 0x201u, 0x401u, 0x101u, 0x002u, 0x102u, 0x300u, 0x600u, 0x003u, 0x202u}) {
                if (taxor_hixf_set_layout(h, code) != TAXOR_OK) continue;
                const taxor_hixf_view *w = taxor_hixf_get_view(h);
                for (size_t i = 0; i < w->n_ixf; ++i) {
                    const taxor_ixf_view &x = w->ixf[i];
                    const uint64_t pitch = x.src_stride ? x.src_stride : x.stride;
                    if (taxor::ixf_src_bytes(code, 3 * x.seg_len, pitch, x.bins) != taxor_hixf_ixf_raw_bytes(h, i)) return 6;
                    sum += taxor::ixf_src_fingerprint(x.data, code, 0, 0, x.seg_len, pitch, x.bins) +
                           taxor::ixf_src_fingerprint(x.data, code, 3 * x.seg_len - 1, x.bins - 1, x.seg_len, pitch, x.bins);
                }
            }
            taxor_hixf_release_data(h);
            taxor_ixf_schema sc; char rep[4096]; taxor_hixf_probe(argv[5], &sc, rep, sizeof rep);
            if (sum == 42) puts("");
            taxor_hixf_free(h); ++ok;
        } else { taxor_ixf_schema sc; char rep[4096]; taxor_hixf_probe(argv[5], &sc, rep, sizeof rep); ++err; }
    }
    printf("ok=%d err=%d\n", ok, err);
}
