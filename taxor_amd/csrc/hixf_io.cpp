// hixf_io.cpp -- .hixf reader / writer (cereal BinaryOutputArchive layout, little endian, no header) and the
// per-read output text of `taxor search`.
//
// Envelope (pinned by the reference):
//   taxor_index::serialize          src/main/index.hpp:208-244
//     u32 version(=1) | u64 window_size | shape | u8 kmer_size | u8 syncmer_size | u8 t_syncmer | u8 parts |
//     bool use_syncmer | u16 scaling | bool compressed | bin_path | species | hixf
//   shape = seqan3::shape = dynamic_bitset<58>: u64 size, u64 bits                      [RECALL seqan3]
//   bin_path: vector<vector<string>>; cereal containers = u64 count + elements, string = u64 len + bytes
//   Species::serialize               src/taxonomy/Species.hpp:40-50  (5 strings, u64 user_bin, u64 seq_len)
//   hixf::serialize                  hierarchical_interleaved_xor_filter.hpp:152-158
//     ixf_vector | next_ixf_id (vector<vector<i64>>) | user_bins
//   user_bins::serialize             :277-282   user_bin_filenames | ixf_bin_to_filename_position
//
// IXF record (UN-VENDORED: seqan3 fork, not in /root/reference).  Schema used here -- the ONE place to change
// when a real file shows the fork's member order:
//     u64 bins | u64 technical_bins (= row stride) | u64 seg_len | u64 bin_words (= technical_bins/64) |
//     u64 seed | u64 ftype (= 8 fingerprint bits) | vector<uint8_t> data (u64 len + len bytes)
// with len == 3 * seg_len * technical_bins.
#include "../../include/taxor_gpu.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_io_err;

struct Cursor {
    const uint8_t *p, *end;
    bool ok = true;
    template <typename T> T get()
    {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; }
        std::memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const uint8_t *bytes(uint64_t n)
    {
        if ((uint64_t)(end - p) < n) { ok = false; return nullptr; }
        const uint8_t *r = p;
        p += n;
        return r;
    }
    std::string str()
    {
        const uint64_t n = get<uint64_t>();
        const uint8_t *b = ok ? bytes(n) : nullptr;
        return b ? std::string((const char *)b, n) : std::string();
    }
};

struct Writer {
    FILE *f;
    bool ok = true;
    template <typename T> void put(T v) { ok = ok && fwrite(&v, sizeof(T), 1, f) == 1; }
    void bytes(const void *p, uint64_t n) { ok = ok && (n == 0 || fwrite(p, 1, n, f) == n); }
    void str(const char *s)
    {
        const uint64_t n = s ? strlen(s) : 0;
        put<uint64_t>(n);
        bytes(s, n);
    }
};

} // namespace

struct taxor_hixf {
    void *map = nullptr;
    size_t map_len = 0;
    std::vector<taxor_ixf_view> ixf;
    std::vector<std::vector<int64_t>> next_ixf, fname_idx; // copies (the file's i64 arrays may be unaligned)
    std::vector<std::vector<uint8_t>> data_copy;           // only for IXFs whose payload is not 16-B aligned
    taxor_hixf_view view{};
    std::vector<std::string> strings;                      // backing store of species / filenames
    std::vector<taxor_species> species;
    std::vector<const char *> filenames;
    taxor_hixf_meta meta{};
    std::map<uint64_t, uint64_t> user_bin_index;           // user_bin -> first species index (taxor_search.cpp:172-178)
};

// taxor_gpu_last_error() lives in api.hip; IO errors are routed through a library-internal hook there
extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);

static int io_fail(int code, const std::string &msg)
{
    taxor_set_last_error(msg.c_str());
    return code;
}

extern "C" int taxor_hixf_load(const char *path, taxor_hixf **out)
{
    if (!path || !out) return io_fail(TAXOR_E_ARG, "hixf_load: null argument");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return io_fail(TAXOR_E_IO, std::string("cannot open index file ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 32) {
        close(fd);
        return io_fail(TAXOR_E_IO, std::string("index file too small: ") + path);
    }
    void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return io_fail(TAXOR_E_IO, std::string("mmap failed for ") + path);
    auto h = new taxor_hixf();
    h->map = m;
    h->map_len = (size_t)sb.st_size;
    Cursor c{(const uint8_t *)m, (const uint8_t *)m + sb.st_size};
    auto bail = [&](const std::string &why) {
        taxor_hixf_free(h);
        return io_fail(TAXOR_E_IO, std::string(path) + ": " + why);
    };

    const uint32_t version = c.get<uint32_t>();                          // index.hpp:211-212
    if (version != 1) return bail("unsupported index version " + std::to_string(version));
    h->meta.window_size = c.get<uint64_t>();                             // :217
    const uint64_t shape_size = c.get<uint64_t>();                       // :218 shape (dynamic_bitset)
    const uint64_t shape_bits = c.get<uint64_t>();
    (void)shape_bits;
    h->view.kmer_size = c.get<uint8_t>();                                // :219
    h->view.syncmer_size = c.get<uint8_t>();                             // :220
    h->view.t_syncmer = c.get<uint8_t>();                                // :221
    h->meta.parts = c.get<uint8_t>();                                    // :222
    h->view.use_syncmer = c.get<uint8_t>();                              // :223
    h->view.scaling = c.get<uint16_t>();                                 // :224
    h->meta.compressed = c.get<uint8_t>();                               // :225
    if (!c.ok || shape_size > 58 || shape_size != h->view.kmer_size)
        return bail("header inconsistent (shape size " + std::to_string(shape_size) + " vs k " +
                    std::to_string(h->view.kmer_size) + ")");
    const uint64_t n_paths = c.get<uint64_t>();                          // :226 bin_path
    if (!c.ok || n_paths > (uint64_t)sb.st_size) return bail("bin_path count implausible");
    for (uint64_t i = 0; i < n_paths && c.ok; ++i) {
        const uint64_t m2 = c.get<uint64_t>();
        if (m2 > (uint64_t)sb.st_size) return bail("bin_path entry implausible");
        for (uint64_t j = 0; j < m2 && c.ok; ++j) (void)c.str();
    }
    const uint64_t n_species = c.get<uint64_t>();                        // :227
    if (!c.ok || n_species > (uint64_t)sb.st_size) return bail("species count implausible");
    std::vector<uint64_t> sp_ub(n_species), sp_len(n_species);
    h->strings.reserve(5 * n_species + 16);
    for (uint64_t i = 0; i < n_species && c.ok; ++i) {                   // Species.hpp:43-49
        for (int j = 0; j < 5; ++j) h->strings.push_back(c.str());
        sp_ub[i] = c.get<uint64_t>();
        sp_len[i] = c.get<uint64_t>();
    }
    if (!c.ok) return bail("truncated in species");
    const uint64_t n_ixf = c.get<uint64_t>();                            // hixf.hpp:155 ixf_vector
    if (!c.ok || n_ixf == 0 || n_ixf > (uint64_t)sb.st_size / 56) return bail("IXF count implausible");
    h->ixf.resize(n_ixf);
    h->data_copy.resize(n_ixf);
    for (uint64_t i = 0; i < n_ixf; ++i) {                               // IXF record, schema in the header comment
        taxor_ixf_view &f = h->ixf[i];
        f.bins = c.get<uint64_t>();
        f.stride = c.get<uint64_t>();
        f.seg_len = c.get<uint64_t>();
        const uint64_t bin_words = c.get<uint64_t>();
        f.seed = c.get<uint64_t>();
        const uint64_t ftype = c.get<uint64_t>();
        const uint64_t len = c.get<uint64_t>();
        if (!c.ok) return bail("truncated in IXF " + std::to_string(i));
        if (ftype != 8 || bin_words * 64 != f.stride || f.stride < f.bins || f.seg_len == 0 ||
            f.seg_len > (1ull << 31) || len / 3 / f.seg_len != f.stride || len != 3 * f.seg_len * f.stride)
            return bail("IXF " + std::to_string(i) + " record inconsistent (bins " + std::to_string(f.bins) + ", stride " +
                        std::to_string(f.stride) + ", seg_len " + std::to_string(f.seg_len) + ", data " + std::to_string(len) +
                        " bytes) -- the IXF schema of this library may differ from the file's (see hixf_io.cpp)");
        f.data = c.bytes(len);
        if (!c.ok) return bail("truncated in IXF " + std::to_string(i) + " data");
    }
    auto read_vv = [&](std::vector<std::vector<int64_t>> &vv, const char *what) -> bool {
        const uint64_t n = c.get<uint64_t>();
        if (!c.ok || n != n_ixf) { g_io_err = std::string(what) + " outer size != IXF count"; return false; }
        vv.resize(n);
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t m2 = c.get<uint64_t>();
            if (!c.ok || m2 != h->ixf[i].bins) { g_io_err = std::string(what) + " inner size != bins of IXF " + std::to_string(i); return false; }
            const uint8_t *b = c.bytes(m2 * 8);
            if (!b) { g_io_err = std::string("truncated in ") + what; return false; }
            vv[i].resize(m2);
            std::memcpy(vv[i].data(), b, m2 * 8);
        }
        return true;
    };
    if (!read_vv(h->next_ixf, "next_ixf_id")) return bail(g_io_err);           // hixf.hpp:156
    const uint64_t n_files = c.get<uint64_t>();                                 // :280 user_bin_filenames
    if (!c.ok || n_files > (uint64_t)sb.st_size) return bail("user_bin_filenames count implausible");
    const size_t first_fn = h->strings.size();
    for (uint64_t i = 0; i < n_files && c.ok; ++i) h->strings.push_back(c.str());
    if (!c.ok) return bail("truncated in user_bin_filenames");
    if (!read_vv(h->fname_idx, "ixf_bin_to_filename_position")) return bail(g_io_err); // :281
    if (c.p != c.end) return bail(std::to_string((size_t)(c.end - c.p)) + " trailing bytes after the index");

    for (uint64_t i = 0; i < n_ixf; ++i) {
        h->ixf[i].next_ixf = h->next_ixf[i].data();
        h->ixf[i].fname_idx = h->fname_idx[i].data();
    }
    h->species.resize(n_species);
    for (uint64_t i = 0; i < n_species; ++i) {
        taxor_species &s = h->species[i];
        s.organism_name = h->strings[5 * i + 0].c_str();
        s.accession_id = h->strings[5 * i + 1].c_str();
        s.taxid = h->strings[5 * i + 2].c_str();
        s.taxnames_string = h->strings[5 * i + 3].c_str();
        s.taxid_string = h->strings[5 * i + 4].c_str();
        s.user_bin = sp_ub[i];
        s.seq_len = sp_len[i];
        h->user_bin_index.emplace(s.user_bin, i); // emplace keeps the first (taxor_search.cpp:174)
    }
    h->filenames.resize(n_files);
    for (uint64_t i = 0; i < n_files; ++i) h->filenames[i] = h->strings[first_fn + i].c_str();
    h->meta.n_species = n_species;
    h->meta.species = h->species.data();
    h->meta.n_user_bin_filenames = n_files;
    h->meta.user_bin_filenames = h->filenames.data();
    h->view.n_ixf = n_ixf;
    h->view.ixf = h->ixf.data();
    h->view.n_user_bins = n_files;
    *out = h;
    return TAXOR_OK;
}

extern "C" void taxor_hixf_free(taxor_hixf *h)
{
    if (!h) return;
    if (h->map) munmap(h->map, h->map_len);
    delete h;
}

extern "C" const taxor_hixf_view *taxor_hixf_get_view(const taxor_hixf *h) { return h ? &h->view : nullptr; }
extern "C" const taxor_hixf_meta *taxor_hixf_get_meta(const taxor_hixf *h) { return h ? &h->meta : nullptr; }

extern "C" int taxor_hixf_store(const char *path, const taxor_hixf_view *v, const taxor_hixf_meta *m)
{
    if (!path || !v || !m) return io_fail(TAXOR_E_ARG, "hixf_store: null argument");
    FILE *f = fopen(path, "wb");
    if (!f) return io_fail(TAXOR_E_IO, std::string("cannot create ") + path);
    Writer w{f};
    w.put<uint32_t>(1);
    w.put<uint64_t>(m->window_size);
    w.put<uint64_t>(v->kmer_size);                                   // shape: ungapped{k}
    w.put<uint64_t>(v->kmer_size >= 64 ? ~0ull : ((1ull << v->kmer_size) - 1ull));
    w.put<uint8_t>(v->kmer_size);
    w.put<uint8_t>(v->syncmer_size);
    w.put<uint8_t>(v->t_syncmer);
    w.put<uint8_t>(m->parts);
    w.put<uint8_t>(v->use_syncmer ? 1 : 0);
    w.put<uint16_t>(v->scaling);
    w.put<uint8_t>(m->compressed ? 1 : 0);
    w.put<uint64_t>(m->n_user_bin_filenames);                        // bin_path: one single-element vector per file
    for (uint64_t i = 0; i < m->n_user_bin_filenames; ++i) {
        w.put<uint64_t>(1);
        w.str(m->user_bin_filenames[i]);
    }
    w.put<uint64_t>(m->n_species);
    for (uint64_t i = 0; i < m->n_species; ++i) {
        const taxor_species &s = m->species[i];
        w.str(s.organism_name);
        w.str(s.accession_id);
        w.str(s.taxid);
        w.str(s.taxnames_string);
        w.str(s.taxid_string);
        w.put<uint64_t>(s.user_bin);
        w.put<uint64_t>(s.seq_len);
    }
    w.put<uint64_t>(v->n_ixf);
    for (uint64_t i = 0; i < v->n_ixf; ++i) {
        const taxor_ixf_view &x = v->ixf[i];
        if (!x.data) {
            fclose(f);
            return io_fail(TAXOR_E_ARG, "hixf_store: IXF without host data");
        }
        w.put<uint64_t>(x.bins);
        w.put<uint64_t>(x.stride);
        w.put<uint64_t>(x.seg_len);
        w.put<uint64_t>(x.stride / 64);
        w.put<uint64_t>(x.seed);
        w.put<uint64_t>(8);
        const uint64_t len = 3 * x.seg_len * x.stride;
        w.put<uint64_t>(len);
        w.bytes(x.data, len);
    }
    w.put<uint64_t>(v->n_ixf);
    for (uint64_t i = 0; i < v->n_ixf; ++i) {
        w.put<uint64_t>(v->ixf[i].bins);
        w.bytes(v->ixf[i].next_ixf, v->ixf[i].bins * 8);
    }
    w.put<uint64_t>(m->n_user_bin_filenames);
    for (uint64_t i = 0; i < m->n_user_bin_filenames; ++i) w.str(m->user_bin_filenames[i]);
    w.put<uint64_t>(v->n_ixf);
    for (uint64_t i = 0; i < v->n_ixf; ++i) {
        w.put<uint64_t>(v->ixf[i].bins);
        w.bytes(v->ixf[i].fname_idx, v->ixf[i].bins * 8);
    }
    const bool ok = w.ok && fclose(f) == 0;
    if (!ok) return io_fail(TAXOR_E_IO, std::string("write failed: ") + path);
    return TAXOR_OK;
}

// taxor_search.cpp:268-305
extern "C" uint64_t taxor_format_read(const taxor_hixf *h, const char *id, uint64_t id_len, uint64_t read_len,
                                      uint32_t n_hashes, const int64_t *user_bin, const uint32_t *count, uint64_t n_tuples,
                                      char *buf, uint64_t cap)
{
    const taxor_hixf_meta *meta = &h->meta;
    std::string out;
    const std::string sid(id, id_len);
    if (n_tuples == 0) {                                                              // :268-273
        out += sid + '\t';
        out += "-\t-\t-\t-\t";
        out += std::to_string(read_len) + "\n";
    } else {
        uint64_t max_count = 0;                                                       // :275-280
        for (uint64_t i = 0; i < n_tuples; ++i)
            if (count[i] > max_count) max_count = count[i];
        for (uint64_t i = 0; i < n_tuples; ++i) {
            if (static_cast<double>(count[i]) < static_cast<double>(max_count) * 0.8) continue; // :285
            // user_bin_index[count.first]: std::map::operator[] default-inserts 0 for an unknown user bin (:289)
            if (meta->n_species == 0) continue;
            const auto it = h->user_bin_index.find((uint64_t)user_bin[i]);
            const uint64_t si = it == h->user_bin_index.end() ? 0 : it->second;
            const taxor_species &s = meta->species[si];
            out += sid + '\t';
            out += s.accession_id;
            out += '\t';
            out += s.organism_name;
            out += '\t';
            out += s.taxid;
            out += '\t';
            out += std::to_string(s.seq_len);
            out += '\t';
            out += std::to_string(read_len);
            out += '\t';
            out += std::to_string(n_hashes);
            out += '\t';
            out += std::to_string(count[i]);
            out += '\t';
            out += s.taxnames_string;
            out += '\t';
            out += s.taxid_string;
            out += '\n';
        }
    }
    if (out.size() <= cap && buf) std::memcpy(buf, out.data(), out.size());
    return out.size();
}
