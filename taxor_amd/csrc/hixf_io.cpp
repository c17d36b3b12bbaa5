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
// How the bytes of `data` are laid out is un-vendored as well: ixf_layout.h lists the layouts a file may follow; the schema
// carries the code (taxor_ixf_schema::layout), a loaded file starts at what its array lengths admit and `taxor pin` /
// `taxor verify --variants` decide (taxor_hixf_set_layout).  Whatever it is, index creation transposes it into the search
// layout on the device (relayout.hip); this file only does the bookkeeping and, for tests and export, the inverse on the host.
#include "../../include/taxor_gpu_tools.h"
#include "ixf_layout.h"
#include "tuning.h"
using taxor::tune_env;

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
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
    int fd = -1;                                           // the file, kept open for the pread() source below
    std::vector<uint64_t> file_off;                        // file offset of every IXF's fingerprint array
    std::vector<uint64_t> raw_len, stored_stride, stored_seg;  // its length; the record's stride / seg_len scalars (0 = not stored)
    taxor_ixf_source source{};                             // view.source: the bytes by pread(), never through the mapping
    std::vector<taxor_ixf_view> ixf;
    std::vector<std::vector<int64_t>> next_ixf, fname_idx; // copies (the file's i64 arrays may be unaligned)
    std::vector<std::vector<uint8_t>> data_copy;           // only for IXFs whose payload is not 16-B aligned
    taxor_hixf_view view{};
    std::vector<std::string> strings;                      // backing store of species / filenames
    std::vector<taxor_species> species;
    std::vector<const char *> filenames;
    taxor_hixf_meta meta{};
    bool pitch_overridden = false;                         // finish_ixfs: the records' stride scalar was contradicted by their lengths
    std::map<uint64_t, uint64_t> user_bin_index;           // user_bin -> first species index (taxor_search.cpp:172-178)
    // the per-species parts of a hit line (taxor_search.cpp:287-305), rendered once: "ACCESSION\tNAME\tTAXID\tREF_LEN\t" and
    // "TAX_STR\tTAX_ID_STR\n"; and user bin -> species index as a flat table for the user bins the index can report
    std::vector<std::string> line_head, line_tail;
    std::vector<uint32_t> ub_species;
};

// taxor_gpu_last_error() lives in api.hip; IO errors are routed through a library-internal hook there
extern "C" __attribute__((visibility("hidden"))) void taxor_set_last_error(const char *msg);

static int io_fail(int code, const std::string &msg)
{
    taxor_set_last_error(msg.c_str());
    return code;
}

namespace {

struct Mapped {
    void *map = nullptr;
    size_t len = 0;
};

int map_file(const char *path, Mapped &m, int *keep_fd = nullptr)
{
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return io_fail(TAXOR_E_IO, std::string("cannot open index file ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 32) {
        close(fd);
        return io_fail(TAXOR_E_IO, std::string("index file too small: ") + path);
    }
    void *p = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (keep_fd && p != MAP_FAILED) *keep_fd = fd;
    else close(fd);
    if (p == MAP_FAILED) return io_fail(TAXOR_E_IO, std::string("mmap failed for ") + path);
    m.map = p;
    m.len = (size_t)sb.st_size;
    return 0;
}

// envelope up to (and including) the IXF count; fills header fields, species and leaves the cursor at the first IXF
// record.  Returns an error string (empty = ok).
std::string parse_envelope(Cursor &c, taxor_hixf *h, std::vector<uint64_t> &sp_ub, std::vector<uint64_t> &sp_len,
                           uint64_t &n_ixf)
{
    const uint64_t fsz = (uint64_t)(c.end - c.p);
    const uint32_t version = c.get<uint32_t>();                          // index.hpp:211-212
    if (version != 1) return "unsupported index version " + std::to_string(version);
    h->meta.window_size = c.get<uint64_t>();                             // :217
    h->view.window_size = h->meta.window_size;
    const uint64_t shape_size = c.get<uint64_t>();                       // :218 shape (dynamic_bitset)
    (void)c.get<uint64_t>();
    h->view.kmer_size = c.get<uint8_t>();                                // :219
    h->view.syncmer_size = c.get<uint8_t>();                             // :220
    h->view.t_syncmer = c.get<uint8_t>();                                // :221
    h->meta.parts = c.get<uint8_t>();                                    // :222
    h->view.use_syncmer = c.get<uint8_t>();                              // :223
    h->view.scaling = c.get<uint16_t>();                                 // :224
    h->meta.compressed = c.get<uint8_t>();                               // :225
    if (!c.ok || shape_size > 58 || shape_size != h->view.kmer_size)
        return "header inconsistent (shape size " + std::to_string(shape_size) + " vs k " + std::to_string(h->view.kmer_size) + ")";
    const uint64_t n_paths = c.get<uint64_t>();                          // :226 bin_path
    if (!c.ok || n_paths > fsz) return "bin_path count implausible";
    for (uint64_t i = 0; i < n_paths && c.ok; ++i) {
        const uint64_t m2 = c.get<uint64_t>();
        if (m2 > fsz) return "bin_path entry implausible";
        for (uint64_t j = 0; j < m2 && c.ok; ++j) (void)c.str();
    }
    const uint64_t n_species = c.get<uint64_t>();                        // :227
    if (!c.ok || n_species > fsz) return "species count implausible";
    sp_ub.resize(n_species);
    sp_len.resize(n_species);
    h->strings.reserve(5 * n_species + 16);
    for (uint64_t i = 0; i < n_species && c.ok; ++i) {                   // Species.hpp:43-49
        for (int j = 0; j < 5; ++j) h->strings.push_back(c.str());
        sp_ub[i] = c.get<uint64_t>();
        sp_len[i] = c.get<uint64_t>();
    }
    if (!c.ok) return "truncated in species";
    n_ixf = c.get<uint64_t>();                                           // hixf.hpp:155 ixf_vector
    if (!c.ok || n_ixf == 0 || n_ixf > fsz / 16) return "IXF count implausible";
    return "";
}

// pinned tail after the IXF records: next_ixf_id | user_bin_filenames | ixf_bin_to_filename_position, to end of file
std::string parse_tail(Cursor &c, taxor_hixf *h, uint64_t n_ixf, size_t &first_fn, uint64_t &n_files, bool keep)
{
    const uint64_t fsz = (uint64_t)(c.end - c.p) + 16;
    auto read_vv = [&](std::vector<std::vector<int64_t>> &vv, const char *what, bool first) -> std::string {
        const uint64_t n = c.get<uint64_t>();
        if (!c.ok || n != n_ixf) return std::string(what) + " outer size != IXF count";
        vv.resize(n);
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t m2 = c.get<uint64_t>();
            if (!c.ok || m2 == 0 || m2 > fsz / 8) return std::string(what) + " inner size implausible";
            if (!first && m2 != h->next_ixf[i].size()) return std::string(what) + " inner size != next_ixf_id's for IXF " + std::to_string(i);
            const uint8_t *b = c.bytes(m2 * 8);
            if (!b) return std::string("truncated in ") + what;
            vv[i].resize(m2);
            if (keep || first) std::memcpy(vv[i].data(), b, m2 * 8);
        }
        return "";
    };
    std::string e = read_vv(h->next_ixf, "next_ixf_id", true);                  // hixf.hpp:156
    if (!e.empty()) return e;
    n_files = c.get<uint64_t>();                                                // :280 user_bin_filenames
    if (!c.ok || n_files > fsz) return "user_bin_filenames count implausible";
    first_fn = h->strings.size();
    for (uint64_t i = 0; i < n_files && c.ok; ++i) {
        std::string t = c.str();
        if (keep) h->strings.push_back(std::move(t));
    }
    if (!c.ok) return "truncated in user_bin_filenames";
    e = read_vv(h->fname_idx, "ixf_bin_to_filename_position", false);           // :281
    if (!e.empty()) return e;
    if (c.p != c.end) return std::to_string((size_t)(c.end - c.p)) + " trailing bytes after the index";
    return "";
}

inline uint64_t ceil64(uint64_t x) { return (x + 63) / 64 * 64; }

// IXF records under `sc`; bins are cross-checked against next_ixf_id's inner sizes by the caller
std::string parse_ixfs(Cursor &c, const taxor_ixf_schema &sc, taxor_hixf *h, uint64_t n_ixf,
                       std::vector<std::vector<uint64_t>> *scalars_out = nullptr, std::vector<uint64_t> *lens_out = nullptr)
{
    h->ixf.assign(n_ixf, taxor_ixf_view{});
    h->raw_len.assign(n_ixf, 0);
    h->stored_stride.assign(n_ixf, 0);
    h->stored_seg.assign(n_ixf, 0);
    const uint32_t ns = sc.n_before + sc.n_after;
    std::vector<uint64_t> sv(ns);
    for (uint64_t i = 0; i < n_ixf; ++i) {
        for (uint32_t j = 0; j < sc.n_before; ++j) sv[j] = c.get<uint64_t>();
        if (sc.skip_before_len) (void)c.bytes(sc.skip_before_len);
        const uint64_t count = c.get<uint64_t>();
        if (sc.skip_after_len) (void)c.bytes(sc.skip_after_len);
        if (!c.ok) return "truncated in IXF " + std::to_string(i);
        // bytes the vector occupies in the file, and fingerprint bytes of them (bits in whole words: up to 7 bytes of padding)
        if (count > (1ull << 60)) return "IXF " + std::to_string(i) + ": fingerprint vector length implausible";
        const uint64_t in_file = sc.len_unit == 8 ? count * 8 : sc.len_unit == 64 ? (count + 63) / 64 * 8 : count;
        const uint64_t len = sc.len_unit == 64 ? count / 8 : in_file;
        if (sc.len_unit == 64 && count % 8 != 0) return "IXF " + std::to_string(i) + ": a length of " + std::to_string(count) + " bits is not whole fingerprints";
        const uint8_t *data = c.bytes(in_file);
        if (!data) return "IXF " + std::to_string(i) + ": fingerprint vector of " + std::to_string(in_file) + " bytes runs past the end of the file";
        for (uint32_t j = 0; j < sc.n_after; ++j) sv[sc.n_before + j] = c.get<uint64_t>();
        if (!c.ok) return "truncated after IXF " + std::to_string(i);
        taxor_ixf_view &f = h->ixf[i];
        f.data = data;
        f.bins = sc.idx_bins >= 0 ? sv[sc.idx_bins] : 0;                 // 0: filled from next_ixf_id later
        f.seed = sc.idx_seed >= 0 ? sv[sc.idx_seed] : sc.default_seed;
        h->raw_len[i] = len;
        h->stored_stride[i] = sc.idx_stride >= 0 ? sv[sc.idx_stride] : 0;
        h->stored_seg[i] = sc.idx_seg_len >= 0 ? (sc.seg_len_is_rows ? sv[sc.idx_seg_len] / 3 : sv[sc.idx_seg_len]) : 0;
        if (sc.idx_seg_len >= 0 && h->stored_seg[i] == 0) return "IXF " + std::to_string(i) + ": stored segment length 0";
        if (scalars_out) scalars_out->push_back(sv);
        if (lens_out) lens_out->push_back(len);
    }
    return "";
}

// stride / seg_len / src_stride of IXF i under layout `code` from its array length, its bin count and the record's scalars; the
// error text names what does not fit
std::string apply_layout(taxor_hixf *h, size_t i, uint32_t code)
{
    taxor_ixf_view &f = h->ixf[i];
    const uint64_t len = h->raw_len[i], bins = f.bins, S = ceil64(bins), stored = h->stored_stride[i];
    const uint32_t kind = taxor::ixf_layout_kind(code), rule = code & taxor::IXF_PITCH_MASK;
    uint64_t pitch = S;
    if (kind != taxor::IXF_KIND_BIT_SLICED) {
        pitch = rule == taxor::IXF_PITCH_BINS ? bins : rule == taxor::IXF_PITCH_STORED ? stored : S;
        if (rule == taxor::IXF_PITCH_STORED && stored == 0) return "IXF " + std::to_string(i) + ": the layout takes its pitch from the record, which stores none";
    }
    if (pitch < bins || pitch == 0 || len % pitch != 0)
        return "IXF " + std::to_string(i) + ": " + (kind == taxor::IXF_KIND_BIN_MAJOR ? "column count " : "row pitch ") + std::to_string(pitch) +
               " inconsistent with " + std::to_string(bins) + " bins / " + std::to_string(len) + " fingerprint bytes";
    const uint64_t rows = len / pitch;
    const uint64_t seg = h->stored_seg[i] ? h->stored_seg[i] : rows / 3;
    if (seg == 0 || seg > (1ull << 31) || 3 * seg != rows)
        return "IXF " + std::to_string(i) + ": " + std::to_string(rows) + " rows are not 3 x segment length " + std::to_string(seg);
    f.seg_len = seg;
    // a row-interleaved source whose pitch is a legal stride is searched as it lies; everything else lands at ceil64(bins)
    const bool as_is = kind == taxor::IXF_KIND_ROWS && !(code & taxor::IXF_ROWS_POSITION_MAJOR) && pitch % 64 == 0;
    f.stride = as_is ? pitch : S;
    f.src_stride = pitch;
    return "";
}

std::string finish_ixfs(const taxor_ixf_schema &sc, taxor_hixf *h)
{
    if (!taxor::ixf_layout_valid(sc.layout)) return "schema names an unknown fingerprint layout code " + std::to_string(sc.layout);
    for (size_t i = 0; i < h->ixf.size(); ++i) {
        taxor_ixf_view &f = h->ixf[i];
        const uint64_t bins = h->next_ixf[i].size();
        if (sc.idx_bins >= 0 && f.bins != bins)
            return "IXF " + std::to_string(i) + ": stored bin count " + std::to_string(f.bins) + " != next_ixf_id's " + std::to_string(bins);
        f.bins = bins;
        f.next_ixf = h->next_ixf[i].data();
        f.fname_idx = h->fname_idx[i].data();
    }
    // the pitch rule: what the schema says; where it says nothing (0) the record's stride scalar if there is one, else bins padded
    // to 64, else -- if no IXF's length fits that -- exactly bins
    uint32_t code = sc.layout;
    if ((code & taxor::IXF_PITCH_MASK) == 0 && taxor::ixf_layout_kind(code) != taxor::IXF_KIND_BIT_SLICED) {
        std::vector<uint32_t> rules;
        if (sc.idx_stride >= 0) rules.push_back(taxor::IXF_PITCH_STORED);
        rules.push_back(taxor::IXF_PITCH_PADDED);
        rules.push_back(taxor::IXF_PITCH_BINS);
        std::string first_err;
        for (uint32_t r : rules) {
            std::string e;
            for (size_t i = 0; i < h->ixf.size() && e.empty(); ++i) e = apply_layout(h, i, code | r);
            if (e.empty()) {
                h->view.ixf_layout = code | r;
                // the record carries a stride scalar and the array lengths contradict it: the file loads under a pitch its own scalar
                // does not name.  Said out loud and remembered (meta.foreign_schema), not dropped -- a file of this library's own
                // schema that arrives here is damaged or was written by something else
                if (sc.idx_stride >= 0 && r != taxor::IXF_PITCH_STORED) {
                    h->pitch_overridden = true;
                    fprintf(stderr, "[TAXOR SEARCH WARNING] the stored row pitch of the IXF records contradicts their array lengths (%s); loaded with the pitch %s\n",
                            first_err.c_str(), r == taxor::IXF_PITCH_PADDED ? "bins padded to 64" : "exactly bins");
                }
                return "";
            }
            if (first_err.empty()) first_err = e;
        }
        return first_err;
    }
    for (size_t i = 0; i < h->ixf.size(); ++i) {
        const std::string e = apply_layout(h, i, code);
        if (!e.empty()) return e;
    }
    h->view.ixf_layout = code;
    return "";
}

void finish_meta(taxor_hixf *h, const std::vector<uint64_t> &sp_ub, const std::vector<uint64_t> &sp_len, size_t first_fn,
                 uint64_t n_files)
{
    const uint64_t n_species = sp_ub.size();
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
    h->line_head.resize(n_species);
    h->line_tail.resize(n_species);
    for (uint64_t i = 0; i < n_species; ++i) {
        const taxor_species &s = h->species[i];
        h->line_head[i] = std::string(s.accession_id) + '\t' + s.organism_name + '\t' + s.taxid + '\t' + std::to_string(s.seq_len) + '\t';
        h->line_tail[i] = std::string(s.taxnames_string) + '\t' + s.taxid_string + '\n';
    }
    h->ub_species.assign(n_files, 0u);                     // unknown user bin -> species[0], like std::map::operator[] (:289)
    for (uint64_t i = n_species; i-- > 0;)                 // descending, so that the first species of a user bin wins (:174)
        if (h->species[i].user_bin < n_files) h->ub_species[h->species[i].user_bin] = (uint32_t)i;
    h->filenames.resize(n_files);
    for (uint64_t i = 0; i < n_files; ++i) h->filenames[i] = h->strings[first_fn + i].c_str();
    h->meta.n_species = n_species;
    h->meta.species = h->species.data();
    h->meta.n_user_bin_filenames = n_files;
    h->meta.user_bin_filenames = h->filenames.data();
    h->view.n_ixf = h->ixf.size();
    h->view.ixf = h->ixf.data();
    h->view.n_user_bins = n_files;
}

// taxor_ixf_source::read of a loaded file: pread() at the array's file offset.  The mapping stays for whoever wants to
// look at the bytes from the host (its pages are only faulted in if someone does); index creation comes through here.
int hixf_pread(void *ctx, uint64_t ixf, uint64_t offset, uint64_t len, void *dst)
{
    const taxor_hixf *h = static_cast<const taxor_hixf *>(ctx);
    if (ixf >= h->file_off.size() || h->fd < 0) return -1;
    uint64_t done = 0;
    while (done < len) {
        const ssize_t n = pread(h->fd, static_cast<char *>(dst) + done, len - done, (off_t)(h->file_off[ixf] + offset + done));
        if (n <= 0) return -1;
        done += (uint64_t)n;
    }
    return 0;
}

// 0 ok; TAXOR_E_IO with *schema_problem=true when only the IXF records did not fit the schema
int load_with(const char *path, const taxor_ixf_schema &sc, taxor_hixf **out, bool *schema_problem)
{
    if (schema_problem) *schema_problem = false;
    Mapped m;
    int fd = -1;
    if (int rc = map_file(path, m, &fd)) return rc;
    auto h = new taxor_hixf();
    h->map = m.map;
    h->map_len = m.len;
    h->fd = fd;
    Cursor c{(const uint8_t *)m.map, (const uint8_t *)m.map + m.len};
    auto bail = [&](const std::string &why) {
        taxor_hixf_free(h);
        return io_fail(TAXOR_E_IO, std::string(path) + ": " + why);
    };
    std::vector<uint64_t> sp_ub, sp_len;
    uint64_t n_ixf = 0, n_files = 0;
    size_t first_fn = 0;
    std::string e = parse_envelope(c, h, sp_ub, sp_len, n_ixf);
    if (!e.empty()) return bail(e);
    e = parse_ixfs(c, sc, h, n_ixf);
    if (e.empty()) e = parse_tail(c, h, n_ixf, first_fn, n_files, true);
    if (e.empty()) e = finish_ixfs(sc, h);
    if (!e.empty()) {
        if (schema_problem) *schema_problem = true;
        return bail(e + " -- the IXF record layout of this file may differ from the schema used (taxor_hixf_probe)");
    }
    finish_meta(h, sp_ub, sp_len, first_fn, n_files);
    h->file_off.resize(h->ixf.size());
    for (size_t i = 0; i < h->ixf.size(); ++i) h->file_off[i] = (uint64_t)(h->ixf[i].data - (const uint8_t *)m.map);
    h->source.read = hixf_pread;
    h->source.ctx = h;
    static const bool use_map = [] { const char *e = tune_env("TAXOR_HIXF_UPLOAD_FROM_MAP"); return e && atoi(e) != 0; }();   // A/B knob: round-2 path
    h->view.source = use_map ? nullptr : &h->source;
    *out = h;
    return TAXOR_OK;
}

} // namespace

extern "C" void taxor_ixf_schema_default(taxor_ixf_schema *out)
{
    if (!out) return;
    *out = taxor_ixf_schema{6, 0, 0, 1, 2, 4, 0, 13572355802537770549ull, 0, 1, 0, 0};
}

extern "C" int taxor_hixf_load_schema(const char *path, const taxor_ixf_schema *schema, taxor_hixf **out)
{
    if (!path || !schema || !out) return io_fail(TAXOR_E_ARG, "hixf_load_schema: null argument");
    const int ns = (int)(schema->n_before + schema->n_after);
    if (schema->n_before > 32 || schema->n_after > 32 || schema->idx_bins >= ns || schema->idx_stride >= ns ||
        schema->idx_seg_len >= ns || schema->idx_seed >= ns || (schema->len_unit > 1 && schema->len_unit != 8 && schema->len_unit != 64) ||
        schema->skip_before_len > 8 || schema->skip_after_len > 8)
        return io_fail(TAXOR_E_ARG, "hixf_load_schema: schema indices out of range");
    const int rc = load_with(path, *schema, out, nullptr);
    if (rc == TAXOR_OK) {            // records framed otherwise than this library writes them: written by other software (or for a test of that)
        taxor_ixf_schema own;
        taxor_ixf_schema_default(&own);
        (*out)->meta.foreign_schema = schema->n_before != own.n_before || schema->n_after != own.n_after || schema->idx_bins != own.idx_bins ||
                                      schema->idx_stride != own.idx_stride || schema->idx_seg_len != own.idx_seg_len || schema->idx_seed != own.idx_seed ||
                                      (schema->len_unit > 1) || schema->skip_before_len || schema->skip_after_len || (*out)->pitch_overridden;
    }
    return rc;
}

extern "C" int taxor_hixf_probe(const char *path, taxor_ixf_schema *out, char *report, uint64_t cap)
{
    if (!path || !out) return io_fail(TAXOR_E_ARG, "hixf_probe: null argument");
    std::string rep;
    auto emit = [&](int rc) {
        if (report && cap) {
            const size_t n = std::min<size_t>(rep.size(), (size_t)cap - 1);
            std::memcpy(report, rep.data(), n);
            report[n] = 0;
        }
        return rc;
    };
    Mapped m;
    if (int rc = map_file(path, m)) return rc;
    taxor_hixf scratch;
    scratch.map = nullptr;
    Cursor c0{(const uint8_t *)m.map, (const uint8_t *)m.map + m.len};
    std::vector<uint64_t> sp_ub, sp_len;
    uint64_t n_ixf = 0;
    std::string e = parse_envelope(c0, &scratch, sp_ub, sp_len, n_ixf);
    if (!e.empty()) {
        munmap(m.map, m.len);
        return io_fail(TAXOR_E_IO, std::string(path) + ": " + e);
    }
    rep += "envelope ok: k=" + std::to_string(scratch.view.kmer_size) + " s=" + std::to_string(scratch.view.syncmer_size) +
           " t=" + std::to_string(scratch.view.t_syncmer) + " scaling=" + std::to_string(scratch.view.scaling) + " species=" +
           std::to_string(sp_ub.size()) + " ixf_count=" + std::to_string(n_ixf) + "\n";
    // 1. record framing: fewest scalars first
    bool found = false;
    taxor_ixf_schema sc{};
    std::vector<std::vector<uint64_t>> scal;
    std::vector<uint64_t> lens;
    taxor_hixf tail;
    // framing of the vector itself, likeliest first: a byte count (cereal's std::vector<uint8_t>), 64-bit words, bits in whole
    // words (sdsl), each also with one byte (an int_vector's width) before or behind the length word
    struct VecFrame { uint32_t unit, before, after; };
    static const VecFrame frames[] = {{1, 0, 0}, {8, 0, 0}, {64, 0, 0}, {64, 0, 1}, {64, 1, 0}, {1, 0, 1}, {1, 1, 0}, {8, 0, 1}, {8, 1, 0}};
    for (const VecFrame &vf : frames)
    for (uint32_t total = 0; total <= 16 && !found; ++total)
        for (uint32_t na = 0; na <= std::min<uint32_t>(total, 4) && !found; ++na) {
            taxor_ixf_schema t{total - na, na, -1, -1, -1, -1, 0, 13572355802537770549ull, 0, vf.unit, vf.before, vf.after};
            Cursor c = c0;
            taxor_hixf trial;
            scal.clear();
            lens.clear();
            size_t first_fn = 0;
            uint64_t n_files = 0;
            if (!parse_ixfs(c, t, &trial, n_ixf, &scal, &lens).empty()) continue;
            if (!parse_tail(c, &trial, n_ixf, first_fn, n_files, false).empty()) continue;
            found = true;
            sc = t;
            tail.next_ixf = trial.next_ixf;
        }
    if (!found) {
        munmap(m.map, m.len);
        rep += "no (n_before, n_after) framing re-parses to the end of the file\n";
        emit(0);
        return io_fail(TAXOR_E_IO, std::string(path) + ": IXF record framing not recognised");
    }
    rep += "framing: " + std::to_string(sc.n_before) + " u64 scalars | fingerprint vector | " + std::to_string(sc.n_after) + " u64 scalars\n";
    rep += std::string("vector: u64 length in ") + (sc.len_unit == 8 ? "64-bit words" : sc.len_unit == 64 ? "bits (whole 64-bit words stored)" : "bytes") +
           (sc.skip_before_len ? ", " + std::to_string(sc.skip_before_len) + " byte(s) before the length word" : std::string()) +
           (sc.skip_after_len ? ", " + std::to_string(sc.skip_after_len) + " byte(s) behind it" : std::string()) + "\n";
    // 2. which scalar is which
    const uint32_t ns = sc.n_before + sc.n_after;
    auto all = [&](auto pred) {
        std::vector<int> r;
        for (uint32_t j = 0; j < ns; ++j) {
            bool ok = true;
            for (uint64_t i = 0; i < n_ixf && ok; ++i) ok = pred(i, scal[i][j]);
            if (ok) r.push_back((int)j);
        }
        return r;
    };
    auto bins_of = [&](uint64_t i) { return (uint64_t)tail.next_ixf[i].size(); };
    const std::vector<int> cb = all([&](uint64_t i, uint64_t v) { return v == bins_of(i); });
    if (!cb.empty()) sc.idx_bins = cb[0];
    std::vector<int> cs = all([&](uint64_t i, uint64_t v) { return v >= bins_of(i) && v != 0 && lens[i] % v == 0 && v < bins_of(i) + 64; });
    for (int j : cs)
        if (j != sc.idx_bins) { sc.idx_stride = j; break; }
    // without a stored pitch: bins padded to 64 where every array length admits it, else exactly bins
    bool padded_fits = true;
    for (uint64_t i = 0; i < n_ixf; ++i) padded_fits = padded_fits && lens[i] % ceil64(bins_of(i)) == 0 && (lens[i] / ceil64(bins_of(i))) % 3 == 0;
    auto stride_of = [&](uint64_t i) { return sc.idx_stride >= 0 ? scal[i][sc.idx_stride] : padded_fits ? ceil64(bins_of(i)) : bins_of(i); };
    bool rows_ok = true;
    for (uint64_t i = 0; i < n_ixf; ++i) rows_ok = rows_ok && lens[i] % stride_of(i) == 0 && (lens[i] / stride_of(i)) % 3 == 0;
    const std::vector<int> cseg = all([&](uint64_t i, uint64_t v) { return 3 * v == lens[i] / stride_of(i); });
    const std::vector<int> crow = all([&](uint64_t i, uint64_t v) { return v == lens[i] / stride_of(i) && v != bins_of(i); });
    if (!cseg.empty()) sc.idx_seg_len = cseg[0];
    else if (!crow.empty()) { sc.idx_seg_len = crow[0]; sc.seg_len_is_rows = 1; }
    // seed: an unused scalar that looks like a 64-bit random value (or the reference's fixed start seed) everywhere
    uint64_t best_min = 0;
    for (uint32_t j = 0; j < ns; ++j) {
        if ((int)j == sc.idx_bins || (int)j == sc.idx_stride || (int)j == sc.idx_seg_len) continue;
        uint64_t mn = ~0ull;
        for (uint64_t i = 0; i < n_ixf; ++i) mn = std::min(mn, scal[i][j]);
        if (mn >= (1ull << 32) && mn > best_min) { best_min = mn; sc.idx_seed = (int)j; }
    }
    rep += "bins: " + (sc.idx_bins >= 0 ? "scalar " + std::to_string(sc.idx_bins) : std::string("not stored (taken from next_ixf_id)")) + "\n";
    rep += "row stride: " + (sc.idx_stride >= 0 ? "scalar " + std::to_string(sc.idx_stride) : std::string("not stored (ceil(bins/64)*64)")) + "\n";
    rep += "segment length: " + (sc.idx_seg_len >= 0 ? "scalar " + std::to_string(sc.idx_seg_len) + (sc.seg_len_is_rows ? " (holds 3*seg_len)" : "")
                                                     : std::string("not stored (rows/3)")) + "\n";
    rep += "seed: " + (sc.idx_seed >= 0 ? "scalar " + std::to_string(sc.idx_seed) : std::string("not stored -> default 13572355802537770549 (xorfilter.hpp:153) -- VERIFY with a positive control")) + "\n";
    if (!rows_ok) rep += "WARNING: fingerprint bytes are not 3 x seg_len x stride under this reading; the fork's layout differs\n";
    {   // which layouts the array lengths admit (the bytes themselves decide: `taxor verify --variants` / `taxor pin`)
        bool unpadded_fits = true, any_unaligned = false;
        for (uint64_t i = 0; i < n_ixf; ++i) {
            unpadded_fits = unpadded_fits && lens[i] % bins_of(i) == 0 && (lens[i] / bins_of(i)) % 3 == 0;
            any_unaligned = any_unaligned || bins_of(i) % 64 != 0;
        }
        rep += std::string("array lengths admit: ") + (padded_fits ? "pitch = bins padded to 64 (row-interleaved, bin-major or bit-sliced)" : "") +
               (padded_fits && unpadded_fits && any_unaligned ? "; " : "") + (unpadded_fits && any_unaligned ? "pitch = exactly bins (row-interleaved or bin-major)" : "") +
               (!any_unaligned ? " [every bin count is a multiple of 64: padded and unpadded coincide]" : "") + "\n";
    }
    for (uint32_t j = 0; j < ns; ++j) {
        rep += "  scalar " + std::to_string(j) + " of IXF 0: " + std::to_string(scal[0][j]) + "\n";
    }
    munmap(m.map, m.len);
    *out = sc;
    emit(0);
    if (!rows_ok) return io_fail(TAXOR_E_IO, std::string(path) + ": framing found but the fingerprint array does not factor as 3*seg_len*stride");
    return TAXOR_OK;
}

extern "C" int taxor_hixf_load(const char *path, taxor_hixf **out)
{
    if (!path || !out) return io_fail(TAXOR_E_ARG, "hixf_load: null argument");
    taxor_ixf_schema sc;
    taxor_ixf_schema_default(&sc);
    bool schema_problem = false;
    const int rc = load_with(path, sc, out, &schema_problem);
    if (rc == TAXOR_OK && (*out)->pitch_overridden) (*out)->meta.foreign_schema = 1;      // own framing, but not this library's pitch: see finish_ixfs
    if (rc == TAXOR_OK || !schema_problem) return rc;
    // the records did not fit this library's schema: probe the file and retry with what it found
    taxor_ixf_schema probed;
    if (taxor_hixf_probe(path, &probed, nullptr, 0) != TAXOR_OK) return TAXOR_E_IO;
    const int rc2 = load_with(path, probed, out, nullptr);
    if (rc2 == TAXOR_OK) (*out)->meta.foreign_schema = 1;
    return rc2;
}

// Give the fingerprint pages of the mapping back, slice by slice (MADV_DONTNEED takes the address-space lock shared and
// only for a slice at a time; one munmap of 113 GB of populated mapping holds it exclusively for seconds and stalls every
// page fault and allocation of the process meanwhile).  After this the view's data pointers must not be read through
// any more by the caller's own code -- index creation and the source reader do not need them.
extern "C" void taxor_hixf_release_data(taxor_hixf *h)
{
    if (!h || !h->map) return;
    const uintptr_t base = (uintptr_t)h->map;
    for (size_t i = 0; i < h->ixf.size(); ++i) {
        const uint64_t len = h->raw_len[i];
        uintptr_t a = (base + h->file_off[i] + 4095) & ~(uintptr_t)4095, e = (base + h->file_off[i] + len) & ~(uintptr_t)4095;
        for (; a < e; a += (256ull << 20)) madvise((void *)a, std::min<uintptr_t>(256ull << 20, e - a), MADV_DONTNEED);
    }
}

extern "C" void taxor_hixf_free(taxor_hixf *h)
{
    if (!h) return;
    if (h->map) munmap(h->map, h->map_len);
    if (h->fd >= 0) close(h->fd);
    delete h;
}

extern "C" void taxor_hixf_set_arith(taxor_hixf *h, uint32_t arith)
{
    if (h) h->view.ixf_arith = arith;
}

extern "C" int taxor_hixf_set_layout(taxor_hixf *h, uint32_t layout)
{
    if (!h) return io_fail(TAXOR_E_ARG, "hixf_set_layout: null handle");
    if (!taxor::ixf_layout_valid(layout)) return io_fail(TAXOR_E_ARG, "hixf_set_layout: unknown layout code " + std::to_string(layout));
    std::vector<taxor_ixf_view> before = h->ixf;
    for (size_t i = 0; i < h->ixf.size(); ++i) {
        const std::string e = apply_layout(h, i, layout);
        if (!e.empty()) {
            h->ixf = before;
            h->view.ixf = h->ixf.data();
            return io_fail(TAXOR_E_ARG, "hixf_set_layout: " + e);
        }
    }
    h->view.ixf_layout = layout;
    return TAXOR_OK;
}

extern "C" uint64_t taxor_hixf_ixf_raw_bytes(const taxor_hixf *h, uint64_t ixf) { return h && ixf < h->raw_len.size() ? h->raw_len[ixf] : 0; }

extern "C" int taxor_ixf_layout_parse(const char *spec, uint32_t *code)
{
    if (!spec || !code) return io_fail(TAXOR_E_ARG, "ixf_layout_parse: null argument");
    uint32_t c = 0;
    const std::string s(spec);
    for (size_t a = 0; a <= s.size();) {
        size_t b = s.find(',', a);
        if (b == std::string::npos) b = s.size();
        const std::string t = s.substr(a, b - a);
        a = b + 1;
        if (t.empty()) continue;
        if (t == "interleaved") c = (c & ~taxor::IXF_KIND_MASK) | taxor::IXF_KIND_ROWS;
        else if (t == "bin-major") c = (c & ~taxor::IXF_KIND_MASK) | taxor::IXF_KIND_BIN_MAJOR;
        else if (t == "bit-sliced") c = (c & ~taxor::IXF_KIND_MASK) | taxor::IXF_KIND_BIT_SLICED;
        else if (t == "padded") c = (c & ~taxor::IXF_PITCH_MASK) | taxor::IXF_PITCH_PADDED;
        else if (t == "unpadded") c = (c & ~taxor::IXF_PITCH_MASK) | taxor::IXF_PITCH_BINS;
        else if (t == "stored-pitch") c = (c & ~taxor::IXF_PITCH_MASK) | taxor::IXF_PITCH_STORED;
        else if (t == "segment-major") c &= ~taxor::IXF_ROWS_POSITION_MAJOR;
        else if (t == "position-major") c |= taxor::IXF_ROWS_POSITION_MAJOR;
        else return io_fail(TAXOR_E_ARG, "ixf_layout_parse: unknown token '" + t + "' (interleaved | bin-major | bit-sliced, padded | unpadded | stored-pitch, segment-major | position-major)");
    }
    if (!taxor::ixf_layout_valid(c)) return io_fail(TAXOR_E_ARG, "ixf_layout_parse: '" + s + "' is not a layout (bit-sliced words have no pitch choice)");
    *code = c;
    return TAXOR_OK;
}

extern "C" uint64_t taxor_ixf_layout_describe(uint32_t code, char *buf, uint64_t cap)
{
    if (!buf || !cap) return 0;
    static const char *kind[] = {"interleaved", "bin-major", "bit-sliced"};
    const uint32_t k = taxor::ixf_layout_kind(code), rule = code & taxor::IXF_PITCH_MASK;
    std::string t = k <= 2 ? kind[k] : "?";
    if (k != taxor::IXF_KIND_BIT_SLICED) t += rule == taxor::IXF_PITCH_BINS ? ",unpadded" : rule == taxor::IXF_PITCH_STORED ? ",stored-pitch" : ",padded";
    t += (code & taxor::IXF_ROWS_POSITION_MAJOR) ? ",position-major" : ",segment-major";
    const size_t n = std::min<size_t>(t.size(), (size_t)cap - 1);
    std::memcpy(buf, t.data(), n);
    buf[n] = 0;
    return n;
}

extern "C" const taxor_hixf_view *taxor_hixf_get_view(const taxor_hixf *h) { return h ? &h->view : nullptr; }
extern "C" const taxor_hixf_meta *taxor_hixf_get_meta(const taxor_hixf *h) { return h ? &h->meta : nullptr; }

extern "C" int taxor_hixf_store_schema(const char *path, const taxor_hixf_view *v, const taxor_hixf_meta *m,
                                       const taxor_ixf_schema *sc)
{
    if (!path || !v || !m || !sc) return io_fail(TAXOR_E_ARG, "hixf_store: null argument");
    if (!taxor::ixf_layout_valid(sc->layout)) return io_fail(TAXOR_E_ARG, "hixf_store: unknown layout code " + std::to_string(sc->layout));
    for (uint64_t i = 0; i < v->n_ixf; ++i)
        if (taxor::ixf_layout_kind(v->ixf_layout) != taxor::IXF_KIND_ROWS || (v->ixf_layout & taxor::IXF_ROWS_POSITION_MAJOR) ||
            (v->ixf[i].src_stride != 0 && v->ixf[i].src_stride != v->ixf[i].stride))
            return io_fail(TAXOR_E_ARG, "hixf_store: the view's bytes must be in the search layout (data[row * stride + bin])");
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
    const uint32_t ns = sc->n_before + sc->n_after;
    std::vector<uint64_t> sv(ns);
    const uint32_t code = sc->layout, kind = taxor::ixf_layout_kind(code), rule = code & taxor::IXF_PITCH_MASK;
    for (uint64_t i = 0; i < v->n_ixf; ++i) {
        const taxor_ixf_view &x = v->ixf[i];
        if (!x.data && !v->source) {
            fclose(f);
            return io_fail(TAXOR_E_ARG, "hixf_store: IXF without host data");
        }
        // the file's pitch (row pitch / bin columns stored) under the schema's layout; the search layout's own is written as it lies
        const uint64_t rows = 3 * x.seg_len;
        const uint64_t pitch = kind == taxor::IXF_KIND_BIT_SLICED ? ceil64(x.bins) : rule == taxor::IXF_PITCH_BINS ? x.bins : rule == taxor::IXF_PITCH_STORED ? x.stride
                               : (kind == taxor::IXF_KIND_ROWS ? x.stride : ceil64(x.bins));
        const bool as_is = kind == taxor::IXF_KIND_ROWS && !(code & taxor::IXF_ROWS_POSITION_MAJOR) && pitch == x.stride;
        // scalars the schema does not name carry the other members such a class would hold
        const uint64_t filler[4] = {x.stride / 64, 8, x.bins ? (x.bins + 63) / 64 : 0, 3};
        uint32_t fi = 0;
        for (uint32_t j = 0; j < ns; ++j) {
            if ((int)j == sc->idx_bins) sv[j] = x.bins;
            else if ((int)j == sc->idx_stride) sv[j] = pitch;
            else if ((int)j == sc->idx_seg_len) sv[j] = sc->seg_len_is_rows ? 3 * x.seg_len : x.seg_len;
            else if ((int)j == sc->idx_seed) sv[j] = x.seed;
            else sv[j] = filler[fi++ % 4];
        }
        for (uint32_t j = 0; j < sc->n_before; ++j) w.put<uint64_t>(sv[j]);
        const uint64_t native_len = rows * x.stride, len = as_is ? native_len : taxor::ixf_src_bytes(code, rows, pitch, x.bins);
        {   // the vector's framing under the schema: length word in its unit, optional filler bytes around it, words padded with zeros
            static const uint8_t filler8[8] = {8, 0, 0, 0, 0, 0, 0, 0};       // (an int_vector's width byte)
            if (sc->skip_before_len) w.bytes(filler8, sc->skip_before_len);
            w.put<uint64_t>(sc->len_unit == 8 ? (len + 7) / 8 : sc->len_unit == 64 ? len * 8 : len);
            if (sc->skip_after_len) w.bytes(filler8, sc->skip_after_len);
        }
        const uint64_t word_pad = sc->len_unit == 8 || sc->len_unit == 64 ? (8 - len % 8) % 8 : 0;
        if (sc->len_unit == 8 && len % 8 != 0) {
            fclose(f);
            return io_fail(TAXOR_E_ARG, "hixf_store: a vector of 64-bit words cannot hold " + std::to_string(len) + " fingerprint bytes");
        }
        if (!as_is) {
            // another writer's layout: the inverse of what index creation does on the device (relayout.hip), on the host, one IXF
            // in memory at a time -- tests of the re-layout and export; columns beyond `bins` are written as zeros
            std::vector<uint8_t> native;
            const uint8_t *src = x.data;
            if (v->source) {
                native.resize(native_len);
                if (v->source->read(v->source->ctx, i, 0, native_len, native.data()) != 0) {
                    fclose(f);
                    return io_fail(TAXOR_E_IO, "hixf_store: the source failed to deliver IXF " + std::to_string(i));
                }
                src = native.data();
            }
            std::vector<uint8_t> out(len, 0);
            const uint64_t groups = (x.bins + 63) / 64;
            // tiles of 64 rows, shared out over a few threads (every source row maps to bytes of its own: no two tiles write the same byte)
            const uint64_t n_tiles = (rows + 63) / 64;
            const unsigned T = (unsigned)std::min<uint64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), std::max<uint64_t>(1, len >> 22));
            std::atomic<uint64_t> next{0};
            auto work = [&] {
                for (uint64_t t = next.fetch_add(1); t < n_tiles; t = next.fetch_add(1)) {
                    const uint64_t ra = t * 64, rb = std::min(rows, ra + 64);
                    if (kind == taxor::IXF_KIND_BIN_MAJOR) {
                        for (uint64_t b = 0; b < x.bins; ++b)
                            for (uint64_t r = ra; r < rb; ++r) out[b * rows + taxor::ixf_src_row(code, r, x.seg_len)] = src[r * x.stride + b];
                        continue;
                    }
                    for (uint64_t r = ra; r < rb; ++r) {
                        const uint64_t rs = taxor::ixf_src_row(code, r, x.seg_len);
                        const uint8_t *row = src + r * x.stride;
                        if (kind == taxor::IXF_KIND_ROWS) std::memcpy(out.data() + rs * pitch, row, x.bins);
                        else
                            for (uint64_t b = 0; b < x.bins; ++b)
                                for (uint32_t p2 = 0; p2 < 8; ++p2)
                                    if ((row[b] >> p2) & 1u) out[(rs * groups + b / 64) * 64 + p2 * 8 + ((b & 63) >> 3)] |= (uint8_t)(1u << (b & 7));
                    }
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < T; ++t) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
            w.bytes(out.data(), len);
        } else if (v->source) {                                      // bytes that are not in host memory as a whole (e.g. an
            std::vector<uint8_t> piece((size_t)std::min<uint64_t>(len, 64ull << 20));   // index resident on a GPU): piece by piece
            for (uint64_t o = 0; o < len && w.ok; o += piece.size()) {
                const uint64_t n = std::min<uint64_t>(piece.size(), len - o);
                if (v->source->read(v->source->ctx, i, o, n, piece.data()) != 0) {
                    fclose(f);
                    return io_fail(TAXOR_E_IO, "hixf_store: the source failed to deliver IXF " + std::to_string(i));
                }
                w.bytes(piece.data(), n);
            }
        } else
            w.bytes(x.data, len);
        if (word_pad) { static const uint8_t z[8] = {0}; w.bytes(z, word_pad); }
        for (uint32_t j = 0; j < sc->n_after; ++j) w.put<uint64_t>(sv[sc->n_before + j]);
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

extern "C" int taxor_hixf_store(const char *path, const taxor_hixf_view *v, const taxor_hixf_meta *m)
{
    taxor_ixf_schema sc;
    taxor_ixf_schema_default(&sc);
    return taxor_hixf_store_schema(path, v, m, &sc);
}

namespace {

inline char *put_u64(char *p, uint64_t v)
{
    char tmp[20];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline uint64_t species_of(const taxor_hixf *h, int64_t user_bin)
{
    if ((uint64_t)user_bin < h->ub_species.size()) return h->ub_species[(uint64_t)user_bin];
    const auto it = h->user_bin_index.find((uint64_t)user_bin);
    return it == h->user_bin_index.end() ? 0 : it->second;
}

// bytes the lines of one read need (exact)
inline uint64_t read_text_size(const taxor_hixf *h, uint64_t id_len, uint64_t read_len, uint32_t n_hashes, const int64_t *user_bin,
                               const uint32_t *count, uint64_t n_tuples)
{
    auto digits = [](uint64_t v) { uint64_t d = 1; while (v >= 10) { v /= 10; ++d; } return d; };
    if (n_tuples == 0) return id_len + 1 + 8 + digits(read_len) + 1;
    if (h->meta.n_species == 0) return 0;
    uint32_t max_count = 0;
    for (uint64_t i = 0; i < n_tuples; ++i) max_count = std::max(max_count, count[i]);
    const uint64_t fixed = id_len + 1 + digits(read_len) + 1 + digits(n_hashes) + 1;
    uint64_t need = 0;
    for (uint64_t i = 0; i < n_tuples; ++i) {
        if (static_cast<double>(count[i]) < static_cast<double>(max_count) * 0.8) continue;
        const uint64_t si = species_of(h, user_bin[i]);
        need += fixed + h->line_head[si].size() + digits(count[i]) + 1 + h->line_tail[si].size();
    }
    return need;
}

// taxor_search.cpp:268-305; p has room for read_text_size() bytes
inline char *write_read_text(const taxor_hixf *h, const char *id, uint64_t id_len, uint64_t read_len, uint32_t n_hashes,
                             const int64_t *user_bin, const uint32_t *count, uint64_t n_tuples, char *p)
{
    if (n_tuples == 0) {                                                              // :268-273
        std::memcpy(p, id, id_len); p += id_len;
        std::memcpy(p, "\t-\t-\t-\t-\t", 9); p += 9;
        p = put_u64(p, read_len);
        *p++ = '\n';
        return p;
    }
    if (h->meta.n_species == 0) return p;
    uint32_t max_count = 0;                                                           // :275-280
    for (uint64_t i = 0; i < n_tuples; ++i) max_count = std::max(max_count, count[i]);
    for (uint64_t i = 0; i < n_tuples; ++i) {
        if (static_cast<double>(count[i]) < static_cast<double>(max_count) * 0.8) continue; // :285
        const uint64_t si = species_of(h, user_bin[i]);                               // :289
        std::memcpy(p, id, id_len); p += id_len;
        *p++ = '\t';
        const std::string &hd = h->line_head[si];
        std::memcpy(p, hd.data(), hd.size()); p += hd.size();
        p = put_u64(p, read_len); *p++ = '\t';
        p = put_u64(p, n_hashes); *p++ = '\t';
        p = put_u64(p, count[i]); *p++ = '\t';
        const std::string &tl = h->line_tail[si];
        std::memcpy(p, tl.data(), tl.size()); p += tl.size();
    }
    return p;
}

} // namespace

// taxor_search.cpp:268-305
extern "C" uint64_t taxor_format_read(const taxor_hixf *h, const char *id, uint64_t id_len, uint64_t read_len,
                                      uint32_t n_hashes, const int64_t *user_bin, const uint32_t *count, uint64_t n_tuples,
                                      char *buf, uint64_t cap)
{
    const uint64_t need = read_text_size(h, id_len, read_len, n_hashes, user_bin, count, n_tuples);
    if (need <= cap && buf) write_read_text(h, id, id_len, read_len, n_hashes, user_bin, count, n_tuples, buf);
    return need;
}

extern "C" uint64_t taxor_format_reads(const taxor_hixf *h, uint64_t n_reads, const char *const *ids, const uint64_t *id_len,
                                       const uint64_t *read_len, const uint32_t *n_hashes, const uint64_t *read_off,
                                       const int64_t *user_bin, const uint32_t *count, char *buf, uint64_t cap)
{
    uint64_t need = 0;
    for (uint64_t r = 0; r < n_reads; ++r)
        need += read_text_size(h, id_len[r], read_len[r], n_hashes[r], user_bin + read_off[r], count + read_off[r], read_off[r + 1] - read_off[r]);
    if (need > cap || !buf) return need;
    char *p = buf;
    for (uint64_t r = 0; r < n_reads; ++r)
        p = write_read_text(h, ids[r], id_len[r], read_len[r], n_hashes[r], user_bin + read_off[r], count + read_off[r],
                            read_off[r + 1] - read_off[r], p);
    return (uint64_t)(p - buf);
}
