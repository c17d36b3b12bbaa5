// search_main.cpp -- `taxor search` on MI355X: same command line, same .hixf index, same per-read TSV as the
// reference's subcommand (src/main/taxor_search.cpp), with the chunk loop replaced by the C ABI
// (include/taxor_gpu.h).  Host work here: argument parsing (:32-80), sanity checks (:97-151), FASTA/FASTQ(.gz)
// reading (:181-184), batching (:315-326) and output (:268-311, :343).
#include "../../include/taxor_gpu.h"

#include <zlib.h>

#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Config {                              // taxor_search_configuration.hpp:8-20
    std::string index_file, query_file, report_file;
    double threshold = -1.0, error_rate = 0.04;
    unsigned threads = 1;
    int gpu = 0;
    uint64_t batch_reads = 65536, batch_bases = 1ull << 30;
};

std::vector<std::string> str_split(const std::string &s, char delim)         // taxor_search.cpp:82-95
{
    std::vector<std::string> out;
    size_t a = 0;
    while (a <= s.size()) {
        const size_t b = s.find(delim, a);
        if (b == std::string::npos) { if (a < s.size()) out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}

bool file_exists(const std::string &p)
{
    struct stat sb;
    return stat(p.c_str(), &sb) == 0;
}

[[noreturn]] void die(const std::string &msg)
{
    fprintf(stderr, "[TAXOR SEARCH ERROR] %s\n", msg.c_str());             // :380-384
    exit(-1);
}

void usage()
{
    fprintf(stderr,
            "taxor search - Queries files of DNA sequences against a list of HIXF index files (MI355X)\n"
            "  --index-file <f[,f..]>   taxor index file(s) containing HIXF index and reference information (required)\n"
            "  --query-file <f[,f..]>   file(s) containing sequences to query against the index\n"
            "  --output-file <f>        file name for the resulting output\n"
            "  --threads <1..32>        host worker threads (accepted for compatibility)\n"
            "  --percentage <0..1>      if set, this threshold is used instead of the syncmer model\n"
            "  --error-rate <0..1>      expected error rate of the reads (default 0.04)\n"
            "  --gpu <id>               device ordinal (default 0)\n"
            "  --batch-reads <n>        reads per GPU batch (default 65536)\n");
}

// ---- minimal FASTA / FASTQ reader over zlib (plain or .gz); ids are the full header line --------------------
struct FastxReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool eof = false;
    std::string pending; // header line read ahead (FASTA)

    bool open(const std::string &path)
    {
        f = gzopen(path.c_str(), "rb");
        buf.resize(1 << 20);
        return f != nullptr;
    }
    ~FastxReader() { if (f) gzclose(f); }
    bool getline(std::string &line)
    {
        line.clear();
        for (;;) {
            if (pos == len) {
                if (eof) return !line.empty();
                const int n = gzread(f, buf.data(), (unsigned)buf.size());
                if (n <= 0) { eof = true; return !line.empty(); }
                pos = 0;
                len = (size_t)n;
            }
            const char *s = buf.data() + pos;
            const char *e = (const char *)memchr(s, '\n', len - pos);
            if (e) {
                line.append(s, e - s);
                pos += (size_t)(e - s) + 1;
                if (!line.empty() && line.back() == '\r') line.pop_back();
                return true;
            }
            line.append(s, len - pos);
            pos = len;
        }
    }
    // returns false at end of file
    bool next(std::string &id, std::string &seq)
    {
        std::string line;
        if (pending.empty()) {
            do {
                if (!getline(line)) return false;
            } while (line.empty());
        } else {
            line.swap(pending);
            pending.clear();
        }
        seq.clear();
        if (line[0] == '>') {
            id = line.substr(1);
            while (getline(line)) {
                if (!line.empty() && line[0] == '>') { pending = line; break; }
                seq += line;
            }
            return true;
        }
        if (line[0] == '@') {
            id = line.substr(1);
            if (!getline(seq)) die("truncated FASTQ record: " + id);
            std::string plus, qual;
            if (!getline(plus) || plus.empty() || plus[0] != '+' || !getline(qual)) die("malformed FASTQ record: " + id);
            return true;
        }
        die("query file is neither FASTA nor FASTQ");
    }
};

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

} // namespace

int main(int argc, char **argv)
{
    int a = 1;
    if (argc > 1 && strcmp(argv[1], "search") == 0) a = 2;                 // `taxor search ...` like the reference
    else if (argc > 1 && (strcmp(argv[1], "build") == 0 || strcmp(argv[1], "profile") == 0)) {
        fprintf(stderr, "[TAXOR ERROR] only the `search` subcommand is provided by this build\n");
        return -1;
    }
    Config cfg;
    for (; a < argc; ++a) {
        const std::string k = argv[a];
        auto val = [&]() -> std::string {
            if (a + 1 >= argc) die("Missing value for option " + k);
            return argv[++a];
        };
        if (k == "--index-file") cfg.index_file = val();
        else if (k == "--query-file") cfg.query_file = val();
        else if (k == "--output-file") cfg.report_file = val();
        else if (k == "--threads") {
            const long t = atol(val().c_str());
            if (t < 1 || t > 32) die("Validation failed for option --threads: Value not in range [1,32].");   // :51-55
            cfg.threads = (unsigned)t;
        } else if (k == "--percentage") {
            cfg.threshold = atof(val().c_str());
            if (cfg.threshold < 0.0 || cfg.threshold > 1.0) die("Validation failed for option --percentage: Value not in range [0,1]."); // :57-61
        } else if (k == "--error-rate") {
            cfg.error_rate = atof(val().c_str());
            if (cfg.error_rate < 0.0 || cfg.error_rate > 1.0) die("Validation failed for option --error-rate: Value not in range [0,1]."); // :63-67
        } else if (k == "--gpu") cfg.gpu = atoi(val().c_str());
        else if (k == "--batch-reads") cfg.batch_reads = strtoull(val().c_str(), nullptr, 10);
        else if (k == "-h" || k == "--help") { usage(); return 0; }
        else die("Unknown option " + k + ". In case this is meant to be a non-option/argument/parameter, please specify the start of non-options with '--'.");
    }
    if (cfg.index_file.empty()) die("Option --index-file is required but not set.");

    // ---- sanity checks (taxor_search.cpp:97-151) --------------------------------------------------------------
    printf("checking input ... ");
    fflush(stdout);
    const std::vector<std::string> index_files = str_split(cfg.index_file, ',');
    const std::vector<std::string> query_files = str_split(cfg.query_file, ',');
    for (const auto &f : index_files)
        if (!file_exists(f)) die("Please check the given index file(s). \nThe following index file does not exist: " + f);
    if (index_files.size() > 1) {
        uint8_t k0 = 1, s0 = 0, t0 = 0, syn0 = 0;
        uint64_t w0 = 0;
        uint16_t sc0 = 0;
        for (const auto &f : index_files) {
            taxor_hixf *h = nullptr;
            if (taxor_hixf_load(f.c_str(), &h) != TAXOR_OK) die(taxor_gpu_last_error());
            const taxor_hixf_view *v = taxor_hixf_get_view(h);
            const taxor_hixf_meta *m = taxor_hixf_get_meta(h);
            if (k0 == 1) { k0 = v->kmer_size; s0 = v->syncmer_size; t0 = v->t_syncmer; syn0 = v->use_syncmer; w0 = m->window_size; sc0 = v->scaling; }
            else if (k0 != v->kmer_size || s0 != v->syncmer_size || t0 != v->t_syncmer || syn0 != v->use_syncmer || w0 != m->window_size || sc0 != v->scaling)
                die("At least two index files have been created with different kmer selection schemes.\n Please provide only index files using the same kmer-/syncmer-/window-size!");
            taxor_hixf_free(h);
        }
    }
    for (const auto &f : query_files)
        if (!file_exists(f)) die("Please check the given input query files. \nThe following query file does not exist: " + f);
    printf("done!\n");

    FILE *out = fopen(cfg.report_file.c_str(), "wb");                       // search_hixf, :340-343
    if (!out) die("cannot open output file " + cfg.report_file);
    fputs("#QUERY_NAME\tACCESSION\tREFERENCE_NAME\tTAXID\tREF_LEN\tQUERY_LEN\tQHASH_COUNT\tQHASH_MATCH\tTAX_STR\tTAX_ID_STR\n", out);

    double t_index = 0, t_reads = 0, t_compute = 0;
    uint64_t total_reads = 0, total_bases = 0;
    for (const auto &query : query_files) {
        for (const auto &hixf_file : index_files) {                        // :344-358
            double t0 = now();
            taxor_hixf *h = nullptr;
            if (taxor_hixf_load(hixf_file.c_str(), &h) != TAXOR_OK) die(taxor_gpu_last_error());
            const taxor_hixf_view *view = taxor_hixf_get_view(h);
            taxor_gpu_index *gidx = nullptr;
            if (taxor_gpu_index_create(view, cfg.gpu, &gidx) != TAXOR_OK) die(taxor_gpu_last_error());
            t_index += now() - t0;
            // threshold model (threshold.hpp:22-47)
            const double ratio = taxor_threshold_ratio(view->kmer_size, cfg.error_rate, cfg.threshold);
            if (cfg.threshold > 0.0 && cfg.threshold <= 1.0) printf("use percentage-model\t%g\n", cfg.threshold);
            else printf("use syncmer model\n");
            if (ratio < 0) die("no syncmer threshold model for k=" + std::to_string(view->kmer_size) + " and error rate " + std::to_string(cfg.error_rate));
            taxor_gpu_search_params prm{ratio, 0, 0, 0};
            taxor_gpu_searcher *sr = nullptr;
            if (taxor_gpu_searcher_create(gidx, &prm, &sr) != TAXOR_OK) die(taxor_gpu_last_error());

            FastxReader rd;
            if (!rd.open(query)) die("cannot open query file " + query);
            std::vector<std::string> ids;
            std::string bases, id, seq, text;
            std::vector<uint64_t> offsets;
            std::vector<char> line;
            bool more = true;
            while (more) {
                t0 = now();
                ids.clear();
                bases.clear();
                offsets.assign(1, 0);
                while (ids.size() < cfg.batch_reads && bases.size() < cfg.batch_bases && (more = rd.next(id, seq))) {
                    ids.push_back(id);
                    bases += seq;
                    offsets.push_back(bases.size());
                }
                t_reads += now() - t0;
                if (ids.empty()) break;
                t0 = now();
                taxor_gpu_results res{};
                if (taxor_gpu_search_batch(sr, bases.data(), offsets.data(), ids.size(), &res) != TAXOR_OK) die(taxor_gpu_last_error());
                t_compute += now() - t0;
                text.clear();
                for (size_t r = 0; r < ids.size(); ++r) {
                    const uint64_t lo = res.read_off[r], n = res.read_off[r + 1] - lo;
                    const uint64_t rl = offsets[r + 1] - offsets[r];
                    uint64_t need = taxor_format_read(h, ids[r].data(), ids[r].size(), rl, res.n_hashes[r], res.user_bin + lo,
                                                      res.count + lo, n, line.data(), line.size());
                    if (need > line.size()) {
                        line.resize(need + 1024);
                        need = taxor_format_read(h, ids[r].data(), ids[r].size(), rl, res.n_hashes[r], res.user_bin + lo,
                                                 res.count + lo, n, line.data(), line.size());
                    }
                    text.append(line.data(), need);
                }
                fwrite(text.data(), 1, text.size(), out);
                total_reads += ids.size();
                total_bases += bases.size();
            }
            taxor_gpu_searcher_destroy(sr);
            taxor_gpu_index_destroy(gidx);
            taxor_hixf_free(h);
        }
    }
    fclose(out);
    printf("Index I/O\tReads I/O\tCompute\n%.2f\t%.2f\t%.2f\n", t_index, t_reads, t_compute);   // :328-336
    printf("%llu reads, %llu bases classified\n", (unsigned long long)total_reads, (unsigned long long)total_bases);
    return 0;
}
