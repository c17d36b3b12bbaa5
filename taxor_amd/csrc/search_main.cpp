// search_main.cpp -- `taxor search` on MI355X: same command line, same .hixf index, same per-read TSV as the
// reference's subcommand (src/main/taxor_search.cpp), with the chunk loop replaced by the C ABI
// (include/taxor_gpu.h).  Host work here: argument parsing (:32-80), sanity checks (:97-151), FASTA/FASTQ(.gz)
// reading (:181-184), batching (:315-326) and output (:268-311, :343).
#include "../../include/taxor_gpu.h"

#include <zlib.h>

#include <sys/stat.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// one chunk of records on its way through the pipeline: reader -> GPU -> formatter/writer
struct Batch {
    uint64_t seq = 0;   // position of this chunk in the input
    std::vector<std::string> ids;
    std::string bases;
    std::vector<uint64_t> offsets;
    // results copied out of the searcher (its buffers are reused by the next batch)
    std::vector<uint64_t> read_off;
    std::vector<int64_t> user_bin;
    std::vector<uint32_t> count, n_hashes;
};

template <typename T> class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T v)
    {
        std::unique_lock<std::mutex> lk(m_);
        not_full_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    bool pop(T &out) // false once closed and drained
    {
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        not_empty_.notify_all();
    }

private:
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

struct Config {                              // taxor_search_configuration.hpp:8-20
    std::string index_file, query_file, report_file;
    double threshold = -1.0, error_rate = 0.04;
    unsigned threads = 1;
    std::vector<int> gpus{0};   // devices that classify batches in parallel, each with its own index replica
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
            "  --gpus <n>               use devices 0..n-1: the index is replicated, batches of reads are sharded\n"
            "  --gpu-list <a,b,..>      explicit device list (a device may be listed twice)\n"
            "  --batch-reads <n>        reads per GPU batch (default 65536)\n");
}

// ---- FASTA / FASTQ reader over zlib (plain or .gz); ids are the full header line (seqan3 default).  Lines are
//      located with memchr inside a large refill buffer; sequence lines are appended straight into the batch and
//      quality lines are skipped without being copied.
struct FastxReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, len = 0;
    bool eof = false;
    std::string pending; // header line read ahead (FASTA)

    bool open(const std::string &path)
    {
        f = gzopen(path.c_str(), "rb");
        if (f) gzbuffer(f, 1 << 20);
        buf.resize(8u << 20);
        return f != nullptr;
    }
    ~FastxReader() { if (f) gzclose(f); }
    bool refill()
    {
        if (eof) return false;
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) { eof = true; return false; }
        pos = 0;
        len = (size_t)n;
        return true;
    }
    // next line -> appended to `out` (if non-null); returns false at end of input with nothing read
    bool line_to(std::string *out, bool *nonempty_first_char = nullptr, char *first = nullptr)
    {
        bool any = false;
        for (;;) {
            if (pos == len && !refill()) {
                if (any && out && !out->empty() && out->back() == '\r') out->pop_back();
                return any;
            }
            const char *s = buf.data() + pos;
            const char *e = (const char *)memchr(s, '\n', len - pos);
            const size_t n = e ? (size_t)(e - s) : len - pos;
            if (!any && n && first) { *first = s[0]; if (nonempty_first_char) *nonempty_first_char = true; }
            if (out) out->append(s, n);
            any = any || n || e;
            pos += n + (e ? 1 : 0);
            if (e) {
                if (out && !out->empty() && out->back() == '\r') out->pop_back();
                return true;
            }
        }
    }
    bool getline(std::string &line)
    {
        line.clear();
        return line_to(&line);
    }
    // appends the record's sequence to `bases`; returns false at end of file
    bool next(std::string &id, std::string &bases)
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
        if (line[0] == '>') {
            id.assign(line, 1, std::string::npos);
            for (;;) {
                // peek the first character of the next line
                if (pos == len && !refill()) break;
                if (buf[pos] == '>') { getline(pending); break; }
                line_to(&bases);
            }
            return true;
        }
        if (line[0] == '@') {
            id.assign(line, 1, std::string::npos);
            if (!line_to(&bases)) die("truncated FASTQ record: " + id);
            if (!getline(line) || line.empty() || line[0] != '+') die("malformed FASTQ record: " + id);
            if (!line_to(nullptr)) die("truncated FASTQ record: " + id);   // quality: skipped, never copied
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
    if (argc > 1 && strcmp(argv[1], "probe") == 0) {                       // hixf-probe: report a file's IXF record layout
        const char *path = nullptr;
        for (int i = 2; i < argc; ++i) {
            if (strcmp(argv[i], "--index-file") == 0 && i + 1 < argc) path = argv[++i];
            else if (argv[i][0] != '-') path = argv[i];
        }
        if (!path) die("usage: taxor probe --index-file <file.hixf>");
        taxor_ixf_schema sc;
        std::vector<char> rep(16384);
        const int rc = taxor_hixf_probe(path, &sc, rep.data(), rep.size());
        fputs(rep.data(), stdout);
        if (rc != TAXOR_OK) die(taxor_gpu_last_error());
        printf("schema: n_before=%u n_after=%u idx_bins=%d idx_stride=%d idx_seg_len=%d%s idx_seed=%d\n", sc.n_before, sc.n_after,
               sc.idx_bins, sc.idx_stride, sc.idx_seg_len, sc.seg_len_is_rows ? "(rows)" : "", sc.idx_seed);
        return 0;
    }
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
        } else if (k == "--gpu") cfg.gpus.assign(1, atoi(val().c_str()));
        else if (k == "--gpus") {
            const int n = atoi(val().c_str());
            if (n < 1 || n > 64) die("Validation failed for option --gpus: Value not in range [1,64].");
            cfg.gpus.clear();
            for (int i = 0; i < n; ++i) cfg.gpus.push_back(i);
        } else if (k == "--gpu-list") {
            cfg.gpus.clear();
            for (const auto &t : str_split(val(), ',')) cfg.gpus.push_back(atoi(t.c_str()));
            if (cfg.gpus.empty()) die("--gpu-list is empty");
        }
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
    std::mutex stat_mu;
    for (const auto &query : query_files) {
        for (const auto &hixf_file : index_files) {                        // :344-358
            double t0 = now();
            taxor_hixf *h = nullptr;
            if (taxor_hixf_load(hixf_file.c_str(), &h) != TAXOR_OK) die(taxor_gpu_last_error());
            const taxor_hixf_view *view = taxor_hixf_get_view(h);
            // one index replica + searcher per device (reads are independent, taxor_search.cpp:214: the index is
            // replicated, batches are sharded); replicas are uploaded concurrently
            const size_t ng = cfg.gpus.size();
            std::vector<taxor_gpu_index *> gidx(ng, nullptr);
            {
                std::vector<std::thread> up;
                std::vector<std::string> errs(ng);
                for (size_t g = 0; g < ng; ++g)
                    up.emplace_back([&, g] {
                        if (taxor_gpu_index_create(view, cfg.gpus[g], &gidx[g]) != TAXOR_OK) errs[g] = taxor_gpu_last_error();
                    });
                for (auto &t : up) t.join();
                for (const auto &e : errs)
                    if (!e.empty()) die(e);
            }
            t_index += now() - t0;
            // threshold model (threshold.hpp:22-47)
            const double ratio = taxor_threshold_ratio(view->kmer_size, cfg.error_rate, cfg.threshold);
            if (cfg.threshold > 0.0 && cfg.threshold <= 1.0) printf("use percentage-model\t%g\n", cfg.threshold);
            else printf("use syncmer model\n");
            if (ratio < 0) die("no syncmer threshold model for k=" + std::to_string(view->kmer_size) + " and error rate " + std::to_string(cfg.error_rate));
            taxor_gpu_search_params prm{ratio, 0, 0, 0};
            std::vector<taxor_gpu_searcher *> sr(ng, nullptr);
            for (size_t g = 0; g < ng; ++g)
                if (taxor_gpu_searcher_create(gidx[g], &prm, &sr[g]) != TAXOR_OK) die(taxor_gpu_last_error());

            // Overlapped stages (the reference joins its workers after every 1024 reads, do_parallel.hpp:31-32):
            //   reader thread     : FASTA/FASTQ(.gz) -> numbered chunks of records   (taxor_search.cpp:315-321)
            //   one thread per GPU: chunk -> GPU (streamed upload, kernels, fetch)   (:325)
            //   writer thread     : tuples -> 0.8*max filter -> TSV lines -> file, in chunk order (:266-311)
            // Output stays in input order (the reference's order at --threads 1).
            BoundedQueue<std::unique_ptr<Batch>> q_in(ng + 1), q_out(2 * ng + 2);
            double t_reads_local = 0;
            std::thread reader([&] {
                FastxReader rd;
                if (!rd.open(query)) die("cannot open query file " + query);
                std::string id;
                bool more = true;
                uint64_t seq = 0;
                while (more) {
                    const double t1 = now();
                    auto b = std::make_unique<Batch>();
                    b->seq = seq++;
                    b->offsets.assign(1, 0);
                    b->bases.reserve(std::min<uint64_t>(cfg.batch_bases, 1ull << 30));
                    while (b->ids.size() < cfg.batch_reads && b->bases.size() < cfg.batch_bases && (more = rd.next(id, b->bases))) {
                        b->ids.push_back(id);
                        b->offsets.push_back(b->bases.size());
                    }
                    t_reads_local += now() - t1;
                    if (b->ids.empty()) break;
                    q_in.push(std::move(b));
                }
                q_in.close();
            });
            std::thread writer([&] {
                std::unique_ptr<Batch> b;
                std::map<uint64_t, std::unique_ptr<Batch>> pending; // chunks that arrived ahead of their turn
                uint64_t next_seq = 0;
                std::string text;
                std::vector<char> line(4096);
                while (q_out.pop(b)) {
                    pending.emplace(b->seq, std::move(b));
                    while (!pending.empty() && pending.begin()->first == next_seq) {
                        std::unique_ptr<Batch> cur = std::move(pending.begin()->second);
                        pending.erase(pending.begin());
                        ++next_seq;
                        text.clear();
                        for (size_t r = 0; r < cur->ids.size(); ++r) {
                            const uint64_t lo = cur->read_off[r], n = cur->read_off[r + 1] - lo;
                            const uint64_t rl = cur->offsets[r + 1] - cur->offsets[r];
                            uint64_t need = taxor_format_read(h, cur->ids[r].data(), cur->ids[r].size(), rl, cur->n_hashes[r],
                                                              cur->user_bin.data() + lo, cur->count.data() + lo, n, line.data(), line.size());
                            if (need > line.size()) {
                                line.resize(need + 1024);
                                need = taxor_format_read(h, cur->ids[r].data(), cur->ids[r].size(), rl, cur->n_hashes[r],
                                                         cur->user_bin.data() + lo, cur->count.data() + lo, n, line.data(), line.size());
                            }
                            text.append(line.data(), need);
                        }
                        fwrite(text.data(), 1, text.size(), out);
                    }
                }
            });
            std::vector<std::thread> workers;
            for (size_t g = 0; g < ng; ++g)
                workers.emplace_back([&, g] {
                    std::unique_ptr<Batch> b;
                    while (q_in.pop(b)) {
                        const double t1 = now();
                        taxor_gpu_results res{};
                        if (taxor_gpu_search_batch(sr[g], b->bases.data(), b->offsets.data(), b->ids.size(), &res) != TAXOR_OK) die(taxor_gpu_last_error());
                        b->read_off.assign(res.read_off, res.read_off + res.n_reads + 1);
                        b->user_bin.assign(res.user_bin, res.user_bin + res.n_tuples);
                        b->count.assign(res.count, res.count + res.n_tuples);
                        b->n_hashes.assign(res.n_hashes, res.n_hashes + res.n_reads);
                        {
                            std::lock_guard<std::mutex> lk(stat_mu);
                            t_compute += now() - t1;
                            total_reads += b->ids.size();
                            total_bases += b->bases.size();
                        }
                        q_out.push(std::move(b));
                    }
                });
            for (auto &t : workers) t.join();
            q_out.close();
            reader.join();
            writer.join();
            t_reads += t_reads_local;
            for (size_t g = 0; g < ng; ++g) {
                taxor_gpu_searcher_destroy(sr[g]);
                taxor_gpu_index_destroy(gidx[g]);
            }
            taxor_hixf_free(h);
        }
    }
    fclose(out);
    printf("Index I/O\tReads I/O\tCompute\n%.2f\t%.2f\t%.2f\n", t_index, t_reads, t_compute);   // :328-336
    printf("%llu reads, %llu bases classified\n", (unsigned long long)total_reads, (unsigned long long)total_bases);
    return 0;
}
