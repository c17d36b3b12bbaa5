// search_main.cpp -- `taxor search` on MI355X: same command line, same .hixf index, same per-read TSV as the
// reference's subcommand (src/main/taxor_search.cpp), with the chunk loop replaced by the C ABI
// (include/taxor_gpu.h).  Host work here: argument parsing (:32-80), sanity checks (:97-151), FASTA/FASTQ(.gz)
// reading (:181-184), batching (:315-326) and output (:268-311, :343).
#include "../../include/taxor_gpu_tools.h"
#include "fastx.h"
#include "tuning.h"
#include "ixf_arith.h"
#include "ixf_layout.h"
using taxor::tune_env;
extern char **environ;

#include <sched.h>
#include <spawn.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

using fastx::Batch;

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
    bool pop_for(T &out, double seconds) // like pop, but gives up (false) after `seconds`
    {
        std::unique_lock<std::mutex> lk(m_);
        not_empty_.wait_for(lk, std::chrono::duration<double>(seconds), [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    bool try_pop(T &out) // false if nothing is queued right now
    {
        std::lock_guard<std::mutex> lk(m_);
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    bool done() // closed and drained
    {
        std::lock_guard<std::mutex> lk(m_);
        return closed_ && q_.empty();
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

// Finished batches go back to the producers: their buffers stay mapped and warm, and at most `cap` batches exist at
// a time (get() blocks), which bounds the host memory the pipeline touches -- faulting in and releasing fresh pages
// costs more than parsing into them.
// The bound is in BYTES as well as in buffers: a chunk's sequence buffer is ~128 MB, page-locked, and recycled rather than
// released, so a count sized for the deepest queues (~150 buffers for one device, ~600 for eight) would let a run whose
// parsers outrun the GPUs or the writer pin tens of gigabytes.  Beyond `floor` buffers -- what the stages need to hold one
// each, so that the pipeline can never starve itself -- a new buffer is made only while the bytes of the existing ones stay
// within `byte_budget`, and a buffer that comes back while the pool is over budget is released (unregistered and freed)
// instead of kept.
struct BatchPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::unique_ptr<Batch>> free_;
    size_t cap = 8, made = 0, floor = 8;
    uint64_t byte_budget = ~0ull, bytes = 0;   // bytes: sequence-buffer capacity of every live batch, as of its last return
    uint64_t peak_bytes = 0, trimmed = 0;
    std::unique_ptr<Batch> get()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !free_.empty() || (made < cap && (made < floor || bytes < byte_budget)); });
        if (free_.empty()) { ++made; return std::make_unique<Batch>(); }
        auto b = std::move(free_.back());
        free_.pop_back();
        return b;
    }
    void put(std::unique_ptr<Batch> b)
    {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t now_bytes = b->bases.capacity();
        bytes += now_bytes - b->pool_bytes;
        b->pool_bytes = now_bytes;
        peak_bytes = std::max(peak_bytes, bytes);
        if (bytes > byte_budget && made > floor) {      // over budget: this buffer goes back to the system
            bytes -= now_bytes;
            --made;
            ++trimmed;
            lk.unlock();
            if (b->pinned) { taxor_gpu_host_unregister(b->pinned); b->pinned = nullptr; }
            b.reset();
            cv.notify_one();
            return;
        }
        free_.push_back(std::move(b));
        cv.notify_one();
    }
};

struct Config {                              // taxor_search_configuration.hpp:8-20
    std::string index_file, query_file, report_file;
    double threshold = -1.0, error_rate = 0.04;
    unsigned threads = 0;       // 0 = not given: as many host threads as feed one GPU (see main); the reference's default is ONE worker
    std::vector<int> gpus{0};   // devices that classify batches in parallel, each with its own index replica
    uint64_t batch_reads = 0, batch_bases = 1ull << 30;   // 0 reads: 65536 per batch (sequential reader) or ~128 MB of
                                                           // file per batch (plain file, parsed in parallel)
    std::string expect_file;    // --expect: a TSV the reference wrote for the same reads and index, compared per read
    bool sequential = false;    // --sequential: one reader thread per file, no byte-range cutting (any legal FASTA/FASTQ)
    uint32_t ixf_arith = 0;     // --ixf-arithmetic: the reading of the un-vendored IXF arithmetic the index follows (0 = this library's)
    bool layout_given = false;  // --ixf-layout: how the file stores each IXF's fingerprints (ixf_layout.h); transposed on the device at load
    uint32_t ixf_layout = 0;
    uint64_t group_reads = 0;   // reads per GPU batch, made of queued chunks (0: 131072, or --batch-reads when that is given)
    std::string gather;         // several devices: "rccl" | "host" (taxor_gpu_comm transports) | "none" (independent workers, each
                                // fetching its own results); empty = rccl when the devices are distinct, host when one repeats
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
            "  --threads <1..32>        host threads parsing the query file (plain FASTA/FASTQ; gzip is one stream) and rendering\n"
            "                           the report; default 4-16 by the size of the machine (the reference's default of 1 is its\n"
            "                           classifying thread; here one host thread cannot feed the GPU)\n"
            "  --percentage <0..1>      if set, this threshold is used instead of the syncmer model\n"
            "  --error-rate <0..1>      expected error rate of the reads (default 0.04)\n"
            "  --gpu <id>               device ordinal (default 0)\n"
            "  --gpus <n>               use devices 0..n-1: the index is replicated, batches of reads are sharded\n"
            "  --gpu-list <a,b,..>      explicit device list (a device may be listed twice)\n"
            "  --gather <rccl|host|none> several devices: index broadcast + per-round gather of the results on the first device over\n"
            "                           RCCL/xGMI (default), the same staged through host memory, or independent workers\n"
            "  --batch-reads <n>        reads per parsed chunk (default: about 128 MB of query file, 65536 reads for gzip)\n"
            "  --sequential             read every query file front to back on one thread (FASTQ whose records wrap their lines)\n"
            "  --group-reads <n>        reads per GPU batch, made of whole chunks (default 131072; --batch-reads if that is given)\n"
            "  --ixf-arithmetic <spec>  search an index whose fingerprints follow another reading of the IXF arithmetic than this\n"
            "                           build's (the spec `taxor verify --variants` prints: kh=..,sm=..,rot=..,red=..,fp=..)\n"
            "  --ixf-layout <spec>      search an index whose fingerprint vectors are laid out otherwise than data[row*stride + bin]\n"
            "                           (the spec `taxor verify --variants` / `taxor pin` print, e.g. bin-major,unpadded,segment-major);\n"
            "                           they are transposed on the device while the index is uploaded\n"
            "  --expect <tsv>           compare the output per read with a TSV the reference wrote for the same input (exit 3 if it differs)\n");
}

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// CPU time of the process so far and what the cgroup's CPU quota took away (cpu.max: a container is commonly given fewer CPUs than
// the machine shows -- the pool's GPU box 16 of 256 -- and every thread of the process stops for the rest of a 100-ms period once
// the quota of that period is used up)
struct CpuMark { double user = 0, sys = 0, throttled = 0; uint64_t periods = 0, throttled_periods = 0; };
CpuMark cpu_mark()
{
    CpuMark m;
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    m.user = ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec;
    m.sys = ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r")) {
        char key[64];
        unsigned long long v;
        while (fscanf(f, "%63s %llu", key, &v) == 2) {
            if (!strcmp(key, "nr_periods")) m.periods = v;
            else if (!strcmp(key, "nr_throttled")) m.throttled_periods = v;
            else if (!strcmp(key, "throttled_usec")) m.throttled = v * 1e-6;
        }
        fclose(f);
    }
    return m;
}
// CPUs the container's cgroup allows (cpu.max: quota / period); 0 if there is no quota
double cpu_quota()
{
    double q = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64];
        unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", a, &period) == 2 && strcmp(a, "max") != 0 && period) q = (double)strtoull(a, nullptr, 10) / (double)period;
        fclose(f);
    }
    return q;
}
// CPUs the process may use: the smaller of the affinity mask and the cgroup's quota (0 if unknown)
double cpu_allowance()
{
    const double q = cpu_quota();
    cpu_set_t set;
    CPU_ZERO(&set);
    const double aff = sched_getaffinity(0, sizeof set, &set) == 0 ? (double)CPU_COUNT(&set) : 0.0;
    return q > 0 && aff > 0 ? std::min(q, aff) : std::max(q, aff);
}

// TAXOR_CLI_TRACE=1: wall-clock marks of the pipeline stages on stderr
const double g_t0 = now();
void trace(const char *what)
{
    static const bool on = tune_env("TAXOR_CLI_TRACE") != nullptr;
    if (on) fprintf(stderr, "[trace] %8.3f s  %s\n", now() - g_t0, what);
}

// Numbered batches of records from one query file -> push() (called from several threads, batches may arrive out
// of order; `seq` gives the input order).  A plain file is read and parsed by up to `threads` threads, each taking
// the next byte range of about cfg.batch_reads records; anything else (gzip, pipes) goes through the sequential
// zlib reader, which cuts batches at exactly cfg.batch_reads records.  Returns the wall time spent producing.
double produce_batches(const std::string &query, const Config &cfg, bool allow_ranges, BatchPool &pool,
                       const std::function<void(std::unique_ptr<Batch>)> &push_, uint32_t file = 0, unsigned parse_threads = 0,
                       const std::function<void()> &gate = nullptr)
{
    std::atomic<uint64_t> n_pushed{0};
    auto push = [&](std::unique_ptr<Batch> b) {
        b->file = file;
        b->end_of_file = false;
        ++n_pushed;
        push_(std::move(b));
    };
    auto finish = [&] {      // tells the consumer how many chunks this file had
        auto m = std::make_unique<Batch>();
        m->file = file;
        m->seq = n_pushed.load();
        m->end_of_file = true;
        m->offsets.assign(1, 0);
        push_(std::move(m));
    };
    const double t_begin = now();
    double blocked = 0;   // time the sequential reader spent waiting for the consumers
    fastx::RangedFastx rf;
    bool ranged = false;
    try {
        ranged = allow_ranges && rf.open(query);
    } catch (const std::exception &e) { die(e.what()); }
    if (ranged) {
        if (cfg.batch_reads) rf.plan(cfg.batch_reads, std::min<uint64_t>(cfg.batch_bases, 1ull << 30));
        else rf.range_bytes = rf.kind == '@' ? (128u << 20) : (64u << 20);
        const unsigned nt = parse_threads ? parse_threads : std::max(1u, std::min(cfg.threads, 16u));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&] {
                fastx::FastxReader rd;
                std::string id;
                uint64_t b, e, seq;
                size_t prev_records = 0, prev_id_bytes = 0;
                for (;;) {
                    if (gate) gate();
                    auto bt = pool.get();                   // buffer first: ranges are then taken in the order they can be filled
                    try {
                        if (!rf.next_range(b, e, seq)) { pool.put(std::move(bt)); break; }
                        bt->seq = seq;
                        bt->may_pin = true;
                        ++bt->fills;
                        const size_t need = fastx::bases_bound(e - b, rf.kind);
                        if (need > bt->bases.capacity()) {  // grow before parsing, and only with the old block unpinned
                            if (bt->pinned) { taxor_gpu_host_unregister(bt->pinned); bt->pinned = nullptr; }
                            bt->bases.clear();
                            bt->bases.reserve((need + need / 16 + (2u << 20)) & ~size_t((2u << 20) - 1));
                            {   // a fresh 128-MB buffer is 32768 page faults when it is first written -- for a query file smaller than the
                                // pool, every buffer is fresh; as huge pages it is 64 (where the system allows them on request)
                                const uintptr_t a = ((uintptr_t)&bt->bases[0] + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
                                const uintptr_t e = ((uintptr_t)&bt->bases[0] + bt->bases.capacity()) & ~(uintptr_t)((2u << 20) - 1);
                                if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
                            }
                        }
                        bt->ids.clear();
                        bt->bases.clear();
                        bt->offsets.assign(1, 0);
                        {   // a fresh chunk buffer: size the per-record arrays once, from what the previous range of this thread
                            // held (growing a vector of 10^5 entries by doubling is a dozen reallocations, each an mmap/munmap
                            // and a TLB shootdown across every thread of the process)
                            const size_t guess = prev_records + prev_records / 8 + 1024;
                            if (bt->offsets.capacity() < guess) bt->offsets.reserve(guess + 1);
                            if (bt->ids.off.capacity() < guess || bt->ids.data.capacity() < prev_id_bytes) bt->ids.reserve(guess, prev_id_bytes + prev_id_bytes / 8 + 4096);
                        }
                        rd.open_range(rf.fd, b, e, rf.kind == '@');
                        while (rd.next(id, bt->bases)) {
                            bt->ids.push_back(id);
                            bt->offsets.push_back(bt->bases.size());
                        }
                        prev_records = bt->ids.size();
                        prev_id_bytes = bt->ids.data.size();
                    } catch (const std::exception &ex) { die(ex.what()); }
                    push(std::move(bt));
                }
            });
        for (auto &t : th) t.join();
        finish();
        return now() - t_begin;
    }
    fastx::FastxReader rd;
    fastx::GzMembers members;
    fastx::ParallelGz pgz;
    bool multi = false, single = false;
    // (inflating threads: up to 32 of --threads -- or what the container's CPU quota allows: beyond it the cgroup only stops them all)
    static const unsigned gz_cap = [] { const double q = cpu_quota(); return q >= 1.0 ? std::min(32u, std::max(2u, (unsigned)q)) : 32u; }();
    const unsigned gz_threads = parse_threads ? parse_threads : std::max(1u, std::min(cfg.threads, gz_cap));
    try {
        multi = allow_ranges && members.open(query, gz_threads);
        // one member (what `gzip reads.fastq` writes): inflated speculatively from the middle on all of this file's threads (pgz.h);
        // --sequential and small files keep the one zlib stream
        if (!multi && allow_ranges && gz_threads > 1) {
            single = pgz.open(query, gz_threads);
        }
    } catch (const std::exception &ex) { die(ex.what()); }
    if (single || multi) {
        // (a multi-member file -- bgzip's 64-KB members, concatenated runs -- goes the same way, its members being the chunks)
        auto take_chunk = [&](std::vector<char> &v) { return single ? pgz.take(v) : members.take(v); };
        // The inflated stream arrives in chunks of ~16 MB that change hands without a copy (ParallelGz::take).  This thread only CUTS:
        // it collects chunks up to ~64 MB, finds the last record start in the last one (the range reader's test: an '@' line whose
        // second successor starts with '+', or a '>' line), and hands everything before it -- whole records -- to a parser thread;
        // what follows the cut is carried into the next segment.  The parsers work like those of a plain file's byte ranges, the
        // buffers are their input directly.  A file whose first records are not of the kind the cut can recognise (FASTQ that wraps
        // its lines) is parsed front to back on this thread instead.
        std::vector<char> first;
        bool have = false;
        try {
            while ((have = take_chunk(first)) && first.empty()) {}         // (empty members are legal)
        } catch (const std::exception &ex) { die(ex.what()); }
        size_t i0 = 0;
        while (i0 < first.size() && (first[i0] == '\n' || first[i0] == '\r')) ++i0;
        char kind = have && i0 < first.size() ? first[i0] : 0;
        bool cuttable = kind == '>' || kind == '@';
        if (kind == '@') {
            const char *p = first.data() + i0, *e = first.data() + std::min<size_t>(first.size(), i0 + (1u << 16));
            for (int rec = 0; rec < 16 && p < e && cuttable; ++rec) {
                const char *l3 = fastx::next_line(fastx::next_line(p, e), e);
                if (l3 >= e) break;
                if (*l3 != '+') cuttable = false;
                const char *nx = fastx::next_line(fastx::next_line(l3, e), e);
                if (nx < e && *nx != '@' && *nx != '\n' && *nx != '\r') cuttable = false;
                p = nx;
            }
        }
        if (have && kind && !cuttable && kind != '@' && kind != '>') die("query file is neither FASTA nor FASTQ");
        if (!have || !cuttable) {
            if (single) rd.pgz = &pgz;                       // front to back: the chunk already taken first, then the stream
            else rd.members = &members;
            rd.buf.swap(first);
            rd.pos = 0;
            rd.len = rd.buf.size();
            if (rd.buf.size() < (8u << 20)) rd.buf.resize(8u << 20);
            if (single) pgz.switch_to_read();
        } else {
            struct Job { std::deque<std::vector<char>> parts; uint64_t seq = 0; size_t bytes = 0; };
            std::mutex jmu;
            std::condition_variable jcv_put, jcv_get;
            std::deque<Job> jobs;
            bool jdone = false;
            static const unsigned np_env = [] { const char *e = tune_env("TAXOR_CLI_GZ_PARSERS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 64 ? (unsigned)v : 0u; }();
            const unsigned np = np_env ? np_env : std::max(2u, gz_threads / 4);
            std::vector<std::thread> parsers;
            for (unsigned t = 0; t < np; ++t)
                parsers.emplace_back([&] {
                    fastx::FastxReader prd;
                    if (single) prd.mem_recycle = [&pgz](std::vector<char> &&v) { pgz.recycle(std::move(v)); };
                    std::string pid;
                    size_t prev_records = 0, prev_id_bytes = 0;
                    for (;;) {
                        Job job;
                        {
                            std::unique_lock<std::mutex> lk(jmu);
                            jcv_get.wait(lk, [&] { return !jobs.empty() || jdone; });
                            if (jobs.empty()) return;
                            job = std::move(jobs.front());
                            jobs.pop_front();
                            jcv_put.notify_one();
                        }
                        if (gate) gate();
                        auto bt = pool.get();
                        try {
                            bt->seq = job.seq;
                            bt->may_pin = true;
                            ++bt->fills;
                            const size_t need = fastx::bases_bound(job.bytes, kind);
                            if (need > bt->bases.capacity()) {
                                if (bt->pinned) { taxor_gpu_host_unregister(bt->pinned); bt->pinned = nullptr; }
                                bt->bases.clear();
                                bt->bases.reserve((need + need / 16 + (2u << 20)) & ~size_t((2u << 20) - 1));
                                const uintptr_t a = ((uintptr_t)&bt->bases[0] + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1);
                                const uintptr_t e = ((uintptr_t)&bt->bases[0] + bt->bases.capacity()) & ~(uintptr_t)((2u << 20) - 1);
                                if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
                            }
                            bt->ids.clear();
                            bt->bases.clear();
                            bt->offsets.assign(1, 0);
                            const size_t guess = prev_records + prev_records / 8 + 1024;
                            if (bt->offsets.capacity() < guess) bt->offsets.reserve(guess + 1);
                            if (bt->ids.off.capacity() < guess || bt->ids.data.capacity() < prev_id_bytes) bt->ids.reserve(guess, prev_id_bytes + prev_id_bytes / 8 + 4096);
                            prd.open_mem(std::move(job.parts), kind == '@');
                            while (prd.next(pid, bt->bases)) {
                                bt->ids.push_back(pid);
                                bt->offsets.push_back(bt->bases.size());
                            }
                            prev_records = bt->ids.size();
                            prev_id_bytes = bt->ids.data.size();
                        } catch (const std::exception &ex) { die(ex.what()); }
                        push(std::move(bt));
                    }
                });
            const size_t target = (size_t)std::min<uint64_t>(cfg.batch_bases, kind == '@' ? (64u << 20) : (32u << 20));     // bytes of text per parser job
            Job cur;
            uint64_t seq = 0;
            auto submit = [&](Job &&j) {
                if (j.bytes == 0) return;
                j.seq = seq++;
                std::unique_lock<std::mutex> lk(jmu);
                jcv_put.wait(lk, [&] { return jobs.size() < np + 2; });
                jobs.push_back(std::move(j));
                jcv_get.notify_one();
            };
            auto add = [&](std::vector<char> &&v) { cur.bytes += v.size(); cur.parts.push_back(std::move(v)); };
            try {
                if (i0) first.erase(first.begin(), first.begin() + (long)i0);
                add(std::move(first));
                for (;;) {
                    if (cur.bytes >= target) {
                        // the last record start in the last part, looked for in its final megabyte (a record is tens of kilobytes)
                        std::vector<char> &last = cur.parts.back();
                        const char *b = last.data(), *e = b + last.size();
                        const char *w = last.size() > (1u << 20) ? e - (1u << 20) : b;
                        // (a part starts at a line start only if it is the job's first: jobs begin at record starts; a bgzip member begins anywhere)
                        const char *p = w == b && cur.parts.size() == 1 ? b : fastx::next_line(w, e);
                        const char *cut = nullptr;
                        for (const char *r = fastx::resync(p, e, kind); r < e; r = fastx::resync(fastx::next_line(r, e), e, kind)) cut = r;
                        if (cut && (cut > b || cur.parts.size() > 1)) {
                            std::vector<char> carry(cut, e);
                            cur.bytes -= (size_t)(e - cut);
                            last.resize((size_t)(cut - b));
                            Job next;
                            next.bytes = carry.size();
                            next.parts.push_back(std::move(carry));
                            submit(std::move(cur));
                            cur = std::move(next);
                        }
                    }
                    std::vector<char> ck;
                    if (!take_chunk(ck)) break;
                    if (!ck.empty()) add(std::move(ck));
                }
                submit(std::move(cur));
            } catch (const std::exception &ex) { die(ex.what()); }
            {
                std::lock_guard<std::mutex> lk(jmu);
                jdone = true;
            }
            jcv_get.notify_all();
            for (auto &t : parsers) t.join();
            finish();
            return now() - t_begin;
        }
    } else if (!rd.open(query)) die("cannot open query file " + query);
    std::string id;
    bool more = true;
    uint64_t seq = 0;
    const uint64_t batch_reads = cfg.batch_reads ? cfg.batch_reads : 65536;
    try {
        while (more) {
            if (gate) gate();
            auto b = pool.get();
            // a batch recycled from the ranged reader of an earlier plain file may still be page-locked; this reader
            // appends without a size bound, so the string can reallocate: give the registration up first (a stale
            // registration over freed memory would be a DMA source) and do not pin batches of this path again
            if (b->pinned) { taxor_gpu_host_unregister(b->pinned); b->pinned = nullptr; }
            b->may_pin = false;
            b->seq = seq++;
            b->ids.clear();
            b->bases.clear();
            b->offsets.assign(1, 0);
            while (b->ids.size() < batch_reads && b->bases.size() < cfg.batch_bases && (more = rd.next(id, b->bases))) {
                b->ids.push_back(id);
                b->offsets.push_back(b->bases.size());
            }
            if (b->ids.empty()) { pool.put(std::move(b)); break; }
            const double t1 = now();
            push(std::move(b));
            blocked += now() - t1;
        }
    } catch (const std::exception &ex) { die(ex.what()); }
    finish();
    return now() - t_begin - blocked;
}

// seqan3's FASTA/FASTQ readers drop white space and digits inside sequences (numbered or column-formatted files).  The
// parsers here append sequence lines raw -- a scan per byte would halve their rate for files that never need it -- and
// the device reports any character outside dna15 (TAXOR_E_ALPHABET); only then is the batch cleaned, in place, and run
// again.  Returns false if there was nothing to remove (the batch really holds a foreign character).
bool strip_space_and_digits(Batch &b)
{
    char *d = &b.bases[0];
    size_t w = 0;
    bool any = false;
    for (size_t r = 0; r + 1 < b.offsets.size(); ++r) {
        const size_t lo = b.offsets[r], hi = b.offsets[r + 1];
        b.offsets[r] = w;
        for (size_t i = lo; i < hi; ++i) {
            const unsigned char c = (unsigned char)d[i];
            if (c == ' ' || (c >= '\t' && c <= '\r') || (c >= '0' && c <= '9')) { any = true; continue; }
            d[w++] = (char)c;
        }
    }
    b.offsets.back() = w;
    b.bases.resize(w);          // shrinks: never reallocates, a page-locked buffer stays where it is
    return any;
}

// `--expect ref.tsv`: compare this run's output with a TSV the reference produced for the same reads and index
// (the day a published .hixf and its reference output are at hand).  The reference writes the lines
// of one read together but, with --threads > 1, the reads in no particular order (sync_out.hpp:24-29): reads are
// matched by id, the lines of a read are compared in order (the HIXF's DFS order).  Returns the number of differing reads.
uint64_t compare_tsv(const std::string &ours, const std::string &expect)
{
    auto load = [](const std::string &path, std::map<std::string, std::vector<std::string>> &by_read, std::vector<std::string> &order) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) die("cannot open " + path);
        std::string line;
        std::vector<char> buf(1 << 20);
        std::string carry;
        size_t n;
        auto take = [&](const std::string &l) {
            if (l.empty() || l[0] == '#') return;
            const size_t tab = l.find('\t');
            const std::string id = l.substr(0, tab);
            auto it = by_read.find(id);
            if (it == by_read.end()) { it = by_read.emplace(id, std::vector<std::string>()).first; order.push_back(id); }
            it->second.push_back(l);
        };
        while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) {
            size_t a = 0;
            for (size_t i = 0; i < n; ++i)
                if (buf[i] == '\n') {
                    carry.append(buf.data() + a, i - a);
                    if (!carry.empty() && carry.back() == '\r') carry.pop_back();
                    take(carry);
                    carry.clear();
                    a = i + 1;
                }
            carry.append(buf.data() + a, n - a);
        }
        if (!carry.empty()) take(carry);
        fclose(f);
    };
    std::map<std::string, std::vector<std::string>> a, b;
    std::vector<std::string> oa, ob;
    load(ours, a, oa);
    load(expect, b, ob);
    uint64_t same = 0, differ = 0, only_ours = 0, only_expect = 0, shown = 0;
    for (const auto &id : oa) {
        const auto it = b.find(id);
        if (it == b.end()) { ++only_ours; continue; }
        if (it->second == a[id]) { ++same; continue; }
        ++differ;
        if (shown++ < 10) {
            printf("read %s differs:\n", id.c_str());
            for (const auto &l : a[id]) printf("  ours    %s\n", l.c_str());
            for (const auto &l : it->second) printf("  expect  %s\n", l.c_str());
        }
    }
    for (const auto &id : ob)
        if (!a.count(id)) ++only_expect;
    printf("compared with %s: %llu reads identical, %llu differ, %llu only in this run, %llu only in the expected file\n", expect.c_str(),
           (unsigned long long)same, (unsigned long long)differ, (unsigned long long)only_ours, (unsigned long long)only_expect);
    const uint64_t bad = differ + only_ours + only_expect;
    printf("%s\n", bad == 0 ? "PASS: per-read output identical to the expected TSV"
                            : "FAIL: output differs from the expected TSV (run `taxor verify` to test the index arithmetic)");
    return bad;
}

// "kh=0,sm=1,rot=21,red=1,fp=1" -> arithmetic code (taxor_ixf_arith_code); fields left out keep this library's reading
bool parse_arith_spec(const std::string &spec, uint32_t *code)
{
    taxor_ixf_variant v;
    taxor_ixf_variant_default(&v, 0, 1, 64);
    for (const auto &tok : str_split(spec, ',')) {
        const size_t eq = tok.find('=');
        if (eq == std::string::npos) return false;
        const std::string key = tok.substr(0, eq);
        char *end = nullptr;
        const long x = strtol(tok.c_str() + eq + 1, &end, 10);
        if (!end || *end != '\0' || x < 0) return false;
        if (key == "kh" && x <= 3) v.key_hash = (uint8_t)x;
        else if (key == "sm" && x <= 3) v.seed_mode = (uint8_t)x;
        else if (key == "rot" && x >= 1 && x <= 63) v.rot = (uint8_t)x;
        else if (key == "red" && x <= 2) v.reduce = (uint8_t)x;
        else if (key == "fp" && x <= 3) v.fp_mode = (uint8_t)x;
        else return false;
    }
    *code = taxor_ixf_arith_code(&v);
    return true;
}

std::string arith_spec(const taxor_ixf_variant &v)
{
    return "kh=" + std::to_string(v.key_hash) + ",sm=" + std::to_string(v.seed_mode) + ",rot=" + std::to_string(v.rot) + ",red=" +
           std::to_string(v.reduce) + ",fp=" + std::to_string(v.fp_mode);
}

// the readings of the un-vendored IXF that `taxor verify --variants` and `taxor pin` probe a root IXF's RAW bytes under: seed (the
// file's, the prototype's fixed start seed, none) x shape (layout kind x pitch x row order, each with the segment length the
// array length implies and the one the loader took from the file) x key hash x seed entry x rotation step x range reduction x
// fingerprint fold.  `loaded_layout` = the code the loader settled on (its pitch rule says whether root.src_stride is a stored
// scalar).
std::vector<taxor_ixf_variant> variant_family(const taxor_ixf_view &root, uint64_t raw_len, uint32_t loaded_layout)
{
    std::vector<uint64_t> seeds{root.seed};
    for (uint64_t sd : {13572355802537770549ull, 0ull})
        if (std::find(seeds.begin(), seeds.end(), sd) == seeds.end()) seeds.push_back(sd);
    const uint64_t S = (root.bins + 63) / 64 * 64;
    struct Shape { uint32_t layout; uint64_t pitch, seg; };
    std::vector<Shape> shapes;
    auto add_shape = [&](uint32_t layout, uint64_t pitch) {
        if (!pitch || pitch < root.bins || raw_len % pitch != 0) return;
        for (uint64_t sg : {raw_len / pitch / 3, root.seg_len}) {
            if (!sg || 3 * sg * pitch > raw_len) continue;
            bool dup = false;
            for (const Shape &sh : shapes) dup = dup || ((sh.layout & ~taxor::IXF_PITCH_MASK) == (layout & ~taxor::IXF_PITCH_MASK) && sh.pitch == pitch && sh.seg == sg);
            if (!dup) shapes.push_back({layout, pitch, sg});
        }
    };
    // the pitch rule the loader settled on goes first: it fits EVERY IXF's array length (the loader checked), and where two rules
    // give the root the same pitch (a root of 64 k bins: padded == unpadded) the first one added is the one the shape keeps
    const uint32_t loaded_rule = loaded_layout & taxor::IXF_PITCH_MASK;
    auto pitch_of = [&](uint32_t rule) { return rule == taxor::IXF_PITCH_BINS ? root.bins : rule == taxor::IXF_PITCH_STORED ? (root.src_stride ? root.src_stride : root.stride) : S; };
    for (uint32_t pm : {0u, (uint32_t)taxor::IXF_ROWS_POSITION_MAJOR}) {
        for (uint32_t kind : {(uint32_t)taxor::IXF_KIND_ROWS, (uint32_t)taxor::IXF_KIND_BIN_MAJOR}) {
            add_shape(kind | pm | loaded_rule, pitch_of(loaded_rule));
            for (uint32_t rule : {(uint32_t)taxor::IXF_PITCH_PADDED, (uint32_t)taxor::IXF_PITCH_BINS})
                if (rule != loaded_rule) add_shape(kind | pm | rule, pitch_of(rule));
        }
        add_shape(taxor::IXF_KIND_BIT_SLICED | pm, S);
    }
    std::vector<taxor_ixf_variant> vs;
    for (uint64_t sd : seeds)
        for (const Shape &sh : shapes)
            for (int kh = 0; kh < 4; ++kh)
                for (int sm = 0; sm < 3; ++sm)
                    for (int rot : {21, 16, 32})
                        for (int red = 0; red < 3; ++red)
                            for (int fp = 0; fp < 4; ++fp) {
                                if (sd == 0 && sm != 0) continue;        // without a seed the three seed modes coincide
                                taxor_ixf_variant v;
                                taxor_ixf_variant_default(&v, sd, sh.seg, sh.pitch);
                                v.key_hash = (uint8_t)kh; v.seed_mode = (uint8_t)sm; v.rot = (uint8_t)rot;
                                v.reduce = (uint8_t)red; v.fp_mode = (uint8_t)fp; v.layout = (uint16_t)sh.layout;
                                vs.push_back(v);
                            }
    return vs;
}

// hash lists for a variant scan: at most `cap` hashes per list (the score is a ratio; the scan's cost is lists x variants x bins x hashes)
void cap_hash_lists(const uint64_t *hoff, const uint64_t *hs, uint64_t n_lists, uint64_t cap, std::vector<uint64_t> &off_out, std::vector<uint64_t> &hs_out)
{
    off_out.assign(1, 0);
    hs_out.clear();
    for (uint64_t l = 0; l < n_lists; ++l) {
        const uint64_t n = std::min<uint64_t>(cap, hoff[l + 1] - hoff[l]);
        hs_out.insert(hs_out.end(), hs + hoff[l], hs + hoff[l] + n);
        off_out.push_back(hs_out.size());
    }
}

// this library's own reading of the root as the loader took it: arithmetic code 0 in the search layout
bool variant_is_native(const taxor_ixf_variant &v, const taxor_ixf_view &root)
{
    return taxor_ixf_arith_code(&v) == 0 && taxor::ixf_layout_kind(v.layout) == taxor::IXF_KIND_ROWS && !(v.layout & taxor::IXF_ROWS_POSITION_MAJOR) &&
           v.stride == root.stride && v.seg_len == root.seg_len && v.seed == root.seed;
}

// (median best-bin match ratio over the hash lists, variant index), best first
std::vector<std::pair<float, size_t>> rank_variants(const std::vector<taxor_ixf_variant> &vs, const std::vector<float> &ratio, size_t n_lists)
{
    std::vector<std::pair<float, size_t>> rank;
    for (size_t i = 0; i < vs.size(); ++i) {
        std::vector<float> r(ratio.begin() + i * n_lists, ratio.begin() + (i + 1) * n_lists);
        std::sort(r.begin(), r.end());
        rank.push_back({r[r.size() / 2], i});
    }
    std::sort(rank.begin(), rank.end(), [](const auto &a, const auto &b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
    return rank;
}

uint64_t fnv1a(const char *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
    return h;
}

#include "pin_cmd.h"

} // namespace

int main(int argc, char **argv)
{
    // while the process is still single-threaded: the HIP runtime's hardware-queue count (taxor_amd/csrc/api.hip,
    // runtime_env_once -- the library would set it at its first call, by which time this host has threads that read the
    // environment)
    // A process that will also hold an RCCL communicator (--gpus a,b,... or --gather rccl) asks for 16: RCCL's own streams otherwise
    // push the searchers' streams onto shared queues (21.5 -> 23.7 Gbp/s in the one-device experiment, docs/EXPERIMENTS.md section 4;
    // bench.py's ranks: profiles/r06/hw_queues_dist.txt).  A value the user exported wins.
    bool with_comm = false;
    for (int i = 1; i + 1 < argc; ++i)
        if ((strcmp(argv[i], "--gpus") == 0 && strchr(argv[i + 1], ',')) || (strcmp(argv[i], "--gather") == 0 && strcmp(argv[i + 1], "rccl") == 0)) with_comm = true;
    setenv("GPU_MAX_HW_QUEUES", with_comm ? "16" : "8", 0);
    if (const char *e = tune_env("TAXOR_CLI_NT")) fastx::stream_stores() = atoi(e) != 0;
    if (const char *e = tune_env("TAXOR_CLI_RDBUF_KB")) fastx::range_buffer_bytes() = (size_t)std::max(64, atoi(e)) << 10;
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
    if (argc > 1 && strcmp(argv[1], "verify") == 0) {
        // Positive control for an index whose provenance is not this library: error-free windows cut
        // from a genome that IS in the index must find (nearly) all of their hashes in one user bin.  A match ratio near
        // the false-positive floor instead means that the hash (wyhash / minimiser) or the IXF arithmetic (seed, row
        // stride, segment length) read from the file does not describe this index -- the two un-vendored boundaries.
        std::string index_file, genome_file;
        uint64_t n_reads = 2000, read_len = 5000;
        int device = 0;
        bool scan_variants = false;
        uint32_t verify_arith = 0, verify_layout = 0;
        bool layout_given = false;
        for (int i = 2; i < argc; ++i) {
            if (strcmp(argv[i], "--variants") == 0) { scan_variants = true; continue; }
            if (strcmp(argv[i], "--index-file") == 0 && i + 1 < argc) index_file = argv[++i];
            else if (strcmp(argv[i], "--genome-file") == 0 && i + 1 < argc) genome_file = argv[++i];
            else if (strcmp(argv[i], "--reads") == 0 && i + 1 < argc) n_reads = strtoull(argv[++i], nullptr, 10);
            else if (strcmp(argv[i], "--read-len") == 0 && i + 1 < argc) read_len = strtoull(argv[++i], nullptr, 10);
            else if (strcmp(argv[i], "--gpu") == 0 && i + 1 < argc) device = atoi(argv[++i]);
            else if (strcmp(argv[i], "--ixf-arithmetic") == 0 && i + 1 < argc) {
                if (!parse_arith_spec(argv[++i], &verify_arith)) die("--ixf-arithmetic: expected kh=..,sm=..,rot=..,red=..,fp=..");
            }
            else if (strcmp(argv[i], "--ixf-layout") == 0 && i + 1 < argc) {
                if (taxor_ixf_layout_parse(argv[++i], &verify_layout) != TAXOR_OK) die(taxor_gpu_last_error());
                layout_given = true;
            }
        }
        if (index_file.empty() || genome_file.empty() || !file_exists(index_file) || !file_exists(genome_file) || !n_reads || !read_len)
            die("usage: taxor verify --index-file <x.hixf> --genome-file <fasta of a genome contained in the index> [--reads n] [--read-len l]");
        taxor_hixf *h = nullptr;
        if (taxor_hixf_load(index_file.c_str(), &h) != TAXOR_OK) die(taxor_gpu_last_error());
        taxor_hixf_set_arith(h, verify_arith);
        if (layout_given && taxor_hixf_set_layout(h, verify_layout) != TAXOR_OK) die(taxor_gpu_last_error());
        const taxor_hixf_view *view = taxor_hixf_get_view(h);
        taxor_gpu_index *gi = nullptr;
        if (taxor_gpu_index_create(view, device, &gi) != TAXOR_OK) die(taxor_gpu_last_error());
        taxor_gpu_search_params prm{};
        prm.ratio = 0.05;                                  // report every bin holding >= 5 % of a window's hashes
        prm.model = TAXOR_THR_PERCENTAGE;
        taxor_gpu_searcher *sr = nullptr;
        if (taxor_gpu_searcher_create(gi, &prm, &sr) != TAXOR_OK) die(taxor_gpu_last_error());
        // windows: evenly spaced over the concatenated records, never across a record boundary
        std::string genome, id, bases;
        std::vector<uint64_t> rec_off{0};
        try {
            fastx::FastxReader rd;
            if (!rd.open(genome_file)) die("cannot open " + genome_file);
            while (rd.next(id, genome)) rec_off.push_back(genome.size());
        } catch (const std::exception &e) { die(e.what()); }
        std::vector<uint64_t> offsets{0};
        for (size_t r = 0; r + 1 < rec_off.size(); ++r) {
            const uint64_t len = rec_off[r + 1] - rec_off[r];
            if (len < read_len) continue;
            const uint64_t want = std::max<uint64_t>(1, n_reads * len / std::max<uint64_t>(1, genome.size()));
            const uint64_t step = std::max<uint64_t>(1, (len - read_len) / want);
            for (uint64_t p = 0; p + read_len <= len && offsets.size() <= n_reads; p += step) {
                bases.append(genome, rec_off[r] + p, read_len);
                offsets.push_back(bases.size());
            }
        }
        if (offsets.size() < 2) die("no record of the genome file is as long as --read-len");
        taxor_gpu_results res{};
        if (taxor_gpu_search_batch(sr, bases.data(), offsets.data(), offsets.size() - 1, &res) != TAXOR_OK) die(taxor_gpu_last_error());
        std::vector<double> best;
        std::map<int64_t, uint64_t> votes;
        uint64_t hashes = 0;
        for (uint64_t r = 0; r < res.n_reads; ++r) {
            uint32_t top = 0;
            int64_t top_bin = -1;
            for (uint64_t i = res.read_off[r]; i < res.read_off[r + 1]; ++i)
                if (res.count[i] > top) { top = res.count[i]; top_bin = res.user_bin[i]; }
            if (res.n_hashes[r]) best.push_back((double)top / (double)res.n_hashes[r]);
            if (top_bin >= 0) ++votes[top_bin];
            hashes += res.n_hashes[r];
        }
        std::sort(best.begin(), best.end());
        const double median = best.empty() ? 0.0 : best[best.size() / 2], low = best.empty() ? 0.0 : best[best.size() / 20];
        int64_t winner = -1;
        uint64_t winner_votes = 0;
        for (const auto &kv : votes)
            if (kv.second > winner_votes) { winner = kv.first; winner_votes = kv.second; }
        printf("windows %llu x %llu bp, %.1f hashes per window (%s, k=%u)\n", (unsigned long long)res.n_reads, (unsigned long long)read_len,
               res.n_reads ? (double)hashes / (double)res.n_reads : 0.0, view->use_syncmer ? "open syncmers" : "minimisers", (unsigned)view->kmer_size);
        printf("best-bin match ratio: median %.4f, 5th percentile %.4f (false-positive floor %.4f)\n", median, low, 1.0 / 256.0);
        if (winner >= 0) printf("most frequent best user bin: %lld (%llu of %llu windows)\n", (long long)winner, (unsigned long long)winner_votes, (unsigned long long)res.n_reads);
        const bool pass = median >= 0.9 && low >= 0.5;
        printf("%s\n", pass ? "PASS: the index answers for this genome -- hashing and IXF arithmetic match the file"
                            : "FAIL: an indexed genome must match itself; check `taxor probe` (seed / stride / segment length) and the hash");
        if (scan_variants || !pass) {
            // Which reading of the un-vendored IXF arithmetic does this file follow?  Probe the raw bytes of the root IXF
            // (every indexed genome is in one of its bins, as a leaf or inside a merged bin) under a family of variants.
            const uint64_t n_lists = std::min<uint64_t>(24, offsets.size() - 1);
            const uint64_t *hoff = nullptr, *hs = nullptr;
            if (taxor_gpu_syncmers(sr, bases.data(), offsets.data(), n_lists, &hoff, &hs) != TAXOR_OK) die(taxor_gpu_last_error());
            const taxor_ixf_view &root = view->ixf[0];
            const uint64_t raw_len = taxor_hixf_ixf_raw_bytes(h, 0);
            std::vector<taxor_ixf_variant> vs = variant_family(root, raw_len, view->ixf_layout);
            std::vector<uint64_t> s_off, s_hs;
            cap_hash_lists(hoff, hs, n_lists, 160, s_off, s_hs);
            // the scan copies the root's RAW bytes to the device in one piece (tens of GB under a 4096-bin root): the resident index
            // has answered everything it was needed for, so it goes first instead of sitting beside that copy
            taxor_gpu_searcher_destroy(sr);
            taxor_gpu_index_destroy(gi);
            sr = nullptr;
            gi = nullptr;
            std::vector<float> ratio(vs.size() * n_lists);
            if (taxor_gpu_ixf_variant_scan(device, root.data, raw_len, root.bins, vs.data(), (uint32_t)vs.size(), s_hs.data(), s_off.data(), n_lists, ratio.data()) != TAXOR_OK)
                die(taxor_gpu_last_error());
            const std::vector<std::pair<float, size_t>> rank = rank_variants(vs, ratio, (size_t)n_lists);
            printf("variant scan of the root IXF (%zu readings of its %llu raw fingerprint bytes, %llu hash lists): median best-bin match ratio\n", vs.size(),
                   (unsigned long long)raw_len, (unsigned long long)n_lists);
            char desc[512];
            for (size_t i = 0; i < std::min<size_t>(5, rank.size()); ++i) {
                const taxor_ixf_variant &v = vs[rank[i].second];
                taxor_ixf_variant_describe(&v, desc, sizeof desc);
                printf("  %.4f  %s%s\n", rank[i].first, desc, variant_is_native(v, root) ? "   [this library's reading, ixf_arith.h, in the search layout]" : "");
            }
            if (!rank.empty() && rank[0].first >= 0.9f) {
                const taxor_ixf_variant &bv = vs[rank[0].second];
                const bool is_mine = variant_is_native(bv, root);
                printf("%s\n", is_mine ? "the file follows this library's reading of the IXF arithmetic, in the search layout"
                                       : "the file follows ANOTHER reading of the IXF (first line)");
                if (!is_mine) {
                    char ld[128];
                    taxor_ixf_layout_describe(bv.layout, ld, sizeof ld);
                    if (bv.seed != root.seed)
                        printf("that reading also uses another seed than the loader took from the file: `taxor probe` shows the record layout\n");
                    else
                    {
                        const bool as_it_lies = taxor::ixf_layout_kind(bv.layout) == taxor::IXF_KIND_ROWS && !(bv.layout & taxor::IXF_ROWS_POSITION_MAJOR) && bv.stride == root.stride &&
                                                bv.seg_len == root.seg_len;
                        printf("search it with: taxor search%s%s%s%s ...\n", taxor_ixf_arith_code(&bv) ? " --ixf-arithmetic " : "",
                               taxor_ixf_arith_code(&bv) ? arith_spec(bv).c_str() : "", as_it_lies ? "" : " --ixf-layout ", as_it_lies ? "" : ld);
                    }
                }
            } else {
                printf("no variant answers: the key hash (wyhash / minimiser value) or the genome is not what the index holds\n");
            }
        }
        if (sr) taxor_gpu_searcher_destroy(sr);
        if (gi) taxor_gpu_index_destroy(gi);
        taxor_hixf_free(h);
        return pass ? 0 : 2;
    }
    if (argc > 1 && strcmp(argv[1], "inflate") == 0) {
        // the parallel gzip reader alone: `taxor inflate --query-file x.fastq.gz [--threads n] [--chunk-mb m] [--output-file out]`
        // decompresses (to a file, or nowhere) and reports the rate, the output's CRC-32 and how many chunks had to be decoded twice
        std::string in, out_path;
        unsigned threads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        size_t chunk = 0;
        for (int i = 2; i < argc; ++i) {
            if (strcmp(argv[i], "--query-file") == 0 && i + 1 < argc) in = argv[++i];
            else if (strcmp(argv[i], "--threads") == 0 && i + 1 < argc) threads = (unsigned)atoi(argv[++i]);
            else if (strcmp(argv[i], "--chunk-mb") == 0 && i + 1 < argc) chunk = (size_t)(atof(argv[++i]) * 1048576.0);
            else if (strcmp(argv[i], "--output-file") == 0 && i + 1 < argc) out_path = argv[++i];
        }
        if (in.empty() || !file_exists(in)) die("usage: taxor inflate --query-file <x.gz> [--threads n] [--chunk-mb m] [--output-file out]");
        fastx::ParallelGz g;
        const double t0 = now();
        if (!g.open(in, threads, chunk, 0)) die(in + " is not a gzip file");
        FILE *of = out_path.empty() ? nullptr : fopen(out_path.c_str(), "wb");
        if (!out_path.empty() && !of) die("cannot write " + out_path);
        std::vector<char> buf;
        uint64_t total = 0;
        try {
            // chunk by chunk, the buffers changing hands like in `search` (ParallelGz::take): a copy out of them on this one thread
            // would be the limit (~5 GB/s) before the inflating threads are
            while (g.take(buf)) {
                total += buf.size();
                if (of && fwrite(buf.data(), 1, buf.size(), of) != buf.size()) die("write error on " + out_path);
                g.recycle(std::move(buf));
                buf = std::vector<char>();
            }
        } catch (const std::exception &e) { die(e.what()); }
        if (of) fclose(of);
        const double dt = now() - t0;
        printf("%llu bytes in %.3f s = %.2f GB/s on %u threads; %llu member(s), %llu chunks, %llu decoded again from a corrected start; CRC-32 and length of every member verified\n",
               (unsigned long long)total, dt, total / 1e9 / dt, threads, (unsigned long long)g.members, (unsigned long long)g.chunks_total, (unsigned long long)g.chunks_redecoded);
        printf("worker seconds: block search %.3f, decode %.3f, marker resolution + CRC %.3f\n", g.ns_find / 1e9, g.ns_decode / 1e9, g.ns_resolve / 1e9);
        printf("memory held by the reader at most: %.1f MB (largest chunk %.1f MB)\n", fastx::ParallelGz::memory_high_water() / 1048576.0, g.largest_chunk_bytes() / 1048576.0);
        return 0;
    }
    if (argc > 1 && strcmp(argv[1], "pin") == 0) return pin_command(argc, argv);   // published .hixf + reference TSV -> committed parity fixture
    if (argc > 1 && strcmp(argv[1], "reads") == 0) {                       // reader check: id, length, FNV-1a of every record
        Config cfg;
        bool allow_ranges = true;
        for (int i = 2; i < argc; ++i) {
            if (strcmp(argv[i], "--query-file") == 0 && i + 1 < argc) cfg.query_file = argv[++i];
            else if (strcmp(argv[i], "--threads") == 0 && i + 1 < argc) cfg.threads = (unsigned)atoi(argv[++i]);
            else if (strcmp(argv[i], "--batch-reads") == 0 && i + 1 < argc) cfg.batch_reads = strtoull(argv[++i], nullptr, 10);
            else if (strcmp(argv[i], "--batch-bases") == 0 && i + 1 < argc) cfg.batch_bases = strtoull(argv[++i], nullptr, 10);      // (tests: small parser jobs)
            else if (strcmp(argv[i], "--sequential") == 0) allow_ranges = false;
        }
        if (cfg.threads == 0) cfg.threads = 1;
        if (cfg.query_file.empty() || !file_exists(cfg.query_file)) die("usage: taxor reads --query-file <fasta|fastq[.gz]> [--threads n] [--batch-reads n] [--sequential]");
        std::mutex mu;
        std::map<uint64_t, std::unique_ptr<Batch>> got;
        BatchPool pool;
        pool.cap = ~size_t(0);                                        // this check keeps every batch
        const double dt = produce_batches(cfg.query_file, cfg, allow_ranges, pool, [&](std::unique_ptr<Batch> b) {
            if (b->end_of_file) return;
            std::lock_guard<std::mutex> lk(mu);
            got.emplace(b->seq, std::move(b));
        });
        uint64_t n = 0, nb = 0, expect = 0;
        for (const auto &kv : got) {
            if (kv.first != expect++) die("batch numbering has a gap");
            const Batch &b = *kv.second;
            for (size_t r = 0; r < b.ids.size(); ++r) {
                const uint64_t lo = b.offsets[r], len = b.offsets[r + 1] - lo;
                printf("%s\t%llu\t%016llx\n", b.ids.str(r).c_str(), (unsigned long long)len, (unsigned long long)fnv1a(b.bases.data() + lo, len));
                ++n;
                nb += len;
            }
        }
        fprintf(stderr, "%llu reads, %llu bases, %zu batches, %.3f s\n", (unsigned long long)n, (unsigned long long)nb, got.size(), dt);
        return 0;
    }
    if (argc > 1 && strcmp(argv[1], "search") == 0) a = 2;                 // `taxor search ...` like the reference
    else if (argc > 1 && (strcmp(argv[1], "build") == 0 || strcmp(argv[1], "profile") == 0)) {
        fprintf(stderr, "[TAXOR ERROR] only the `search` subcommand is provided by this build\n");
        return -1;
    }
    Config cfg;
    // seqan3::argument_parser takes a long option's value either as the next argument or attached with '='
    // ("--threads 4" / "--threads=4"); split the second form so that one loop handles both
    std::vector<std::string> args;
    for (int i = a; i < argc; ++i) {
        const std::string tok = argv[i];
        const size_t eq = tok.find('=');
        if (tok.size() > 2 && tok[0] == '-' && tok[1] == '-' && eq != std::string::npos && eq > 2) {
            args.push_back(tok.substr(0, eq));
            args.push_back(tok.substr(eq + 1));
        } else
            args.push_back(tok);
    }
    for (size_t ai = 0; ai < args.size(); ++ai) {
        const std::string k = args[ai];
        auto val = [&]() -> std::string {
            if (ai + 1 >= args.size()) die("Missing value for option " + k);
            return args[++ai];
        };
        auto num = [&](const std::string &v, bool integer) -> double {   // seqan3 rejects "4x" / "abc" instead of reading a prefix
            char *end = nullptr;
            const double d = integer ? (double)strtoll(v.c_str(), &end, 10) : strtod(v.c_str(), &end);
            if (v.empty() || !end || *end != '\0') die("Value parse failed for " + k + ": Argument " + v + " could not be parsed as type " + (integer ? "unsigned 64 bit integer." : "double."));
            return d;
        };
        if (k == "--index-file") cfg.index_file = val();
        else if (k == "--query-file") cfg.query_file = val();
        else if (k == "--output-file") cfg.report_file = val();
        else if (k == "--threads") {
            const long t = (long)num(val(), true);
            if (t < 1 || t > 32) die("Validation failed for option --threads: Value not in range [1,32].");   // :51-55
            cfg.threads = (unsigned)t;
        } else if (k == "--percentage") {
            cfg.threshold = num(val(), false);
            if (cfg.threshold < 0.0 || cfg.threshold > 1.0) die("Validation failed for option --percentage: Value not in range [0,1]."); // :57-61
        } else if (k == "--error-rate") {
            cfg.error_rate = num(val(), false);
            if (cfg.error_rate < 0.0 || cfg.error_rate > 1.0) die("Validation failed for option --error-rate: Value not in range [0,1]."); // :63-67
        }
        // hidden flags of the reference's parser (taxor_search.cpp:68-79): accepted, without effect there as here
        else if (k == "--output-verbose-statistics" || k == "--debug") continue;
        else if (k == "--version") { printf("taxor-search version: 0.2.0 (MI355X build)\n"); return 0; }               // :34
        else if (k == "--") break;
        else if (k == "--gpu") cfg.gpus.assign(1, atoi(val().c_str()));
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
        else if (k == "--expect") cfg.expect_file = val();
        else if (k == "--sequential") cfg.sequential = true;
        else if (k == "--group-reads") cfg.group_reads = strtoull(val().c_str(), nullptr, 10);
        else if (k == "--ixf-arithmetic") {
            const std::string spec = val();
            if (!parse_arith_spec(spec, &cfg.ixf_arith)) die("Validation failed for option --ixf-arithmetic: expected kh=<0-3>,sm=<0-3>,rot=<1-63>,red=<0-2>,fp=<0-3> (as printed by `taxor verify --variants`), got " + spec);
        }
        else if (k == "--ixf-layout") {
            const std::string spec = val();
            if (taxor_ixf_layout_parse(spec.c_str(), &cfg.ixf_layout) != TAXOR_OK) die(std::string("Validation failed for option --ixf-layout: ") + taxor_gpu_last_error());
            cfg.layout_given = true;
        }
        else if (k == "--gather") {
            cfg.gather = val();
            if (cfg.gather != "rccl" && cfg.gather != "host" && cfg.gather != "none") die("Validation failed for option --gather: Value not in {rccl, host, none}.");
        }
        else if (k == "-h" || k == "--help" || k == "-hh" || k == "--advanced-help") { usage(); return 0; }
        else die("Unknown option " + k + ". In case this is meant to be a non-option/argument/parameter, please specify the start of non-options with '--'.");
    }
    if (cfg.index_file.empty()) die("Option --index-file is required but not set.");
    if (cfg.threads == 0) {
        // --threads not given.  The reference's default is one worker thread (taxor_search_configuration.hpp:17) -- there the
        // thread that classifies; here host threads only parse the query file and render text for the GPU, and one of them
        // delivers 1-3 Gbp/s to a device that classifies 25.  Default: a sixteenth of the machine, between 4 and 16 -- or what the
        // container's CPU quota allows, if it has one (threads beyond it only make the cgroup stop all of them for the rest of each
        // scheduler period).  A --threads the user gives is obeyed as it is.
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const double quota = cpu_quota();
        cfg.threads = quota >= 1.0 ? std::min(16u, std::max(2u, (unsigned)quota)) : std::min(16u, std::max(4u, hw / 16u));
    }

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
    if (!cfg.expect_file.empty() && !file_exists(cfg.expect_file)) die("The expected-output file does not exist: " + cfg.expect_file);
    printf("done!\n");
    trace("input checked");

    FILE *out = fopen(cfg.report_file.c_str(), "wb");                       // search_hixf, :340-343
    if (!out) die("cannot open output file " + cfg.report_file);
    fputs("#QUERY_NAME\tACCESSION\tREFERENCE_NAME\tTAXID\tREF_LEN\tQUERY_LEN\tQHASH_COUNT\tQHASH_MATCH\tTAX_STR\tTAX_ID_STR\n", out);

    double t_index = 0, t_reads = 0, t_compute = 0, t_pin = 0, t_search = 0, t_gather = 0, t_search_wall = 0;
    uint64_t n_batches = 0, n_gpu_batches = 0;
    uint64_t total_reads = 0, total_bases = 0;
    std::mutex stat_mu;
    // One index, all of `queries`: the index is loaded and uploaded once; query files are read concurrently (a gzip
    // stream inflates on one thread, so several files are the only way to read gzip input faster) and written in the
    // order they were given.
    auto search_files = [&](const std::string &hixf_file, const std::vector<std::string> &queries, bool last) {
        double t0 = now();
        taxor_hixf *h = nullptr;
        if (taxor_hixf_load(hixf_file.c_str(), &h) != TAXOR_OK) die(taxor_gpu_last_error());
        trace("index file loaded");
        if (cfg.ixf_arith) {
            taxor_hixf_set_arith(h, cfg.ixf_arith);
            taxor_ixf_variant av{};
            taxor_ixf_arith_decode(cfg.ixf_arith, &av);
            char desc[512];
            taxor_ixf_variant_describe(&av, desc, sizeof desc);
            fprintf(stderr, "[TAXOR SEARCH WARNING] %s is searched under --ixf-arithmetic %s, not under this build's own reading of the\n"
                            "  un-vendored IXF arithmetic: %s (seed, segment length and row stride per IXF from the file)\n", hixf_file.c_str(),
                    arith_spec(av).c_str(), desc);
        }
        if (cfg.layout_given) {
            if (taxor_hixf_set_layout(h, cfg.ixf_layout) != TAXOR_OK) die(taxor_gpu_last_error());
            char ld[128];
            taxor_ixf_layout_describe(cfg.ixf_layout, ld, sizeof ld);
            fprintf(stderr, "[TAXOR SEARCH WARNING] %s is read under --ixf-layout %s: its fingerprint vectors are transposed into the search layout on the device\n",
                    hixf_file.c_str(), ld);
        }
        const taxor_hixf_view *view = taxor_hixf_get_view(h);
        if (taxor_hixf_get_meta(h)->foreign_schema)
            fprintf(stderr, "[TAXOR SEARCH WARNING] %s was not written by this library (its IXF records follow another layout, read by probing).\n"
                            "  The arithmetic of seqan3's interleaved_xor_filter is not part of the reference sources; this build's reading of it has\n"
                            "  not been checked against this file.  Run `taxor verify --index-file %s --genome-file <a genome in the index>` first:\n"
                            "  a wrong reading still loads and classifies at the false-positive floor.\n", hixf_file.c_str(), hixf_file.c_str());
        // threshold model (threshold.hpp:22-47)
        taxor_gpu_search_params prm{};
        if (taxor_threshold_select(view, cfg.error_rate, cfg.threshold, &prm) != TAXOR_OK)
            die("no threshold model for k=" + std::to_string(view->kmer_size) + " and error rate " + std::to_string(cfg.error_rate));
        switch (prm.model) {                                            // the reference's messages, threshold.hpp:32-46
        case TAXOR_THR_PERCENTAGE: printf("use percentage-model\t%g\n", cfg.threshold); break;
        case TAXOR_THR_SYNCMER: printf("use syncmer model\n"); break;
        case TAXOR_THR_KMER: printf("use kmer-model\n"); break;
        default: printf("use frac minhash\n"); break;
        }
        const size_t ng = cfg.gpus.size();

        // Overlapped stages (the reference joins its workers after every 1024 reads, do_parallel.hpp:31-32):
        //   reader threads    : FASTA/FASTQ(.gz) -> numbered chunks of records   (taxor_search.cpp:315-321);
        //                       plain files are parsed by up to --threads threads (fastx.h), several files at a time
        //   GPU workers       : chunks -> one GPU batch of useful size (as many queued chunks as make ~131072 reads, handed
        //                       over as segments, no copy) -> streamed upload, kernels, fetch (:325); two per device, so
        //                       that one batch's transfers run beside the other's kernels
        //   sequencer + piece threads : tuples -> 0.8*max filter -> TSV text (:266-306) -> file, in file and chunk order (:311); below
        // Output stays in input order (the reference's order at --threads 1).  The host stages start BEFORE the index goes to
        // the devices: the first chunks are parsed while it uploads.
        const size_t nf = queries.size();
        const unsigned readers = (unsigned)std::max<size_t>(1, std::min<size_t>({nf, cfg.threads, 8}));
        const unsigned parse_threads = std::max(1u, std::min(cfg.threads, 32u) / readers);
        static const unsigned fmt_div = [] { const char *e = tune_env("TAXOR_CLI_FORMATTER_DIV"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? (unsigned)v : 4u; }();
        const unsigned formatters = std::max(1u, std::min(cfg.threads, 32u) / fmt_div);
        static const unsigned workers_per_gpu = [] { const char *e = tune_env("TAXOR_CLI_WORKERS_PER_GPU"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 4 ? (unsigned)v : 2u; }();
        const uint64_t group_reads = cfg.group_reads ? cfg.group_reads : (cfg.batch_reads ? cfg.batch_reads : 131072);
        const size_t group_max_chunks = 32;
        // (q_parsed / q_in are as deep as the pool allows: what bounds the parsers is the pool's byte budget, not a queue -- while the
        // index uploads they parse ahead, and a query file smaller than the budget is parsed completely by the time it is resident)
        BoundedQueue<std::unique_ptr<Batch>> q_parsed(4096), q_in(4096), q_fmt(8);
        BatchPool pool;
        pool.cap = readers * parse_threads + 16 + 2 + 32 + ng * workers_per_gpu * group_max_chunks + 8 + formatters + 8 + 8 + 2 * readers;
        // ... by count for small chunks; by bytes for the usual ~128 MB ones: 6 GiB for the host stages + 6 GiB per device (two GPU
        // batches of ~1.3 GB in flight per device, the chunks being parsed, formatted and written) instead of cap x 128 MB
        // = 19 GB at one device and 75 GB at eight.  The floor lets every stage hold one buffer whatever the budget.
        pool.floor = readers * parse_threads + 2 * readers + ng * workers_per_gpu + formatters + 8;
        {
            const char *e = tune_env("TAXOR_CLI_POOL_GB");
            const double gb = e ? atof(e) : 0.0;
            pool.byte_budget = gb > 0.0 ? (uint64_t)(gb * (double)(1ull << 30)) : (6ull + 6ull * ng) << 30;
        }
        // a file that is not being written yet may run at most two chunks ahead, so that the file whose turn it is can
        // always get buffers
        std::mutex fmu;
        std::condition_variable fcv;
        size_t current_file = 0;
        std::vector<int> ahead(nf, 0);
        std::atomic<size_t> next_file{0};
        std::vector<double> reader_time(readers, 0.0);
        std::vector<std::thread> reader_threads;
        for (unsigned rt = 0; rt < readers; ++rt)
            reader_threads.emplace_back([&, rt] {
                for (;;) {
                    const size_t f = next_file.fetch_add(1);
                    if (f >= nf) break;
                    auto gate = [&, f] {
                        std::unique_lock<std::mutex> lk(fmu);
                        fcv.wait(lk, [&] { return f == current_file || ahead[f] < 2; });
                    };
                    reader_time[rt] += produce_batches(queries[f], cfg, !cfg.sequential, pool, [&, f](std::unique_ptr<Batch> b) {
                        if (!b->end_of_file) {
                            std::lock_guard<std::mutex> lk(fmu);
                            ++ahead[f];
                        }
                        q_parsed.push(std::move(b));
                    }, (uint32_t)f, parse_threads, gate);
                }
            });
        // Page-locking a chunk buffer (once per buffer: chunks are recycled) costs ~50-90 ms per GB inside the driver, and while the
        // driver registers, the GPU workers' submissions wait: RefSeq-class index, 4.2 M x 10 kb, interleaved on one box
        // (profiles/r04/cli_pin_10kb.txt): buffers registered when they come round the second time (rounds 3-4a) 0.82-0.84 x the
        // library's rate; all of them at their first fill, i.e. while the index uploads, 0.86-0.89 and the index resident 0.15-0.2 s
        // later; NEVER 0.90-0.92 -- the library copies from pageable memory at 0.94 of the page-locked rate since the hardware queues
        // are set (round 3).  So chunk buffers stay pageable; TAXOR_CLI_PIN_AFTER=n (under TAXOR_TUNING) registers a buffer at its
        // n-th fill, on a small stage of its own between the parsers and the GPU workers.
        static const uint32_t pin_after = [] { const char *e = tune_env("TAXOR_CLI_PIN_AFTER"); const int v = e ? atoi(e) : 0; return v >= 1 ? (uint32_t)v : ~0u; }();
        std::vector<std::thread> pinners;
        static const int n_pinners = [] { const char *e = tune_env("TAXOR_CLI_PINNERS"); const int v = e ? atoi(e) : 2; return v >= 1 && v <= 32 ? v : 2; }();
        for (int pt = 0; pt < n_pinners; ++pt)
            pinners.emplace_back([&] {
                std::unique_ptr<Batch> b;
                while (q_parsed.pop(b)) {
                    if (!b->end_of_file && b->may_pin && !b->pinned && b->fills >= pin_after && b->bases.capacity() >= (1u << 20) &&
                        taxor_gpu_host_register(&b->bases[0], b->bases.capacity()) == TAXOR_OK)
                        b->pinned = &b->bases[0];       // recycled with the chunk: a DMA source from then on
                    q_in.push(std::move(b));
                }
            });
        std::thread reader([&] {
            for (auto &t : reader_threads) t.join();
            q_parsed.close();
            for (auto &t : pinners) t.join();
            q_in.close();
            trace("readers done");
        });
        // ---- TSV text and the report file.  The report is ONE file and the kernel serialises a file's buffered writes, so there is one
        // stream of write() calls whatever the threads (profiles/r04/pwrite_scaling.txt, a RAM-backed file on the pool's box: pwrite of
        // 1-MiB blocks from 1 / 2 / 8 threads 9.9 / 9.8 / 9.5 GB/s).  Those figures are for blocks that are in the writing core's cache;
        // one writer thread handed 64-MiB texts by formatter threads -- rounds 2-4a -- copies them out of DRAM at 5-6 GB/s, and short
        // reads (a 1-kb read of the family workload is ~620-870 bytes of text per 1000 bases) were bound by exactly that.  So the text is
        // made in PIECES of about a megabyte, each written by the thread that formatted it while it is still in that core's cache:
        //   sequencer       : chunks (GPU workers, any order) -> file and chunk order -> pieces (ranges of a chunk's reads), numbered
        //   piece threads   : tuples -> 0.8*max filter -> TSV text of the piece (:266-306) -> wait for the piece's turn -> write()
        // (Also tried: the piece threads pwrite()-ing at offsets handed out in turn, the writes themselves overlapping: 2.0-3.3 GB/s on
        // 8-32 threads, 34-100 s summed inside pwrite -- the threads spin on the file's lock -- against 6.5 with the write inside the
        // turn, profiles/r04/cli_text_1kb_pwrite.txt; a helper thread preallocating the file a gigabyte ahead of one writer, 5.9 GB/s -- no
        // gain; copies into a mapping of a fully preallocated file, 11 GB/s on four threads, but the preallocation is 21 GB/s of one
        // thread by itself and the mapping costs the GPU calls of the process their address-space lock -- round 3.)
        struct Hold { std::unique_ptr<Batch> b; std::atomic<uint32_t> left{0}; };
        struct Piece { Hold *h = nullptr; uint32_t r0 = 0, r1 = 0; uint64_t ticket = 0; };
        BoundedQueue<Piece> q_piece(4 * formatters + 16);
        // from here on the report goes to the file descriptor: the header line above was the last thing written through the FILE*
        // (flushed now); nothing may use `out`'s stdio buffer again until the pieces are done, or it would land out of order
        fflush(out);
        const int out_fd = fileno(out);
        static const uint64_t piece_bytes = [] { const char *e = tune_env("TAXOR_CLI_PIECE_KB"); const int v = e ? atoi(e) : 0; return (uint64_t)(v >= 16 ? v : 1024) << 10; }();
        std::atomic<uint64_t> out_written{0};
        double t_write = 0;            // seconds inside write()
        std::mutex turn_mu;
        std::condition_variable turn_cv[64];            // a waiting piece thread sleeps on the one of its ticket: passing the turn wakes one thread
        uint64_t next_ticket = 0;
        static const bool turn_spin = [] { const char *e = tune_env("TAXOR_CLI_TURN_SPIN"); return e ? atoi(e) != 0 : true; }();
        std::atomic<uint64_t> turn_now{0};              // next_ticket, readable without the lock: the thread whose turn is next spins on it
        double t_first_write = 0, t_last_write = 0;
        std::atomic<uint64_t> line_bytes_guess{192};   // bytes of text per tuple, learnt from the pieces formatted so far
        auto chunk_done = [&](Hold *hd) {
            {
                std::lock_guard<std::mutex> lk(fmu);
                --ahead[hd->b->file];
                fcv.notify_all();
            }
            pool.put(std::move(hd->b));
            delete hd;
        };
        std::thread sequencer([&] {
            std::unique_ptr<Batch> b;
            std::map<std::pair<uint32_t, uint64_t>, std::unique_ptr<Batch>> pending; // chunks that arrived ahead of their turn
            uint32_t cur_file = 0;
            uint64_t next_seq = 0, ticket = 0;
            while (q_fmt.pop(b)) {
                const auto key = std::make_pair(b->file, b->seq);
                pending.emplace(key, std::move(b));
                while (!pending.empty() && pending.begin()->first == std::make_pair(cur_file, next_seq)) {
                    std::unique_ptr<Batch> cur = std::move(pending.begin()->second);
                    pending.erase(pending.begin());
                    if (cur->end_of_file) {                 // every chunk of this file is on its way: the next file's turn
                        ++cur_file;
                        next_seq = 0;
                        std::lock_guard<std::mutex> lk(fmu);
                        current_file = cur_file;
                        fcv.notify_all();
                        continue;
                    }
                    ++next_seq;
                    Hold *hd = new Hold;
                    hd->b = std::move(cur);
                    const Batch &bt = *hd->b;
                    const size_t n = bt.ids.size();
                    if (n == 0) { chunk_done(hd); continue; }
                    // cut where the estimated text passes piece_bytes
                    const uint64_t per_tuple = line_bytes_guess.load();
                    std::vector<uint32_t> cuts{0};
                    uint64_t est = 0;
                    for (size_t r = 0; r < n; ++r) {
                        est += (bt.read_off[r + 1] - bt.read_off[r]) * per_tuple + bt.ids.len(r) + 32;
                        if (est >= piece_bytes && r + 1 < n) { cuts.push_back((uint32_t)(r + 1)); est = 0; }
                    }
                    cuts.push_back((uint32_t)n);
                    hd->left = (uint32_t)(cuts.size() - 1);
                    for (size_t c = 0; c + 1 < cuts.size(); ++c) q_piece.push(Piece{hd, cuts[c], cuts[c + 1], ticket++});
                }
            }
            q_piece.close();
        });
        std::vector<std::thread> fmt_threads;
        for (unsigned ft = 0; ft < formatters; ++ft)
            fmt_threads.emplace_back([&] {
                Piece pc;
                std::vector<const char *> idp;
                std::vector<uint64_t> idl, rl;
                std::vector<char> text(piece_bytes + piece_bytes / 2);
                while (q_piece.pop(pc)) {
                    const Batch &bt = *pc.h->b;
                    const size_t n = pc.r1 - pc.r0;
                    idp.resize(n); idl.resize(n); rl.resize(n);
                    for (size_t r = 0; r < n; ++r) {
                        idp[r] = bt.ids.ptr(pc.r0 + r);
                        idl[r] = bt.ids.len(pc.r0 + r);
                        rl[r] = bt.offsets[pc.r0 + r + 1] - bt.offsets[pc.r0 + r];
                    }
                    uint64_t need = taxor_format_reads(h, n, idp.data(), idl.data(), rl.data(), bt.n_hashes.data() + pc.r0, bt.read_off.data() + pc.r0,
                                                       bt.user_bin.data(), bt.count.data(), text.data(), text.size());
                    if (need > text.size()) {
                        text.resize(need + need / 8 + 4096);
                        need = taxor_format_reads(h, n, idp.data(), idl.data(), rl.data(), bt.n_hashes.data() + pc.r0, bt.read_off.data() + pc.r0,
                                                  bt.user_bin.data(), bt.count.data(), text.data(), text.size());
                    }
                    const uint64_t tuples = bt.read_off[pc.r1] - bt.read_off[pc.r0];
                    if (tuples > 64) line_bytes_guess = std::max<uint64_t>(64, need / tuples + 16);
                    {
                        // (the thread that is next in line does not go to sleep for the ~0.1 ms its predecessor writes: a wake-up costs
                        // 10-20 us of every 120)
                        if (turn_spin && pc.ticket - turn_now.load(std::memory_order_acquire) == 1)
                            for (int spin = 0; spin < 20000 && turn_now.load(std::memory_order_acquire) != pc.ticket; ++spin) _mm_pause();
                        std::unique_lock<std::mutex> lk(turn_mu);
                        turn_cv[pc.ticket % 64].wait(lk, [&] { return next_ticket == pc.ticket; });
                        const double w0 = now();
                        if (t_first_write == 0) t_first_write = w0;
                        for (uint64_t done = 0; done < need;) {
                            const ssize_t w = ::write(out_fd, text.data() + done, need - done);
                            if (w < 0 && errno == EINTR) continue;            // a signal (SIGSTOP / SIGCONT, a debugger) is not an I/O error
                            if (w <= 0) die("cannot write to " + cfg.report_file + ": " + (w < 0 ? strerror(errno) : "write() returned 0"));
                            done += (uint64_t)w;
                        }
                        t_last_write = now();
                        t_write += t_last_write - w0;
                        ++next_ticket;
                        turn_now.store(next_ticket, std::memory_order_release);
                        turn_cv[next_ticket % 64].notify_all();
                    }
                    out_written += need;
                    if (pc.h->left.fetch_sub(1) == 1) chunk_done(pc.h);
                }
            });

        // ---- the index: one replica per device (reads are independent, taxor_search.cpp:214: the index is replicated,
        // batches are sharded).  Several devices (north star: "reads sharded across the GPUs, per-read results gathered over
        // RCCL/xGMI"): one communicator; the index crosses PCIe once and is broadcast, rounds of ng batches are classified
        // side by side and their results gathered on the first device (taxor_gpu.h, "Several GPUs of one node").  The
        // transport is decided here, once, by rule -- never by a failure.
        std::vector<taxor_gpu_index *> gidx(ng, nullptr);
        taxor_gpu_comm *comm = nullptr;
        std::string gather = cfg.gather;
        if (gather.empty()) {
            bool distinct = true;
            for (size_t i = 0; i < ng; ++i)
                for (size_t j = i + 1; j < ng; ++j) distinct = distinct && cfg.gpus[i] != cfg.gpus[j];
            gather = ng == 1 ? "none" : (distinct ? "rccl" : "host");
        }
        if (gather != "none") {
            if (taxor_gpu_comm_create(cfg.gpus.data(), (uint32_t)ng, gather == "rccl" ? TAXOR_COMM_RCCL : TAXOR_COMM_HOST, &comm) != TAXOR_OK)
                die(std::string(taxor_gpu_last_error()) + "\n(--gather host stages the same transfers through host memory)");
            if (taxor_gpu_index_create_replicated(comm, view, gidx.data()) != TAXOR_OK) die(taxor_gpu_last_error());
            taxor_gpu_comm_stats cs{};
            taxor_gpu_comm_info(comm, &cs);
            if (tune_env("TAXOR_CLI_TRACE"))
                fprintf(stderr, "[trace] index on %zu devices (%s): %.2f GB per replica, %.2f GB over PCIe, %.2f GB by ncclBroadcast, %.3f s\n", ng,
                        gather.c_str(), cs.index_bytes / 1e9, cs.index_upload_bytes / 1e9, cs.index_broadcast_bytes / 1e9, cs.index_seconds);
        } else {
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
        trace("index resident in HBM");
        const CpuMark cpu0 = cpu_mark();
        if (tune_env("TAXOR_CLI_TRACE"))
            fprintf(stderr, "[trace] index: %.2f GB per replica in %.3f s = %.1f GB/s (file open to resident)\n", taxor_gpu_index_data_bytes(gidx[0]) / 1e9,
                    now() - t0, taxor_gpu_index_data_bytes(gidx[0]) / 1e9 / (now() - t0));
        // the host mapping of the index is not needed any more: its pages go back on a helper thread, beside the search
        std::thread releaser([h] { taxor_hixf_release_data(h); });
        const double t_search0 = now();
        const unsigned wpg = workers_per_gpu;       // with a communicator: that many searcher SETS (one searcher per device each), rounds alternate between them
        std::vector<taxor_gpu_searcher *> sr(ng * wpg, nullptr);
        for (size_t g = 0; g < ng; ++g)
            for (unsigned w = 0; w < wpg; ++w)
                if (taxor_gpu_searcher_create(gidx[g], &prm, &sr[g * wpg + w]) != TAXOR_OK) die(taxor_gpu_last_error());

        auto pin = [](Batch &bt) {      // recycled with the chunk: pinned once, DMA source from then on
            if (bt.may_pin && !bt.pinned && bt.fills >= pin_after && bt.bases.capacity() >= (1u << 20) &&
                taxor_gpu_host_register(&bt.bases[0], bt.bases.capacity()) == TAXOR_OK)
                bt.pinned = &bt.bases[0];
        };
        // the CSR of several chunks classified as one batch -> each chunk's own CSR
        auto split_results = [&](std::vector<std::unique_ptr<Batch>> &chunks, const taxor_gpu_results &res) {
            uint64_t r0 = 0;
            for (auto &bt : chunks) {
                const uint64_t n = bt->ids.size(), lo = res.read_off[r0], hi = res.read_off[r0 + n];
                bt->read_off.resize(n + 1);
                for (uint64_t i = 0; i <= n; ++i) bt->read_off[i] = res.read_off[r0 + i] - lo;
                bt->user_bin.assign(res.user_bin + lo, res.user_bin + hi);
                bt->count.assign(res.count + lo, res.count + hi);
                bt->n_hashes.assign(res.n_hashes + r0, res.n_hashes + r0 + n);
                r0 += n;
                std::lock_guard<std::mutex> lk(stat_mu);
                ++n_batches;
                total_reads += n;
                total_bases += bt->bases.size();
            }
            if (r0 != res.n_reads) die("internal: a batch's results do not cover its chunks");
        };
        static const double fill_seconds = [] { const char *e = tune_env("TAXOR_CLI_FILL_MS"); return e ? atof(e) * 1e-3 : 0.020; }();
        // (a third and fourth worker per device that copy their results out and collect the next batch while two "slots" stay taken:
        // 25-32 Gbp/s against 33-35 with two workers, profiles/r04/cli_variants.txt -- more searchers share the same hardware queues)
        static const int fill_running = [] { const char *e = tune_env("TAXOR_CLI_FILL_RUNNING"); return e ? atoi(e) : 1; }();
        std::atomic<int> n_running{0};      // GPU batches in flight (single-device workers)
        std::vector<std::thread> workers;
        std::mutex comm_mu;     // a communicator is single-caller, and its result arrays live until its next gather
        if (comm)
          for (unsigned set = 0; set < wpg; ++set)
            workers.emplace_back([&, set] {
                // searcher of device g in this worker's set (sr is laid out device-major: sr[g * wpg + set])
                std::vector<taxor_gpu_searcher *> mine(ng);
                for (size_t g = 0; g < ng; ++g) mine[g] = sr[g * wpg + set];
                // one round = up to ng GPU batches, batch g on device g, classified side by side; then ONE gather of the round's
                // per-read results on the first device and one copy to the host.  A GPU batch is made of queued chunks like in
                // the single-device workers below (~group_reads reads, handed over as segments); devices without a batch in a
                // round run an empty one, so that every searcher has a finished run to gather.
                std::vector<std::vector<std::unique_ptr<Batch>>> round(ng);
                std::vector<std::unique_ptr<Batch>> eofs, flat;
                std::unique_ptr<Batch> b;
                const uint64_t zero_off[1] = {0};
                bool open = true;
                while (open) {
                    for (auto &g : round) g.clear();
                    eofs.clear();
                    size_t used = 0;
                    // the first chunk of a round is waited for; after that only what is already queued is taken
                    for (size_t g = 0; g < ng; ++g) {
                        uint64_t gr = 0, gb = 0;
                        while ((gr < group_reads || (!cfg.batch_reads && !cfg.group_reads && gb < (1ull << 29) && gr < (1u << 20))) &&
                               gb < (3ull << 30) && round[g].size() < group_max_chunks) {
                            bool got;
                            if (used == 0 && round[g].empty() && eofs.empty()) { got = q_in.pop(b); if (!got) open = false; }
                            else got = q_in.try_pop(b);
                            if (!got) break;
                            if (b->end_of_file) { eofs.push_back(std::move(b)); continue; }
                            gr += b->ids.size();
                            gb += b->bases.size();
                            round[g].push_back(std::move(b));
                            ++used;
                        }
                        if (round[g].empty()) break;
                    }
                    if (used == 0) {
                        for (auto &e : eofs) q_fmt.push(std::move(e));
                        continue;
                    }
                    const double t1 = now();
                    std::vector<std::thread> dev;
                    std::vector<std::string> errs(ng);
                    for (size_t g = 0; g < ng; ++g)
                        dev.emplace_back([&, g] {
                            if (round[g].empty()) {
                                if (taxor_gpu_search_batch_begin(mine[g], nullptr, zero_off, 0) != TAXOR_OK) errs[g] = taxor_gpu_last_error();
                                return;
                            }
                            std::vector<taxor_read_segment> segs;
                            auto run = [&]() -> int {
                                segs.clear();
                                for (auto &bt : round[g]) segs.push_back({bt->bases.data(), bt->offsets.data(), bt->ids.size()});
                                const int rc = taxor_gpu_search_segments_begin(mine[g], segs.data(), segs.size());
                                return rc != TAXOR_OK ? rc : taxor_gpu_batch_sync(mine[g]);
                            };
                            for (auto &bt : round[g]) pin(*bt);
                            int rc = run();
                            if (rc == TAXOR_E_ALPHABET) {
                                bool any = false;
                                for (auto &bt : round[g]) any = strip_space_and_digits(*bt) || any;
                                if (any) rc = run();
                            }
                            if (rc != TAXOR_OK) errs[g] = taxor_gpu_last_error();
                        });
                    for (auto &t : dev) t.join();
                    for (const auto &e : errs)
                        if (!e.empty()) die(e);
                    const double t2 = now();
                    flat.clear();                           // the gathered CSR is in device order, and in chunk order within a device
                    for (auto &g : round)
                        for (auto &bt : g) flat.push_back(std::move(bt));
                    double t3;
                    {
                        std::lock_guard<std::mutex> lk(comm_mu);      // the other set's round keeps the devices busy meanwhile
                        taxor_gpu_results res{};
                        if (taxor_gpu_gather_results(comm, mine.data(), &res) != TAXOR_OK) die(taxor_gpu_last_error());
                        t3 = now();
                        split_results(flat, res);
                    }
                    {
                        std::lock_guard<std::mutex> lk(stat_mu);
                        t_search += t2 - t1;
                        t_gather += t3 - t2;
                        t_compute += now() - t1;
                        n_gpu_batches += std::min(used, ng);
                    }
                    for (auto &bt : flat) q_fmt.push(std::move(bt));
                    for (auto &e : eofs) q_fmt.push(std::move(e));
                }
            });
        else
            for (size_t wi = 0; wi < ng * wpg; ++wi)
                workers.emplace_back([&, wi] {
                    std::vector<std::unique_ptr<Batch>> group, eofs;
                    std::vector<taxor_read_segment> segs;
                    std::unique_ptr<Batch> b;
                    while (q_in.pop(b)) {
                        if (b->end_of_file) { q_fmt.push(std::move(b)); continue; }
                        // one GPU batch = the queued chunks that make up ~group_reads reads: the kernels want >= 10^5 reads in
                        // flight, the parsers want chunks small enough to hand out to many threads
                        group.clear();
                        eofs.clear();
                        uint64_t gr = b->ids.size(), gb = b->bases.size();
                        group.push_back(std::move(b));
                        // (short reads: more of them, until the batch holds 2^29 bases -- the library's sub-batches then reach
                        // their full size)
                        const double t_fill0 = now();
                        while ((gr < group_reads || (!cfg.batch_reads && !cfg.group_reads && gb < (1ull << 29) && gr < (1u << 20))) &&
                               gb < (3ull << 30) && group.size() < group_max_chunks) {
                            if (!q_in.try_pop(b)) {
                                // nothing queued: the GPU is the faster side right now.  Launching what there is makes the batches
                                // small exactly then (a quarter of the size the kernels are fastest at); while enough OTHER batches
                                // are in flight to keep the device busy this worker goes on collecting instead -- for a bounded
                                // time, because the chunks it holds may be what the writer is waiting for
                                if (!(gr < group_reads && gb < (1ull << 29) && now() - t_fill0 < fill_seconds && n_running.load() >= fill_running)) break;
                                if (!q_in.pop_for(b, 100e-6)) { if (q_in.done()) break; continue; }
                            }
                            if (b->end_of_file) { eofs.push_back(std::move(b)); continue; }
                            gr += b->ids.size();
                            gb += b->bases.size();
                            group.push_back(std::move(b));
                        }
                        ++n_running;
                        const double t1 = now();
                        for (auto &bt : group) pin(*bt);
                        const double t2 = now();
                        taxor_gpu_results res{};
                        auto run = [&]() -> int {
                            segs.clear();
                            for (auto &bt : group) segs.push_back({bt->bases.data(), bt->offsets.data(), bt->ids.size()});
                            const int rc = taxor_gpu_search_segments_begin(sr[wi], segs.data(), segs.size());
                            return rc != TAXOR_OK ? rc : taxor_gpu_search_batch_end(sr[wi], &res);
                        };
                        int rc = run();
                        if (rc == TAXOR_E_ALPHABET) {
                            bool any = false;
                            for (auto &bt : group) any = strip_space_and_digits(*bt) || any;
                            if (any) rc = run();
                        }
                        if (rc != TAXOR_OK) die(taxor_gpu_last_error());
                        --n_running;
                        const double t3 = now();
                        split_results(group, res);
                        {
                            std::lock_guard<std::mutex> lk(stat_mu);
                            t_pin += t2 - t1;
                            t_search += t3 - t2;
                            ++n_gpu_batches;
                            t_compute += now() - t1;
                        }
                        for (auto &bt : group) q_fmt.push(std::move(bt));
                        for (auto &e : eofs) q_fmt.push(std::move(e));
                    }
                });
        for (auto &t : workers) t.join();
        trace("GPU workers done");
        t_search_wall += now() - t_search0;
        if (tune_env("TAXOR_CLI_TRACE")) {
            const CpuMark c1 = cpu_mark();
            const double w = now() - t_search0;
            fprintf(stderr, "[trace] search phase: %.3f s wall, %.2f s user + %.2f s system CPU = %.1f CPUs busy (allowance %.0f); the cgroup throttled the process in %llu of %llu periods, %.2f thread-seconds\n",
                    w, c1.user - cpu0.user, c1.sys - cpu0.sys, w > 0 ? (c1.user - cpu0.user + c1.sys - cpu0.sys) / w : 0.0, cpu_allowance(),
                    (unsigned long long)(c1.throttled_periods - cpu0.throttled_periods), (unsigned long long)(c1.periods - cpu0.periods), c1.throttled - cpu0.throttled);
        }
        q_fmt.close();
        sequencer.join();
        for (auto &t : fmt_threads) t.join();
        reader.join();
        trace("writer done");
        if (tune_env("TAXOR_CLI_TRACE"))
            fprintf(stderr, "[trace] report: %.2f GB in %llu pieces by %u threads, %.3f s from the first write to the last = %.1f GB/s (%.3f s inside write() = %.1f GB/s)\n",
                    out_written.load() / 1e9, (unsigned long long)next_ticket, formatters, t_last_write - t_first_write,
                    t_last_write > t_first_write ? out_written.load() / 1e9 / (t_last_write - t_first_write) : 0.0, t_write, t_write > 0 ? out_written.load() / 1e9 / t_write : 0.0);
        t_reads += *std::max_element(reader_time.begin(), reader_time.end());
        for (auto *x : sr) taxor_gpu_searcher_destroy(x);
        if (comm) {
            taxor_gpu_comm_stats cs{};
            taxor_gpu_comm_info(comm, &cs);
            if (tune_env("TAXOR_CLI_TRACE"))
                fprintf(stderr, "[trace] %llu gathers (%s): %.1f MB from peer devices, %.3f s inside taxor_gpu_gather_results\n",
                        (unsigned long long)cs.gathers, gather.c_str(), cs.gather_bytes / 1e6, cs.gather_seconds);
            taxor_gpu_comm_destroy(comm);
        }
        for (size_t g = 0; g < ng; ++g) taxor_gpu_index_destroy(gidx[g]);
        trace("GPU memory released");
        releaser.join();
        taxor_hixf_free(h);
        trace("host index released");
        if (last) {
            // the process is about to end: unpinning and unmapping gigabytes of staging buffers page by page
            // would only delay the exit (~0.1 s per GB)
            for (auto &b : pool.free_) (void)b.release();
        } else {
            for (auto &b : pool.free_)
                if (b->pinned) taxor_gpu_host_unregister(b->pinned);
        }
        pool.free_.clear();
        if (tune_env("TAXOR_CLI_TRACE"))
            fprintf(stderr, "[trace] chunk buffers: %zu made (floor %zu, cap %zu), peak %.2f GB on the books against a budget of %.2f GB, %llu released early\n",
                    pool.made, pool.floor, pool.cap, pool.peak_bytes / 1073741824.0, pool.byte_budget / 1073741824.0, (unsigned long long)pool.trimmed);
        trace("batch buffers released");
    };
    if (index_files.size() == 1) {
        search_files(index_files[0], query_files, true);
    } else {
        // several indexes: the reference's order, every query file against every index in turn (:344-358)
        for (size_t qi = 0; qi < query_files.size(); ++qi)
            for (size_t ii = 0; ii < index_files.size(); ++ii)
                search_files(index_files[ii], {query_files[qi]}, qi + 1 == query_files.size() && ii + 1 == index_files.size());
    }
    fclose(out);
    trace("output closed");
    if (tune_env("TAXOR_CLI_TRACE"))
        fprintf(stderr, "[trace] %llu chunks in %llu GPU batches: pin %.3f s, search %.3f s, gather %.3f s, copy-out %.3f s (summed over workers); "
                        "search phase %.3f s wall after the index was resident = %.1f Mbp/s\n", (unsigned long long)n_batches,
                (unsigned long long)(n_gpu_batches ? n_gpu_batches : n_batches), t_pin, t_search, t_gather, t_compute - t_pin - t_search - t_gather, t_search_wall,
                t_search_wall > 0 ? total_bases / t_search_wall / 1e6 : 0.0);
    printf("Index I/O\tReads I/O\tCompute\n%.2f\t%.2f\t%.2f\n", t_index, t_reads, t_compute);   // :328-336
    printf("%llu reads, %llu bases classified\n", (unsigned long long)total_reads, (unsigned long long)total_bases);
    {   // the reference's main() closes with the process's CPU time and peak resident set (main.cpp:37-49,79-84)
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        const double cpu = ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
        printf("CPU time  : %g sec\nPeak RSS  : %d MByte\n", cpu, (int)((size_t)ru.ru_maxrss * 1024 / (1024 * 1024)));
    }
    int rc = 0;
    if (!cfg.expect_file.empty()) rc = compare_tsv(cfg.report_file, cfg.expect_file) ? 3 : 0;
    fflush(stdout);
    fflush(stderr);
    if (tune_env("TAXOR_CLI_CLEAN_EXIT")) return rc;     // a profiler's exit hooks must run (rocprofv3 writes its files at exit)
    _exit(rc);  // everything is written and closed; skip the runtime's and the allocator's teardown
}
