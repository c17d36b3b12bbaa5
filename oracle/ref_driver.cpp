// ref_driver.cpp -- C entry points around the pieces of the REFERENCE that compile from their own files with nothing
// but the standard library (no seqan3 / cereal / ankerl): built by `make ref` into oracle/_ref/libtaxor_ref.so straight
// from /root/reference (the sources are included / compiled where they lie; nothing is copied).  TEST INFRASTRUCTURE:
// it pins the oracle's restatements against the reference's own code wherever that is possible:
//   * src/hixf/search/syncmer_model.hpp      get_min_syncmer_match_ratio          (on the search path, threshold.hpp:59)
//   * src/hixf/search/{kmer_model,fracminhash_model,gaussian_inverse}.cpp         (thresholds of non-syncmer indexes)
//   * src/hixf/build/adjust_seed.hpp         adjust_seed                          (minimiser seed)
//   * src/hixf/search/do_parallel.hpp        hixf::do_parallel, the chunk loop's scheduler (taxor_search.cpp:325): `threads`
//     std::async tasks over floor(n/threads)-sized slices, the remainder on the last one, one barrier per 1024-record chunk
//   * src/hixf/search/sync_out.hpp           hixf::sync_out, the mutexed result stream (taxor_search.cpp:311)
//   * src/main/xorfilter.hpp + hashutil.hpp  the in-repo XOR-filter prototype: NOT linked into the reference's search
//     (which uses the un-vendored seqan3::interleaved_xor_filter) but the evidence the IXF restatement rests on --
//     murmur64(key + seed), rotl64 by 21*i, multiply-shift reduction, 8-bit fingerprint, 32 + 1.23 n slots.
// Everything else on the path (syncmer.cpp, hierarchical_interleaved_xor_filter.hpp, threshold.hpp, taxor_search.cpp)
// includes the absent third-party headers and cannot be built here.
#include <climits>
#include <cstddef>
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "syncmer_model.hpp"
#include "kmer_model.hpp"
#include "fracminhash_model.hpp"
#include "gaussian_inverse.hpp"
#include "adjust_seed.hpp"
#include "xorfilter.hpp"
#include "do_parallel.hpp"
#include "sync_out.hpp"

#include <string>
#include <thread>

using Proto = xorfilter::XorFilter<uint64_t, uint8_t>;

extern "C" {

double ref_syncmer_match_ratio(size_t kmer_size, double error_rate)
{
    return hixf::threshold::get_min_syncmer_match_ratio(kmer_size, error_rate);
}

size_t ref_nmut_kmer_ci_high(double r, size_t kmer_size, size_t kmer_count, double confidence)
{
    return hixf::threshold::calculate_nmut_kmer_CI(r, kmer_size, kmer_count, confidence).second;
}

double ref_containment_index_ci_low(double r, size_t kmer_size, size_t kmer_count, double scaling_factor, double confidence)
{
    return hixf::threshold::calculate_containment_index_CI(r, kmer_size, kmer_count, scaling_factor, confidence).first;
}

double ref_normal_cdf_inverse(double p) { return hixf::threshold::NormalCDFInverse(p); }

uint64_t ref_adjust_seed(uint8_t kmer_size) { return hixf::adjust_seed(kmer_size); }

// the prototype filter over `n` keys; returns 0 if its construction failed
void *ref_xor_build(const uint64_t *keys, size_t n, uint64_t *seed, uint64_t *block_length, uint64_t *array_length)
{
    auto *f = new Proto(n);
    if (f->AddAll(keys, 0, n) != xorfilter::Ok) {
        delete f;
        return nullptr;
    }
    *seed = f->hasher->seed;
    *block_length = f->blockLength;
    *array_length = f->arrayLength;
    return f;
}

int ref_xor_contain(const void *h, uint64_t key) { return static_cast<const Proto *>(h)->Contain(key) == xorfilter::Ok; }

const uint8_t *ref_xor_fingerprints(const void *h) { return static_cast<const Proto *>(h)->fingerprints; }

void ref_xor_probe(const void *h, uint64_t key, uint64_t rows[3], uint8_t *fp)
{
    const Proto *f = static_cast<const Proto *>(h);
    const uint64_t hash = (*f->hasher)(key);
    for (int i = 0; i < 3; ++i) rows[i] = xorfilter::getHashFromHash(hash, i, (int)f->blockLength);
    *fp = f->fingerprint(hash);
}

void ref_xor_free(void *h) { delete static_cast<Proto *>(h); }

// The reference's chunk loop (taxor_search.cpp:315-326) with the reference's own scheduler: the batch is cut into chunks of
// `chunk` records (views::chunk(1024) there) and hixf::do_parallel runs `worker` over each chunk's slices.  worker(ctx, start,
// end) gets GLOBAL record indices (the reference's worker indexes the chunk's own `records` vector; the base is added here).
// Returns the number of worker invocations; *compute_time accumulates like the reference's (do_parallel.hpp:20,34-35).
uint64_t ref_do_parallel_chunks(void (*worker)(void *, uint64_t, uint64_t), void *ctx, uint64_t n_records, uint64_t chunk, size_t threads,
                                double *compute_time)
{
    uint64_t calls = 0;
    if (chunk == 0) chunk = 1024;
    for (uint64_t base = 0; base < n_records; base += chunk) {
        const size_t n = (size_t)(n_records - base < chunk ? n_records - base : chunk);
        hixf::do_parallel([=](size_t start, size_t end) { worker(ctx, base + start, base + end); }, n, threads, *compute_time);
        calls += threads;
    }
    return calls;
}

// the slices do_parallel hands out for (n, threads): out[2*i], out[2*i+1] = start, end of task i (recorded under a mutex)
void ref_do_parallel_slices(size_t n, size_t threads, uint64_t *out)
{
    std::mutex mu;
    size_t k = 0;
    double t = 0;
    hixf::do_parallel([&](size_t start, size_t end) { std::lock_guard<std::mutex> lk(mu); out[2 * k] = start; out[2 * k + 1] = end; ++k; }, n, threads, t);
}

// hixf::sync_out: `threads` writers, `lines` lines each ("t<thread>:<line>\n") through operator<<; the caller reads the file back
void ref_sync_out_lines(const char *path, int threads, int lines)
{
    hixf::sync_out out{std::filesystem::path{path}};
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&out, t, lines] {
            for (int i = 0; i < lines; ++i) out << ("t" + std::to_string(t) + ":" + std::to_string(i) + "\n");
        });
    for (auto &x : th) x.join();
}

} // extern "C"
