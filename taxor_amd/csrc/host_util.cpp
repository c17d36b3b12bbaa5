// host_util.cpp -- host-side scalars of the path and the tools the tests / bench need around it
// (threshold ratio, classification filter, XOR-filter bin construction, synthetic reads).
#include "../../include/taxor_gpu_tools.h"
#include "ixf_arith.h"

#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// src/hixf/search/syncmer_model.hpp:14-36 -- minimal matching ratios; rows = read accuracy 80..100 %,
// columns = k 12,14,...,30.  (Data of the reference's empirical model; needed for identical thresholds.)
const double kMatchingRatios[21][10] = {
    {0.552077, 0.195989, 0.151428, 0.118475, 0.0946177, 0.0797244, 0.0604658, 0.0480255, 0.0367569, 0.0252911},
    {0.552385, 0.207533, 0.161204, 0.127368, 0.103704, 0.0881939, 0.0689396, 0.0556991, 0.044185, 0.0298818},
    {0.552239, 0.220393, 0.17382, 0.139866, 0.113736, 0.0966358, 0.0783558, 0.0639223, 0.0523452, 0.0389549},
    {0.552682, 0.236329, 0.188152, 0.152267, 0.126191, 0.106106, 0.0876917, 0.0730642, 0.0621864, 0.0489249},
    {0.553172, 0.254091, 0.202686, 0.165344, 0.137087, 0.116649, 0.098822, 0.0831266, 0.0703342, 0.0582562},
    {0.553716, 0.271183, 0.219848, 0.181959, 0.152163, 0.130048, 0.110622, 0.0942414, 0.0810792, 0.0688187},
    {0.554532, 0.292154, 0.240059, 0.199738, 0.168952, 0.144956, 0.122726, 0.105878, 0.0940805, 0.0777557},
    {0.557957, 0.313553, 0.260912, 0.220014, 0.186567, 0.16101, 0.137399, 0.119867, 0.10453, 0.0900014},
    {0.563925, 0.338316, 0.283689, 0.2401, 0.206963, 0.179541, 0.155347, 0.135128, 0.121575, 0.104741},
    {0.568519, 0.364594, 0.310373, 0.267578, 0.231083, 0.20088, 0.174376, 0.153111, 0.139339, 0.120042},
    {0.579726, 0.395595, 0.338947, 0.295287, 0.258713, 0.22876, 0.200759, 0.175309, 0.161306, 0.139616},
    {0.599258, 0.430241, 0.371291, 0.325596, 0.289651, 0.257329, 0.228011, 0.201799, 0.186956, 0.164794},
    {0.611572, 0.468953, 0.410482, 0.363923, 0.325828, 0.293046, 0.26167, 0.235216, 0.216716, 0.192162},
    {0.624341, 0.510411, 0.452122, 0.407016, 0.370022, 0.334601, 0.303413, 0.275232, 0.254563, 0.227871},
    {0.655724, 0.555245, 0.498564, 0.453201, 0.416285, 0.381883, 0.352291, 0.322556, 0.299739, 0.271481},
    {0.694872, 0.608367, 0.552085, 0.509395, 0.471692, 0.437803, 0.405938, 0.377117, 0.354352, 0.325132},
    {0.742071, 0.669034, 0.613738, 0.57366, 0.539215, 0.50832, 0.476855, 0.449152, 0.42683, 0.397277},
    {0.795543, 0.733694, 0.68341, 0.647737, 0.617382, 0.588448, 0.56083, 0.533714, 0.514757, 0.486399},
    {0.853121, 0.802585, 0.763169, 0.733734, 0.708902, 0.684331, 0.660171, 0.637633, 0.621567, 0.596993},
    {0.918163, 0.882314, 0.854479, 0.835831, 0.819643, 0.804269, 0.788526, 0.771895, 0.763059, 0.742114},
    {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0}};

inline uint64_t splitmix64(uint64_t &x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

} // namespace

extern "C" {

double taxor_threshold_ratio(uint32_t kmer_size, double error_rate, double percentage)
{
    if (percentage > 0.0 && percentage <= 1.0) return percentage;           // threshold.hpp:27-32,76-79
    // the asserts at syncmer_model.hpp:40-44 are compiled out of the reference's Release build and the
    // lookup would read out of bounds; reject instead
    if (kmer_size % 2 != 0 || kmer_size < 12 || kmer_size > 30) return -1.0;
    if (!(error_rate >= 0.0) || !(error_rate <= 0.2)) return -1.0;
    const size_t row = (size_t)std::ceil((1.0 - error_rate) * 100.0 - 80.0); // syncmer_model.hpp:47
    const size_t col = kmer_size - 10 - ((kmer_size - 10) / 2) - 1;          // :48
    if (row > 20 || col > 9) return -1.0;
    return kMatchingRatios[row][col];
}

int taxor_threshold_kind(int use_syncmer, uint32_t kmer_size, uint64_t window_size, double percentage)
{
    const uint64_t kmers_per_window = window_size - kmer_size + 1;          // threshold.hpp:26
    if (percentage > 0.0 && percentage <= 1.0) return TAXOR_THR_PERCENTAGE; // :28
    if (use_syncmer) return TAXOR_THR_SYNCMER;                              // :34
    // the `fracminhash` member is always false here (search_arguments.hpp:61-75), so the window decides
    return kmers_per_window == 1 ? TAXOR_THR_KMER : TAXOR_THR_FRACMINHASH;  // :39-47
}

namespace {

// static_cast<size_t>(double) as the reference's stock build performs it.  For very short reads the reference casts
// NaN (sqrt of a negative variance) or a negative bound to size_t, which is undefined in C++; its build sets no
// -march (src/CMakeLists.txt:17-29), so GCC emits the baseline x86-64 cvttsd2si sequence, whose results are spelled
// out here instead of being left to whatever this compiler and this CPU would do.
uint64_t to_size_like_reference(double x)
{
    constexpr double two63 = 9223372036854775808.0;
    constexpr uint64_t indefinite = 0x8000000000000000ull;
    if (std::isnan(x) || x <= -two63) return indefinite;
    if (x < two63) return (uint64_t)(int64_t)x;
    const double y = x - two63;
    return y >= two63 ? 0 : ((uint64_t)(int64_t)y ^ indefinite);
}

double normal_cdf_inverse(double p)                                         // gaussian_inverse.cpp:13-50
{
    auto approx = [](double t) {
        constexpr double c0 = 2.515517, c1 = 0.802853, c2 = 0.010328, d0 = 1.432788, d1 = 0.189269, d2 = 0.001308;
        return t - ((c2 * t + c1) * t + c0) / (((d2 * t + d1) * t + d0) * t + 1.0);
    };
    return p < 0.5 ? -approx(std::sqrt(-2.0 * std::log(p))) : approx(std::sqrt(-2.0 * std::log(1.0 - p)));
}

// moments of the number of mutated k-mers (Blanca et al.), kmer_model.cpp:26-46; the operand order is the
// reference's, because double arithmetic is not associative and the result is truncated to an integer threshold
struct NmutMoments {
    double q, expected, variance;
    NmutMoments(double r, double k, double n)
    {
        q = 1.0 - std::pow(1.0 - r, k);
        expected = n * q;
        variance = n * (1.0 - q) * (q * (2.0 * k + (2.0 / r) - 1.0) - 2.0 * k) + k * (k - 1.0) * std::pow((1.0 - q), 2.0) +
                   (2.0 * (1.0 - q) / (std::pow(r, 2.0))) * ((1.0 + (k - 1.0) * (1.0 - q)) * r - q);
    }
};

} // namespace

uint64_t taxor_threshold_model(int kind, uint64_t count, uint32_t kmer_size, double error_rate, double percentage,
                               double scaling_factor)
{
    const uint64_t fp_correction = (uint64_t)((double)count * 0.0039);      // threshold.hpp:53
    const double n = (double)count, k = (double)kmer_size;
    switch (kind) {
    case TAXOR_THR_SYNCMER: return (uint64_t)(n * taxor_threshold_ratio(kmer_size, error_rate, -1.0)); // :57-61
    case TAXOR_THR_KMER: {                                                  // :62-67, kmer_model.cpp:10-23
        const NmutMoments m(error_rate, k, n);
        const double z = normal_cdf_inverse(1.0 - (1 - 0.95) / 2.0);
        const uint64_t high = to_size_like_reference(std::ceil(n * m.q + z * std::sqrt(m.variance)));
        return count - high - fp_correction;                               // size_t arithmetic: may wrap
    }
    case TAXOR_THR_FRACMINHASH: {                                           // :68-75, fracminhash_model.cpp:8-33
        const NmutMoments m(error_rate, k, n);
        const double z = normal_cdf_inverse(1.0 - (1.0 - 0.95) / 2.0);
        const double term3 = m.variance / std::pow(n, 2);
        const double term2 = n * m.expected - (std::pow(m.expected, 2) + m.variance);
        const double denominator = scaling_factor * std::pow(n, 3) * std::pow(1.0 - std::pow(1.0 - scaling_factor, n), 2);
        const double term1 = (1.0 - scaling_factor) / denominator;
        const double c_low = std::pow((1.0 - error_rate), k) - z * std::sqrt(term1 * term2 + term3);
        return to_size_like_reference(c_low * n) - fp_correction;
    }
    default: return (uint64_t)(n * percentage);                             // :76-79
    }
}

int taxor_threshold_select(const taxor_hixf_view *view, double error_rate, double percentage, taxor_gpu_search_params *prm)
{
    if (!view || !prm) return TAXOR_E_ARG;
    prm->model = (uint32_t)taxor_threshold_kind(view->use_syncmer, view->kmer_size, view->window_size, percentage);
    prm->error_rate = error_rate;
    prm->ratio = 0.0;
    if (prm->model == TAXOR_THR_PERCENTAGE) prm->ratio = percentage;
    else if (prm->model == TAXOR_THR_SYNCMER) {
        prm->ratio = taxor_threshold_ratio(view->kmer_size, error_rate, -1.0);
        if (prm->ratio < 0.0) return TAXOR_E_ARG;
    } else if (!(error_rate > 0.0) || !(error_rate < 1.0)) return TAXOR_E_ARG; // the models divide by r and take log-free powers of 1-r
    return TAXOR_OK;
}

uint64_t taxor_threshold(uint64_t hash_count, double ratio)
{
    return (uint64_t)((double)hash_count * ratio); // threshold.hpp:60
}

void taxor_classify_filter(const uint32_t *count, uint64_t n, uint8_t *keep)
{
    uint64_t max_count = 0; // taxor_search.cpp:275-280
    for (uint64_t i = 0; i < n; ++i)
        if (count[i] > max_count) max_count = count[i];
    for (uint64_t i = 0; i < n; ++i) // :285
        keep[i] = !(static_cast<double>(count[i]) < static_cast<double>(max_count) * 0.8);
}

uint64_t taxor_ixf_seg_len(uint64_t max_bin_elements) { return taxor::ixf_seg_len(max_bin_elements); }

// XOR-filter construction for one bin: peel the 3-uniform hypergraph, assign fingerprints in reverse
// (the algorithm family of src/main/xorfilter.hpp:142-334; queue formulation).  The dense per-row scratch is
// kept per thread and only the touched rows are cleared, so sparse bins of a very tall IXF cost O(n).
int taxor_ixf_build_bin(const uint64_t *keys, uint64_t n, uint64_t seed, uint64_t seg_len, uint8_t *column)
{
    return taxor_ixf_build_bin_arith(keys, n, seed, seg_len, 0, column);
}

int taxor_ixf_build_bin_arith(const uint64_t *keys, uint64_t n, uint64_t seed, uint64_t seg_len, uint32_t arith, uint8_t *column)
{
    const uint64_t rows = 3 * seg_len;
    std::memset(column, 0, rows);
    if (n == 0) return 0;
    if (seg_len == 0 || seg_len > 0x55555555ull) return 1;
    static thread_local std::vector<uint32_t> cnt;
    static thread_local std::vector<uint64_t> xr;
    if (cnt.size() < rows) {
        cnt.assign(rows, 0);
        xr.assign(rows, 0);
    }
    std::vector<uint32_t> touched;
    touched.reserve(3 * n);
    for (uint64_t i = 0; i < n; ++i) {
        const taxor::ixf_probe p = taxor::ixf_probe_key_arith(keys[i], seed, (uint32_t)seg_len, arith);
        for (int j = 0; j < 3; ++j) {
            if (cnt[p.row[j]]++ == 0) touched.push_back(p.row[j]);
            xr[p.row[j]] ^= keys[i];
        }
    }
    std::vector<uint32_t> queue;
    queue.reserve(touched.size());
    for (uint32_t r : touched)
        if (cnt[r] == 1) queue.push_back(r);
    std::vector<uint64_t> st_key;
    std::vector<uint32_t> st_row;
    st_key.reserve(n);
    st_row.reserve(n);
    while (!queue.empty()) {
        const uint32_t r = queue.back();
        queue.pop_back();
        if (cnt[r] != 1) continue;
        const uint64_t key = xr[r];
        st_key.push_back(key);
        st_row.push_back(r);
        const taxor::ixf_probe p = taxor::ixf_probe_key_arith(key, seed, (uint32_t)seg_len, arith);
        for (int j = 0; j < 3; ++j) {
            const uint32_t rr = p.row[j];
            cnt[rr]--;
            xr[rr] ^= key;
            if (cnt[rr] == 1) queue.push_back(rr);
        }
    }
    for (uint32_t r : touched) { // leave the scratch clean for the next call
        cnt[r] = 0;
        xr[r] = 0;
    }
    if (st_key.size() != n) return 1; // not peelable under this seed (or duplicate keys)
    for (size_t i = st_key.size(); i-- > 0;) {
        const taxor::ixf_probe p = taxor::ixf_probe_key_arith(st_key[i], seed, (uint32_t)seg_len, arith);
        uint8_t v = (uint8_t)(p.fp4 & 0xFFu);
        for (int j = 0; j < 3; ++j)
            if (p.row[j] != st_row[i]) v ^= column[p.row[j]];
        column[st_row[i]] = v;
    }
    return 0;
}

int taxor_synth_reads(const char *genomes, const uint64_t *genome_off, uint64_t n_genomes, uint64_t n_reads,
                      uint32_t read_len, double error_rate, double frac_random, double frac_reverse,
                      uint64_t seed, int threads, char *bases, uint64_t cap, uint64_t *offsets, int32_t *origin)
{
    if ((uint64_t)read_len * n_reads > cap) return TAXOR_E_ARG;
    if (threads < 1) threads = 1;
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    auto comp = [](char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; };
    auto work = [&](uint64_t lo, uint64_t hi) {
        for (uint64_t i = lo; i < hi; ++i) {
            // every read gets its own stream: the state is a hash of (seed, i).  (splitmix64 advances its state by a
            // constant, so seeding read i with seed + i * that constant made read i replay read 0's stream i draws
            // later -- error positions were then correlated across reads and the error rate varied along the read.)
            uint64_t s0 = seed ^ (i * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
            uint64_t st = splitmix64(s0) ^ (splitmix64(s0) << 1);
            char *out = bases + i * read_len;
            offsets[i] = i * read_len;
            const double u0 = (double)(splitmix64(st) >> 11) * (1.0 / 9007199254740992.0);
            if (n_genomes == 0 || u0 < frac_random) {
                if (origin) origin[i] = -1;
                for (uint32_t j = 0; j < read_len; ++j) out[j] = ACGT[splitmix64(st) & 3];
                continue;
            }
            const uint64_t g = splitmix64(st) % n_genomes;
            if (origin) origin[i] = (int32_t)g;
            const char *G = genomes + genome_off[g];
            const uint64_t glen = genome_off[g + 1] - genome_off[g];
            const uint64_t span = glen > read_len ? glen - read_len : 0;
            const uint64_t start = span ? splitmix64(st) % span : 0;
            const bool rev = (double)(splitmix64(st) >> 11) * (1.0 / 9007199254740992.0) < frac_reverse;
            uint64_t src = 0; // offset within the sampled stretch
            uint32_t j = 0;
            while (j < read_len) {
                char c;
                if (!rev) {
                    const uint64_t pos = start + src;
                    c = pos < glen ? G[pos] : ACGT[splitmix64(st) & 3];
                } else { // walk the stretch [start, start+read_len) backwards, complemented
                    const int64_t pos = (int64_t)start + (int64_t)read_len - 1 - (int64_t)src;
                    c = (pos >= 0 && (uint64_t)pos < glen) ? comp(G[pos]) : ACGT[splitmix64(st) & 3];
                }
                ++src;
                const double x = (double)(splitmix64(st) >> 11) * (1.0 / 9007199254740992.0);
                if (x < error_rate * 0.4) out[j++] = ACGT[splitmix64(st) & 3];          // substitution
                else if (x < error_rate * 0.7) {                                           // insertion
                    out[j++] = c;
                    if (j < read_len) out[j++] = ACGT[splitmix64(st) & 3];
                } else if (x < error_rate) continue;                                       // deletion
                else out[j++] = c;
            }
        }
    };
    std::vector<std::thread> pool;
    const uint64_t per = (n_reads + (uint64_t)threads - 1) / (uint64_t)threads;
    for (int t = 0; t < threads; ++t) {
        const uint64_t lo = per * (uint64_t)t, hi = lo + per > n_reads ? n_reads : lo + per;
        if (lo < hi) pool.emplace_back(work, lo, hi);
    }
    for (auto &th : pool) th.join();
    offsets[n_reads] = n_reads * (uint64_t)read_len;
    return 0;
}

} // extern "C"
