/*
 * taxor_oracle.c -- CPU ORACLE (test infrastructure, see taxor_oracle.h for scope and pinning status).
 * Plain C restatement of the reference `taxor search` hot path; citations are file:line into
 * /root/reference (JensUweUlrich/Taxor @ 2025-05-23).
 */
#include "taxor_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* =============================================================================================== */
/* hashing                                                                                         */
/* =============================================================================================== */

/* martinus/unordered_dense v3.0.1, detail::wyhash: mum() = 128-bit product split into lo/hi,
 * mix(a,b) = lo ^ hi, hash(uint64_t x) = mix(x, 0x9E3779B97F4A7C15).  [un-vendored; published algorithm] */
uint64_t orc_wyhash_u64(uint64_t x)
{
    __uint128_t r = (__uint128_t)x * (__uint128_t)UINT64_C(0x9E3779B97F4A7C15);
    return (uint64_t)r ^ (uint64_t)(r >> 64);
}

int orc_dna4_normalise(char *seq, size_t len)
{
    /* rank_to_char of dna4 is "ACGT"; char_to_rank sends IUPAC codes to their first base, U to T and
     * every other legal character (N) to rank 0 = 'A'.  [RECALL seqan3 alphabet/nucleotide/dna4.hpp] */
    static char map[256];
    static int init = 0;
    if (!init) {
        memset(map, 0, sizeof map);
        const char *from = "ACGTURYSWKMBDHVN";
        const char *to   = "ACGTTACCAGACAAAA";
        for (int i = 0; from[i]; ++i) {
            map[(unsigned char)from[i]] = to[i];
            map[(unsigned char)(from[i] + 32)] = to[i];
        }
        init = 1;
    }
    for (size_t i = 0; i < len; ++i) {
        char m = map[(unsigned char)seq[i]];
        if (!m) return -1;
        seq[i] = m;
    }
    return 0;
}

/* syncmer.cpp:53-70 -- A/a 0, C/c 1, G/g 2, T/t/U/u 3, bytes 0..3 map to themselves, else 4 */
static unsigned char nt4(unsigned char c)
{
    switch (c) {
    case 0: case 'A': case 'a': return 0;
    case 1: case 'C': case 'c': return 1;
    case 2: case 'G': case 'g': return 2;
    case 3: case 'T': case 't': case 'U': case 'u': return 3;
    default: return 4;
    }
}

/* insertion-ordered set of uint64 (ankerl::unordered_dense::set<size_t> keeps a dense vector in
 * insertion order; taxor_search.cpp:236 copies it out front to back) */
typedef struct {
    uint64_t *dense;
    size_t n, dense_cap;
    uint32_t *slots; /* index+1 into dense, 0 = empty */
    size_t mask;
} oset;

static void oset_init(oset *s)
{
    s->dense_cap = 256;
    s->dense = (uint64_t *)malloc(s->dense_cap * sizeof(uint64_t));
    s->n = 0;
    s->mask = 1023;
    s->slots = (uint32_t *)calloc(s->mask + 1, sizeof(uint32_t));
}

static void oset_free(oset *s)
{
    free(s->dense);
    free(s->slots);
}

static size_t oset_slot(uint64_t v, size_t mask)
{
    v ^= v >> 29;
    v *= UINT64_C(0xBF58476D1CE4E5B9);
    v ^= v >> 32;
    return (size_t)v & mask;
}

static void oset_insert(oset *s, uint64_t v)
{
    size_t p = oset_slot(v, s->mask);
    while (s->slots[p]) {
        if (s->dense[s->slots[p] - 1] == v) return;
        p = (p + 1) & s->mask;
    }
    if (s->n == s->dense_cap) {
        s->dense_cap *= 2;
        s->dense = (uint64_t *)realloc(s->dense, s->dense_cap * sizeof(uint64_t));
    }
    s->dense[s->n++] = v;
    s->slots[p] = (uint32_t)s->n;
    if (s->n * 2 > s->mask) { /* grow + rehash */
        size_t nm = (s->mask + 1) * 4 - 1;
        uint32_t *ns = (uint32_t *)calloc(nm + 1, sizeof(uint32_t));
        for (size_t i = 0; i < s->n; ++i) {
            size_t q = oset_slot(s->dense[i], nm);
            while (ns[q]) q = (q + 1) & nm;
            ns[q] = (uint32_t)(i + 1);
        }
        free(s->slots);
        s->slots = ns;
        s->mask = nm;
    }
}

/* make_string_to_hashvalues_open_syncmers_canonical, syncmer.cpp:80-155.
 * Positions are size_t in the reference and wrap (qs_min_pos = -1, i - k with i < k); uint64_t
 * arithmetic reproduces that exactly. */
static void syncmers_into(const char *seq, size_t len, uint64_t k, uint64_t s, uint64_t t, oset *set)
{
    const uint64_t kmask = (k < 32) ? ((UINT64_C(1) << (2 * k)) - 1) : UINT64_MAX;   /* :86 */
    const uint64_t smask = (UINT64_C(1) << (2 * s)) - 1;                               /* :87 */
    const uint64_t kshift = (k - 1) * 2, sshift = (s - 1) * 2;                        /* :88-89 */
    const uint64_t w = k - s + 1;
    uint64_t *qs = (uint64_t *)malloc((w + 2) * sizeof(uint64_t)); /* std::deque, :90 */
    size_t q_head = 0, q_size = 0;
    const size_t q_cap = (size_t)w + 2;
    uint64_t qs_min_val = UINT64_MAX;                                                  /* :91 */
    uint64_t qs_min_pos = (uint64_t)-1;                                                /* :92 */
    uint64_t l = 0, xk[2] = {0, 0}, xs[2] = {0, 0};                                    /* :94-96 */

    for (uint64_t i = 0; i < len; ++i) {
        unsigned c = nt4((unsigned char)seq[i]);                                       /* :99 */
        if (c < 4) {
            xk[0] = (xk[0] << 2 | c) & kmask;                                          /* :101 */
            xk[1] = xk[1] >> 2 | (uint64_t)(3 - c) << kshift;                          /* :102 */
            xs[0] = (xs[0] << 2 | c) & smask;                                          /* :103 */
            xs[1] = xs[1] >> 2 | (uint64_t)(3 - c) << sshift;                          /* :104 */
            if (++l < s) continue;                                                     /* :105 */
            uint64_t ys = xs[0] < xs[1] ? xs[0] : xs[1];                               /* :109 */
            qs[(q_head + q_size) % q_cap] = ys;                                        /* :111 */
            ++q_size;
            if (q_size < w) continue;                                                  /* :113 */
            if (q_size == w) {                                                         /* :116 */
                for (uint64_t j = 0; j < q_size; ++j) {
                    uint64_t v = qs[(q_head + j) % q_cap];
                    if (v < qs_min_val) {                                              /* :118 */
                        qs_min_val = v;
                        qs_min_pos = i - k + j + 1;
                    }
                }
            } else {
                q_head = (q_head + 1) % q_cap;                                         /* :126 */
                --q_size;
                if (qs_min_pos == i - k) {                                             /* :128 */
                    qs_min_val = UINT64_MAX;
                    qs_min_pos = i - s + 1;
                    for (int64_t j = (int64_t)q_size - 1; j >= 0; --j) {               /* :131 */
                        uint64_t v = qs[(q_head + (size_t)j) % q_cap];
                        if (v < qs_min_val) {
                            qs_min_val = v;
                            qs_min_pos = i - k + (uint64_t)j + 1;
                        }
                    }
                } else if (ys < qs_min_val) {                                          /* :137 */
                    qs_min_val = ys;
                    qs_min_pos = i - s + 1;
                }
            }
            if (qs_min_pos == i - k + t) {                                             /* :142 */
                uint64_t yk = xk[0] < xk[1] ? xk[0] : xk[1];
                oset_insert(set, orc_wyhash_u64(yk));                                  /* :145 */
            }
        } else {                                                                       /* :147-153 */
            qs_min_val = UINT64_MAX;
            qs_min_pos = (uint64_t)-1;
            l = xs[0] = xs[1] = xk[0] = xk[1] = 0;
            q_head = q_size = 0;
        }
    }
    free(qs);
}

size_t orc_seq_to_syncmers(const char *seq, size_t len, int k, int s, int t, uint64_t *out, size_t cap)
{
    oset set;
    oset_init(&set);
    syncmers_into(seq, len, (uint64_t)k, (uint64_t)s, (uint64_t)t, &set);
    size_t n = set.n;
    memcpy(out, set.dense, (n < cap ? n : cap) * sizeof(uint64_t));
    oset_free(&set);
    return n;
}

/* =============================================================================================== */
/* thresholds                                                                                      */
/* =============================================================================================== */

/* syncmer_model.hpp:14-36 -- rows: read accuracy 80..100 %, columns: k = 12,14,...,30 (data table) */
static const double matching_ratios[21][10] = {
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

double orc_syncmer_match_ratio(size_t kmer_size, double error_rate)
{
    /* asserts at syncmer_model.hpp:40-44 are compiled out in the reference's Release build; fence them */
    if (kmer_size % 2 != 0 || kmer_size < 12 || kmer_size > 30) return -1.0;
    if (!(error_rate >= 0.0) || !(error_rate <= 0.2)) return -1.0;
    size_t row_index = (size_t)ceil((1.0 - error_rate) * 100.0 - 80.0);               /* :47 */
    size_t col_index = kmer_size - 10 - ((kmer_size - 10) / 2) - 1;                    /* :48 */
    if (row_index > 20 || col_index > 9) return -1.0;
    return matching_ratios[row_index][col_index];                                      /* :49 */
}

size_t orc_threshold(size_t hash_count, size_t kmer_size, double error_rate, double percentage)
{
    if (percentage > 0.0 && percentage <= 1.0)                                         /* threshold.hpp:27 */
        return (size_t)((double)hash_count * percentage);                              /* :76-79 */
    double ratio = orc_syncmer_match_ratio(kmer_size, error_rate);                     /* :59 */
    return (size_t)((double)hash_count * ratio);                                       /* :60 */
}


/* ---- k-mer and FracMinHash threshold models (used for indexes built without --use-syncmer) ------------- */

/* static_cast<size_t>(double) as the reference's stock build executes it.  Converting a negative or NaN double to
 * size_t is undefined in C++; the reference does it for very short reads (variance < 0 -> sqrt = NaN, or a negative
 * containment bound).  Its build uses no -march flag (src/CMakeLists.txt:17-29), so GCC emits the baseline x86-64
 * sequence around cvttsd2si; this function states that sequence's result explicitly so that the oracle does not
 * depend on the machine it is compiled for (AVX-512's vcvttsd2usi would return all-ones instead). */
static size_t orc_f64_to_size(double x)
{
    const double two63 = 9223372036854775808.0;
    if (x != x) return (size_t)UINT64_C(0x8000000000000000);                           /* NaN: "integer indefinite" */
    if (x >= two63) {
        double y = x - two63;
        if (y >= two63) return 0;                                                      /* indefinite ^ sign bit */
        return (size_t)((uint64_t)(int64_t)y ^ UINT64_C(0x8000000000000000));
    }
    if (x <= -two63) return (size_t)UINT64_C(0x8000000000000000);
    return (size_t)(uint64_t)(int64_t)x;                                               /* truncates toward zero; negatives wrap */
}

/* RationalApproximation / NormalCDFInverse, gaussian_inverse.cpp:13-50 (Abramowitz-Stegun 26.2.23).
 * The reference throws for p outside (0,1); the callers below only pass 0.975. */
static double orc_rational_approximation(double t)
{
    const double c[] = {2.515517, 0.802853, 0.010328};
    const double d[] = {1.432788, 0.189269, 0.001308};
    return t - ((c[2] * t + c[1]) * t + c[0]) / (((d[2] * t + d[1]) * t + d[0]) * t + 1.0);
}

double orc_normal_cdf_inverse(double p)
{
    if (p < 0.5) return -orc_rational_approximation(sqrt(-2.0 * log(p)));              /* :41-44 */
    return orc_rational_approximation(sqrt(-2.0 * log(1.0 - p)));                      /* :45-49 */
}

/* variance_nmut_kmer / expected_nmut_kmer / expected_nmut_kmer_squared, kmer_model.cpp:26-46; the
 * expression order is the reference's (double arithmetic is not associative). */
static double orc_expected_nmut_kmer(double r, size_t kmer_size, size_t kmer_count)
{
    double q = 1.0 - pow(1.0 - r, kmer_size);
    return kmer_count * q;
}

static double orc_variance_nmut_kmer(double r, size_t kmer_size, size_t kmer_count)
{
    double q = 1.0 - pow(1.0 - r, kmer_size);
    double varN = (double)kmer_count * (1.0 - q) * (q * (2.0 * (double)kmer_size + (2.0 / r) - 1.0) - 2.0 * (double)kmer_size)
                  + (double)kmer_size * ((double)kmer_size - 1.0) * pow((1.0 - q), 2.0)
                  + (2.0 * (1.0 - q) / (pow(r, 2.0))) * ((1.0 + ((double)kmer_size - 1.0) * (1.0 - q)) * r - q);
    return varN;
}

static double orc_expected_nmut_kmer_squared(double r, size_t kmer_size, size_t kmer_count)
{
    return pow(orc_expected_nmut_kmer(r, kmer_size, kmer_count), 2) + orc_variance_nmut_kmer(r, kmer_size, kmer_count);
}

/* calculate_nmut_kmer_CI(...).second, kmer_model.cpp:10-23 (only the upper bound is used, threshold.hpp:65) */
size_t orc_nmut_kmer_ci_high(double r, size_t kmer_size, size_t kmer_count, double confidence)
{
    double q = 1.0 - pow(1.0 - r, kmer_size);
    double varN = orc_variance_nmut_kmer(r, kmer_size, kmer_count);                    /* same expression, :13-15 */
    double alpha = 1 - confidence;
    double z = orc_normal_cdf_inverse(1.0 - alpha / 2.0);
    return orc_f64_to_size(ceil(kmer_count * q + z * sqrt(varN)));                     /* :19 */
}

/* calculate_containment_index_CI(...).first, fracminhash_model.cpp:8-33 */
double orc_containment_index_ci_low(double r, size_t kmer_size, size_t kmer_count, double scaling_factor, double confidence)
{
    double z_alpha = orc_normal_cdf_inverse(1.0 - (1.0 - confidence) / 2.0);
    double term3 = orc_variance_nmut_kmer(r, kmer_size, kmer_count) / pow(kmer_count, 2);
    double term2 = kmer_count * orc_expected_nmut_kmer(r, kmer_size, kmer_count) - orc_expected_nmut_kmer_squared(r, kmer_size, kmer_count);
    double denominator = scaling_factor * pow(kmer_count, 3) * pow(1.0 - pow(1.0 - scaling_factor, kmer_count), 2);
    double term1 = (1.0 - scaling_factor) / denominator;
    double var = term1 * term2 + term3;
    return pow((1.0 - r), kmer_size) - z_alpha * sqrt(var);
}

int orc_threshold_kind(int use_syncmer, size_t kmer_size, size_t window_size, double percentage)
{
    size_t kmers_per_window = window_size - kmer_size + 1;                             /* threshold.hpp:26 */
    if (percentage > 0.0 && percentage <= 1.0) return ORC_THR_PERCENTAGE;              /* :28 */
    if (use_syncmer) return ORC_THR_SYNCMER;                                           /* :34 */
    if (kmers_per_window == 1) return ORC_THR_KMER;    /* :39; `fracminhash` is always false, search_arguments.hpp:72 */
    return ORC_THR_FRACMINHASH;                                                        /* :44 */
}

size_t orc_threshold_model(int kind, size_t minimiser_count, size_t kmer_size, double error_rate, double percentage,
                           double scaling_factor)
{
    size_t fp_correction = (size_t)((double)minimiser_count * 0.0039);                 /* :53 */
    switch (kind) {
    case ORC_THR_SYNCMER:
        return (size_t)((double)minimiser_count * orc_syncmer_match_ratio(kmer_size, error_rate)); /* :57-61 */
    case ORC_THR_KMER:
        /* size_t arithmetic: wraps (to a threshold no bin can reach) when the bound exceeds the count, :64-66 */
        return minimiser_count - orc_nmut_kmer_ci_high(error_rate, kmer_size, minimiser_count, 0.95) - fp_correction;
    case ORC_THR_FRACMINHASH:
        return orc_f64_to_size(orc_containment_index_ci_low(error_rate, kmer_size, minimiser_count, scaling_factor, 0.95) *
                               (double)minimiser_count) - fp_correction;               /* :68-75 */
    default:
        return (size_t)((double)minimiser_count * percentage);                         /* :76-79 */
    }
}

/* =============================================================================================== */
/* interleaved XOR filter  [un-vendored; restated from xorfilter.hpp + hashutil.hpp evidence]      */
/* =============================================================================================== */

static uint64_t murmur64(uint64_t h) /* hashutil.hpp:50-57 */
{
    h ^= h >> 33;
    h *= UINT64_C(0xff51afd7ed558ccd);
    h ^= h >> 33;
    h *= UINT64_C(0xc4ceb9fe1a85ec53);
    h ^= h >> 33;
    return h;
}

static uint64_t rotl64(uint64_t n, unsigned c) /* xorfilter.hpp:22-28 */
{
    c &= 63;
    return (n << c) | (n >> ((-c) & 63));
}

static uint32_t reduce32(uint32_t hash, uint32_t n) /* xorfilter.hpp:36-39 */
{
    return (uint32_t)(((uint64_t)hash * n) >> 32);
}

uint64_t orc_ixf_seg_len(uint64_t max_bin_elements)
{
    size_t array_len = (size_t)(32 + 1.23 * (double)max_bin_elements);                 /* xorfilter.hpp:67 */
    return array_len / 3;                                                              /* :68 */
}

/* another reading of the un-vendored arithmetic (orc_ixf::arith != 0), decoded field by field */
static void ixf_probe_arith(const orc_ixf *f, uint64_t key, uint64_t rows[3], uint8_t *fp)
{
    const unsigned a = f->arith, kh = a & 3u, sm = (a >> 2) & 3u, red = (a >> 4) & 3u, fpm = (a >> 6) & 3u, rot = ((a >> 8) & 0xFFu) ^ 21u;
    uint64_t x = key, h;
    if (sm == 0) x = key + f->seed;
    else if (sm == 1) x = key ^ f->seed;
    if (kh == 0) h = murmur64(x);
    else if (kh == 1) h = x;
    else if (kh == 2) h = orc_wyhash_u64(x);
    else {
        uint64_t z = x + UINT64_C(0x9E3779B97F4A7C15);
        z = (z ^ (z >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)) * UINT64_C(0x94D049BB133111EB);
        h = z ^ (z >> 31);
    }
    if (sm == 2) h += f->seed;
    *fp = fpm == 0 ? (uint8_t)(h ^ (h >> 32)) : fpm == 1 ? (uint8_t)h : fpm == 2 ? (uint8_t)(h >> 56) : (uint8_t)(h >> 32);
    for (int i = 0; i < 3; ++i) {
        const uint64_t r = rotl64(h, rot * (unsigned)i);
        uint64_t row;
        if (red == 0) row = ((uint64_t)(uint32_t)r * f->seg_len) >> 32;
        else if (red == 1) row = (uint64_t)(uint32_t)r % f->seg_len;
        else row = (uint64_t)(((unsigned __int128)r * f->seg_len) >> 64);
        rows[i] = row + (uint64_t)i * f->seg_len;
    }
}

void orc_ixf_probe(const orc_ixf *f, uint64_t key, uint64_t rows[3], uint8_t *fp)
{
    if (f->arith) { ixf_probe_arith(f, key, rows, fp); return; }
    uint64_t hash = murmur64(key + f->seed);                                           /* hashutil.hpp:59-61 */
    *fp = (uint8_t)(hash ^ (hash >> 32));                                              /* xorfilter.hpp:60-62 */
    for (int i = 0; i < 3; ++i) {                                                      /* :42-45,340-347 */
        uint32_t r = (uint32_t)rotl64(hash, (unsigned)(21 * i));
        rows[i] = (uint64_t)reduce32(r, (uint32_t)f->seg_len) + (uint64_t)i * f->seg_len;
    }
}

void orc_ixf_bulk_count(const orc_ixf *f, const uint64_t *hashes, size_t n, uint32_t *counts)
{
    memset(counts, 0, f->bins * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
        uint64_t rows[3];
        uint8_t fp;
        orc_ixf_probe(f, hashes[i], rows, &fp);
        const uint8_t *r0 = f->data + rows[0] * f->stride;
        const uint8_t *r1 = f->data + rows[1] * f->stride;
        const uint8_t *r2 = f->data + rows[2] * f->stride;
        for (uint64_t j = 0; j < f->bins; ++j)
            counts[j] += (uint32_t)(fp == (uint8_t)(r0[j] ^ r1[j] ^ r2[j]));           /* xorfilter.hpp:348-349 */
    }
}

/* Checker of BUILT filters (the GPU builder's full-size test): the synthetic keys first .. first + n - 1 (a bijection of i + salt,
 * the splitmix64 finaliser -- the same function the device generates them with, taxor_amd/csrc/ixf_arith.h synth_key) against
 * filter f.  Every key is looked up in column `bin` with orc_ixf_bulk_count's rule (fp == D[h0][bin]^D[h1][bin]^D[h2][bin]); the
 * return value is the number found there -- n when no key is missing.  Every sample_step-th key (sample_step > 0) also goes
 * through orc_ixf_bulk_count itself, over all bins: counts[bins] accumulates (the caller zeroes it), *sampled the keys that did. */
static uint64_t orc_synth_key(uint64_t i, uint64_t salt)
{
    uint64_t z = i + salt;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_keys(uint64_t first, uint64_t n, uint64_t salt, uint64_t *out)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = orc_synth_key(first + i, salt);
}

uint64_t orc_ixf_synth_keys_found(const orc_ixf *f, uint64_t bin, uint64_t first, uint64_t n, uint64_t salt, uint64_t sample_step,
                                  uint64_t *counts, uint64_t *sampled, int threads)
{
    uint64_t found = 0, n_sampled = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : found, n_sampled)
    {
        uint32_t *mine = counts ? (uint32_t *)calloc(f->bins ? f->bins : 1, sizeof(uint32_t)) : NULL;
        uint32_t *one = counts ? (uint32_t *)malloc((f->bins ? f->bins : 1) * sizeof(uint32_t)) : NULL;
        /* (a 4096-bin root is tens of GB: three cache misses per key.  Keys go in groups of 32 -- probes first, with a prefetch of the
         *  three bytes each, then the comparisons -- so that the misses overlap; the rule applied is the one above) */
#pragma omp for schedule(static)
        for (uint64_t g = 0; g < (n + 31) / 32; ++g) {
            uint64_t keys[32], rows[32][3];
            uint8_t fps[32];
            const uint64_t i0 = g * 32, m = n - i0 < 32 ? n - i0 : 32;
            for (uint64_t q = 0; q < m; ++q) {
                keys[q] = orc_synth_key(first + i0 + q, salt);
                orc_ixf_probe(f, keys[q], rows[q], &fps[q]);
                for (int r = 0; r < 3; ++r) __builtin_prefetch(f->data + rows[q][r] * f->stride + bin, 0, 0);
            }
            for (uint64_t q = 0; q < m; ++q) {
                const uint8_t v = (uint8_t)(f->data[rows[q][0] * f->stride + bin] ^ f->data[rows[q][1] * f->stride + bin] ^ f->data[rows[q][2] * f->stride + bin]);
                found += (uint64_t)(v == fps[q]);
                if (counts && sample_step && (i0 + q) % sample_step == 0) {
                    orc_ixf_bulk_count(f, &keys[q], 1, one);
                    for (uint64_t j = 0; j < f->bins; ++j) mine[j] += one[j];
                    ++n_sampled;
                }
            }
        }
        if (counts) {
#pragma omp critical
            for (uint64_t j = 0; j < f->bins; ++j) counts[j] += mine[j];
        }
        free(mine);
        free(one);
    }
    if (sampled) *sampled += n_sampled;
    return found;
}

/* =============================================================================================== */
/* hierarchical IXF traversal -- hierarchical_interleaved_xor_filter.hpp:303-340                   */
/* =============================================================================================== */

typedef struct {
    int64_t *ub;
    uint32_t *cnt;
    size_t n, cap;
    uint64_t bytes;
} resbuf;

/* The reference constructs a fresh `bins`-sized counting vector per visited IXF per read (:307).  The port keeps one per
 * recursion depth and thread and reuses it: same values, no allocator traffic -- at 256 threads the malloc/free pairs of
 * the literal form serialise in the allocator and the all-cores baseline measured the allocator, not the path. */
#define ORC_MAX_DEPTH 64
static _Thread_local uint32_t *tl_counts[ORC_MAX_DEPTH];
static _Thread_local size_t tl_counts_cap[ORC_MAX_DEPTH];

static uint32_t *counts_at_depth(int depth, size_t bins)
{
    if (depth >= ORC_MAX_DEPTH) return (uint32_t *)malloc((bins ? bins : 1) * sizeof(uint32_t));
    if (tl_counts_cap[depth] < bins || !tl_counts[depth]) {
        free(tl_counts[depth]);
        tl_counts_cap[depth] = bins ? bins : 1;
        tl_counts[depth] = (uint32_t *)malloc(tl_counts_cap[depth] * sizeof(uint32_t));
    }
    return tl_counts[depth];
}

static void bulk_contains_rec(const orc_hixf *h, const uint64_t *hashes, size_t n, int64_t ixf_idx,
                              size_t threshold, resbuf *rb, int depth);

static void bulk_contains_impl(const orc_hixf *h, const uint64_t *hashes, size_t n, int64_t ixf_idx,
                               size_t threshold, resbuf *rb)
{
    bulk_contains_rec(h, hashes, n, ixf_idx, threshold, rb, 0);
}

static void bulk_contains_rec(const orc_hixf *h, const uint64_t *hashes, size_t n, int64_t ixf_idx,
                              size_t threshold, resbuf *rb, int depth)
{
    const orc_ixf *f = &h->ixf[ixf_idx];
    uint32_t *result = counts_at_depth(depth, f->bins);                                /* :307 */
    orc_ixf_bulk_count(f, hashes, n, result);                                          /* :309 */
    rb->bytes += (uint64_t)n * 3u * f->bins;
    uint32_t sum = 0;                                                                  /* :310 */
    const int64_t *fname = h->fname_idx[ixf_idx];
    for (size_t bin = 0; bin < f->bins; ++bin) {                                       /* :313 */
        sum += result[bin];                                                            /* :315 */
        int64_t cur = fname[bin];                                                      /* :317 */
        if (cur < 0) {                                                                 /* :319 merged bin */
            if ((size_t)sum >= threshold)                                              /* :321 */
                bulk_contains_rec(h, hashes, n, h->next_ixf[ixf_idx][bin], threshold, rb, depth + 1);
            sum = 0u;
        } else if (bin + 1u == f->bins || cur != fname[bin + 1]) {                     /* :325-326 */
            if ((size_t)sum >= threshold) {                                            /* :328 */
                if (rb->n < rb->cap) {
                    rb->ub[rb->n] = cur;                                               /* :330 */
                    rb->cnt[rb->n] = sum;
                }
                rb->n++;
            }
            sum = 0u;
        }
    }
    if (depth >= ORC_MAX_DEPTH) free(result);
}

size_t orc_bulk_contains(const orc_hixf *h, const uint64_t *hashes, size_t n, size_t threshold,
                         int64_t *user_bin, uint32_t *count, size_t cap, uint64_t *visited_bytes)
{
    resbuf rb = {user_bin, count, 0, cap, 0};
    bulk_contains_impl(h, hashes, n, 0, threshold, &rb);                               /* :391 */
    if (visited_bytes) *visited_bytes = rb.bytes;
    return rb.n;
}

/* =============================================================================================== */
/* per-read driver -- taxor_search.cpp:196-313                                                     */
/* =============================================================================================== */


/* ---- minimiser / k-mer hashing (indexes built without --use-syncmer) -------------------------------------
 * seqan3::views::minimiser_hash(ungapped{k}, window_size{w}, seed{adjust_seed(k)}), call sites
 * taxor_search.cpp:210-212,241-256 and taxor_build.cpp:316-318.  The view lives in the un-vendored seqan3 fork:
 * restated from the published seqan3 3.x algorithm -- parity unpinned for w > k (which of several equal minima
 * is kept); for w == k every canonical k-mer value is emitted and no tie rule is involved.
 *   value of k-mer i   = min(fwd_i ^ seed, rc_i ^ seed), fwd/rc = 2-bit rank encoding (A0 C1 G2 T3) of the k-mer
 *                        and of its reverse complement (kmer_hash of the complemented, reversed text)
 *   window             = w-k+1 consecutive k-mer values (all of them if the read has fewer)
 *   first window       : the RIGHTMOST minimum is the minimiser (min_element with less_equal) -> emitted
 *   each shift         : minimiser left the window -> rightmost minimum of the new window, emitted;
 *                        else new value < minimiser -> it becomes the minimiser, emitted; else nothing. */
uint64_t orc_adjust_seed(int k) { return UINT64_C(0x8F3F73B5CF1C9ADE) >> (64u - 2u * (unsigned)k); } /* adjust_seed.hpp:40-44 */

size_t orc_minimiser_hash(const char *seq, size_t len, int k, int w, uint64_t *out, size_t cap)
{
    if (k < 1 || k > 32 || w < k || len < (size_t)k) return 0;
    const uint64_t seed = orc_adjust_seed(k);
    const uint64_t mask = k == 32 ? ~UINT64_C(0) : ((UINT64_C(1) << (2 * k)) - 1);
    const unsigned shift = 2u * (unsigned)(k - 1);
    const size_t nk = len - (size_t)k + 1;
    uint64_t *val = (uint64_t *)malloc(nk * sizeof(uint64_t));
    uint64_t f = 0, r = 0;
    for (size_t i = 0; i < len; ++i) {
        uint64_t c = nt4((unsigned char)seq[i]) & 3u;           /* reads are dna4 here (dna4_traits.hpp:15-18) */
        f = ((f << 2) | c) & mask;
        r = (r >> 2) | ((3u - c) << shift);
        if (i + 1 >= (size_t)k) {
            uint64_t a = f ^ seed, b = r ^ seed;
            val[i + 1 - (size_t)k] = a < b ? a : b;
        }
    }
    size_t W = (size_t)(w - k + 1);
    if (W > nk) W = nk;
    size_t n = 0;
    /* first window: rightmost minimum */
    size_t pos = 0;
    for (size_t j = 1; j < W; ++j)
        if (val[j] <= val[pos]) pos = j;
    if (n < cap) out[n] = val[pos];
    ++n;
    for (size_t j = 1; j + W <= nk; ++j) {                      /* window j = val[j .. j+W-1] */
        const size_t newest = j + W - 1;
        int emit = 0;
        if (pos < j) {                                          /* minimiser_position_offset == 0 before the shift */
            pos = j;
            for (size_t x = j + 1; x <= newest; ++x)
                if (val[x] <= val[pos]) pos = x;
            emit = 1;
        } else if (val[newest] < val[pos]) {
            pos = newest;
            emit = 1;
        }
        if (emit) {
            if (n < cap) out[n] = val[pos];
            ++n;
        }
    }
    free(val);
    return n;
}

size_t orc_search_read(const orc_hixf *h, const orc_search_params *p, const char *seq, size_t len,
                       uint32_t *n_hashes, int64_t *user_bin, uint32_t *count, size_t cap,
                       uint64_t *visited_bytes)
{
    if (p->window > 0) {                                                               /* !compute_syncmer, :239-260 */
        size_t capn = len + 1, m = 0;
        uint64_t *hs = (uint64_t *)malloc(capn * sizeof(uint64_t));
        size_t cnt = orc_minimiser_hash(seq, len, p->k, p->window, hs, capn);
        for (size_t i = 0; i < cnt; ++i) {                                             /* no dedup: every emitted value counts */
            if (p->scaling > 1) {
                uint64_t v = orc_wyhash_u64(hs[i]);
                if (!((double)v <= (double)UINT64_MAX / (double)p->scaling)) continue; /* :243-249 */
            }
            hs[m++] = hs[i];
        }
        size_t hash_count = m;                                                         /* :261 */
        int kind = orc_threshold_kind(0, (size_t)p->k, (size_t)p->window, p->percentage);
        double sf = (double)hash_count / ((double)len - (double)p->k + 1.0);           /* :263 */
        size_t thr = orc_threshold_model(kind, hash_count, (size_t)p->k, p->error_rate, p->percentage, sf);
        size_t nt = orc_bulk_contains(h, hs, hash_count, thr, user_bin, count, cap, visited_bytes);
        if (n_hashes) *n_hashes = (uint32_t)hash_count;
        free(hs);
        return nt;
    }
    oset set;
    oset_init(&set);
    syncmers_into(seq, len, (uint64_t)p->k, (uint64_t)p->s, (uint64_t)p->t, &set);     /* :222 */
    if (p->scaling > 1) {                                                              /* :223-233 */
        size_t m = 0;
        for (size_t i = 0; i < set.n; ++i) {
            uint64_t v = orc_wyhash_u64(set.dense[i]);
            if ((double)v <= (double)UINT64_MAX / (double)p->scaling) set.dense[m++] = set.dense[i];
        }
        set.n = m;
    }
    size_t hash_count = set.n;                                                         /* :261 */
    size_t thr = orc_threshold(hash_count, (size_t)p->k, p->error_rate, p->percentage); /* :263 */
    size_t n = orc_bulk_contains(h, set.dense, hash_count, thr, user_bin, count, cap, visited_bytes); /* :265 */
    if (n_hashes) *n_hashes = (uint32_t)hash_count;
    oset_free(&set);
    return n;
}

/* ---- the chunk loop's worker (taxor_search.cpp:196-313) as a callable over a slice [start, end) of the batch, so that the
 * SCHEDULER can be anybody's: the OpenMP loop of orc_search_batch below, or the reference's own hixf::do_parallel compiled
 * into oracle/_ref (ref_driver.cpp, ref_do_parallel_chunks).  A slice fills a private growing buffer (the reference's per-thread
 * result_string); slices are registered in the context and assembled in read order by orc_batch_finish. */
typedef struct { uint64_t start, end; int64_t *ub; uint32_t *cnt; uint64_t n, cap, vbytes; } orc_slice;
struct orc_batch_ctx {
    const orc_hixf *h;
    const orc_search_params *p;
    const char *bases;
    const uint64_t *offsets;
    uint64_t n_reads;
    uint32_t *n_hashes;
    uint64_t *sizes;          /* tuples per read */
    orc_slice *slices;
    uint64_t n_slices, cap_slices;
    volatile int lock;
};

orc_batch_ctx *orc_batch_begin(const orc_hixf *h, const orc_search_params *p, const char *bases, const uint64_t *offsets,
                               uint64_t n_reads, uint32_t *n_hashes)
{
    orc_batch_ctx *c = (orc_batch_ctx *)calloc(1, sizeof(orc_batch_ctx));
    c->h = h; c->p = p; c->bases = bases; c->offsets = offsets; c->n_reads = n_reads; c->n_hashes = n_hashes;
    c->sizes = (uint64_t *)calloc(n_reads + 1, sizeof(uint64_t));
    c->cap_slices = 64;
    c->slices = (orc_slice *)calloc(c->cap_slices, sizeof(orc_slice));
    return c;
}

void orc_batch_worker(orc_batch_ctx *c, uint64_t start, uint64_t end)     /* thread-safe; reads [start, end) of the batch */
{
    orc_slice b;
    memset(&b, 0, sizeof b);
    b.start = start; b.end = end;
    b.cap = 1024;
    b.ub = (int64_t *)malloc(b.cap * sizeof(int64_t));
    b.cnt = (uint32_t *)malloc(b.cap * sizeof(uint32_t));
    for (uint64_t r = start; r < end && r < c->n_reads; ++r) {
        const char *seq = c->bases + c->offsets[r];
        size_t len = (size_t)(c->offsets[r + 1] - c->offsets[r]);
        uint32_t nh = 0;
        uint64_t vb = 0;
        for (;;) {
            size_t n = orc_search_read(c->h, c->p, seq, len, &nh, b.ub + b.n, b.cnt + b.n, (size_t)(b.cap - b.n), &vb);
            if (b.n + n <= b.cap) { c->sizes[r] = n; b.n += n; break; }
            while (b.n + n > b.cap) b.cap *= 2;      /* rare: re-run this read with room */
            b.ub = (int64_t *)realloc(b.ub, b.cap * sizeof(int64_t));
            b.cnt = (uint32_t *)realloc(b.cnt, b.cap * sizeof(uint32_t));
        }
        c->n_hashes[r] = nh;
        b.vbytes += vb;
    }
    while (__sync_lock_test_and_set(&c->lock, 1)) {}
    if (c->n_slices == c->cap_slices) {
        c->cap_slices *= 2;
        c->slices = (orc_slice *)realloc(c->slices, c->cap_slices * sizeof(orc_slice));
    }
    c->slices[c->n_slices++] = b;
    __sync_lock_release(&c->lock);
}

static int slice_cmp(const void *a, const void *b)
{
    const orc_slice *x = (const orc_slice *)a, *y = (const orc_slice *)b;
    return x->start < y->start ? -1 : x->start > y->start;
}

int orc_batch_finish(orc_batch_ctx *c, uint64_t *out_off, int64_t *user_bin, uint32_t *count, uint64_t cap, uint64_t *visited_bytes)
{
    out_off[0] = 0;
    for (uint64_t r = 0; r < c->n_reads; ++r) out_off[r + 1] = out_off[r] + c->sizes[r];
    int rc = out_off[c->n_reads] > cap ? -1 : 0;
    qsort(c->slices, c->n_slices, sizeof(orc_slice), slice_cmp);                     /* slice order = read order */
    uint64_t pos = 0, vsum = 0, covered = 0;
    for (uint64_t i = 0; i < c->n_slices; ++i) {
        orc_slice *b = &c->slices[i];
        if (b->start != covered && rc == 0) rc = -2;                                   /* the scheduler left a gap or an overlap */
        covered = b->end;
        if (rc == 0 && b->n) {
            memcpy(user_bin + pos, b->ub, b->n * sizeof(int64_t));
            memcpy(count + pos, b->cnt, b->n * sizeof(uint32_t));
        }
        pos += b->n;
        vsum += b->vbytes;
        free(b->ub);
        free(b->cnt);
    }
    if (covered != c->n_reads && rc == 0) rc = -2;
    if (visited_bytes) *visited_bytes = vsum;
    free(c->slices);
    free(c->sizes);
    free(c);
    return rc;
}

int orc_search_batch(const orc_hixf *h, const orc_search_params *p, const char *bases,
                     const uint64_t *offsets, uint64_t n_reads, int threads, uint32_t *n_hashes,
                     uint64_t *out_off, int64_t *user_bin, uint32_t *count, uint64_t cap,
                     uint64_t *visited_bytes)
{
    /* this file's own scheduler: `threads` contiguous slices of the whole batch under OpenMP (one barrier per batch; the
     * reference cuts 1024-record chunks and runs do_parallel.hpp:17-36 on each -- that shape, with the reference's own
     * scheduler code, is ref_do_parallel_chunks in oracle/ref_driver.cpp) */
    if (threads < 1) threads = 1;
    orc_batch_ctx *c = orc_batch_begin(h, p, bases, offsets, n_reads, n_hashes);
    const uint64_t per = (n_reads + (uint64_t)threads - 1) / (uint64_t)threads;
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int tsk = 0; tsk < threads; ++tsk) {
        uint64_t lo = per * (uint64_t)tsk, hi = lo + per > n_reads ? n_reads : lo + per;
        if (lo > n_reads) lo = n_reads;
        orc_batch_worker(c, lo, hi);
    }
    return orc_batch_finish(c, out_off, user_bin, count, cap, visited_bytes);
}

void orc_classify_filter(const uint32_t *count, size_t n, uint8_t *keep)
{
    uint64_t max_count = 0;                                                            /* :275-280 */
    for (size_t i = 0; i < n; ++i)
        if (count[i] > max_count) max_count = count[i];
    for (size_t i = 0; i < n; ++i)                                                     /* :285 */
        keep[i] = !((double)count[i] < (double)max_count * 0.8);
}
