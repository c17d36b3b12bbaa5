#!/usr/bin/env python3
"""Generates tests/golden/*.json -- known-answer vectors for the CPU oracle and the HIP path.

The reference (JensUweUlrich/Taxor) is C++ that cannot be built or imported in this image (its seqan3 /
cereal / ankerl dependencies are fetched from the network at configure time), and it ships no tests or
fixtures.  These vectors therefore come from a SECOND, independent restatement of the reference's
readable sources, written here in plain Python (deque / bigint arithmetic, no numpy, nothing shared with
oracle/taxor_oracle.c).  Agreement between the two restatements guards against transcription slips; it
does not pin the two un-vendored boundaries (wyhash mix, IXF arithmetic) -- see oracle/taxor_oracle.h.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Run:  python tests/golden/make_golden.py        (rewrites the JSON files next to this script)
"""
import json
import math
import os
import random
from collections import deque

M64 = (1 << 64) - 1
HERE = os.path.dirname(os.path.abspath(__file__))


# ---- ankerl::unordered_dense v3.0.1 detail::wyhash::hash(uint64_t) (call site src/hashing/syncmer.cpp:75)
def wyhash(x):
    r = (x & M64) * 0x9E3779B97F4A7C15
    return (r & M64) ^ (r >> 64)


# ---- src/hashing/syncmer.cpp:53-70
def nt4(ch):
    return {"A": 0, "a": 0, "C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3, "U": 3, "u": 3}.get(ch, 4)


# ---- src/hashing/syncmer.cpp:80-165 (size_t arithmetic reproduced with & M64)
def seq_to_syncmers(seq, k, s, t):
    kmask = (1 << (2 * k)) - 1
    smask = (1 << (2 * s)) - 1
    kshift = (k - 1) * 2
    sshift = (s - 1) * 2
    qs = deque()
    min_val = M64
    min_pos = M64  # (size_t)-1
    l = 0
    xk = [0, 0]
    xs = [0, 0]
    out = []
    seen = set()
    for i, ch in enumerate(seq):
        c = nt4(ch)
        if c < 4:
            xk[0] = ((xk[0] << 2) | c) & kmask
            xk[1] = (xk[1] >> 2) | ((3 - c) << kshift)
            xs[0] = ((xs[0] << 2) | c) & smask
            xs[1] = (xs[1] >> 2) | ((3 - c) << sshift)
            l += 1
            if l < s:
                continue
            ys = min(xs)
            qs.append(ys)
            if len(qs) < k - s + 1:
                continue
            if len(qs) == k - s + 1:
                for j in range(len(qs)):
                    if qs[j] < min_val:
                        min_val = qs[j]
                        min_pos = (i - k + j + 1) & M64
            else:
                qs.popleft()
                if min_pos == (i - k) & M64:
                    min_val = M64
                    min_pos = (i - s + 1) & M64
                    for j in range(len(qs) - 1, -1, -1):
                        if qs[j] < min_val:
                            min_val = qs[j]
                            min_pos = (i - k + j + 1) & M64
                elif ys < min_val:
                    min_val = ys
                    min_pos = (i - s + 1) & M64
            if min_pos == (i - k + t) & M64:
                h = wyhash(min(xk))
                if h not in seen:
                    seen.add(h)
                    out.append(h)
        else:
            min_val = M64
            min_pos = M64
            l = 0
            xs = [0, 0]
            xk = [0, 0]
            qs.clear()
    return out


# ---- seqan3 dna4 char mapping [RECALL], src/hixf/build/dna4_traits.hpp:15-18
DNA4 = dict(zip("ACGTURYSWKMBDHVN", "ACGTTACCAGACAAAA"))
DNA4.update({a.lower(): b for a, b in list(DNA4.items())})


def dna4(seq):
    return "".join(DNA4[c] for c in seq)


# ---- src/hixf/search/syncmer_model.hpp:14-50 : only the k=22 column is reproduced here as an independent
#      spot check of the table transcription (rows = accuracy 80..100 %)
RATIO_K22 = [0.0797244, 0.0881939, 0.0966358, 0.106106, 0.116649, 0.130048, 0.144956, 0.16101, 0.179541,
             0.20088, 0.22876, 0.257329, 0.293046, 0.334601, 0.381883, 0.437803, 0.50832, 0.588448,
             0.684331, 0.804269, 1.0]


def threshold_k22(n, err):
    row = math.ceil((1.0 - err) * 100.0 - 80.0)          # syncmer_model.hpp:47
    return int(n * RATIO_K22[row])                       # threshold.hpp:60


# ---- IXF arithmetic: src/main/hashutil.hpp:50-61, src/main/xorfilter.hpp:22-45,60-68,338-350
def murmur64(h):
    h ^= h >> 33
    h = (h * 0xFF51AFD7ED558CCD) & M64
    h ^= h >> 33
    h = (h * 0xC4CEB9FE1A85EC53) & M64
    h ^= h >> 33
    return h


def rotl64(n, c):
    c &= 63
    return ((n << c) | (n >> ((-c) & 63))) & M64


def probe(key, seed, seg_len):
    h = murmur64((key + seed) & M64)
    fp = (h ^ (h >> 32)) & 0xFF
    rows = []
    for i in range(3):
        r = rotl64(h, 21 * i) & 0xFFFFFFFF
        rows.append(((r * seg_len) >> 32) + i * seg_len)
    return rows, fp


def seg_len_for(n):
    return int(32 + 1.23 * n) // 3


def build_bin(keys, seed, seg_len):
    """XOR-filter peeling for one bin (xorfilter.hpp:142-334, simplified queue form). Returns column
    bytes (len 3*seg_len) or None if peeling fails for this seed."""
    rows_n = 3 * seg_len
    cnt = [0] * rows_n
    xr = [0] * rows_n
    info = {}
    for kx in keys:
        rows, fp = probe(kx, seed, seg_len)
        info[kx] = (rows, fp)
        for r in rows:
            cnt[r] += 1
            xr[r] ^= kx
    stack = []
    q = [r for r in range(rows_n) if cnt[r] == 1]
    while q:
        r = q.pop()
        if cnt[r] != 1:
            continue
        kx = xr[r]
        stack.append((kx, r))
        for rr in info[kx][0]:
            cnt[rr] -= 1
            xr[rr] ^= kx
            if cnt[rr] == 1:
                q.append(rr)
    if len(stack) != len(keys):
        return None
    col = [0] * rows_n
    for kx, r in reversed(stack):
        rows, fp = info[kx]
        v = fp
        for rr in rows:
            if rr != r:
                v ^= col[rr]
        col[r] = v
    return col


def bulk_count(ixf, hashes):
    cnt = [0] * ixf["bins"]
    st = ixf["stride"]
    d = ixf["data"]
    for h in hashes:
        rows, fp = probe(h, ixf["seed"], ixf["seg_len"])
        for j in range(ixf["bins"]):
            if fp == d[rows[0] * st + j] ^ d[rows[1] * st + j] ^ d[rows[2] * st + j]:
                cnt[j] += 1
    return cnt


# ---- src/hixf/build/hierarchical_interleaved_xor_filter.hpp:303-340
def bulk_contains(hixf, hashes, thr, ixf_idx=0, out=None):
    if out is None:
        out = []
    ixf = hixf["ixfs"][ixf_idx]
    result = bulk_count(ixf, hashes)
    fn = hixf["fname_idx"][ixf_idx]
    s = 0
    for b in range(len(result)):
        s = (s + result[b]) & 0xFFFFFFFF
        cur = fn[b]
        if cur < 0:
            if s >= thr:
                bulk_contains(hixf, hashes, thr, hixf["next_ixf"][ixf_idx][b], out)
            s = 0
        elif b + 1 == len(result) or cur != fn[b + 1]:
            if s >= thr:
                out.append([cur, s])
            s = 0
    return out


def make_ixf(rng, bin_keys, bins, seed0):
    """bin_keys: dict bin -> list of keys; other bins random bytes."""
    stride = ((bins + 63) // 64) * 64
    mx = max([len(v) for v in bin_keys.values()] + [8])
    seg = seg_len_for(mx)
    seed = seed0
    while True:
        cols = {b: build_bin(ks, seed, seg) for b, ks in bin_keys.items()}
        if all(c is not None for c in cols.values()):
            break
        seed = (seed * 6364136223846793005 + 1442695040888963407) & M64
    data = [rng.randrange(256) for _ in range(3 * seg * stride)]
    for b, col in cols.items():
        for r in range(3 * seg):
            data[r * stride + b] = col[r]
    return {"bins": bins, "stride": stride, "seg_len": seg, "seed": seed, "data": data}


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(rng, seq, e):
    out = []
    for ch in seq:
        x = rng.random()
        if x < e * 0.4:
            out.append(rng.choice("ACGT"))
        elif x < e * 0.7:
            out.append(ch)
            out.append(rng.choice("ACGT"))
        elif x < e:
            continue
        else:
            out.append(ch)
    return "".join(out)



# ---- indexes built without --use-syncmer: seqan3::views::minimiser_hash + the k-mer / FracMinHash threshold models
def adjust_seed(k):  # src/hixf/build/adjust_seed.hpp:40-44
    return 0x8F3F73B5CF1C9ADE >> (64 - 2 * k)


def minimiser_hash(seq, k, w):
    """seqan3::views::minimiser_hash(ungapped{k}, window_size{w}, seed{adjust_seed(k)}) as called at
    src/main/taxor_search.cpp:210-212 (un-vendored seqan3: restated from the published 3.x iterator, which keeps a
    deque of the window's values and the offset of the current minimiser in it)."""
    rank = {"A": 0, "C": 1, "G": 2, "T": 3}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    seed = adjust_seed(k)
    n = len(seq) - k + 1
    if n <= 0:
        return []

    def kmer_hash(text):
        v = 0
        for ch in text:
            v = v * 4 + rank[ch]
        return v

    fwd = [kmer_hash(seq[i:i + k]) ^ seed for i in range(n)]
    rc_text = "".join(comp[c] for c in reversed(seq))            # views::complement | views::reverse
    rc = [kmer_hash(rc_text[i:i + k]) ^ seed for i in range(n)][::-1]   # ... | kmer_hash | ^seed | views::reverse
    values = [min(a, b) for a, b in zip(fwd, rc)]
    window = min(w - k + 1, n)
    dq = deque(values[:window])
    # window_first: min_element with less_equal -> the rightmost minimum
    m_val, m_off = dq[0], 0
    for i, x in enumerate(dq):
        if x <= m_val:
            m_val, m_off = x, i
    out = [m_val]
    for x in values[window:]:
        dq.popleft()
        dq.append(x)
        if m_off == 0:
            m_val, m_off = dq[0], 0
            for i, y in enumerate(dq):
                if y <= m_val:
                    m_val, m_off = y, i
            out.append(m_val)
        elif x < m_val:
            m_val, m_off = x, len(dq) - 1
            out.append(m_val)
        else:
            m_off -= 1
    return out


def f64_to_size(x):
    """static_cast<size_t>(double) of the reference's stock x86-64 build (no -march: cvttsd2si sequence), stated
    explicitly because negative / NaN inputs (very short reads) are undefined behaviour in C++"""
    if x != x:
        return 1 << 63
    two63 = 9223372036854775808.0
    if x >= two63:
        y = x - two63
        return 0 if y >= two63 else (int(y) ^ (1 << 63))
    if x <= -two63:
        return 1 << 63
    return int(x) & M64


def normal_cdf_inverse(p):  # src/hixf/search/gaussian_inverse.cpp:13-50
    def ra(t):
        c = (2.515517, 0.802853, 0.010328)
        d = (1.432788, 0.189269, 0.001308)
        return t - ((c[2] * t + c[1]) * t + c[0]) / (((d[2] * t + d[1]) * t + d[0]) * t + 1.0)
    if p < 0.5:
        return -ra(math.sqrt(-2.0 * math.log(p)))
    return ra(math.sqrt(-2.0 * math.log(1.0 - p)))


def _sqrt(x):
    return math.sqrt(x) if x >= 0 else float("nan")


def variance_nmut_kmer(r, k, n):  # src/hixf/search/kmer_model.cpp:32-39
    q = 1.0 - math.pow(1.0 - r, k)
    return (float(n) * (1.0 - q) * (q * (2.0 * float(k) + (2.0 / r) - 1.0) - 2.0 * float(k))
            + float(k) * (float(k) - 1.0) * math.pow((1.0 - q), 2.0)
            + (2.0 * (1.0 - q) / (math.pow(r, 2.0))) * ((1.0 + (float(k) - 1.0) * (1.0 - q)) * r - q))


def threshold_kmer_model(n, k, err):  # threshold.hpp:62-66 + kmer_model.cpp:10-23
    q = 1.0 - math.pow(1.0 - err, k)
    z = normal_cdf_inverse(1.0 - (1 - 0.95) / 2.0)
    high = f64_to_size(math.ceil(n * q + z * _sqrt(variance_nmut_kmer(err, k, n)))) if variance_nmut_kmer(err, k, n) >= 0 else (1 << 63)
    return (n - high - f64_to_size(n * 0.0039)) & M64


def threshold_fracminhash(n, k, err, sf):  # threshold.hpp:68-75 + fracminhash_model.cpp:8-33
    z = normal_cdf_inverse(1.0 - (1.0 - 0.95) / 2.0)
    q = 1.0 - math.pow(1.0 - err, k)
    e_n = n * q
    var_n = variance_nmut_kmer(err, k, n)
    try:
        term3 = var_n / math.pow(n, 2)
        term2 = n * e_n - (math.pow(e_n, 2) + var_n)
        den = sf * math.pow(n, 3) * math.pow(1.0 - math.pow(1.0 - sf, n), 2)
        term1 = (1.0 - sf) / den
        var = term1 * term2 + term3
        clow = math.pow((1.0 - err), k) - z * _sqrt(var)
        prod = clow * float(n)
    except (ZeroDivisionError, ValueError, OverflowError):
        return None                                       # IEEE inf/nan territory: left to the C restatement
    return (f64_to_size(prod) - f64_to_size(n * 0.0039)) & M64


def main():
    rng = random.Random(20250523)
    k, s, t = 22, 12, 5

    # ---------------- syncmer vectors (incl. tie-heavy, N, lower case, short) -----------------------
    seqs = {
        "random_300": rand_seq(rng, 300),
        "random_1000": rand_seq(rng, 1000),
        "homopolymer_A_120": "A" * 120,
        "homopolymer_T_57": "T" * 57,
        "dinuc_AT_150": "AT" * 75,
        "trinuc_CAG_180": "CAG" * 60,
        "telomere_TTAGGG_240": "TTAGGG" * 40,
        "palindrome": "ACGTACGTACGTTGCATGCATGCAACGTACGTACGTTGCATGCATGCA" * 3,
        "mixed_lowcomplex": rand_seq(rng, 60) + "A" * 40 + rand_seq(rng, 30) + "GT" * 30 + rand_seq(rng, 60),
        "with_N": rand_seq(rng, 80) + "N" + rand_seq(rng, 90) + "NN" + rand_seq(rng, 25),
        "lower_case": rand_seq(rng, 100).lower(),
        "len_lt_k": rand_seq(rng, 21),
        "len_eq_k": rand_seq(rng, 22),
        "len_k_plus_1": rand_seq(rng, 23),
        "empty": "",
        "repeat_unit_11": rand_seq(rng, 11) * 12,
        "two_copies": (lambda x: x + x)(rand_seq(rng, 200)),
    }
    sync = {"k": k, "s": s, "t": t, "cases": []}
    for name, sq in seqs.items():
        sync["cases"].append({"name": name, "seq": sq, "hashes": [str(h) for h in seq_to_syncmers(sq, k, s, t)]})
    # other (k,s,t) to exercise generic parameters: t = ceil((k-s+1)/2) with integer division (taxor_build.cpp:510)
    for (kk, ss) in [(16, 8), (20, 10), (28, 14), (30, 12)]:
        tt = math.ceil((kk - ss + 1) // 2)
        sq = rand_seq(rng, 400) + "C" * 50 + rand_seq(rng, 100)
        sync["cases"].append({"name": f"k{kk}_s{ss}", "k": kk, "s": ss, "t": tt, "seq": sq,
                              "hashes": [str(h) for h in seq_to_syncmers(sq, kk, ss, tt)]})
    # dna4 mapping cases (search path): IUPAC / N become bases before the selector sees them
    iu = "ACGTNRYSWKMBDHVUacgtnryswkmbdhvu" * 4 + rand_seq(rng, 80)
    sync["dna4"] = [{"raw": iu, "mapped": dna4(iu), "hashes": [str(h) for h in seq_to_syncmers(dna4(iu), k, s, t)]}]
    json.dump(sync, open(os.path.join(HERE, "syncmers.json"), "w"), indent=0)

    # ---------------- wyhash / threshold known answers ---------------------------------------------
    kat = {
        "wyhash": [[str(x), str(wyhash(x))] for x in
                   [0, 1, 2, 0xFFFFFFFFFFFFFFFF, 0x0123456789ABCDEF, (1 << 44) - 1, 0x9E3779B97F4A7C15,
                    rng.getrandbits(44), rng.getrandbits(44), rng.getrandbits(64)]],
        # threshold.hpp:60 with syncmer_model.hpp:47-49 at k = 22; (430, 0.04) -> 218 and (0, .) -> 0 were also
        # observed from the reference's own headers in the survey session
        "threshold_k22": [[n, e, threshold_k22(n, e)] for n in [0, 1, 2, 98, 158, 224, 344, 430, 435, 870, 100000]
                          for e in [0.0, 0.01, 0.04, 0.05, 0.1, 0.15, 0.2]],
        # README.md:206-210 sample rows (QHASH_COUNT, QHASH_MATCH): every reported match count must reach
        # the threshold for SOME error rate <= 0.2 -- a weak sanity pin from the reference's own docs
        "readme_rows": [[98, 56], [224, 104], [158, 38], [344, 158]],
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=0)

    # ---------------- toy HIXF: split bin, merged chain of depth 3, planted genomes ------------------
    genomes = [rand_seq(rng, 700) for _ in range(6)]
    gh = [seq_to_syncmers(g, k, s, t) for g in genomes]
    # layout (user bins 0..5):
    #  root (ixf0, 8 bins): [ub0 split over bins 0,1,2] [bin3 merged->ixf1] [ub1 bin4] [bin5 merged->ixf2] [ub2 bins 6,7]
    #  ixf1 (4 bins): [ub3 bin0] [bin1 merged->ixf3] [ub3 again? no] -> [ub3][merged][ub4 split bins 2,3]
    #  ixf3 (3 bins): [ub5 bins 0,1] [ub5? ] -> [ub5 split 0,1][ub5' none]  (depth-3 chain root->ixf1->ixf3)
    #  ixf2 (2 bins): [ub1' none] -> holds copies so that bin5 of root matches ub2-like reads: [ub2 bin0][ub1 bin1]
    def split(keys, n):
        return [keys[i::n] for i in range(n)]

    ixf3 = make_ixf(rng, {0: split(gh[5], 2)[0], 1: split(gh[5], 2)[1], 2: gh[3][:40]}, 3, 11)
    ixf1 = make_ixf(rng, {0: gh[3], 1: gh[5] + gh[3][:40], 2: split(gh[4], 2)[0], 3: split(gh[4], 2)[1]}, 4, 12)
    ixf2 = make_ixf(rng, {0: gh[2][:60], 1: gh[1][:60]}, 2, 13)
    r0 = split(gh[0], 3)
    r2 = split(gh[2], 2)
    ixf0 = make_ixf(rng, {0: r0[0], 1: r0[1], 2: r0[2], 3: gh[3] + gh[4] + gh[5], 4: gh[1],
                          5: gh[2][:60] + gh[1][:60], 6: r2[0], 7: r2[1]}, 8, 14)
    hixf = {
        "ixfs": [ixf0, ixf1, ixf2, ixf3],
        "next_ixf": [[0, 0, 0, 1, 0, 2, 0, 0], [1, 3, 1, 1], [2, 2], [3, 3, 3]],
        "fname_idx": [[0, 0, 0, -1, 1, -1, 2, 2], [3, -1, 4, 4], [2, 1], [5, 5, 3]],
    }
    reads = []
    for gi, g in enumerate(genomes):
        for e in (0.0, 0.04, 0.1):
            st = rng.randrange(0, 200)
            reads.append(mutate(rng, g[st:st + 450], e))
    reads.append(rand_seq(rng, 400))          # true negative
    reads.append(rand_seq(rng, 15))           # zero hashes -> threshold 0 -> every user bin with count 0
    reads.append(genomes[2][:60])             # few hashes
    cases = []
    for rd in reads:
        hs = seq_to_syncmers(rd, k, s, t)
        for err in (0.04, 0.1):
            thr = threshold_k22(len(hs), err)
            cases.append({"read": rd, "err": err, "n_hashes": len(hs), "thr": thr,
                          "result": bulk_contains(hixf, hs, thr)})
    # explicit low thresholds to force wide traversal
    hs = seq_to_syncmers(reads[0], k, s, t)
    for thr in (0, 1, 2):
        cases.append({"read": reads[0], "err": None, "n_hashes": len(hs), "thr": thr,
                      "result": bulk_contains(hixf, hs, thr)})
    counts0 = bulk_count(ixf0, seq_to_syncmers(reads[1], k, s, t))
    out = {"k": k, "s": s, "t": t, "hixf": hixf, "cases": cases,
           "root_counts_read1": counts0}
    json.dump(out, open(os.path.join(HERE, "toy_hixf.json"), "w"))
    print("golden vectors written:", sorted(f for f in os.listdir(HERE) if f.endswith(".json")))

    # ---------------- minimiser / k-mer mode (indexes built without --use-syncmer) --------------------
    mseqs = {
        "random_400": rand_seq(rng, 400),
        "homopolymer_A_90": "A" * 90,
        "dinuc_AT_120": "AT" * 60,
        "trinuc_CAG_150": "CAG" * 50,
        "palindrome": "ACGTACGTACGTTGCATGCATGCAACGTACGTACGTTGCATGCATGCA" * 2,
        "mixed_lowcomplex": rand_seq(rng, 70) + "C" * 45 + rand_seq(rng, 40) + "GA" * 25 + rand_seq(rng, 50),
        "repeat_unit_7": rand_seq(rng, 7) * 20,
        "len_lt_k": rand_seq(rng, 19),
        "len_eq_k": rand_seq(rng, 20),
        "len_between_k_and_w": rand_seq(rng, 26),
        "len_eq_w": rand_seq(rng, 32),
        "empty": "",
    }
    mini = {"cases": [], "thresholds": []}
    for (kk, ww) in [(20, 20), (20, 32), (22, 22), (16, 24), (31, 40), (32, 32)]:
        for name, sq in mseqs.items():
            mini["cases"].append({"name": name, "k": kk, "w": ww, "seq": sq, "hashes": [str(h) for h in minimiser_hash(sq, kk, ww)]})
    for kk in (20, 22, 31):
        for err in (0.01, 0.04, 0.1):
            for n in (0, 1, 2, 5, 10, 50, 100, 435, 1000, 4981, 9979, 99979):
                mini["thresholds"].append({"model": "kmer", "k": kk, "err": err, "n": n, "thr": str(threshold_kmer_model(n, kk, err))})
                for sf in (0.05, 0.125, 0.5):
                    v = threshold_fracminhash(n, kk, err, sf)
                    if v is not None:
                        mini["thresholds"].append({"model": "fracminhash", "k": kk, "err": err, "n": n, "sf": sf, "thr": str(v)})
    json.dump(mini, open(os.path.join(HERE, "minimisers.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
