"""CPU oracle (oracle/taxor_oracle.c) against the committed golden vectors (tests/golden/*.json, produced by
the independent pure-Python restatement in tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_wyhash_kat(golden_dir):
    for x, h in _load(golden_dir, "kat.json")["wyhash"]:
        assert orc.wyhash(int(x)) == int(h)
    # algebraic identities of mix(x, C) = lo ^ hi of the 128-bit product
    assert orc.wyhash(0) == 0
    assert orc.wyhash(1) == 0x9E3779B97F4A7C15


def test_threshold_table(golden_dir):
    kat = _load(golden_dir, "kat.json")
    for n, e, thr in kat["threshold_k22"]:
        assert orc.threshold(n, 22, e) == thr, (n, e)
    # values observed from the reference's own headers in the survey session
    assert orc.threshold(430, 22, 0.04) == 218
    assert orc.threshold(0, 22, 0.04) == 0
    assert orc.threshold(435, 22, 0.04) == 221
    assert orc.syncmer_match_ratio(22, 0.04) == 0.50832
    # percentage model (threshold.hpp:27,76-79)
    assert orc.threshold(435, 22, 0.04, 0.5) == 217
    # out-of-range inputs are fenced, not read out of bounds
    assert orc.syncmer_match_ratio(21, 0.04) == -1.0
    assert orc.syncmer_match_ratio(22, 0.5) == -1.0
    assert orc.syncmer_match_ratio(32, 0.04) == -1.0


def test_readme_rows_reachable(golden_dir):
    # README.md:206-210: each reported QHASH_MATCH must be >= threshold(QHASH_COUNT) for some error rate <= 0.2
    for n, m in _load(golden_dir, "kat.json")["readme_rows"]:
        assert any(orc.threshold(n, 22, e / 100.0) <= m for e in range(0, 21))


def test_syncmers_golden(golden_dir):
    g = _load(golden_dir, "syncmers.json")
    for c in g["cases"]:
        k, s, t = c.get("k", g["k"]), c.get("s", g["s"]), c.get("t", g["t"])
        got = orc.seq_to_syncmers(c["seq"].encode(), k, s, t)
        want = np.array([int(h) for h in c["hashes"]], dtype=np.uint64)
        assert got.tolist() == want.tolist(), c["name"]
    for c in g["dna4"]:
        mapped = orc.dna4_normalise(c["raw"].encode())
        assert mapped.decode() == c["mapped"]
        got = orc.seq_to_syncmers(mapped, g["k"], g["s"], g["t"])
        assert got.tolist() == [int(h) for h in c["hashes"]]
    with pytest.raises(ValueError):
        orc.dna4_normalise(b"ACGT!ACGT")


def test_syncmer_density():
    # README.md:206-210 implies about one distinct syncmer per 11.5 bp at k22/s12
    rng = np.random.default_rng(1)
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=200000).tobytes()
    n = orc.seq_to_syncmers(seq).size
    assert 200000 / 12.5 < n < 200000 / 10.5


def _toy(golden_dir):
    g = _load(golden_dir, "toy_hixf.json")
    hx = g["hixf"]
    ixfs = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"],
                 data=np.array(f["data"], dtype=np.uint8)) for f in hx["ixfs"]]
    return g, orc.Hixf(ixfs, hx["next_ixf"], hx["fname_idx"])


def test_toy_hixf_golden(golden_dir):
    g, h = _toy(golden_dir)
    k, s, t = g["k"], g["s"], g["t"]
    saw_split = saw_deep = saw_zero = False
    for c in g["cases"]:
        hs = orc.seq_to_syncmers(c["read"].encode(), k, s, t)
        assert hs.size == c["n_hashes"]
        if c["err"] is not None:
            assert orc.threshold(hs.size, k, c["err"]) == c["thr"]
        ub, cnt, vbytes = h.bulk_contains(hs, c["thr"])
        assert [[int(a), int(b)] for a, b in zip(ub, cnt)] == c["result"]
        assert vbytes >= hs.size * 3 * 8
        ubs = [r[0] for r in c["result"]]
        saw_split |= 0 in ubs or 4 in ubs
        saw_deep |= 5 in ubs
        saw_zero |= c["n_hashes"] == 0 and len(ubs) == 9  # every leaf run of every IXF reported, count 0
    assert saw_split and saw_deep and saw_zero
    cnt0 = h.ixf_bulk_count(0, orc.seq_to_syncmers(g["cases"][2]["read"].encode(), k, s, t))
    assert cnt0.tolist() == g["root_counts_read1"]


def test_search_batch_matches_per_read(golden_dir):
    g, h = _toy(golden_dir)
    reads = [c["read"].encode() for c in g["cases"] if c["err"] == 0.04]
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    want = [c for c in g["cases"] if c["err"] == 0.04]
    for threads in (1, 3):
        nh, off, ub, cnt, vb = h.search_batch(bases, offs, threads=threads)
        for i, c in enumerate(want):
            assert nh[i] == c["n_hashes"]
            lo, hi = int(off[i]), int(off[i + 1])
            assert [[int(a), int(b)] for a, b in zip(ub[lo:hi], cnt[lo:hi])] == c["result"]


def test_classify_filter():
    keep = orc.classify_filter([100, 80, 79, 0])
    assert keep.tolist() == [True, True, False, False]
    assert orc.classify_filter([0, 0]).tolist() == [True, True]   # zero-hash quirk: 0 < 0*0.8 is false


def test_minimiser_hash_golden(golden_dir):
    """indexes built without --use-syncmer: seqan3 minimiser_hash restated twice (C oracle, Python deque iterator)"""
    g = _load(golden_dir, "minimisers.json")
    for c in g["cases"]:
        got = orc.minimiser_hash(c["seq"].encode(), c["k"], c["w"])
        assert got.tolist() == [int(h) for h in c["hashes"]], (c["name"], c["k"], c["w"])
    # w == k: every canonical k-mer, one per position, no tie rule involved
    seq = b"ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGATCCA"
    assert len(orc.minimiser_hash(seq, 20, 20)) == len(seq) - 19
    # strand symmetry of the VALUES (not of the syncmer selection): a read and its reverse complement give the
    # same multiset of canonical k-mer values
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    rc = seq.translate(comp)[::-1]
    assert sorted(orc.minimiser_hash(seq, 20, 20).tolist()) == sorted(orc.minimiser_hash(rc, 20, 20).tolist())
    assert orc.adjust_seed(20) == 0x8F3F73B5CF1C9ADE >> 24 and orc.adjust_seed(32) == 0x8F3F73B5CF1C9ADE


def test_threshold_models_golden(golden_dir):
    g = _load(golden_dir, "minimisers.json")
    for t in g["thresholds"]:
        kind = orc.THR_KMER if t["model"] == "kmer" else orc.THR_FRACMINHASH
        got = orc.threshold_model(kind, t["n"], t["k"], t["err"], -1.0, t.get("sf", 1.0))
        assert got == int(t["thr"]), t
    # kind selection, threshold.hpp:22-47
    assert orc.threshold_kind(True, 22, 22, -1.0) == orc.THR_SYNCMER
    assert orc.threshold_kind(False, 20, 20, -1.0) == orc.THR_KMER
    assert orc.threshold_kind(False, 20, 32, -1.0) == orc.THR_FRACMINHASH
    assert orc.threshold_kind(False, 20, 20, 0.7) == orc.THR_PERCENTAGE and orc.threshold_kind(True, 22, 22, 1.0) == orc.THR_PERCENTAGE
    # a 10-kb read at 4 % error keeps ~40 % of its 20-mers: the k-mer model asks for a little less than that
    thr = orc.threshold_model(orc.THR_KMER, 9981, 20, 0.04)
    assert 0.35 * 9981 < thr < 0.44 * 9981
    assert abs(orc.lib().orc_normal_cdf_inverse(0.975) - 1.96) < 0.001
