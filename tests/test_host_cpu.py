"""CPU-side checks that need no GPU: the C-ABI library loads and exports every symbol include/taxor_gpu.h and
include/taxor_gpu_tools.h declare, and the host-side scalars / construction helpers agree with the oracle."""
import os
import re

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import _lib, search, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("taxor_gpu.h", "taxor_gpu_tools.h"))
    declared = set(re.findall(r"\b(taxor_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)))
    declared -= {"taxor_status"}
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libtaxor_gpu.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_threshold_ratio_matches_oracle():
    for k in range(12, 31, 2):
        for e in [0.0, 0.01, 0.04, 0.05, 0.1, 0.15, 0.2]:
            assert search.threshold_ratio(k, e) == orc.syncmer_match_ratio(k, e)
            for n in (0, 1, 98, 435, 870, 12345):
                assert search.threshold(n, search.threshold_ratio(k, e)) == orc.threshold(n, k, e)
    assert search.threshold_ratio(22, 0.04, 0.37) == 0.37
    assert search.threshold(435, 0.37) == orc.threshold(435, 22, 0.04, 0.37)
    for bad in [(21, 0.04), (22, 0.3), (32, 0.04), (10, 0.04)]:
        with pytest.raises(ValueError):
            search.threshold_ratio(*bad)


def test_classify_filter_matches_oracle():
    rng = np.random.default_rng(0)
    for n in (1, 2, 7, 100):
        c = rng.integers(0, 500, size=n, dtype=np.uint32)
        assert search.classify_filter(c).tolist() == orc.classify_filter(c).tolist()
    assert search.classify_filter([0, 0, 0]).tolist() == [True, True, True]


def test_seg_len():
    for n in (0, 1, 10, 1000, 123457, 10**7):
        assert synth.seg_len_for(n) == orc.ixf_seg_len(n) == int(32 + 1.23 * n) // 3


def test_build_bin_members_always_match():
    """Construction (product host code) checked through the oracle's query: planted keys hit, others at ~2^-8."""
    rng = np.random.default_rng(3)
    keys = {b: np.unique(rng.integers(0, 2**63, size=n, dtype=np.uint64)) for b, n in [(0, 5000), (5, 1), (70, 300)]}
    bins, stride = 100, 128
    seg = synth.seg_len_for(5000)
    seed, cols = synth.build_columns(keys, seg, 12345)
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    for b, c in cols.items():
        data[:, b] = c
    h = orc.Hixf([dict(bins=bins, stride=stride, seg_len=seg, seed=seed, data=data.reshape(-1))],
                 [np.zeros(bins, np.int64)], [np.arange(bins)])
    for b, ks in keys.items():
        cnt = h.ixf_bulk_count(0, ks)
        assert cnt[b] == ks.size
        others = np.delete(cnt, b)
        assert others.max() <= max(8, ks.size // 256 * 4 + 8)
    neg = rng.integers(0, 2**63, size=20000, dtype=np.uint64)
    cnt = h.ixf_bulk_count(0, neg)
    assert abs(cnt.mean() - 20000 / 256) < 20


def test_synth_reads_deterministic_and_plausible():
    g, go = synth.random_genomes(3, 20000, seed=7)
    a = synth.synth_reads(g, go, 200, 1500, 0.04, 0.1, seed=9, threads=1)
    b = synth.synth_reads(g, go, 200, 1500, 0.04, 0.1, seed=9, threads=4)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert set(np.unique(a[0]).tolist()) <= set(b"ACGT")
    assert a[1][-1] == 200 * 1500
    frac_neg = (a[2] < 0).mean()
    assert 0.02 < frac_neg < 0.25
    # a read drawn from genome g at 2 % error keeps enough syncmers to pass the default (--error-rate 0.04)
    # threshold; at 4 % uniform error (1-e)^22 ~ 0.41 of the 22-mers survive, below the model's 0.508
    c = synth.synth_reads(g, go, 200, 1500, 0.02, 0.1, seed=9, threads=2)
    gh = [set(orc.seq_to_syncmers(g[int(go[i]):int(go[i + 1])].tobytes()).tolist()) for i in range(3)]
    ok = tot = 0
    for r in range(0, 200, 7):
        rd = c[0][int(c[1][r]):int(c[1][r + 1])].tobytes()
        hs = orc.seq_to_syncmers(rd)
        if c[2][r] >= 0:
            shared = sum(int(x) in gh[c[2][r]] for x in hs)
            tot += 1
            ok += shared >= orc.threshold(hs.size, 22, 0.04)
        else:
            assert sum(int(x) in gh[0] for x in hs) < 5
    assert ok >= 0.8 * tot, (ok, tot)


def test_layout_paths_cover_split_and_depth3():
    rng = np.random.default_rng(1)
    planted = [np.unique(rng.integers(0, 2**63, size=400, dtype=np.uint64)) for _ in range(7)]
    lay = synth.make_layout(planted, root_bins=70, child_bins=40, n_children=3, seed=5)
    assert lay["depth"] == 3 and len(lay["ixfs"]) == 1 + 3 + 1
    host = synth.materialize_host(lay)
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    for i, ks in enumerate(planted):
        ub, cnt, _ = h.bulk_contains(ks, ks.size)        # threshold = all hashes: only the true path survives
        assert lay["planted_user_bin"][i] in ub.tolist()
        j = ub.tolist().index(lay["planted_user_bin"][i])
        assert ks.size <= cnt[j] <= ks.size + 16      # split bins add a few cross-part false positives


def test_threshold_models_match_oracle():
    """k-mer / FracMinHash threshold models (threshold.hpp:51-81) of the host library == oracle, including the wrapped
    thresholds of very short reads"""
    from taxor_amd import search as ts
    from taxor_amd import _lib
    rng = np.random.default_rng(9)
    ns = [0, 1, 2, 3, 5, 7, 10, 20, 50, 99, 100, 435, 871, 4981, 9979, 99979, 1000000] + [int(x) for x in rng.integers(0, 200000, 300)]
    for k in (16, 20, 22, 31, 32):
        for err in (0.001, 0.01, 0.04, 0.1, 0.2, 0.5):
            for n in ns:
                assert ts.threshold_model(_lib.THR_KMER, n, k, err) == orc.threshold_model(orc.THR_KMER, n, k, err), (k, err, n)
                for sf in (1e-3, 0.05, 1 / 7, 0.5, 0.999):
                    a = ts.threshold_model(_lib.THR_FRACMINHASH, n, k, err, -1.0, sf)
                    b = orc.threshold_model(orc.THR_FRACMINHASH, n, k, err, -1.0, sf)
                    assert a == b, (k, err, n, sf)
    assert ts.threshold_model(_lib.THR_PERCENTAGE, 435, 22, 0.04, 0.5) == 217
    assert ts.threshold_model(_lib.THR_SYNCMER, 435, 22, 0.04) == 221
    for use_syn, k, w, pct in ((1, 22, 22, -1.0), (0, 20, 20, -1.0), (0, 20, 32, -1.0), (0, 20, 20, 0.7), (1, 22, 22, 1.0), (0, 20, 21, 0.0)):
        assert ts.threshold_kind(use_syn, k, w, pct) == orc.threshold_kind(use_syn, k, w, pct)
