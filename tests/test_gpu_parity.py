"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.
Bit-exact: everything on this path is integer / byte / index work."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import GpuIndex, Searcher, synth
from taxor_amd._lib import TaxorError

pytestmark = pytest.mark.gpu


def _dummy_index(k=22, s=12, t=5):
    bins, stride, seg = 64, 64, 16
    data = np.zeros(3 * seg * stride, dtype=np.uint8)
    return GpuIndex([dict(bins=bins, stride=stride, seg_len=seg, seed=1, next_ixf=np.zeros(bins, np.int64),
                          fname_idx=np.arange(bins), data=data)], bins, k, s, t)


def _cat(reads):
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, np.uint8)
    offs = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    return bases, offs


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------------ syncmers
def test_syncmers_golden(golden_dir):
    g = _load(golden_dir, "syncmers.json")
    by_kst = {}
    for c in g["cases"]:
        if "N" in c["seq"]:
            continue  # the search path never sees N: dna4 turns it into A before the selector (see dna4 case)
        by_kst.setdefault((c.get("k", g["k"]), c.get("s", g["s"]), c.get("t", g["t"])), []).append(c)
    for (k, s, t), cases in by_kst.items():
        idx = _dummy_index(k, s, t)
        sr = Searcher(idx, ratio=0.5)
        bases, offs = _cat([c["seq"].encode() for c in cases])
        hoff, hashes = sr.seq_to_syncmers(bases, offs)
        for i, c in enumerate(cases):
            got = hashes[int(hoff[i]):int(hoff[i + 1])].tolist()
            assert got == [int(h) for h in c["hashes"]], (c["name"], k, s, t)
        sr.close()
        idx.close()


def test_syncmers_dna4_mapping(golden_dir):
    g = _load(golden_dir, "syncmers.json")
    idx = _dummy_index()
    sr = Searcher(idx, ratio=0.5)
    for c in g["dna4"]:
        hoff, hashes = sr.seq_to_syncmers(*_cat([c["raw"].encode()]))
        assert hashes.tolist() == [int(h) for h in c["hashes"]]
    # with_N through the search path == oracle on the dna4-normalised read
    n_case = [c for c in g["cases"] if c["name"] == "with_N"][0]
    hoff, hashes = sr.seq_to_syncmers(*_cat([n_case["seq"].encode()]))
    assert hashes.tolist() == orc.seq_to_syncmers(orc.dna4_normalise(n_case["seq"].encode())).tolist()
    with pytest.raises(TaxorError) as e:
        sr.seq_to_syncmers(*_cat([b"ACGTACGTACGTACGTACGTACGTAC#GT"]))
    assert e.value.code == -3
    sr.close()
    idx.close()


def _lowcomplex(rng, n):
    parts, tot = [], 0
    while tot < n:
        c = rng.random()
        if c < 0.35:
            p = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(5, 80))).astype(np.uint8))
        elif c < 0.6:
            p = bytes([int(rng.choice(list(b"ACGT")))]) * int(rng.integers(5, 120))
        else:
            u = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(2, 12))).astype(np.uint8))
            p = u * int(rng.integers(3, 60))
        parts.append(p)
        tot += len(p)
    return b"".join(parts)[:n]


@pytest.mark.parametrize("kst", [(22, 12, 5), (16, 8, 4), (30, 12, 9), (28, 14, 7), (22, 12, 1), (22, 12, 11), (32, 16, 8)])
def test_syncmers_vs_oracle_ragged_and_tie_heavy(kst):
    k, s, t = kst
    rng = np.random.default_rng(k * 100 + s)
    reads = [b"", b"A", b"ACGT" * 5]
    reads += [bytes(rng.choice(list(b"ACGT"), size=n).astype(np.uint8)) for n in
              (k - 1, k, k + 1, 2 * k, 100, 2047, 2048, 2049, 2048 + k - 1, 2048 + k, 4096 + k - 1, 5000, 20011)]
    reads += [_lowcomplex(rng, int(n)) for n in rng.integers(k, 9000, size=40)]
    reads += [b"A" * 7000, b"AC" * 3000, b"TTAGGG" * 1500, (b"ACGTTGCA" * 2 + b"G") * 400]
    idx = _dummy_index(k, s, t)
    sr = Searcher(idx, ratio=0.5, sub_batch_reads=7)      # several sub-batches
    hoff, hashes = sr.seq_to_syncmers(*_cat(reads))
    for i, rd in enumerate(reads):
        want = orc.seq_to_syncmers(rd, k, s, t)
        got = hashes[int(hoff[i]):int(hoff[i + 1])]
        assert got.tolist() == want.tolist(), (i, len(rd), got.size, want.size)
    sr.close()
    idx.close()


def test_syncmers_wave_kernel_edges():
    """short reads go through the wave-per-read kernel (k_syncmers_wave: 512-window tiles, register-carried tie state,
    all-pairs dedup): tile edges (nwin = 511, 512, 513, 1024, 1025), the candidate-capacity boundary where reads switch
    back to the block kernel (~2.5 kb), duplicates and tie-heavy low-complexity sequence, mixed with long reads in one
    sub-batch (the long ones form the prefix of the processing order) -- all against the sequential oracle"""
    k, s, t = 22, 12, 5
    rng = np.random.default_rng(2025)
    rnd = lambda n: bytes(rng.choice(list(b"ACGT"), size=int(n)).astype(np.uint8))
    reads = [rnd(n) for n in (k - 1, k, k + 1, 511 + k - 1, 512 + k - 1, 513 + k - 1, 1024 + k - 1, 1025 + k - 1, 1536 + k - 1,
                              2000, 2500, 2540, 2550, 2560, 2570, 2580, 2600, 2700, 6000, 30000)]
    unit = rnd(300)
    reads += [unit * 3, unit * 8, unit[:150] + rnd(200) + unit[:150], b"A" * 900, b"AC" * 700, b"TTAGGG" * 400,
              (b"ACGTTGCA" * 2 + b"G") * 120]
    reads += [_lowcomplex(rng, int(n)) for n in rng.integers(k, 2560, size=60)]
    reads += [rnd(n) for n in rng.integers(k, 2560, size=200)]
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    idx = _dummy_index(k, s, t)
    for sub in (0, 13):                                    # one sub-batch / many small ones
        sr = Searcher(idx, ratio=0.5, sub_batch_reads=sub)
        hoff, hashes = sr.seq_to_syncmers(*_cat(reads))
        for i, rd in enumerate(reads):
            want = orc.seq_to_syncmers(rd, k, s, t)
            got = hashes[int(hoff[i]):int(hoff[i + 1])]
            assert got.tolist() == want.tolist(), (i, len(rd), got.size, want.size)
        sr.close()
    idx.close()


def test_syncmers_long_reads_partitioned_and_global_dedup():
    """reads with more selected syncmers than one LDS table holds dedup in partitioned passes (with and without
    duplicates); beyond ~750 kb the per-block table in global memory takes over"""
    rng = np.random.default_rng(11)
    unit = bytes(rng.choice(list(b"ACGT"), size=30000).astype(np.uint8))
    mb = bytes(rng.choice(list(b"ACGT"), size=900000).astype(np.uint8))
    reads = [bytes(rng.choice(list(b"ACGT"), size=300000).astype(np.uint8)),
             unit * 6,                                   # every hash occurs 6 times
             bytes(rng.choice(list(b"ACGT"), size=70000).astype(np.uint8)),
             unit[:25000] + unit[:700] + unit[20000:],   # a few duplicates in a read of two passes
             mb + mb[100000:400000],                     # 1.2 Mb: global table, with duplicates
             bytes(rng.choice(list(b"ACGT"), size=18000).astype(np.uint8))]
    idx = _dummy_index()
    sr = Searcher(idx, ratio=0.5)
    hoff, hashes = sr.seq_to_syncmers(*_cat(reads))
    for i, rd in enumerate(reads):
        want = orc.seq_to_syncmers(rd)
        got = hashes[int(hoff[i]):int(hoff[i + 1])]
        assert got.size == want.size and np.array_equal(got, want), (i, got.size, want.size)
    sr.close()
    idx.close()


# ---------------------------------------------------------------------------------------------- bulk_count
@pytest.mark.parametrize("bins", [1, 63, 64, 65, 192, 200, 1000, 1024, 4096, 5000])
def test_ixf_bulk_count_vs_oracle(bins):
    rng = np.random.default_rng(bins)
    stride = ((bins + 63) // 64) * 64
    planted = {b: np.unique(rng.integers(0, 2**63, size=500, dtype=np.uint64))
               for b in sorted({0, bins // 2, bins - 1})}
    seg = synth.seg_len_for(700)
    seed, cols = synth.build_columns(planted, seg, 99 + bins)
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    for b, c in cols.items():
        data[:, b] = c
    ixf = dict(bins=bins, stride=stride, seg_len=seg, seed=seed, next_ixf=np.zeros(bins, np.int64),
               fname_idx=np.arange(bins), data=data.reshape(-1))
    idx = GpuIndex([ixf], bins)
    sr = Searcher(idx, ratio=0.5)
    h = orc.Hixf([ixf], [ixf["next_ixf"]], [ixf["fname_idx"]])
    for n in (0, 1, 2, 239, 240, 241, 481, 1500):
        keys = rng.integers(0, 2**63, size=n, dtype=np.uint64)
        if n >= 240:
            keys[: n // 2] = planted[0][: n // 2] if n // 2 <= planted[0].size else keys[: n // 2]
        got = sr.ixf_bulk_count(0, keys)
        want = h.ixf_bulk_count(0, keys)
        assert np.array_equal(got, want), (bins, n)
    for b, ks in planted.items():
        assert sr.ixf_bulk_count(0, ks)[b] == ks.size
    assert np.array_equal(idx.download_ixf(0), data.reshape(-1))
    sr.close()
    idx.close()


# ------------------------------------------------------------------------------------------- bulk_contains
def test_toy_hixf_golden(golden_dir):
    g = _load(golden_dir, "toy_hixf.json")
    hx = g["hixf"]
    ixfs = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"],
                 next_ixf=np.array(hx["next_ixf"][i]), fname_idx=np.array(hx["fname_idx"][i]),
                 data=np.array(f["data"], dtype=np.uint8)) for i, f in enumerate(hx["ixfs"])]
    idx = GpuIndex(ixfs, 6)
    assert idx.depth == 3 and idx.leaf_runs == 9
    sr = Searcher(idx, ratio=0.5)
    for c in g["cases"]:
        hoff, hashes = sr.seq_to_syncmers(*_cat([c["read"].encode()]))
        assert hashes.size == c["n_hashes"]
        ub, cnt = sr.bulk_contains(hashes, c["thr"])
        assert [[int(a), int(b)] for a, b in zip(ub, cnt)] == c["result"]
    # and the whole driver at the two error rates
    for err in (0.04, 0.1):
        cases = [c for c in g["cases"] if c["err"] == err]
        s2 = Searcher(idx, error_rate=err, sub_batch_reads=5)
        res = s2.search_batch(*_cat([c["read"].encode() for c in cases]))
        for i, c in enumerate(cases):
            assert res.n_hashes[i] == c["n_hashes"]
            assert [list(x) for x in res.tuples(i)] == c["result"]
        s2.close()
    sr.close()
    idx.close()


def _planted_setup(seed, n_genomes=9, glen=30000, root_bins=70, child_bins=48, n_children=3):
    g, go = synth.random_genomes(n_genomes, glen, seed=seed)
    hidx = _dummy_index()
    hs = Searcher(hidx, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(n_genomes)]
    hs.close()
    hidx.close()
    lay = synth.make_layout(planted, root_bins=root_bins, child_bins=child_bins, n_children=n_children, seed=seed)
    host = synth.materialize_host(lay)
    return g, go, lay, host


def _compare(res, oracle_out, n):
    nh, off, ub, cnt, _ = oracle_out
    assert np.array_equal(res.n_hashes, nh)
    assert np.array_equal(res.read_off, off)
    assert np.array_equal(res.user_bin, ub)
    assert np.array_equal(res.count, cnt)


def test_search_batch_vs_oracle_planted_hierarchy():
    g, go, lay, host = _planted_setup(5)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, origin = synth.synth_reads(g, go, 600, 1800, error_rate=0.02, frac_random=0.15, seed=3)
    # ragged extras: empty read, shorter than k (threshold 0 -> every leaf run reported), == k, IUPAC codes
    extra = [b"", b"ACGTACGTAC", bytes(g[:22]), bytes(g[100:160]), b"ACGTNRYKMSWBDHVNacgtnn" * 20,
             bytes(g[int(go[3]):int(go[3]) + 5000])]
    all_reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(600)] + extra
    B, O = _cat(all_reads)
    Bn = np.frombuffer(orc.dna4_normalise(B.tobytes()), dtype=np.uint8)
    for err, sub in ((0.04, 0), (0.04, 37), (0.1, 64), (0.2, 1000)):
        sr = Searcher(idx, error_rate=err, sub_batch_reads=sub)
        res = sr.search_batch(B, O)
        want = h.search_batch(Bn, O, err=err, threads=4)
        _compare(res, want, len(all_reads))
        # positive control: reads drawn from a planted genome report that genome's user bin
        hit = tot = 0
        for i in range(600):
            if origin[i] >= 0:
                tot += 1
                hit += lay["planted_user_bin"][origin[i]] in [u for u, _ in res.tuples(i)]
        assert hit > 0.8 * tot, (err, hit, tot)
        # the short reads report every leaf run of the hierarchy, count 0 or small
        assert len(res.tuples(601)) == idx.leaf_runs
        sr.close()
    # percentage model
    sr = Searcher(idx, percentage=0.3)
    _compare(sr.search_batch(B, O), h.search_batch(Bn, O, percentage=0.3, threads=4), len(all_reads))
    sr.close()
    idx.close()


def test_threshold_zero_flood_grows_buffers():
    """many reads shorter than k: each reports every leaf run (taxor quirk) -> hit/queue/tuple
    buffers overflow their first sizing and the library must grow and rerun, never truncate"""
    g, go, lay, host = _planted_setup(8, n_genomes=5, glen=8000, root_bins=200, child_bins=64, n_children=40)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    reads = [b"ACGTACGTACGTACG"] * 300 + [bytes(g[:3000])] * 3
    B, O = _cat(reads)
    sr = Searcher(idx, sub_batch_reads=128)
    res = sr.search_batch(B, O)
    _compare(res, h.search_batch(B, O, threads=4), len(reads))
    assert int(res.read_off[-1]) >= 300 * idx.leaf_runs
    # idempotence: same batch again on the same searcher
    res2 = sr.search_batch(B, O)
    assert np.array_equal(res.user_bin, res2.user_bin) and np.array_equal(res.count, res2.count)
    sr.close()
    idx.close()


def test_wide_ixf_many_tuples_sorted_segments():
    """reads with > 64 tuples take the block-wide sorter; order must still be the reference's DFS order"""
    rng = np.random.default_rng(2)
    bins = 700
    planted = {b: np.unique(rng.integers(0, 2**63, size=64, dtype=np.uint64)) for b in (3,)}
    seg = synth.seg_len_for(100)
    seed, cols = synth.build_columns(planted, seg, 7)
    child_bins = 130
    root = dict(bins=bins, stride=704, seg_len=seg, seed=seed, next_ixf=np.zeros(bins, np.int64),
                fname_idx=np.arange(bins, dtype=np.int64),
                data=rng.integers(0, 256, size=3 * seg * 704, dtype=np.uint8))
    # two merged bins in the middle -> children with their own leaves
    root["fname_idx"][100] = -1
    root["next_ixf"][100] = 1
    root["fname_idx"][400] = -1
    root["next_ixf"][400] = 2
    ch = [dict(bins=child_bins, stride=192, seg_len=seg, seed=5 + i, next_ixf=np.full(child_bins, i + 1, np.int64),
               fname_idx=np.arange(bins + i * child_bins, bins + (i + 1) * child_bins, dtype=np.int64),
               data=rng.integers(0, 256, size=3 * seg * 192, dtype=np.uint8)) for i in range(2)]
    ixfs = [root] + ch
    idx = GpuIndex(ixfs, bins + 2 * child_bins)
    h = orc.Hixf(ixfs, [f["next_ixf"] for f in ixfs], [f["fname_idx"] for f in ixfs])
    sr = Searcher(idx, ratio=0.5)
    for n, thr in ((300, 0), (300, 1), (300, 2), (2000, 8), (2000, 9)):
        keys = rng.integers(0, 2**63, size=n, dtype=np.uint64)
        ub, cnt = sr.bulk_contains(keys, thr)
        wub, wcnt, _ = h.bulk_contains(keys, thr)
        assert np.array_equal(ub, wub) and np.array_equal(cnt, wcnt), (n, thr, ub.size, wub.size)
    sr.close()
    idx.close()


def test_device_built_index_matches_host_copy():
    """index created directly in HBM (fill kernel + column uploads) == what download hands to the oracle"""
    g, go, lay, _ = _planted_setup(21, n_genomes=6, glen=12000)
    idx = synth.device_index(lay)
    host = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], next_ixf=f["next_ixf"],
                 fname_idx=f["fname_idx"], data=idx.download_ixf(i)) for i, f in enumerate(lay["ixfs"])]
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, origin = synth.synth_reads(g, go, 300, 2500, error_rate=0.02, frac_random=0.1, seed=4)
    sr = Searcher(idx, time_kernels=True)
    sr.upload(bases, offs)
    sr.run()
    sr.sync()
    res = sr.fetch()
    want = h.search_batch(bases, offs, threads=4)
    _compare(res, want, 300)
    st = sr.stats()
    assert st["n_reads"] == 300 and st["n_bases"] == 300 * 2500
    assert st["n_hashes"] == int(want[0].sum()) and st["n_tuples"] == want[2].size
    assert st["query_bytes"] == want[4]                       # algorithmic gather bytes, SURVEY 8(d)
    assert st["algorithmic_bytes"] == want[4] + 300 * 625 + 8 * 300 + 12 * want[2].size
    assert st["query_ms"] > 0 and st["query_launches"] == idx.depth
    sr.close()
    idx.close()


def test_pruning_long_split_runs_and_thresholds():
    """threshold-aware pruning: dead runs are skipped, alive runs counted exactly.  A user bin split over 640
    technical bins keeps > 32 units alive (dense fallback); short runs exercise the sparse phase; results must
    equal the oracle for thresholds on both sides of every boundary."""
    rng = np.random.default_rng(12)
    bins, stride = 700, 704
    seg = synth.seg_len_for(3000)
    keys_a = np.unique(rng.integers(0, 2**63, size=2500, dtype=np.uint64))     # planted in the long split run
    keys_b = np.unique(rng.integers(0, 2**63, size=2000, dtype=np.uint64))     # planted in a 3-bin run
    keys_c = np.unique(rng.integers(0, 2**63, size=1500, dtype=np.uint64))     # planted in a single bin
    planted = {b: keys_a[b::640] for b in range(0, 640, 7)}
    planted.update({650 + j: keys_b[j::3] for j in range(3)})
    planted[690] = keys_c
    seed, cols = synth.build_columns(planted, seg, 5)
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    for b, c in cols.items():
        data[:, b] = c
    fname = np.arange(bins, dtype=np.int64)
    fname[:640] = 0
    fname[650:653] = 650
    ixf = dict(bins=bins, stride=stride, seg_len=seg, seed=seed, next_ixf=np.zeros(bins, np.int64), fname_idx=fname,
               data=data.reshape(-1))
    idx = GpuIndex([ixf], bins)
    h = orc.Hixf([ixf], [ixf["next_ixf"]], [ixf["fname_idx"]])
    sr = Searcher(idx, ratio=0.5)
    noise = rng.integers(0, 2**63, size=600, dtype=np.uint64)
    for name, q in (("split640", np.concatenate([keys_a[:700], noise[:300]])),
                    ("split3", np.concatenate([keys_b[:500], noise[:500]])),
                    ("single", np.concatenate([keys_c[:400], noise])),
                    ("mixed", np.concatenate([keys_a[:300], keys_b[:300], keys_c[:300], noise[:100]])),
                    ("noise", noise)):
        for thr in (1, 5, 17, 100, 299, 300, 301, 400, 401, 500, 501, 699, 700, 701, q.size - 1, q.size, q.size + 1, q.size + 40):
            ub, cnt = sr.bulk_contains(q, thr)
            wub, wcnt, _ = h.bulk_contains(q, thr)
            assert np.array_equal(ub, wub) and np.array_equal(cnt, wcnt), (name, thr, ub.tolist(), wub.tolist())
    # raw bulk_count is never pruned
    assert np.array_equal(sr.ixf_bulk_count(0, noise), h.ixf_bulk_count(0, noise))
    sr.close()
    idx.close()


def test_empty_and_degenerate_batches():
    g, go, lay, host = _planted_setup(13, n_genomes=5, glen=6000, root_bins=66, child_bins=24, n_children=2)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    sr = Searcher(idx)
    # zero reads
    res = sr.search_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert res.read_off.tolist() == [0] and res.user_bin.size == 0 and res.n_hashes.size == 0
    # only empty reads: threshold 0 -> every leaf run, count 0 (reference quirk), per read
    res = sr.search_batch(np.zeros(0, np.uint8), np.zeros(4, np.uint64))
    assert res.n_hashes.tolist() == [0, 0, 0] and int(res.read_off[-1]) == 3 * idx.leaf_runs
    assert not res.count.any()
    want = h.search_batch(np.zeros(0, np.uint8), np.zeros(4, np.uint64))
    assert np.array_equal(res.user_bin, want[2])
    # one read longer than a tile, offsets not starting at 0 (a slice of a larger buffer)
    big = np.concatenate([np.frombuffer(b"TTTT", dtype=np.uint8), g[:5000]])
    offs = np.array([4, 5004], dtype=np.uint64)
    res = sr.search_batch(big, offs)
    want = h.search_batch(g[:5000], np.array([0, 5000], dtype=np.uint64))
    assert np.array_equal(res.n_hashes, want[0]) and np.array_equal(res.user_bin, want[2]) and np.array_equal(res.count, want[3])
    # API misuse is an error, not a crash
    s2 = Searcher(idx)
    with pytest.raises(TaxorError):
        s2.fetch()
    with pytest.raises(TaxorError):
        Searcher(idx, ratio=1.5)
    s2.close()
    sr.close()
    idx.close()


def test_index_validation_errors():
    bins, stride, seg = 64, 64, 16
    base = dict(bins=bins, stride=stride, seg_len=seg, seed=1, next_ixf=np.zeros(bins, np.int64),
                fname_idx=np.arange(bins), data=np.zeros(3 * seg * stride, np.uint8))
    for mutate, needle in [(lambda d: d.update(stride=96), "malformed"),
                           (lambda d: d["fname_idx"].__setitem__(3, 10**6), "out of range"),
                           (lambda d: (d["fname_idx"].__setitem__(3, -1), d["next_ixf"].__setitem__(3, 7)), "bad child")]:
        d = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
        mutate(d)
        if d["stride"] != stride:
            d["data"] = np.zeros(3 * seg * d["stride"], np.uint8)
        with pytest.raises(TaxorError) as e:
            GpuIndex([d], bins)
        assert needle in str(e.value)
    # a minimiser index whose window is shorter than k is rejected loudly (tests/test_gpu_minimiser.py has the rest)
    with pytest.raises(TaxorError):
        GpuIndex([base], bins, use_syncmer=False, window_size=10)
    # a merged bin whose child is referenced twice is not a tree
    a = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    a["fname_idx"][0] = -1
    a["next_ixf"][0] = 1
    a["fname_idx"][1] = -1
    a["next_ixf"][1] = 1
    b = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    b["next_ixf"] = np.full(bins, 1, np.int64)
    with pytest.raises(TaxorError) as e:
        GpuIndex([a, b], bins)
    assert "not a tree" in str(e.value)


def test_fracminhash_scaling():
    """scaled syncmer index (taxor_search.cpp:223-233): hashes are kept iff double(wyhash(h)) <= 2^64/scaling;
    QHASH_COUNT and the threshold use the filtered count"""
    g, go, lay, host = _planted_setup(17, n_genomes=6, glen=20000)
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, origin = synth.synth_reads(g, go, 300, 3000, error_rate=0.01, frac_random=0.1, seed=8)
    for scaling in (2, 10):
        idx = GpuIndex(host, lay["n_user_bins"], scaling=scaling)
        sr = Searcher(idx)
        res = sr.search_batch(bases, offs)
        want = h.search_batch(bases, offs, scaling=scaling, threads=4)
        _compare(res, want, 300)
        full = h.search_batch(bases, offs, threads=4)
        assert 0.5 / scaling < res.n_hashes.sum() / full[0].sum() < 2.0 / scaling
        hoff, hashes = sr.seq_to_syncmers(bases[:3000], offs[:2])
        ref = orc.seq_to_syncmers(bases[:3000].tobytes())
        keep = [int(x) for x in ref if float(orc.wyhash(int(x))) <= float(2**64 - 1) / scaling]
        assert hashes.tolist() == keep
        sr.close()
        idx.close()


def test_deep_hierarchy_chain():
    """a depth-7 merged chain with side leaves at every level: the level loop, queue ping-pong and DFS ordering
    beyond three levels.  Tuples must come out in the reference's recursion order."""
    rng = np.random.default_rng(33)
    depth, bins = 7, 70
    keys = np.unique(rng.integers(0, 2**63, size=900, dtype=np.uint64))
    side = [np.unique(rng.integers(0, 2**63, size=200, dtype=np.uint64)) for _ in range(depth)]
    seg = synth.seg_len_for(2600)
    ixfs = []
    ub = 0
    for lvl in range(depth):
        fname = np.zeros(bins, dtype=np.int64)
        nxt = np.full(bins, lvl, dtype=np.int64)
        planted = {}
        merged_bin = 5 + lvl            # the chain moves to a different bin at every level
        for b in range(bins):
            if lvl + 1 < depth and b == merged_bin:
                fname[b] = -1
                nxt[b] = lvl + 1
                planted[b] = np.unique(np.concatenate([keys] + side[lvl + 1:]))   # union of everything below
            else:
                fname[b] = ub
                ub += 1
        planted[2] = side[lvl]                                          # a side leaf before the merged bin
        if lvl == depth - 1:
            planted[60] = keys                                           # the deepest leaf holds the planted keys
        else:
            planted[65] = keys[: 300 + 40 * lvl]                         # partial copies after the merged bin
        seed, cols = synth.build_columns(planted, seg, 100 + lvl)
        data = rng.integers(0, 256, size=(3 * seg, 128), dtype=np.uint8)
        for b, c in cols.items():
            data[:, b] = c
        ixfs.append(dict(bins=bins, stride=128, seg_len=seg, seed=seed, next_ixf=nxt, fname_idx=fname, data=data.reshape(-1)))
    idx = GpuIndex(ixfs, ub)
    assert idx.depth == depth
    h = orc.Hixf(ixfs, [f["next_ixf"] for f in ixfs], [f["fname_idx"] for f in ixfs])
    sr = Searcher(idx, ratio=0.5)
    q = np.concatenate([keys[:700], side[0][:100], side[3][:150], rng.integers(0, 2**63, size=50, dtype=np.uint64)])
    for thr in (0, 1, 90, 140, 160, 290, 350, 420, 460, 500, 699, 700, 701):
        ubg, cntg = sr.bulk_contains(q, thr)
        ubo, cnto, _ = h.bulk_contains(q, thr)
        assert np.array_equal(ubg, ubo) and np.array_equal(cntg, cnto), (thr, ubg.tolist(), ubo.tolist())
    ubg, cntg = sr.bulk_contains(q, 300)
    assert ubg.size >= 3      # side leaf, deep leaf and partial copies all reported, in DFS order
    sr.close()
    idx.close()


def test_mixed_read_lengths_longest_first_order():
    """ONT-like length skew (200 bp .. 150 kb) in one batch: reads are processed longest-first inside a
    sub-batch but results stay in input order and bit-identical"""
    g, go, lay, host = _planted_setup(41, n_genomes=6, glen=160000, root_bins=70, child_bins=48, n_children=3)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    rng = np.random.default_rng(3)
    reads = []
    for L, n in ((200, 60), (1500, 60), (9000, 30), (40000, 6), (150000, 2), (21, 5), (22, 5)):
        b, o, _ = synth.synth_reads(g, go, n, L, error_rate=0.02, frac_random=0.2, seed=int(rng.integers(1, 10**6)))
        reads += [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(n)]
    perm = rng.permutation(len(reads))
    reads = [reads[i] for i in perm]
    B, O = _cat(reads)
    for sub in (0, 50):
        sr = Searcher(idx, sub_batch_reads=sub)
        res = sr.search_batch(B, O)
        _compare(res, h.search_batch(B, O, threads=8), len(reads))
        sr.close()
    idx.close()


def test_single_wave_query_blocks_short_reads_with_long_outliers():
    """a batch whose MEAN read length is short (the levels of narrow IXFs -- here all of them: 66-bin root, 40-bin
    children -- run the single-wave instantiation k_query_level<.., 64, 256>) but which contains reads with more hashes
    than its 256 probe slots hold (unstaged tiles), reads that fill them exactly, threshold-0 reads and empty ones;
    thresholds on both sides, several sub-batch sizes -- against the oracle"""
    g, go, lay, host = _planted_setup(43, n_genomes=6, glen=40000, root_bins=66, child_bins=40, n_children=3)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    rng = np.random.default_rng(9)
    reads = []
    for L, n in ((700, 500), (1500, 150), (2940, 40), (3000, 20), (12000, 6), (35000, 3), (21, 4), (0, 2)):
        if L < 22:
            reads += [bytes(g[5:5 + L])] * n
            continue
        b, o, _ = synth.synth_reads(g, go, n, L, error_rate=0.02, frac_random=0.2, seed=int(rng.integers(1, 10**6)))
        reads += [bytes(b[int(o[i]):int(o[i + 1])]) for i in range(n)]
    perm = rng.permutation(len(reads))
    reads = [reads[i] for i in perm]
    B, O = _cat(reads)
    assert O[-1] / len(reads) < 2600                       # the condition under which the library picks that instantiation
    for err, sub in ((0.04, 0), (0.1, 97), (0.0, 1000)):
        sr = Searcher(idx, error_rate=err, sub_batch_reads=sub)
        res = sr.search_batch(B, O)
        _compare(res, h.search_batch(B, O, err=err, threads=8), len(reads))
        assert res.user_bin.size > 500
        sr.close()
    idx.close()


def test_ten_thousand_ixfs_queue_grouping_beyond_the_lds_histogram():
    """a hierarchy of 10 101 small IXFs (root -> 100 -> 10 000): the work queues of the deeper levels are grouped by IXF
    id with a counting sort whose per-block histogram covers ids below 8192 in LDS and takes the rest through global
    atomics -- genomes planted under IXFs on both sides of that boundary, against the oracle"""
    rng = np.random.default_rng(77)
    g, go = synth.random_genomes(6, 6000, seed=77)
    planted = [np.unique(orc.seq_to_syncmers(bytes(g[int(go[i]):int(go[i + 1])]))) for i in range(6)]
    next_ub = [0]

    def new_ub():
        next_ub[0] += 1
        return next_ub[0] - 1

    def new_ixf(bins):
        return dict(bins=bins, stride=64 * ((bins + 63) // 64), keys={}, next_ixf=None, fname_idx=np.full(bins, -2, dtype=np.int64),
                    child_of={}, max_elems=None)

    ixfs = [new_ixf(100)]
    mids = []
    for b in range(100):
        ixfs.append(new_ixf(100))
        mids.append(len(ixfs) - 1)
        ixfs[0]["fname_idx"][b] = -1
        ixfs[0]["child_of"][b] = mids[-1]
    leaves = {}
    for m in mids:
        for b in range(100):
            ixfs.append(new_ixf(8))
            leaves[(m, b)] = len(ixfs) - 1
            ixfs[m]["fname_idx"][b] = -1
            ixfs[m]["child_of"][b] = leaves[(m, b)]
    assert len(ixfs) == 10101
    spots = [(mids[0], 0, 3), (mids[40], 7, 0), (mids[79], 99, 5), (mids[81], 0, 1), (mids[99], 50, 7), (mids[99], 99, 2)]
    ids = [leaves[(m, b)] for m, b, _ in spots]
    assert min(ids) < 8192 < max(ids) and sum(i >= 8192 for i in ids) >= 3
    planted_ub = []
    for (m, b, leaf_bin), keys in zip(spots, planted):
        ub = new_ub()
        planted_ub.append(ub)
        ixfs[leaves[(m, b)]]["keys"][leaf_bin] = keys
        ixfs[leaves[(m, b)]]["fname_idx"][leaf_bin] = ub
    lay = dict(ixfs=synth._finalize_layout(ixfs, new_ub, rng, "host"), n_user_bins=next_ub[0])
    host = synth.materialize_host(lay)
    idx = GpuIndex(host, lay["n_user_bins"])
    assert idx.depth == 3 and idx.n_ixf == 10101
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, origin = synth.synth_reads(g, go, 400, 1500, error_rate=0.02, frac_random=0.1, seed=5)
    for sub in (0, 57):
        sr = Searcher(idx, sub_batch_reads=sub, group_always=True)
        res = sr.search_batch(bases, offs)
        _compare(res, h.search_batch(bases, offs, threads=8), 400)
        hit = sum(planted_ub[origin[i]] in [u for u, _ in res.tuples(i)] for i in range(400) if origin[i] >= 0)
        assert hit > 0.85 * int((origin >= 0).sum())
        sr.close()
    idx.close()


def test_pruning_with_rows_wider_than_one_block_pass():
    """5000 bins = 313 units: the dense loop needs two column passes and the alive-unit bitmap spans ten words;
    pruning, sparse probing and the tally must still equal the oracle"""
    rng = np.random.default_rng(44)
    bins, stride = 5000, 5056
    seg = synth.seg_len_for(1200)
    planted = {b: np.unique(rng.integers(0, 2**63, size=900, dtype=np.uint64)) for b in (7, 2500, 4099, 4999)}
    seed, cols = synth.build_columns(planted, seg, 3)
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    for b, c in cols.items():
        data[:, b] = c
    fname = np.arange(bins, dtype=np.int64)
    fname[4096:4100] = 4096                      # a split run straddling the first column pass boundary
    ixf = dict(bins=bins, stride=stride, seg_len=seg, seed=seed, next_ixf=np.zeros(bins, np.int64), fname_idx=fname,
               data=data.reshape(-1))
    idx = GpuIndex([ixf], bins)
    h = orc.Hixf([ixf], [ixf["next_ixf"]], [ixf["fname_idx"]])
    sr = Searcher(idx, ratio=0.5)
    noise = rng.integers(0, 2**63, size=500, dtype=np.uint64)
    for name, q in (("b7", np.concatenate([planted[7][:600], noise[:300]])),
                    ("b4099", np.concatenate([planted[4099][:700], noise[:200]])),
                    ("two", np.concatenate([planted[2500][:450], planted[4999][:450], noise[:100]])),
                    ("noise", noise)):
        for thr in (1, 3, 200, 440, 450, 451, 600, 700, 701, q.size):
            ub, cnt = sr.bulk_contains(q, thr)
            wub, wcnt, _ = h.bulk_contains(q, thr)
            assert np.array_equal(ub, wub) and np.array_equal(cnt, wcnt), (name, thr)
    sr.close()
    idx.close()


@pytest.mark.gpu
def test_gather_ceiling_runs_on_every_row_width():
    """measurement aid: the random-row reader must run (and report a sane rate) for narrow, odd and widest rows"""
    rng = np.random.default_rng(0)
    for bins in (1, 64, 80, 1000, 4096):
        stride = (bins + 63) // 64 * 64
        seg = 4096
        idx = GpuIndex([dict(bins=bins, stride=stride, seg_len=seg, seed=1, next_ixf=np.zeros(bins, np.int64),
                             fname_idx=np.arange(bins), data=rng.integers(0, 256, 3 * seg * stride, dtype=np.uint8))], bins)
        gbps, row_bytes = idx.gather_ceiling(0, want_bytes=1 << 28, reps=2)
        assert row_bytes == (bins + 15) // 16 * 16 and gbps > 1.0
        idx.close()


def test_search_batch_begin_end_overlaps_two_searchers():
    """the drop-in call in two halves: two batches in flight on two searchers of one index, pageable and registered
    host buffers; results equal the blocking call"""
    import ctypes as C
    from taxor_amd import _lib
    g, go, lay, host = _planted_setup(9)
    idx = GpuIndex(host, lay["n_user_bins"])
    b1, o1, _ = synth.synth_reads(g, go, 700, 2500, error_rate=0.02, frac_random=0.2, seed=1)
    b2, o2, _ = synth.synth_reads(g, go, 900, 1200, error_rate=0.03, frac_random=0.1, seed=2)
    s1, s2 = Searcher(idx, sub_batch_reads=128), Searcher(idx, sub_batch_reads=300)
    want1, want2 = s1.search_batch(b1, o1), s2.search_batch(b2, o2)
    assert _lib.lib().taxor_gpu_host_register(b2.ctypes.data, b2.size) == 0
    for _ in range(3):
        s1.search_batch_begin(b1, o1)
        s2.search_batch_begin(b2, o2)
        r2 = s2.search_batch_end()
        r1 = s1.search_batch_end()
        for got, want in ((r1, want1), (r2, want2)):
            assert np.array_equal(got.read_off, want.read_off) and np.array_equal(got.user_bin, want.user_bin)
            assert np.array_equal(got.count, want.count) and np.array_equal(got.n_hashes, want.n_hashes)
    assert _lib.lib().taxor_gpu_host_unregister(b2.ctypes.data) == 0
    with pytest.raises(TaxorError):
        Searcher(idx).search_batch_end()                    # nothing in flight
    s1.close()
    s2.close()
    idx.close()


def test_tuning_knobs_are_ignored_without_the_gate(monkeypatch):
    """TAXOR_QUERY_PRUNE=0 in the environment of a process that loads the library must not switch pruning off unless
    TAXOR_TUNING=1 says the environment is to be read (taxor_amd/csrc/tuning.h); the per-searcher flag always works"""
    g, go, lay, host = _planted_setup(21)
    idx = GpuIndex(host, lay["n_user_bins"])
    bases, offs, _ = synth.synth_reads(g, go, 300, 4000, error_rate=0.02, frac_random=0.1, seed=4)

    def stats(**kw):
        sr = Searcher(idx, time_kernels=True, **kw)
        sr.upload(bases, offs)
        sr.run()
        res, st = sr.fetch(), sr.stats()
        sr.close()
        return res, st

    want, st0 = stats()
    _, st_dense = stats(prune=False)          # what the kernel requests with every hash against every bin (the flag always works)
    assert st0["query_touched_bytes"] != st_dense["query_touched_bytes"]
    monkeypatch.delenv("TAXOR_TUNING", raising=False)
    monkeypatch.setenv("TAXOR_QUERY_PRUNE", "0")
    r1, st1 = stats()
    assert st1["query_touched_bytes"] == st0["query_touched_bytes"]                            # the stray variable changed nothing
    monkeypatch.setenv("TAXOR_TUNING", "1")
    r2, st2 = stats()
    assert st2["query_touched_bytes"] == st_dense["query_touched_bytes"]                       # gate open: the knob is read
    monkeypatch.delenv("TAXOR_TUNING")
    monkeypatch.delenv("TAXOR_QUERY_PRUNE")
    r3, st3 = stats(prune=False)
    assert st3["query_touched_bytes"] == st_dense["query_touched_bytes"]
    for r in (r1, r2, r3):
        assert np.array_equal(r.read_off, want.read_off) and np.array_equal(r.user_bin, want.user_bin) and np.array_equal(r.count, want.count)
    idx.close()


# ------------------------------------------------------------------------------------------------ small calls: lanes + column parts
def _family_like_index(seed, root_bins=1024, child_bins=128):
    """a wide root (so that its rows can be cut into column parts) over a few children, planted genomes on full paths, split bins"""
    g, go = synth.random_genomes(10, 40000, seed=seed)
    hidx = _dummy_index()
    hs = Searcher(hidx, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(10)]
    hs.close()
    hidx.close()
    lay = synth.make_layout(planted, root_bins=root_bins, child_bins=child_bins, n_children=6, seed=seed)
    host = synth.materialize_host(lay)
    return g, go, lay, host


def _mixed_reads(g, go, n, seed, read_len=3000):
    """synthetic reads of one length plus the awkward ones: shorter than k, exactly k, IUPAC / lower case, long, empty"""
    bases, offs, _ = synth.synth_reads(g, go, n, read_len, error_rate=0.02, frac_random=0.15, seed=seed)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(n)]
    extra = [b"", b"ACGTACGTAC", bytes(g[100:122]), b"acgtnnryACGT" * 40, bytes(g[5000:5000 + 12345]), bytes(g[:60])]
    rng = np.random.default_rng(seed)
    for e in extra:
        reads.insert(int(rng.integers(0, len(reads) + 1)), e)
    return reads


@pytest.mark.parametrize("n_reads", [1, 7, 250, 257, 1018, 3000])
def test_small_calls_through_the_lanes_equal_the_oracle(n_reads):
    """calls of up to 16384 reads (the reference's chunk is 1024 records, taxor_search.cpp:315) run as pieces on lanes with the
    root's items in column parts and one fused finalize per piece: same tuples as the oracle and as the pipeline of large batches"""
    g, go, lay, host = _family_like_index(41)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    reads = _mixed_reads(g, go, n_reads, seed=n_reads)
    B, O = _cat(reads)
    Bn = np.frombuffer(orc.dna4_normalise(B.tobytes()), dtype=np.uint8)
    want = h.search_batch(Bn, O, threads=8)
    lanes, plain = Searcher(idx), Searcher(idx, small_path=False)
    for rep in range(3):              # lanes are reused: their counters must come back cleared
        res = lanes.search_batch(B, O)
        _compare(res, want, len(reads))
    _compare(plain.search_batch(B, O), want, len(reads))
    assert res.user_bin.size > 0 or n_reads < 8
    st = lanes.stats()
    assert st["n_reads"] == len(reads) and st["n_tuples"] == res.user_bin.size and st["n_hashes"] == int(res.n_hashes.sum())
    # begin/end halves and the device-resident export after a lane run
    lanes.search_batch_begin(B, O)
    assert lanes.result_sizes() == (len(reads), int(want[1][-1]))
    _compare(lanes.search_batch_end(), want, len(reads))
    lanes.close(); plain.close(); idx.close()


def test_small_call_overflow_falls_back_per_piece():
    """reads without a single hash have threshold 0 and report EVERY leaf run: a few hundred of them
    overflow a lane's hit buffer and its result area -- the piece is classified again through the large-batch pipeline, which
    grows the buffers; reads with more than 64 tuples also take the block-wide sort of the small finalize"""
    g, go, lay, host = _family_like_index(43, root_bins=256, child_bins=64)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    reads = _mixed_reads(g, go, 300, seed=3, read_len=1500)
    few = reads + [b"ACGT"] * 3                    # three threshold-0 reads: > 64 tuples each, fits the lane
    B, O = _cat(few)
    sr = Searcher(idx)
    _compare(sr.search_batch(B, O), h.search_batch(np.frombuffer(orc.dna4_normalise(B.tobytes()), np.uint8), O, threads=8), len(few))
    many = reads + [b"ACGT"] * 700                 # 700 x every leaf run: beyond the hit buffer and the 65536-tuple result area
    B, O = _cat(many)
    want = h.search_batch(np.frombuffer(orc.dna4_normalise(B.tobytes()), np.uint8), O, threads=8)
    assert int(want[1][-1]) > 65536 * 2
    for _ in range(2):
        _compare(sr.search_batch(B, O), want, len(many))
    _compare(sr.search_batch(*_cat(few)), h.search_batch(np.frombuffer(orc.dna4_normalise(_cat(few)[0].tobytes()), np.uint8), _cat(few)[1], threads=8), len(few))
    with pytest.raises(TaxorError):
        sr.search_batch(*_cat([b"ACGT" * 30, b"ACGT!ACGT" * 30]))
    _compare(sr.search_batch(*_cat(few)), h.search_batch(np.frombuffer(orc.dna4_normalise(_cat(few)[0].tobytes()), np.uint8), _cat(few)[1], threads=8), len(few))
    sr.close(); idx.close()


@pytest.mark.parametrize("n_reads", [40, 1018, 3000])
def test_stalled_tree_launch_is_rerun_level_by_level(n_reads):
    """the one-launch traversal of a small piece waits for queue entries with a watchdog; when it fires (forced here:
    TAXOR_SEARCH_FORCE_TREE_STALL makes every block give up at its first empty poll) the piece is classified again through the
    level-by-level pipeline -- identical tuples, counted in tree_stalls_recovered, and the lanes stay usable afterwards"""
    g, go, lay, host = _family_like_index(53)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    reads = _mixed_reads(g, go, n_reads, seed=n_reads + 1)
    B, O = _cat(reads)
    want = h.search_batch(np.frombuffer(orc.dna4_normalise(B.tobytes()), np.uint8), O, threads=8)
    assert int(want[1][-1]) > 0
    stalled, healthy = Searcher(idx, force_tree_stall=True), Searcher(idx)
    for rep in range(3):
        res = stalled.search_batch(B, O)
        _compare(res, want, len(reads))
        st = stalled.stats()
        # pieces of up to 1024 reads take the one-launch traversal (api.hip small_enqueue); larger ones never wait on a queue
        assert st["tree_stalls_recovered"] >= 1 or n_reads > 2048, st["tree_stalls_recovered"]
        assert st["n_tuples"] == res.user_bin.size and st["n_hashes"] == int(res.n_hashes.sum())
    _compare(healthy.search_batch(B, O), want, len(reads))
    assert healthy.stats()["tree_stalls_recovered"] == 0
    stalled.close(); healthy.close(); idx.close()


@pytest.mark.parametrize("root_bins", [128, 200, 1024, 4096])
def test_root_items_in_column_parts_equal_whole_rows(root_bins):
    """QueryArgs::parts forced on for a batch of any size (TAXOR_SEARCH_SPLIT_ALWAYS): a root work item cut into column ranges at
    run boundaries prunes, tallies and reports per part -- the tuples are those of whole rows, pruned and unpruned, through the
    resident pipeline and through the lanes"""
    g, go, lay, host = _family_like_index(47, root_bins=root_bins, child_bins=64)
    idx = GpuIndex(host, lay["n_user_bins"])
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    reads = _mixed_reads(g, go, 600, seed=9, read_len=4000)
    B, O = _cat(reads)
    want = h.search_batch(np.frombuffer(orc.dna4_normalise(B.tobytes()), np.uint8), O, threads=8)
    for kw in (dict(split_always=True, small_path=False), dict(split_always=True, small_path=False, prune=False), dict(split_always=True),
               dict(split_always=True, small_path=False, sub_batch_reads=97)):
        sr = Searcher(idx, **kw)
        _compare(sr.search_batch(B, O), want, len(reads))
        sr.upload(B, O)
        sr.run()
        _compare(sr.fetch(), want, len(reads))
        sr.close()
    idx.close()
