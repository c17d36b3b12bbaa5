"""GPU construction of IXF fingerprint columns (taxor_gpu_index_build_ixf): every key of every
bin must match in its bin -- checked through the GPU query kernel AND through the CPU oracle on the downloaded bytes --
non-members hit at the 2^-8 rate, untouched bins keep their content, duplicate keys fail loudly."""
import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import GpuIndex, Searcher, synth
from taxor_amd._lib import TaxorError

pytestmark = pytest.mark.gpu


def _empty_index(bins, max_elems, seed=3):
    stride = ((bins + 63) // 64) * 64
    seg = synth.seg_len_for(max_elems)
    rng = np.random.default_rng(seed)
    data = rng.integers(0, 256, size=3 * seg * stride, dtype=np.uint8)
    ixf = dict(bins=bins, stride=stride, seg_len=seg, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins), data=data)
    return ixf, GpuIndex([ixf], bins)


@pytest.mark.parametrize("bins,max_elems", [(64, 3000), (200, 20000), (5, 400000)])
def test_build_ixf_members_match(bins, max_elems):
    rng = np.random.default_rng(bins)
    ixf, idx = _empty_index(bins, max_elems)
    before = idx.download_ixf(0).reshape(-1, ixf["stride"]).copy()
    sizes = {0: max_elems, 1: 1, 2: max_elems // 3, bins - 1: max_elems // 2, bins // 2: 17}
    keys = {b: np.unique(rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)) for b, n in sizes.items()}
    keys[2] = np.concatenate([keys[2], np.array([0], dtype=np.uint64)])     # wyhash(poly-A k-mer) = 0 is a legal key
    seed, rounds = idx.build_ixf(0, keys, seed0=12345)
    assert rounds >= 1
    sr = Searcher(idx, ratio=0.5)
    after = idx.download_ixf(0)
    h = orc.Hixf([dict(ixf, seed=seed, data=after)], [ixf["next_ixf"]], [ixf["fname_idx"]])
    for b, ks in keys.items():
        cnt = sr.ixf_bulk_count(0, ks)
        assert cnt[b] == ks.size, (b, int(cnt[b]), ks.size)
        assert np.array_equal(cnt, h.ixf_bulk_count(0, ks))
    neg = rng.integers(0, 2**64 - 1, size=40000, dtype=np.uint64)
    cnt = sr.ixf_bulk_count(0, neg)
    assert abs(cnt[0] / 40000 - 1 / 256) < 0.002 and abs(cnt.mean() / 40000 - 1 / 256) < 0.001
    # bins without keys keep their previous content; built bins are zero outside their assigned rows
    a2 = after.reshape(-1, ixf["stride"])
    untouched = [b for b in range(bins) if b not in keys]
    assert np.array_equal(a2[:, untouched], before[:, untouched])
    # deterministic: same input, same seed -> same bytes
    _, idx2 = _empty_index(bins, max_elems)
    seed2, _ = idx2.build_ixf(0, keys, seed0=12345)
    assert seed2 == seed
    b2 = idx2.download_ixf(0).reshape(-1, ixf["stride"])
    for b in keys:
        assert np.array_equal((a2[:, b] != 0).sum(), (b2[:, b] != 0).sum())
    sr.close()
    idx.close()
    idx2.close()


def test_build_ixf_matches_cpu_builder_semantics_and_search():
    """an index whose planted columns are built on the GPU classifies reads exactly like the oracle says"""
    g, go = synth.random_genomes(4, 30000, seed=77)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    hs.close()
    dummy.close()
    planted = {3 + 5 * i: hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(4)}
    ixf, idx = _empty_index(100, 4000, seed=9)
    seed, _ = idx.build_ixf(0, planted, seed0=99)
    after = idx.download_ixf(0)
    h = orc.Hixf([dict(ixf, seed=seed, data=after)], [ixf["next_ixf"]], [ixf["fname_idx"]])
    bases, offs, origin = synth.synth_reads(g, go, 200, 2000, error_rate=0.02, frac_random=0.2, seed=5)
    sr = Searcher(idx)
    res = sr.search_batch(bases, offs)
    nh, off, ub, cnt, _ = h.search_batch(bases, offs, threads=4)
    assert np.array_equal(res.read_off, off) and np.array_equal(res.user_bin, ub) and np.array_equal(res.count, cnt)
    hit = sum((3 + 5 * origin[i]) in [u for u, _ in res.tuples(i)] for i in range(200) if origin[i] >= 0)
    assert hit > 0.8 * (origin >= 0).sum()
    sr.close()
    idx.close()


def test_build_ixf_errors():
    ixf, idx = _empty_index(64, 100)
    dup = np.array([5, 7, 7, 9], dtype=np.uint64)
    with pytest.raises(TaxorError) as e:
        idx.build_ixf(0, {0: dup})
    assert e.value.code == -4 and "duplicate" in str(e.value)
    with pytest.raises(TaxorError):
        idx.build_ixf(0, {0: np.arange(10**6, dtype=np.uint64)})    # more keys than rows
    idx.close()


def test_layout_built_on_gpu_equals_oracle_search():
    """synth.make_layout(build='gpu') + device_index: the whole planted hierarchy constructed by the GPU builder"""
    g, go = synth.random_genomes(7, 15000, seed=5)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    hs.close()
    dummy.close()
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(7)]
    lay = synth.make_layout(planted, root_bins=70, child_bins=40, n_children=3, seed=6, build="gpu")
    idx = synth.device_index(lay)
    host = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], next_ixf=f["next_ixf"],
                 fname_idx=f["fname_idx"], data=idx.download_ixf(i)) for i, f in enumerate(lay["ixfs"])]
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, origin = synth.synth_reads(g, go, 300, 2000, error_rate=0.02, frac_random=0.1, seed=2)
    sr = Searcher(idx)
    res = sr.search_batch(bases, offs)
    nh, off, ub, cnt, _ = h.search_batch(bases, offs, threads=4)
    assert np.array_equal(res.read_off, off) and np.array_equal(res.user_bin, ub) and np.array_equal(res.count, cnt)
    hit = sum(lay["planted_user_bin"][origin[i]] in [u for u, _ in res.tuples(i)] for i in range(300) if origin[i] >= 0)
    assert hit > 0.8 * (origin >= 0).sum()
    sr.close()
    idx.close()


def _tree_layout(rng, n_genomes=10, glen=5000):
    """random hierarchy with LEAF key sets only (what a build front end hands over after the layout step)"""
    g, go = synth.random_genomes(n_genomes, glen, seed=int(rng.integers(1, 2**31)))
    planted = [orc.seq_to_syncmers(bytes(g[int(go[i]):int(go[i + 1])])) for i in range(n_genomes)]
    while True:     # a hierarchy worth building: several IXFs, most genomes placed
        lay = synth.random_layout(planted, rng, max_depth=4)
        if len(lay["ixfs"]) >= 3 and sum(p is not None for p in lay["planted_user_bin"]) >= n_genomes - 2:
            return g, go, planted, lay


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_hierarchical_build_on_gpu(seed):
    """taxor_gpu_index_build_hixf: leaf key sets in, merged bins = union of their child IXF computed on the device,
    every IXF peeled on the device.  Checked three ways: every leaf key is found in its bin and in the merged bin of
    every ancestor (oracle on the downloaded bytes), searches agree with the oracle, and with a host-built index of
    the same layout they report the same user bins for the control reads."""
    rng = np.random.default_rng(seed)
    g, go, planted, lay = _tree_layout(rng)
    ix = lay["ixfs"]
    # leaf keys per (ixf, bin) from the layout's columns' key sets: rebuild them from the planted assignment
    leaf = {}
    for i, f in enumerate(ix):
        for b, keys in f.get("leaf_keys", {}).items():
            leaf[(i, b)] = keys
    assert leaf, "layout carries no leaf keys"
    # rows sized for the unions: merged bins hold whole subtrees
    def subtree_keys(i):
        ks = [k for (ii, b), k in leaf.items() if ii == i]
        for b in range(ix[i]["bins"]):
            if ix[i]["fname_idx"][b] == -1:
                ks.append(subtree_keys(int(ix[i]["next_ixf"][b])))
        return np.unique(np.concatenate(ks)) if ks else np.zeros(0, np.uint64)
    shapes = []
    for i, f in enumerate(ix):
        biggest = max([len(k) for (ii, b), k in leaf.items() if ii == i] +
                      [len(subtree_keys(int(f["next_ixf"][b]))) for b in range(f["bins"]) if f["fname_idx"][b] == -1] + [1])
        shapes.append(dict(bins=f["bins"], stride=f["stride"], seg_len=synth.seg_len_for(biggest), seed=7 + i,
                           next_ixf=f["next_ixf"], fname_idx=f["fname_idx"], data=None))
    idx = GpuIndex(shapes, lay["n_user_bins"])
    for i in range(len(shapes)):
        idx.fill_random(i, 100 + i)
    rounds = idx.build_hixf(leaf, seed0=11)
    assert rounds >= 1
    host = [dict(s, seed=idx.ixf_seed(i), data=idx.download_ixf(i)) for i, s in enumerate(shapes)]
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    # (1) membership: leaf keys in their bin, and in the merged bin of the parent, grandparent, ...
    parent = {}
    for i, f in enumerate(host):
        for b in range(f["bins"]):
            if f["fname_idx"][b] == -1:
                parent[int(f["next_ixf"][b])] = (i, b)
    for (i, b), keys in leaf.items():
        cnt = h.ixf_bulk_count(i, keys)
        assert cnt[b] == len(keys)
        node = i
        while node in parent:
            pi, pb = parent[node]
            assert h.ixf_bulk_count(pi, keys)[pb] == len(keys), (i, b, pi, pb)
            node = pi
    # (2) search parity on the GPU-built index, (3) control reads classify
    bases, offs, origin = synth.synth_reads(g, go, 200, 1500, error_rate=0.01, frac_random=0.1, seed=seed)
    sr = Searcher(idx, error_rate=0.04)
    res = sr.search_batch(bases, offs)
    Bn = np.frombuffer(orc.dna4_normalise(bases.tobytes()), dtype=np.uint8)
    nh, off, ub, cnt, _ = h.search_batch(Bn, offs, threads=4)
    assert np.array_equal(res.read_off, off) and np.array_equal(res.user_bin, ub) and np.array_equal(res.count, cnt)
    hit = tot = 0
    for r in range(200):
        if origin[r] >= 0 and lay["planted_user_bin"][origin[r]] is not None:
            tot += 1
            hit += lay["planted_user_bin"][origin[r]] in ub[int(off[r]):int(off[r + 1])].tolist()
    assert tot > 50 and hit > 0.8 * tot, (hit, tot)
    sr.close()
    idx.close()


def test_hierarchical_build_rejects_keys_on_merged_bins():
    nx = np.array([1, 0, 0, 0], dtype=np.int64)
    fn = np.array([-1, 0, 1, 2], dtype=np.int64)
    shapes = [dict(bins=4, stride=64, seg_len=64, seed=1, next_ixf=nx, fname_idx=fn, data=None),
              dict(bins=4, stride=64, seg_len=64, seed=2, next_ixf=np.full(4, 1, np.int64), fname_idx=np.array([3, 4, 5, 6], np.int64), data=None)]
    idx = GpuIndex(shapes, 7)
    with pytest.raises(TaxorError):
        idx.build_hixf({(0, 0): np.arange(5, dtype=np.uint64)})
    idx.build_hixf({(1, 1): np.arange(1, 40, dtype=np.uint64), (0, 2): np.arange(100, 130, dtype=np.uint64)})
    idx.close()


@pytest.mark.parametrize("kh,sm,rot,red,fp", [(0, 1, 21, 1, 1), (2, 2, 16, 2, 2), (3, 0, 16, 0, 3), (1, 3, 21, 1, 0)])
def test_arithmetic_code_through_builder_query_and_oracle(kh, sm, rot, red, fp):
    """an index created with another reading of the un-vendored IXF arithmetic (taxor_hixf_view::ixf_arith): the GPU builder,
    the query kernel (raw bulk_count, and bulk_contains with pruning on) and the oracle parametrised with the same code agree
    bit for bit; the default reading does not find the keys"""
    from taxor_amd.search import arith_code
    code = arith_code(kh, sm, rot, red, fp)
    rng = np.random.default_rng(code)
    bins, max_elems = 130, 6000
    stride = 192
    seg = synth.seg_len_for(max_elems)
    data = rng.integers(0, 256, size=3 * seg * stride, dtype=np.uint8)
    ixf = dict(bins=bins, stride=stride, seg_len=seg, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins), data=data)
    idx = GpuIndex([ixf], bins, arith=code)
    keys = {b: np.unique(rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)) for b, n in {0: 6000, 64: 2500, 129: 900}.items()}
    seed, _ = idx.build_ixf(0, keys, seed0=777)
    after = idx.download_ixf(0)
    h = orc.Hixf([dict(ixf, seed=seed, data=after)], [ixf["next_ixf"]], [ixf["fname_idx"]], arith=code)
    h0 = orc.Hixf([dict(ixf, seed=seed, data=after)], [ixf["next_ixf"]], [ixf["fname_idx"]])
    sr = Searcher(idx, ratio=0.5)
    for b, ks in keys.items():
        cnt = sr.ixf_bulk_count(0, ks)
        assert cnt[b] == ks.size
        assert np.array_equal(cnt, h.ixf_bulk_count(0, ks))
        assert h0.ixf_bulk_count(0, ks)[b] < ks.size * 0.05
        mix = np.concatenate([ks[:700], rng.integers(0, 2**64 - 1, size=300, dtype=np.uint64)])
        for thr in (1, 500, 690, 700, 712, 1000):
            ub, ct = sr.bulk_contains(mix, thr)
            wub, wct, _ = h.bulk_contains(mix, thr)
            assert np.array_equal(ub, wub) and np.array_equal(ct, wct), (b, thr)
    sr.close()
    idx.close()


def test_two_builds_are_byte_identical():
    """which of a key's singleton rows wins the claim is a race; the free row that is ASSIGNED is the lowest-segment row listed for the
    key's round, so the columns are a function of (keys, seed) alone -- ranks that each build their replica of an index hold the
    same bytes (bench.py's strong-scaling digest rests on this)"""
    rng = np.random.default_rng(12)
    keys = {b: np.unique(rng.integers(0, 2**64 - 1, size=n, dtype=np.uint64)) for b, n in {0: 300000, 1: 290000, 7: 5, 63: 120000}.items()}
    got = []
    for _ in range(3):
        ixf, idx = _empty_index(64, 300000, seed=3)
        seed, rounds = idx.build_ixf(0, keys, seed0=99)
        got.append((seed, idx.download_ixf(0)))
        idx.close()
    assert got[0][0] == got[1][0] == got[2][0]
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][1], got[2][1])


@pytest.mark.parametrize("n", [(1 << 24) + 12345, (1 << 26) + 777])
def test_bins_of_more_than_16M_keys(n):
    """a merged bin high in a large hierarchy: with 2^24 .. 2^26 - 1 keys its degree | index-sum words stay 32 bits wide with the degree in 6 bits
    (the sum modulo 2^26 is the last key's index), from 2^26 keys on they are 64 bits wide"""
    keys = synth.synth_keys_host(0, n, 5)
    bins = 3
    stride, seg = 64, synth.seg_len_for(n)
    ixf = dict(bins=bins, stride=stride, seg_len=seg, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins),
               data=np.zeros(3 * seg * stride, dtype=np.uint8))
    idx = GpuIndex([ixf], bins)
    seed, rounds = idx.build_ixf(0, {1: keys, 2: keys[:1000]}, seed0=4)
    del keys
    after = idx.download_ixf(0)
    idx.close()
    h = orc.Hixf([dict(ixf, seed=seed, data=after)], [ixf["next_ixf"]], [ixf["fname_idx"]])
    found, _ = h.synth_keys_found(0, 1, 0, n, 5)
    assert found == n
    found, _ = h.synth_keys_found(0, 2, 0, 1000, 5)
    assert found == 1000
    found, _ = h.synth_keys_found(0, 2, 1000, 200000, 5)                  # non-members of bin 2
    assert abs(found / 200000 - 1 / 256) < 0.001


def test_ixf_larger_than_the_scratch_is_built_in_chunks_of_its_bins(monkeypatch):
    """TAXOR_TUNING=1 TAXOR_BUILD_SCRATCH_MB=1: the budget is raised to what ONE bin needs, so this IXF of 150 bins goes through the engine a few bins
    at a time; the hierarchy above it is built from its union as usual"""
    monkeypatch.setenv("TAXOR_TUNING", "1")                  # (the knob sits behind the library's one gate, tuning.h)
    monkeypatch.setenv("TAXOR_BUILD_SCRATCH_MB", "1")
    nc, cb, kpb = 3, 150, 20000
    shapes, ub, counts = synth.full_hierarchy_shapes(nc, cb, kpb, slack=1.1)
    idx = GpuIndex(shapes, ub)
    st, off = idx.build_hixf_synth(counts, salt=77, seed0=3)
    assert st["chunks"] > 30 and st["keys_inserted"] == 2 * nc * cb * kpb and st["scratch_bytes"] < 100 << 20        # (one root bin of 3.3 M keys is what sets the size)
    host = [dict(s, seed=idx.ixf_seed(i), data=idx.download_ixf(i)) for i, s in enumerate(shapes)]
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    rb = shapes[0]["bins"]
    for c in range(1, nc + 1):
        for b in range(cb):
            g = rb + (c - 1) * cb + b
            assert h.synth_keys_found(c, b, int(off[g]), kpb, 77)[0] == kpb
            assert h.synth_keys_found(0, c - 1, int(off[g]), kpb, 77)[0] == kpb
    idx.close()


def test_many_small_ixfs_share_chunks_and_only_a_failed_ixf_is_reseeded():
    """400 child IXFs of 4 bins with 40 keys each: one chunk holds all of them (one launch sequence for 400 IXFs); tiny
    hypergraphs fail to peel now and then -- such an IXF alone is redone under a redrawn seed, every key is found in the end, and
    the duplicate keys shared by neighbouring bins are merged in the unions"""
    nc, cb, kpb = 400, 4, 40
    shapes, ub, counts = synth.full_hierarchy_shapes(nc, cb, kpb, slack=2.0)     # (a roomy root: its 400 small bins must all peel under ONE seed)
    for f in shapes[1:]:
        f["seg_len"] = 22                       # 66 rows for 40 keys, without the +32 slack: a bin peels under 88 % of the seeds, four under 60 %
    idx = GpuIndex(shapes, ub)
    rng = np.random.default_rng(8)
    leaf = {}
    for c in range(1, nc + 1):
        shared = rng.integers(1, 2**63, size=8, dtype=np.uint64)                    # the same 8 keys in every bin of a child
        for b in range(cb):
            leaf[(c, b)] = np.unique(np.concatenate([shared, rng.integers(1, 2**63, size=kpb - 8, dtype=np.uint64)]))
    rounds = idx.build_hixf(leaf, seed0=21)
    seeds = [idx.ixf_seed(i) for i in range(1, nc + 1)]
    redrawn = sum(1 for i, sd in enumerate(seeds) if sd != (21 + 0x9E3779B97F4A7C15 * (i + 1)) % 2**64)
    print(f"\n{redrawn} of {nc} IXFs were redone under a redrawn seed")
    assert 0 < redrawn < nc, redrawn                                                 # some failed and were redone, not all
    sr = Searcher(idx, ratio=0.5)
    for c in (1, 17, nc):
        allk = np.unique(np.concatenate([leaf[(c, b)] for b in range(cb)]))
        assert allk.size == cb * (kpb - 8) + 8
        assert sr.ixf_bulk_count(0, allk)[c - 1] == allk.size
        for b in (0, 1, cb - 1):
            assert sr.ixf_bulk_count(c, leaf[(c, b)])[b] == leaf[(c, b)].size
    sr.close()
    idx.close()


def test_generated_keys_need_no_memory_and_mix_with_real_ones():
    """taxor_gpu_index_build_hixf_gen: leaf bins whose keys are GENERATED by the kernels (synth_key of a running index) next to bins
    that bring real keys.  A child of generated bins with consecutive index ranges hands its parent a generated range (nothing is
    materialised); a child that mixes both kinds, or whose ranges leave a gap, gets its union written out.  Checked by the oracle:
    every key of every leaf bin is found in its bin and in the root's merged bin above it."""
    import ctypes as C
    from taxor_amd import _lib
    nc, cb, kpb, salt = 5, 8, 3000, 424242
    shapes, ub, _ = synth.full_hierarchy_shapes(nc, cb, kpb, slack=1.5)
    rb = shapes[0]["bins"]
    n_bins = rb + nc * cb
    rng = np.random.default_rng(77)
    gen_first = np.zeros(n_bins, np.uint64)
    gen_count = np.zeros(n_bins, np.uint64)
    real = {}
    nxt = 10**9                                             # running index of the generated keys
    for c in range(1, nc + 1):
        for b in range(cb):
            g = rb + (c - 1) * cb + b
            n = int(rng.integers(kpb // 2, kpb + 1))
            kind = "gen"
            if c == 2 and b == 3: kind = "real"             # a child that mixes both kinds
            if c == 3: kind = "real"                        # a child of real keys only
            if c == 5 and b == 4: kind = "empty"
            if kind == "real":
                real[g] = np.unique(rng.integers(1, 2**63, size=n, dtype=np.uint64))
            elif kind == "gen":
                if c == 4 and b == 5: nxt += 12345          # a gap in child 4's index ranges: no single generated range any more
                gen_first[g], gen_count[g] = nxt, n
                nxt += n
    off = np.zeros(n_bins + 1, np.uint64)
    for g, k in real.items():
        off[g + 1] = k.size
    off = np.cumsum(off).astype(np.uint64)
    keys = np.concatenate([real[g] for g in sorted(real)])
    idx = GpuIndex(shapes, ub)
    for i in range(len(shapes)):
        idx.fill_random(i, 5 + i)
    st = _lib.BuildStats()
    _lib.check(_lib.lib().taxor_gpu_index_build_hixf_gen(idx._h, keys.ctypes.data_as(C.c_void_p), 0, off.ctypes.data_as(C.c_void_p),
                                                         gen_first.ctypes.data_as(C.c_void_p), gen_count.ctypes.data_as(C.c_void_p), salt, 9, C.byref(st)))
    n_leaf = int(gen_count.sum()) + keys.size
    assert st.keys_inserted == 2 * n_leaf
    host = [dict(s, seed=idx.ixf_seed(i), data=idx.download_ixf(i)) for i, s in enumerate(shapes)]
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    for c in range(1, nc + 1):
        for b in range(cb):
            g = rb + (c - 1) * cb + b
            if gen_count[g]:
                n = int(gen_count[g])
                assert h.synth_keys_found(c, b, int(gen_first[g]), n, salt)[0] == n and h.synth_keys_found(0, c - 1, int(gen_first[g]), n, salt)[0] == n, (c, b)
            elif g in real:
                assert h.ixf_bulk_count(c, real[g])[b] == real[g].size and h.ixf_bulk_count(0, real[g])[c - 1] == real[g].size, (c, b)
    # a bin may bring keys or have them generated, not both; merged bins neither
    bad = gen_count.copy()
    bad[sorted(real)[0]] = 5
    with pytest.raises(TaxorError):
        _lib.check(_lib.lib().taxor_gpu_index_build_hixf_gen(idx._h, keys.ctypes.data_as(C.c_void_p), 0, off.ctypes.data_as(C.c_void_p),
                                                             gen_first.ctypes.data_as(C.c_void_p), bad.ctypes.data_as(C.c_void_p), salt, 9, C.byref(st)))
    idx.close()
