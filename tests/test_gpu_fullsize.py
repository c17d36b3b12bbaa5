"""BASELINE.json configs[1] at full size: viral-class index on one MI355X, 1 M synthetic 5 kb reads.
The oracle checks a sample bit for bit; the whole batch is checked through size-independent properties:
sub-batch invariance, pruning on/off equality, idempotence, and counter checksums."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import GpuIndex, Searcher, synth

pytestmark = pytest.mark.gpu


def _csr_equal(a, b):
    return (np.array_equal(a.read_off, b.read_off) and np.array_equal(a.user_bin, b.user_bin)
            and np.array_equal(a.count, b.count) and np.array_equal(a.n_hashes, b.n_hashes))


def test_viral_class_one_million_reads():
    n_reads, read_len = 1_000_000, 5000
    g, go = synth.random_genomes(64, 100000, seed=synth.DEFAULT_SEED)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    hs.close()
    dummy.close()
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(64)]
    # 373 MB: root 256 bins (40 %), 252 children + 1 grandchild of 64 bins
    total = 373e6
    root_max = int((total * 0.4 / 256 - 32) / 1.23)
    child_max = max(int((total * 0.6 / (253 * 64) - 32) / 1.23), max(len(p) for p in planted) + 64)
    lay = synth.make_layout(planted, root_bins=256, child_bins=64, n_children=252, root_max_elems=root_max,
                            child_max_elems=child_max, seed=synth.DEFAULT_SEED)
    idx = synth.device_index(lay)
    assert 0.3e9 < idx.data_bytes < 0.5e9
    bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=0.02, frac_random=0.1,
                                            seed=synth.DEFAULT_SEED, threads=os.cpu_count() or 8)

    sr = Searcher(idx, time_kernels=True)
    sr.upload(bases, offs)
    sr.run()
    res = sr.fetch()
    st = sr.stats()
    assert st["n_reads"] == n_reads and st["n_bases"] == n_reads * read_len
    assert st["n_hashes"] == int(res.n_hashes.astype(np.int64).sum())
    assert st["n_tuples"] == res.user_bin.size == int(res.read_off[-1])
    assert np.all(np.diff(res.read_off.astype(np.int64)) >= 0)
    if os.environ.get("TAXOR_QUERY_PRUNE") != "0":
        assert st["query_touched_bytes"] < st["query_bytes"]      # pruning removed traffic
    # idempotence on the resident batch
    sr.run()
    assert _csr_equal(res, sr.fetch())
    sr.close()

    # sub-batch invariance and pruning on/off: different launch partitions, same answer
    s2 = Searcher(idx, sub_batch_reads=50021)
    s2.upload(bases, offs)
    s2.run()
    assert _csr_equal(res, s2.fetch())
    s2.close()
    os.environ["TAXOR_QUERY_PRUNE"] = "0"
    try:
        s3 = Searcher(idx, sub_batch_reads=131072)
    finally:
        del os.environ["TAXOR_QUERY_PRUNE"]
    s3.upload(bases, offs)
    s3.run()
    r3 = s3.fetch()
    st3 = s3.stats()
    assert _csr_equal(res, r3)
    assert st3["query_touched_bytes"] == st3["query_bytes"] == st["query_bytes"]
    s3.close()

    # oracle, bit for bit, on a 5 % sample spread over the batch (first, middle, last reads)
    host = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], next_ixf=f["next_ixf"],
                 fname_idx=f["fname_idx"], data=idx.download_ixf(i)) for i, f in enumerate(lay["ixfs"])]
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    for lo in (0, n_reads // 2 - 8000, n_reads - 17000):
        hi = lo + 17000
        sub_off = offs[lo:hi + 1] - offs[lo]
        nh, off, ub, cnt, _ = h.search_batch(bases[int(offs[lo]):int(offs[hi])], sub_off, threads=min(64, os.cpu_count() or 8))
        a, b = int(res.read_off[lo]), int(res.read_off[hi])
        assert np.array_equal(res.n_hashes[lo:hi], nh)
        assert np.array_equal(res.read_off[lo:hi + 1] - res.read_off[lo], off)
        assert np.array_equal(res.user_bin[a:b], ub) and np.array_equal(res.count[a:b], cnt)
    # positive control: planted reads report their genome's user bin
    per = np.diff(res.read_off.astype(np.int64))
    planted_reads = origin >= 0
    assert (per[planted_reads] > 0).mean() > 0.9
    assert (per[~planted_reads] > 0).mean() < 0.01
    idx.close()
