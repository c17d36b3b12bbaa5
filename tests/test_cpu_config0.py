"""BASELINE.json configs[0]: the viral-class k22/s12 index (373 MB, README.md:50) and 10 k synthetic 5 kb reads through
the CPU path end to end -- no GPU: index built on the host, written to and read back from a `.hixf`, reads classified
by the oracle in the reference's do_parallel shape (taxor_search.cpp:196-326), per-read lines produced by the library's
host formatter (taxor_search.cpp:268-305) and parsed the way `taxor profile` does.  Plumbing: the same layout, seeds
and reads are what the GPU test of configs[1] (tests/test_gpu_fullsize.py) compares bit for bit."""
import os

import numpy as np

from oracle import oracle as orc
from taxor_amd import synth
from taxor_amd.hixf_file import HixfFile, store_hixf
from tests.test_hixf_file_cpu import HEADER, expected_lines, make_species, parse_search_results_like_taxor_profile


def test_viral_class_ten_thousand_reads_cpu_path(tmp_path):
    n_reads, read_len = 10_000, 5000
    g, go = synth.random_genomes(64, 100000, seed=synth.DEFAULT_SEED)
    planted = [orc.seq_to_syncmers(bytes(g[int(go[i]):int(go[i + 1])])) for i in range(64)]
    total = 373e6                         # root 256 bins (40 %), 252 children + 1 grandchild of 64 bins
    root_max = int((total * 0.4 / 256 - 32) / 1.23)
    child_max = max(int((total * 0.6 / (253 * 64) - 32) / 1.23), max(len(p) for p in planted) + 64)
    lay = synth.make_layout(planted, root_bins=256, child_bins=64, n_children=252, root_max_elems=root_max,
                            child_max_elems=child_max, seed=synth.DEFAULT_SEED)
    host = synth.materialize_host(lay)
    assert 0.3e9 < sum(f["data"].size for f in host) < 0.5e9 and len(host) == 254
    sp = make_species(lay)
    path = tmp_path / "viral_class.hixf"
    store_hixf(path, host, lay["n_user_bins"], sp)
    assert 0.3e9 < os.path.getsize(path) < 0.5e9
    del host
    hx = HixfFile(path)                   # the loaded file drives the search, like load_index.hpp:27-38
    assert (hx.k, hx.s, hx.t, hx.use_syncmer) == (22, 12, 5, True) and not hx.foreign_schema
    h = orc.Hixf(hx.ixfs, [f["next_ixf"] for f in hx.ixfs], [f["fname_idx"] for f in hx.ixfs])

    bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=0.02, frac_random=0.1,
                                            seed=synth.DEFAULT_SEED, threads=os.cpu_count() or 8)
    nh, off, ub, cnt, visited = h.search_batch(bases, offs, err=0.04, threads=os.cpu_count() or 8)
    # thread count and chunking do not change a tuple (do_parallel slices reads, taxor_search.cpp:315-326)
    nh1, off1, ub1, cnt1, _ = h.search_batch(bases[: int(offs[400])], offs[:401], err=0.04, threads=1)
    t = int(off[400])
    assert np.array_equal(nh[:400], nh1) and np.array_equal(off[:401], off1) and np.array_equal(ub[:t], ub1) and np.array_equal(cnt[:t], cnt1)
    per = np.diff(off.astype(np.int64))
    planted_reads = origin >= 0
    assert (per[planted_reads] > 0).mean() > 0.9 and (per[~planted_reads] > 0).mean() < 0.01
    own = np.array([lay["planted_user_bin"][o] if o >= 0 else -1 for o in origin])
    assert np.mean([own[i] in ub[int(off[i]):int(off[i + 1])].tolist() for i in np.flatnonzero(planted_reads)[:2000]]) > 0.9
    assert 400 < nh.mean() < 470                              # ~1 distinct syncmer per 11.5 bp (README.md:206-210)
    assert visited > int(nh.astype(np.int64).sum()) * 3 * 256   # every read pays the root; classified ones a child too

    # per-read text -> TSV -> what `taxor profile` reads back
    text = HEADER
    for i in range(n_reads):
        lo, hi = int(off[i]), int(off[i + 1])
        line = hx.format_read(f"read_{i} ch={i % 512}", read_len, int(nh[i]), ub[lo:hi], cnt[lo:hi])
        if i < 300:
            assert line == expected_lines(sp, f"read_{i} ch={i % 512}", read_len, int(nh[i]), [(int(a), int(b)) for a, b in zip(ub[lo:hi], cnt[lo:hi])])
        text += line
    out = tmp_path / "out.tsv"
    out.write_text(text)
    results, taxpath = parse_search_results_like_taxor_profile(out.read_text())
    assert len(results) == n_reads
    hits = [r for rs in results.values() for r in rs if r["accession_id"] != "-"]
    assert len(hits) >= int(planted_reads.sum() * 0.9) and all(r["query_len"] == read_len for rs in results.values() for r in rs)
    assert all(r["query_hash_match"] >= orc.threshold(r["query_hash_count"], 22, 0.04) for r in hits)
    hx.close()

    # the same reads at BASELINE.md's nominal 4 % read error: 45 % of their 22-mers survive, below the model's 0.508
    b4, o4, g4 = synth.synth_reads(g, go, 500, read_len, error_rate=0.04, frac_random=0.0, seed=7, threads=4)
    hx = HixfFile(path)
    h = orc.Hixf(hx.ixfs, [f["next_ixf"] for f in hx.ixfs], [f["fname_idx"] for f in hx.ixfs])
    _, off4, _, _, _ = h.search_batch(b4, o4, err=0.04, threads=os.cpu_count() or 8)
    assert (np.diff(off4.astype(np.int64)) > 0).mean() < 0.2
    hx.close()
