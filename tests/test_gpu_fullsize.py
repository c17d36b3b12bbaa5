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
    s3 = Searcher(idx, sub_batch_reads=131072, prune=False)
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


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[2] and configs[3] at full size: RefSeq-class (9.9 GB) and GTDB-class (113 GB) indexes resident
# in HBM, 10 M synthetic 10 kb reads each (ten batches of 1 M, like the CLI feeds them), the workload bench.py measures
# (families of related strains, read error 0.02 -- bench.py's default --read-error; the search runs at --error-rate 0.04).  Bit-exact oracle samples at the first / middle / last reads, pruning
# on/off equality over ALL reads, sub-batch invariance and idempotence, counter checksums, positive/negative controls.
# ---------------------------------------------------------------------------------------------------------------------
def _digest(res):
    """order-sensitive checksum of a CSR result (a checksum of checksums over batches is compared between runs)"""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for a in (res.read_off, res.user_bin, res.count, res.n_hashes):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _full_size_class(workload, min_index_bytes, max_index_bytes):
    import psutil
    import torch

    import bench

    free_hbm, _ = torch.cuda.mem_get_info(0)
    if free_hbm < max_index_bytes + 40e9:
        pytest.skip(f"needs {max_index_bytes/1e9 + 40:.0f} GB of free HBM, have {free_hbm/1e9:.0f}")
    n_batches, reads_per_batch, read_len = 10, 1_000_000, 10_000
    avail = psutil.virtual_memory().available
    while n_batches > 1 and n_batches * reads_per_batch * read_len * 1.3 + 60e9 > avail:
        n_batches -= 1                                           # the GPU pool's hosts have 3 TB; a small host runs fewer batches
    args = bench.parse_args(["--workload", workload, "--reads", str(reads_per_batch), "--read-len", str(read_len),
                             "--batches", str(n_batches)])
    wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
    assert min_index_bytes < idx.data_bytes < max_index_bytes
    assert info["fam_size"] == 16 and lay["depth"] == 3 and lay["split_runs"] > 8

    host = None
    # oracle samples: 5 000 consecutive reads out of EVERY batch (50 000 in all, at a different place in each batch; the first
    # reads of the first batch and the last reads of the last one among them)
    S = 5000
    checks = []           # (batch, lo, hi)
    for b_ in range(n_batches):
        lo_ = 0 if b_ == 0 else reads_per_batch - S if b_ == n_batches - 1 else (b_ * 97003) % (reads_per_batch - S)
        checks.append((b_, lo_, lo_ + S))

    sr = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
    dense = Searcher(idx, error_rate=args.error_rate, sub_batch_reads=65536, prune=False)
    planted_ub = np.array([u if u is not None else -1 for u in lay["planted_user_bin"]], dtype=np.int64)
    total_reads = total_tuples = total_hashes = 0
    hit_own = n_planted = n_random = random_hit = 0
    digests, digests_dense = [], []
    samples = {}
    for b, (bases, offs) in enumerate(batches):
        sr.upload(bases, offs)
        sr.run()
        res = sr.fetch()
        st = sr.stats()
        n = offs.size - 1
        assert st["n_reads"] == n and st["n_bases"] == int(offs[-1])
        assert st["n_hashes"] == int(res.n_hashes.astype(np.int64).sum())
        assert st["n_tuples"] == res.user_bin.size == int(res.read_off[-1])
        assert np.all(np.diff(res.read_off.astype(np.int64)) >= 0)
        assert st["query_touched_bytes"] < st["query_bytes"]                  # pruning removed traffic
        total_reads += n
        total_tuples += res.user_bin.size
        total_hashes += st["n_hashes"]
        digests.append(_digest(res))
        # pruning off, another sub-batch partition: the identical CSR for every one of the 10 M reads
        dense.upload(bases, offs)
        dense.run()
        rd = dense.fetch()
        sd = dense.stats()
        assert sd["query_touched_bytes"] == sd["query_bytes"] == st["query_bytes"]
        digests_dense.append(_digest(rd))
        if b == 0:
            sr.run()                                                            # idempotence on the resident batch
            assert _csr_equal(res, sr.fetch())
        # controls: a planted read reports its own genome's user bin; random reads report nothing
        origin = info["origins"][b]
        per = np.diff(res.read_off.astype(np.int64))
        is_planted = origin >= 0
        own = planted_ub[np.where(is_planted, origin, 0)]
        rid = np.repeat(np.arange(n), per)
        got_own = np.zeros(n, dtype=bool)
        got_own[rid[res.user_bin == own[rid]]] = True
        hit_own += int((got_own & is_planted).sum())
        n_planted += int(is_planted.sum())
        n_random += int((~is_planted).sum())
        random_hit += int((per[~is_planted] > 0).sum())
        for (cb, lo, hi) in checks:
            if cb == b and (cb, lo) not in samples:
                a, e = int(res.read_off[lo]), int(res.read_off[hi])
                samples[(cb, lo)] = (bases[int(offs[lo]):int(offs[hi])].copy(), offs[lo:hi + 1] - offs[lo], res.n_hashes[lo:hi].copy(),
                                     res.read_off[lo:hi + 1] - res.read_off[lo], res.user_bin[a:e].copy(), res.count[a:e].copy())
    # the `layouts` leg strand_mixed of bench.py: planted reads from either strand (a reverse-strand read stops at the root)
    mb, mo, m_origin = synth.synth_reads(info["genomes"], info["genome_off"], S, read_len, error_rate=args.read_error, frac_random=0.1,
                                         seed=synth.DEFAULT_SEED + 55000, threads=os.cpu_count() or 8, frac_reverse=0.5)
    mres = sr.search_batch(mb, mo)
    samples[("strand_mixed", 0)] = (mb, mo, mres.n_hashes.copy(), mres.read_off.copy(), mres.user_bin.copy(), mres.count.copy())
    sr.close()
    dense.close()
    assert total_reads == n_batches * reads_per_batch
    assert digests == digests_dense, "pruning / sub-batch partition changed results"
    assert len(set(digests)) == n_batches                                       # ten DISTINCT batches
    assert hit_own / n_planted > 0.8, hit_own / n_planted
    assert random_hit / max(1, n_random) < 0.01
    assert total_tuples / total_reads > 2.0                                     # related strains: several tuples per read
    batches.clear()

    # the oracle, bit for bit, on the samples (host copy of the IXFs the traversal can enter)
    needed = {0} | {i for i, f in enumerate(lay["ixfs"]) if f.get("key_sets") or f["columns"]}
    host = []
    for i, f in enumerate(lay["ixfs"]):
        nbytes = 3 * f["seg_len"] * f["stride"]
        data = idx.download_ixf(i) if i in needed else np.empty(nbytes, dtype=np.uint8)    # untouched virtual memory
        host.append(dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], data=data))
    h = orc.Hixf(host, [f["next_ixf"] for f in lay["ixfs"]], [f["fname_idx"] for f in lay["ixfs"]])
    assert len(samples) == len(set((c[0], c[1]) for c in checks)) + 1
    import time
    t0 = time.time()
    sched = "reference" if orc.ref_lib() is not None else "openmp"          # the reference's own do_parallel where oracle/_ref is present
    n_checked = t_checked = 0
    for key, (sb, so, g_nh, g_off, g_ub, g_cnt) in samples.items():
        nh, off, ub, cnt, _ = h.search_batch(sb, so, err=args.error_rate, threads=min(32, os.cpu_count() or 8), scheduler=sched)
        assert np.array_equal(g_nh, nh), key
        assert np.array_equal(g_off, off), key
        assert np.array_equal(g_ub, ub) and np.array_equal(g_cnt, cnt), key
        assert ub.size > 0
        n_checked += nh.size
        t_checked += ub.size
    print(f"\n{workload}: oracle ({sched} scheduler) agrees on {n_checked} reads / {t_checked} tuples out of {n_batches} batches + the strand-mixed sample, "
          f"{time.time() - t0:.1f} s")
    assert n_checked >= 50000 + S or n_batches < 10
    idx.close()


@pytest.mark.parametrize("label,root_bins,child_bins", [("chopper_1024", 1024, 1024), ("root_4096", 4096, 128)])
def test_gtdb_class_layout_leg_shapes_against_the_oracle(label, root_bins, child_bins):
    """bench.py's `layouts` legs measure two more index shapes at the 113-GB footprint -- children as wide as the root (the reference's
    layout step applies one t_max at every level) and a 4096-bin root; here 5 000 reads each are compared with the oracle tuple by
    tuple (the third leg, strand-mixed reads, is a sample of the two class tests above)"""
    import torch

    import bench

    free_hbm, _ = torch.cuda.mem_get_info(0)
    if free_hbm < 160e9:
        pytest.skip(f"needs 160 GB of free HBM, have {free_hbm/1e9:.0f}")
    args = bench.parse_args(["--workload", "gtdb", "--reads", "5000", "--batches", "1", "--root-bins", str(root_bins), "--child-bins", str(child_bins)])
    wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
    assert 95e9 < idx.data_bytes < 125e9 and wl["root_bins"] == root_bins and wl["child_bins"] == child_bins
    bases, offs = batches[0]
    sr = Searcher(idx, error_rate=args.error_rate)
    res = sr.search_batch(bases, offs)
    sr.close()
    needed = {0} | {i for i, f in enumerate(lay["ixfs"]) if f.get("key_sets") or f["columns"]}
    host = []
    for i, f in enumerate(lay["ixfs"]):
        nbytes = 3 * f["seg_len"] * f["stride"]
        data = idx.download_ixf(i) if i in needed else np.empty(nbytes, dtype=np.uint8)    # untouched virtual memory
        host.append(dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], data=data))
    idx.close()
    h = orc.Hixf(host, [f["next_ixf"] for f in lay["ixfs"]], [f["fname_idx"] for f in lay["ixfs"]])
    nh, off, ub, cnt, _ = h.search_batch(bases, offs, err=args.error_rate, threads=min(32, os.cpu_count() or 8))
    assert np.array_equal(res.n_hashes, nh) and np.array_equal(res.read_off, off)
    assert np.array_equal(res.user_bin, ub) and np.array_equal(res.count, cnt)
    assert ub.size > 5000                                                       # related strains: several tuples per read
    print(f"\n{label}: oracle agrees on {nh.size} reads / {ub.size} tuples ({idx.n_ixf if False else len(host)} IXFs, {sum(1 for i in needed)} of them downloaded)")


def test_refseq_class_ten_million_reads():
    """BASELINE.json configs[2]: RefSeq-ABFV-class k22/s12 index (9.9 GB, README.md:52) resident in HBM, 10 M x 10 kb"""
    _full_size_class("refseq", 9.0e9, 11e9)


def test_gtdb_class_ten_million_reads():
    """BASELINE.json configs[3]: GTDB-220-class k22/s12 index (113 GB, README.md:51) resident in one GPU's HBM, 10 M x 10 kb"""
    _full_size_class("gtdb", 105e9, 120e9)


def test_refseq_class_hixf_file_through_the_cli(tmp_path):
    """BASELINE configs[2] through the whole drop-in chain: the RefSeq-class index written as an 11 GB `.hixf` with the
    library's writer, loaded and searched by the C++ `taxor search` CLI (parallel FASTQ parsing, streamed batches), its
    TSV byte-identical to the library formatter over the Python searcher's tuples (profiles/cli_e2e_class.py)"""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.disk_usage(str(tmp_path)).free < 20e9:
        pytest.skip("needs 20 GB of scratch space")
    cp = subprocess.run([sys.executable, os.path.join(root, "profiles", "cli_e2e_class.py"), "refseq", "100000"], capture_output=True, text=True,
                        timeout=900, env=dict(os.environ, TAXOR_E2E_TMP=str(tmp_path)))
    assert cp.returncode == 0, cp.stdout[-2000:] + cp.stderr[-2000:]
    assert "identical to formatter(searcher results): True" in cp.stdout
    assert cp.stdout.count(" rc 0 ") == 3


def test_cli_search_phase_keeps_up_with_the_library(tmp_path):
    """the drop-in CLI must not be slower than the library it wraps.  RefSeq-class `.hixf` (11 GB) and 1.3 M x
    10 kb reads as FASTA (round 4: 4.2 M reads, 42 GB) in tmpfs: once the index is resident, the CLI's search phase (parse -> GPU batches made of
    parsed chunks -> TSV text -> file) keeps up with the library's own host-fed `sustained` rate on the same reads, and after
    the last line is written the command is done within 0.3 s (no host mapping of the index to tear down).  The first run
    reads tmpfs pages that were written a moment ago (every page is promoted on the LRU under 32 readers) and is not the one
    judged.  (FASTA, because the library alone sustains ~38 Gbp/s on this small index: as FASTQ that is 76 GB/s of file to
    parse, beyond the parsers; the GTDB-class FASTQ run is profiles/r03/cli_e2e_gtdb.txt.)"""
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scratch = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 120e9 else str(tmp_path)
    if shutil.disk_usage(scratch).free < 120e9:
        pytest.skip("needs 120 GB of scratch space")
    # 4 M reads (42 GB of FASTA): the search phase is then ~1 s.  Round 3 ran 1.3 M reads, a phase of 0.4 s of which the first GPU
    # batch -- which nothing can overlap -- is a fifth; the library figure it is held against is a steady-state rate over 2 M reads
    cp = subprocess.run([sys.executable, os.path.join(root, "profiles", "cli_e2e_class.py"), "refseq", "4194304"], capture_output=True, text=True,
                        timeout=1800, env=dict(os.environ, TAXOR_E2E_TMP=scratch, TAXOR_E2E_RUNS="32,32,32,16", TAXOR_E2E_FORMAT="fasta"))
    assert cp.returncode == 0, cp.stdout[-3000:] + cp.stderr[-2000:]
    assert "identical to formatter(searcher results): True" in cp.stdout
    rates = [(float(m.group(1)), float(m.group(2)), float(m.group(3)))
             for m in re.finditer(r"RATE .*?search phase ([0-9.]+) Mbp/s = ([0-9.]+) x library sustained; teardown .*? ([0-9.]+) s;", cp.stdout)]
    assert len(rates) == 4, cp.stdout[-3000:]
    print(cp.stdout[-2500:])
    # The review's bar is >= 20 Gbp/s and >= 0.8 x the library's sustained rate on the MEDIAN run.  Measured with the
    # round's last pipeline (profiles/r04/cli_nopin_10kb.txt, cli_pin_10kb.txt): 36.5-38.8 Gbp/s = 0.88-0.93 x on every judged run (32
    # and 16 threads), the library alone at 41.5 Gbp/s.  Round 3 on 1.3 M reads: 0.58-0.80 and `max` over the runs; this round, step by
    # step and side by side on one box each time: 0.72-0.79 (start) -> 0.81-0.85 (sequence buffers of their own kind instead of
    # std::strings: 2-MiB boundaries, huge pages, streaming stores; bounded batch collecting) -> 0.88-0.93 (chunk buffers are not
    # page-locked any more: every registration made the GPU workers' submissions wait in the driver).  Asserted: the MEDIAN of the
    # judged runs reaches the review's bar.
    judged = sorted(rates[1:])
    med_rate = sorted(v for v, _, _ in judged)[len(judged) // 2]
    med_ratio = sorted(r for _, r, _ in judged)[len(judged) // 2]
    assert med_rate >= 30000.0, rates
    assert med_ratio >= 0.8, rates
    assert max(t for _, _, t in rates) < 0.3, rates


def test_gtdb_class_index_with_every_bin_a_real_filter_against_the_oracle():
    """The headline layout at its full 113 GB once more, this time with EVERY bin a real XOR filter constructed on the GPU
    (synth.exact_fill_index -> taxor_gpu_index_build_hixf_gen: planted bins from their genomes' hashes, 130 000 decoy leaf bins from
    generated keys -- 7e10 insertions, no key memory --, merged bins from the union of their child), read back in full and searched
    by the oracle: 5 000 reads, tuple for tuple; planted reads report their genome's user bin."""
    import psutil
    import torch

    import bench

    free_hbm, _ = torch.cuda.mem_get_info(0)
    if free_hbm < 160e9 or psutil.virtual_memory().available < 200e9:
        pytest.skip("needs 160 GB of free HBM and 200 GB of host memory")
    args = bench.parse_args(["--workload", "gtdb", "--reads", "5000", "--batches", "1"])
    wl, idx0, lay, batches, info = bench.build_workload(args, 0, 0, 1)
    idx0.close()                                                         # (only the layout and the reads are needed)
    idx, st = synth.exact_fill_index(lay, fill_frac=0.95)
    assert 105e9 < idx.data_bytes < 120e9 and st["keys_inserted"] > 5e10
    print(f"\nexact-fill GTDB-class index: {idx.data_bytes / 1e9:.1f} GB, {st['keys_inserted'] / 1e9:.1f} G insertions in {st['seconds_total']:.1f} s "
          f"({st['keys_inserted'] / st['seconds_total'] / 1e9:.2f} G/s), {st['chunks']} chunks, {st['reseeds']} IXFs redone")
    bases, offs = batches[0]
    sr = Searcher(idx, error_rate=args.error_rate)
    res = sr.search_batch(bases, offs)
    sr.close()
    host = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=idx.ixf_seed(i), data=idx.download_ixf(i)) for i, f in enumerate(lay["ixfs"])]
    idx.close()
    h = orc.Hixf(host, [f["next_ixf"] for f in lay["ixfs"]], [f["fname_idx"] for f in lay["ixfs"]])
    nh, off, ub, cnt, _ = h.search_batch(bases, offs, err=args.error_rate, threads=min(32, os.cpu_count() or 8))
    assert np.array_equal(res.n_hashes, nh) and np.array_equal(res.read_off, off)
    assert np.array_equal(res.user_bin, ub) and np.array_equal(res.count, cnt)
    origin = info["origins"][0]
    planted_ub = np.array([u if u is not None else -1 for u in lay["planted_user_bin"]], dtype=np.int64)
    per = np.diff(res.read_off.astype(np.int64))
    own = sum(int(planted_ub[origin[r]]) in res.user_bin[int(res.read_off[r]):int(res.read_off[r + 1])].tolist() for r in range(5000) if origin[r] >= 0)
    assert own > 0.8 * int((origin >= 0).sum()) and (per[origin < 0] > 0).mean() < 0.01
    print(f"oracle agrees on {nh.size} reads / {ub.size} tuples of the index in which every bin is a real filter")
