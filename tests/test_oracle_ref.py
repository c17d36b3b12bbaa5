"""The oracle AND the product's host functions against the REAL reference wherever the reference compiles on its own:
oracle/_ref/libtaxor_ref.so is built by `make -C oracle ref` straight from /root/reference (syncmer threshold table,
k-mer / FracMinHash threshold models, adjust_seed, the in-repo XOR-filter prototype, the chunk loop's scheduler
do_parallel.hpp and the result stream sync_out.hpp; see oracle/ref_driver.cpp).
The library travels to the GPU box prebuilt; where neither it nor /root/reference exists these tests skip."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import _lib
from taxor_amd import search as ts

REF = orc.ref_lib()
pytestmark = pytest.mark.skipif(REF is None, reason="oracle/_ref/libtaxor_ref.so not built and /root/reference absent")


def test_syncmer_match_ratio_is_the_references_table():
    """get_min_syncmer_match_ratio (syncmer_model.hpp:38-50): every (k, error rate) the reference accepts"""
    for k in range(12, 31, 2):
        for e100 in range(0, 2001):
            err = e100 / 10000.0
            want = REF.ref_syncmer_match_ratio(k, err)
            assert orc.syncmer_match_ratio(k, err) == want, (k, err)
            assert ts.threshold_ratio(k, err) == want, (k, err)
    for err in (0.15, 0.05, 0.04, 0.1, 0.2, 0.0, 0.07, 0.29 / 2):         # fp-rounding-sensitive row indices
        assert orc.syncmer_match_ratio(22, err) == REF.ref_syncmer_match_ratio(22, err)
    for n in (0, 1, 98, 435, 871, 10**6):
        assert orc.threshold(n, 22, 0.04) == int(n * REF.ref_syncmer_match_ratio(22, 0.04))


def test_threshold_model_components_are_the_references():
    """calculate_nmut_kmer_CI(...).second and calculate_containment_index_CI(...).first from the reference's own
    translation units, incl. the NaN / negative casts of very short reads (compiled as the reference compiles them)"""
    rng = np.random.default_rng(3)
    ns = [0, 1, 2, 3, 5, 7, 10, 20, 50, 99, 100, 435, 871, 4981, 9979, 99979, 10**6] + [int(x) for x in rng.integers(0, 300000, 400)]
    for k in (16, 20, 22, 31, 32):
        for err in (0.001, 0.01, 0.04, 0.1, 0.2, 0.5):
            for n in ns:
                high = REF.ref_nmut_kmer_ci_high(err, k, n, 0.95)
                assert orc.lib().orc_nmut_kmer_ci_high(err, k, n, 0.95) == high, (k, err, n)
                fp = int(n * 0.0039)
                want_kmer = (n - high - fp) % 2**64                                           # threshold.hpp:62-66
                assert orc.threshold_model(orc.THR_KMER, n, k, err) == want_kmer
                assert ts.threshold_model(_lib.THR_KMER, n, k, err) == want_kmer
                for sf in (1e-3, 0.05, 1 / 7, 0.5, 0.999):
                    low = REF.ref_containment_index_ci_low(err, k, n, sf, 0.95)
                    mine = orc.lib().orc_containment_index_ci_low(err, k, n, sf, 0.95)
                    assert (low == mine) or (low != low and mine != mine), (k, err, n, sf)     # NaN == NaN
    assert orc.lib().orc_normal_cdf_inverse(0.975) == REF.ref_normal_cdf_inverse(0.975)
    for k in range(1, 33):
        assert orc.adjust_seed(k) == REF.ref_adjust_seed(k)


def test_ixf_arithmetic_is_the_in_repo_prototypes():
    """src/main/xorfilter.hpp (the evidence for the un-vendored IXF): same rows, same fingerprint, same sizing; a
    filter the prototype builds answers identically through the oracle's lookup and through the product's builder"""
    rng = np.random.default_rng(5)
    for n in (1, 2, 10, 1000, 50000):
        keys = np.unique(rng.integers(0, 2**63, size=n, dtype=np.uint64))
        seed, blk, arr = C.c_uint64(), C.c_uint64(), C.c_uint64()
        h = REF.ref_xor_build(keys.ctypes.data, keys.size, C.byref(seed), C.byref(blk), C.byref(arr))
        assert h, "prototype construction failed"
        assert orc.ixf_seg_len(keys.size) == blk.value == ts_seg_len(keys.size)           # (32 + 1.23 n) / 3
        fps = np.ctypeslib.as_array(REF.ref_xor_fingerprints(h), shape=(int(arr.value),)).copy()
        data = np.zeros(3 * blk.value, dtype=np.uint8)
        data[:] = fps[:3 * blk.value]
        ixf = dict(bins=1, stride=1, seg_len=int(blk.value), seed=int(seed.value), data=data)
        hx = orc.Hixf([ixf], [np.zeros(1, np.int64)], [np.zeros(1, np.int64)])
        rows, fp = (C.c_uint64 * 3)(), C.c_uint8()
        mine_rows, mine_fp = np.zeros(3, np.uint64), np.zeros(1, np.uint8)
        probe = list(keys[:200]) + [int(x) for x in rng.integers(0, 2**64, size=200, dtype=np.uint64)]
        for key in probe:
            REF.ref_xor_probe(h, int(key), rows, C.byref(fp))
            orc.lib().orc_ixf_probe(C.byref(hx._ixf[0]), C.c_uint64(int(key)), mine_rows.ctypes.data, mine_fp.ctypes.data)
            assert list(rows) == mine_rows.tolist() and fp.value == int(mine_fp[0]), key
        # membership through the oracle's bulk_count == the prototype's Contain
        assert int(hx.ixf_bulk_count(0, keys)[0]) == keys.size
        others = rng.integers(0, 2**64, size=20000, dtype=np.uint64)
        want = sum(REF.ref_xor_contain(h, int(x)) for x in others[:3000])
        assert int(hx.ixf_bulk_count(0, others[:3000])[0]) == want
        # the product's own host builder under the prototype's seed produces a filter the prototype's lookup rule accepts
        col = np.zeros(3 * blk.value, dtype=np.uint8)
        rc = _lib.lib().taxor_ixf_build_bin(keys.ctypes.data, keys.size, int(seed.value), int(blk.value), col.ctypes.data)
        if rc == 0:
            ixf2 = dict(ixf, data=col)
            hx2 = orc.Hixf([ixf2], [np.zeros(1, np.int64)], [np.zeros(1, np.int64)])
            assert int(hx2.ixf_bulk_count(0, keys)[0]) == keys.size
        REF.ref_xor_free(h)


def ts_seg_len(n):
    return int(_lib.lib().taxor_ixf_seg_len(int(n)))


def test_do_parallel_slices_are_the_references():
    """hixf::do_parallel (do_parallel.hpp:22-29): `threads` tasks, floor(n / threads) records each, the remainder on the LAST task
    -- observed from the reference's own object code -- and the oracle's worker driven by it over 1024-record chunks
    (taxor_search.cpp:315-326) gives the same CSR as the oracle's OpenMP slices."""
    for n, th in ((1024, 32), (1000, 32), (31, 32), (5, 3), (1, 1), (0, 4), (1024, 7)):
        out = np.zeros(2 * th, dtype=np.uint64)
        REF.ref_do_parallel_slices(n, th, out.ctypes.data_as(C.c_void_p))
        got = sorted((int(out[2 * i]), int(out[2 * i + 1])) for i in range(th))
        per = n // th
        want = sorted((per * i, n if i == th - 1 else per * (i + 1)) for i in range(th))
        assert got == want, (n, th)
    from taxor_amd import synth
    rng = np.random.default_rng(11)
    g, go = synth.random_genomes(5, 8000, seed=11)
    planted = [np.unique(orc.seq_to_syncmers(bytes(g[int(go[i]):int(go[i + 1])]))) for i in range(5)]
    lay = synth.make_layout(planted, root_bins=64, child_bins=32, n_children=3, seed=11)
    host = synth.materialize_host(lay)
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, _ = synth.synth_reads(g, go, 2500, 900, error_rate=0.02, frac_random=0.2, seed=12)
    want = h.search_batch(bases, offs, threads=4)
    for th, chunk in ((32, 1024), (3, 1024), (5, 700), (1, 1024)):
        tm = {}
        got = h.search_batch(bases, offs, threads=th, scheduler="reference", chunk=chunk, timing=tm)
        for a, b in zip(got[:4], want[:4]):
            assert np.array_equal(a, b), (th, chunk)
        assert got[4] == want[4] and tm["compute_time"] > 0
    assert want[2].size > 1000


def test_sync_out_serialises_whole_writes(tmp_path):
    """hixf::sync_out (sync_out.hpp:24-36): every operator<< is one mutexed write -- lines of concurrent writers never interleave"""
    p = tmp_path / "out.txt"
    REF.ref_sync_out_lines(str(p).encode(), 8, 500)
    lines = open(p).read().split("\n")
    assert lines[-1] == "" and len(lines) == 8 * 500 + 1
    assert sorted(lines[:-1]) == sorted(f"t{t}:{i}" for t in range(8) for i in range(500))
    for t in range(8):                        # each writer's own lines stay in its order
        assert [l for l in lines if l.startswith(f"t{t}:")] == [f"t{t}:{i}" for i in range(500)]
