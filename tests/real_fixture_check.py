"""Consumers of the fixtures `taxor pin` writes (tests/golden/real_<name>.json; taxor_amd/csrc/pin_cmd.h): everything in such a
file is re-derived here and held against what the REFERENCE wrote for the same read -- QHASH_COUNT (the number of distinct
open-syncmer hashes, taxor_search.cpp:261,298) and QHASH_MATCH (the tally of the reported user bin, :265,299) -- so that a
fixture pins the un-vendored boundaries (wyhash, the IXF arithmetic) of the CPU oracle and of the HIP path without the index
file itself, which is hundreds of megabytes and cannot be committed."""
import base64
import ctypes as C
import json

import numpy as np

from oracle import oracle as orc


def load(path):
    with open(path) as f:
        fx = json.load(f)
    assert fx["format"] == 1
    return fx


def _arr(b64, dtype):
    return np.frombuffer(base64.b64decode(b64), dtype=dtype)


def hashes_of(fx, rd):
    """the fixture's stored distinct hashes of one read, in first-insertion order"""
    return _arr(rd["hashes_u64_b64"], np.uint64)


def oracle_hashes(fx, seq):
    ix = fx["index"]
    hs = orc.seq_to_syncmers(orc.dna4_normalise(seq.encode()), ix["k"], ix["s"], ix["t"])
    if ix["scaling"] > 1:         # FracMinHash down-sampling, taxor_search.cpp:223-233
        hs = np.array([h for h in hs.tolist() if float(orc.wyhash(h)) <= float(2**64 - 1) / float(ix["scaling"])], dtype=np.uint64)
    return hs


def check_cpu(fx):
    """oracle vs fixture vs the reference's own numbers; returns (reads checked, probes checked)"""
    assert fx["pinned"] is True, "the fixture was written by a pin run whose output differed from the reference's"
    assert fx["summary"]["differing_or_missing"] == 0 and fx["summary"]["identical"] == fx["summary"]["reads_expected"]
    arith = int(fx["index"]["ixf_arith"])
    n_probes = 0
    for rd in fx["reads"]:
        hs = oracle_hashes(fx, rd["seq"])
        assert np.array_equal(hs, hashes_of(fx, rd)), rd["id"]
        assert rd["ours"] == rd["expect"], rd["id"]
        fields = [l.split("\t") for l in rd["expect"]]
        for f in fields:
            if len(f) >= 10 and f[1] != "-":
                assert int(f[6]) == hs.size, (rd["id"], "QHASH_COUNT")        # the reference's distinct-hash count
                assert int(f[5]) == len(rd["seq"]), (rd["id"], "QUERY_LEN")
            else:
                assert int(f[5]) == len(rd["seq"]), (rd["id"], "QUERY_LEN of a miss line")
        leaf_sum = {}
        for X in rd["ixfs"]:
            rows = _arr(X["rows_u32_b64"], np.uint32).reshape(-1, 3)
            fps = _arr(X["fingerprints_u8_b64"], np.uint8)
            assert rows.shape[0] == hs.size == fps.size
            ixf = orc._Ixf(X["bins"], X["stride"], X["seg_len"], X["seed"], None, arith)
            r3, fp1 = np.zeros(3, np.uint64), np.zeros(1, np.uint8)
            for i in range(hs.size):                                          # the oracle's reading of the IXF arithmetic
                orc.lib().orc_ixf_probe(C.byref(ixf), C.c_uint64(int(hs[i])), r3.ctypes.data_as(C.c_void_p), fp1.ctypes.data_as(C.c_void_p))
                assert (int(r3[0]), int(r3[1]), int(r3[2])) == tuple(int(x) for x in rows[i]) and int(fp1[0]) == int(fps[i]), (rd["id"], X["ixf"], i)
                assert all(int(x) < 3 * X["seg_len"] for x in rows[i])
            for bp in X["bins_probed"]:
                b = _arr(bp["bytes_u8_b64"], np.uint8).reshape(-1, 3)
                cnt = int(((b[:, 0] ^ b[:, 1] ^ b[:, 2]) == fps).sum())
                assert cnt == bp["count"], (rd["id"], X["ixf"], bp["bin"])
                n_probes += 1
                if not bp["merged"]:
                    leaf_sum[bp["expect_line"]] = leaf_sum.get(bp["expect_line"], 0) + cnt
        for li, s in leaf_sum.items():                                        # split bins summed: hixf.hpp:315,325-326
            assert s == int(fields[li][7]), (rd["id"], li, "QHASH_MATCH")     # the reference's tally of that user bin
    return len(fx["reads"]), n_probes


def check_gpu(fx):
    """the HIP path's hashes for the fixture's reads equal the stored ones (and with them the reference's QHASH_COUNT)"""
    from taxor_amd import GpuIndex, Searcher
    ix = fx["index"]
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins),
                           data=np.zeros(3 * 16 * 64, np.uint8))], bins, ix["k"], ix["s"], ix["t"], scaling=ix["scaling"])
    sr = Searcher(dummy, ratio=0.5)
    reads = [rd["seq"].encode() for rd in fx["reads"]]
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    hoff, hashes = sr.seq_to_syncmers(bases, offs)
    for i, rd in enumerate(fx["reads"]):
        assert np.array_equal(hashes[int(hoff[i]):int(hoff[i + 1])], hashes_of(fx, rd)), rd["id"]
    sr.close()
    dummy.close()
    return len(reads)
