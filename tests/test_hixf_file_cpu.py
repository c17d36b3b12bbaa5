"""`.hixf` envelope writer/reader round trip, loud failure on damaged files, and the per-read output text
(taxor_search.cpp:268-305) against a line-by-line Python restatement kept in this test."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import synth
from taxor_amd._lib import TaxorError
from taxor_amd.hixf_file import HixfFile, store_hixf

HEADER = "#QUERY_NAME\tACCESSION\tREFERENCE_NAME\tTAXID\tREF_LEN\tQUERY_LEN\tQHASH_COUNT\tQHASH_MATCH\tTAX_STR\tTAX_ID_STR\n"


def make_species(lay):
    sp = []
    for ub in range(lay["n_user_bins"]):
        sp.append(dict(organism_name=f"Organism {ub}", accession_id=f"GCF_{ub:09d}.1", taxid=str(1000 + ub),
                       taxnames_string=f"k__Bacteria;p__P{ub % 7};s__Organism {ub}", taxid_string=f"2;{ub % 7};{1000 + ub}",
                       user_bin=ub, seq_len=1000000 + ub))
    return sp


def expected_lines(species, read_id, read_len, n_hashes, tuples):
    """taxor_search.cpp:268-305 restated: '-' line, or the 0.8*max filter and ten tab-separated columns"""
    ubi = {}
    for i, s in enumerate(species):
        ubi.setdefault(s["user_bin"], i)
    if not tuples:
        return f"{read_id}\t-\t-\t-\t-\t{read_len}\n"
    mx = max(c for _, c in tuples)
    out = ""
    for ub, c in tuples:
        if float(c) < float(mx) * 0.8:
            continue
        s = species[ubi.get(ub, 0)]
        out += "\t".join([read_id, s["accession_id"], s["organism_name"], s["taxid"], str(s["seq_len"]), str(read_len),
                          str(n_hashes), str(c), s["taxnames_string"], s["taxid_string"]]) + "\n"
    return out


def small_layout(seed=4):
    rng = np.random.default_rng(seed)
    planted = [np.unique(rng.integers(0, 2**63, size=300, dtype=np.uint64)) for _ in range(6)]
    lay = synth.make_layout(planted, root_bins=66, child_bins=24, n_children=2, seed=seed)
    return lay, synth.materialize_host(lay), planted


def test_store_load_roundtrip(tmp_path):
    lay, host, planted = small_layout()
    sp = make_species(lay)
    p = tmp_path / "toy.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp)
    h = HixfFile(p)
    assert (h.k, h.s, h.t, h.use_syncmer, h.scaling, h.window_size) == (22, 12, 5, True, 1, 22)
    assert h.n_user_bins == lay["n_user_bins"] and len(h.ixfs) == len(host)
    for a, b in zip(h.ixfs, host):
        for key in ("bins", "stride", "seg_len", "seed"):
            assert a[key] == b[key]
        assert np.array_equal(a["data"], b["data"])
        assert np.array_equal(a["next_ixf"], b["next_ixf"]) and np.array_equal(a["fname_idx"], b["fname_idx"])
    assert h.species == sp
    assert h.filenames == [f"user_bin_{i}.fna" for i in range(lay["n_user_bins"])]
    # the loaded arrays drive the oracle exactly like the originals
    ho = orc.Hixf(h.ixfs, [f["next_ixf"] for f in h.ixfs], [f["fname_idx"] for f in h.ixfs])
    ub, cnt, _ = ho.bulk_contains(planted[1], planted[1].size)
    assert lay["planted_user_bin"][1] in ub.tolist()
    del ho
    h.close()


def test_envelope_bytes_follow_index_hpp(tmp_path):
    """first bytes of the file: u32 version=1, u64 window, shape(u64 size, u64 bits), k, s, t, parts, use_syncmer,
    u16 scaling, compressed (src/main/index.hpp:211-225)"""
    lay, host, _ = small_layout()
    p = tmp_path / "toy.hixf"
    store_hixf(p, host, lay["n_user_bins"], make_species(lay))
    raw = open(p, "rb").read(64)
    assert int.from_bytes(raw[0:4], "little") == 1
    assert int.from_bytes(raw[4:12], "little") == 22
    assert int.from_bytes(raw[12:20], "little") == 22 and int.from_bytes(raw[20:28], "little") == (1 << 22) - 1
    assert list(raw[28:33]) == [22, 12, 5, 1, 1]
    assert int.from_bytes(raw[33:35], "little") == 1 and raw[35] == 0
    assert int.from_bytes(raw[36:44], "little") == lay["n_user_bins"]     # bin_path outer size


def test_damaged_files_fail_loudly(tmp_path):
    lay, host, _ = small_layout()
    p = tmp_path / "toy.hixf"
    store_hixf(p, host, lay["n_user_bins"], make_species(lay))
    raw = open(p, "rb").read()
    for name, blob in [("truncated", raw[: len(raw) // 2]), ("trailing", raw + b"\0" * 8),
                       ("version", b"\x02\0\0\0" + raw[4:]), ("tiny", raw[:16])]:
        q = tmp_path / f"{name}.hixf"
        open(q, "wb").write(blob)
        with pytest.raises(TaxorError) as e:
            HixfFile(q)
        assert e.value.code == -5, name
    with pytest.raises(TaxorError):
        HixfFile(tmp_path / "does_not_exist.hixf")


def test_format_read_matches_reference_text(tmp_path):
    lay, host, _ = small_layout()
    sp = make_species(lay)
    sp[3]["user_bin"] = 9999          # user bin 3 has no species entry -> the reference falls back to species[0]
    p = tmp_path / "toy.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp)
    h = HixfFile(p)
    cases = [("read_0 runid=abc", 5000, 435, []),
             ("r1", 1139, 98, [(2, 56)]),
             ("r2", 2589, 224, [(1, 104), (5, 104), (4, 83), (0, 84)]),      # 83 < 0.8*104 = 83.2 dropped, 84 kept
             ("r3", 30, 0, [(0, 0), (1, 0), (2, 0)]),                        # zero-hash quirk: everything, count 0
             ("r4", 4000, 300, [(3, 200)])]
    for rid, rl, nh, tup in cases:
        got = h.format_read(rid, rl, nh, [u for u, _ in tup], [c for _, c in tup])
        assert got == expected_lines(sp, rid, rl, nh, tup), rid
    # the chunk formatter (what the CLI's formatter threads call) writes the same text for all of them at once
    off = np.cumsum([0] + [len(t) for _, _, _, t in cases])
    text = h.format_reads([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], off,
                          [u for c in cases for u, _ in c[3]], [k for c in cases for _, k in c[3]])
    assert text == "".join(expected_lines(sp, rid, rl, nh, tup) for rid, rl, nh, tup in cases)
    assert h.format_reads([], [], [], [0], [], []) == ""
    h.close()


def test_probe_recovers_foreign_ixf_schemas(tmp_path):
    """The IXF record layout of the seqan3 fork is un-vendored; `taxor_hixf_probe` must find framing and fields of
    files written with other member orders (scalars before/after the vector, fields not stored at all), and
    `taxor_hixf_load` must fall back to it on its own."""
    from taxor_amd.hixf_file import default_schema, make_schema, probe_hixf
    lay, host, planted = small_layout(9)
    sp = make_species(lay)
    cases = {
        "default": None,
        # seqan3-IBF-like member order: bins, technical_bins, bin_size(=rows), hash_shift-ish filler, bin_words, seed
        "ibf_like_rows": make_schema(7, 0, 0, 1, 2, 6, seg_len_is_rows=1),
        # seed first, data in the middle, stride after the vector
        "seed_first_stride_after": make_schema(3, 2, 1, 3, 2, 0),
        # nothing but the seed and the vector: bins / stride / seg_len derived
        "minimal": make_schema(1, 0, -1, -1, -1, 0),
        # no seed stored at all -> default start seed, flagged in the report
        "no_seed": make_schema(2, 0, 0, 1, -1, -1),
    }
    dflt = default_schema()
    assert (dflt.n_before, dflt.n_after, dflt.idx_bins, dflt.idx_stride, dflt.idx_seg_len, dflt.idx_seed) == (6, 0, 0, 1, 2, 4)
    for name, sc in cases.items():
        p = tmp_path / f"{name}.hixf"
        store_hixf(p, host, lay["n_user_bins"], sp, schema=sc)
        got, report = probe_hixf(p)
        want = sc if sc is not None else dflt
        assert (got.n_before, got.n_after) == (want.n_before, want.n_after), (name, report)
        assert "framing:" in report and "seed:" in report
        if name == "no_seed":
            assert got.idx_seed == -1 and "VERIFY" in report
        h = HixfFile(p)                      # default schema first, probe fallback when the records do not fit
        for a, b in zip(h.ixfs, host):
            for key in ("bins", "stride", "seg_len"):
                assert a[key] == b[key], (name, key)
            if name != "no_seed":
                assert a["seed"] == b["seed"], name
            assert np.array_equal(a["data"], b["data"]), name
            assert np.array_equal(a["next_ixf"], b["next_ixf"]) and np.array_equal(a["fname_idx"], b["fname_idx"])
        assert h.species == sp
        # a file whose records needed the probed layout is flagged: `taxor search` warns that it was written by other
        # software and that `taxor verify` should run first
        assert h.foreign_schema == (sc is not None and (sc.n_before, sc.n_after, sc.idx_bins, sc.idx_stride, sc.idx_seg_len, sc.idx_seed)
                                    != (dflt.n_before, dflt.n_after, dflt.idx_bins, dflt.idx_stride, dflt.idx_seg_len, dflt.idx_seed)), name
        h.close()
        h2 = HixfFile(p, schema=got)
        assert len(h2.ixfs) == len(host)
        h2.close()
    # a file that is not a .hixf at all is still an error, not a guess
    junk = tmp_path / "junk.hixf"
    open(junk, "wb").write(b"\x01\0\0\0" + bytes(range(256)) * 8)
    with pytest.raises(TaxorError):
        probe_hixf(junk)


def test_loader_survives_random_corruption(tmp_path):
    """fuzz: flipped bytes, truncations and spliced garbage must produce a clean error or a consistent index --
    never a crash or an out-of-bounds view (the reference swallows read errors, index.hpp:235-238)"""
    lay, host, _ = small_layout(11)
    sp = make_species(lay)[:20]
    p = tmp_path / "base.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp)
    raw = bytearray(open(p, "rb").read())
    data_start = raw.find(bytes(host[0]["data"][:64]))          # fingerprints may be corrupted freely: not validated
    assert data_start > 0
    rng = np.random.default_rng(5)
    meta_positions = list(range(0, data_start)) + list(range(len(raw) - 4000, len(raw)))
    outcomes = {"ok": 0, "err": 0}
    for trial in range(250):
        blob = bytearray(raw)
        kind = trial % 4
        if kind == 0:      # flip a few metadata bytes
            for _ in range(int(rng.integers(1, 4))):
                blob[int(rng.choice(meta_positions))] ^= int(rng.integers(1, 256))
        elif kind == 1:    # overwrite an aligned u64 in the metadata with a huge / tiny value
            pos = int(rng.choice(meta_positions)) & ~7
            blob[pos:pos + 8] = [0, 1, 2**63, 2**64 - 1, 2**40][int(rng.integers(0, 5))].to_bytes(8, "little")
        elif kind == 2:    # truncate
            blob = blob[: int(rng.integers(1, len(blob)))]
        else:              # splice garbage
            pos = int(rng.integers(0, len(blob)))
            blob[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8))
        q = tmp_path / "fuzz.hixf"
        open(q, "wb").write(bytes(blob))
        try:
            h = HixfFile(q)
        except TaxorError as e:
            assert e.code in (-5, -1)
            outcomes["err"] += 1
            continue
        # accepted: every view must be internally consistent
        for f in h.ixfs:
            assert f["data"].size == 3 * f["seg_len"] * f["stride"] and f["stride"] % 64 == 0 and f["stride"] >= f["bins"]
            assert f["next_ixf"].size == f["bins"] and f["fname_idx"].size == f["bins"]
        h.close()
        outcomes["ok"] += 1
    assert outcomes["err"] > 100          # most corruptions of the metadata are detected


def parse_search_results_like_taxor_profile(text):
    """src/main/taxor_profile.cpp:93-163 (parse_search_results) restated: the header line is skipped; fields are split
    at tabs; the read id is cut at its first blank; a '-' in column 1 is a miss whose read length sits in column 5; a hit
    line is read at the indices 1, 3, 4, 5, 6, 7 (accession, taxid, ref_len, query_len, hash count, hash match) and 9 / 8
    (tax id path / tax name path); a miss is not added to a read that already has a reference assignment."""
    results, taxpath = {}, {}
    for n, line in enumerate(text.split("\n")):
        if n == 0 or not line:
            continue
        f = line.split("\t")                                        # :111-116
        read_id = f[0].split(" ")[0] if " " in f[0] else f[0]       # :120-123
        if f[1] == "-":                                             # :126-130
            res = dict(accession_id="-", query_len=int(f[5]))
        else:                                                       # :131-145
            res = dict(accession_id=f[1], tax_id=f[3], ref_len=int(f[4]), query_len=int(f[5]),
                       query_hash_count=int(f[6]), query_hash_match=int(f[7]))
            taxpath.setdefault(f[1], (f[9], f[8]))
        lst = results.setdefault(read_id, [])                       # :147-150
        if lst and res["accession_id"] == "-":                      # :153-156
            continue
        lst.append(res)
    return results, taxpath


def test_tsv_is_what_taxor_profile_parses(tmp_path):
    """the search TSV is `taxor profile`'s input.  The text written by the library's formatter must
    parse with the column indices taxor_profile.cpp uses -- a 6-column miss line with the read length at index 5, a
    10-column hit line -- and carry the values the searcher produced."""
    lay, host, _ = small_layout()
    sp = make_species(lay)
    p = tmp_path / "toy.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp)
    h = HixfFile(p)
    text = HEADER
    want = {}
    cases = [("read_1 runid=7 ch=3", 5000, 430, [(2, 300), (5, 290), (1, 100)]),      # three tuples, one below 0.8*max
             ("read_2", 7123, 612, []),                                                  # miss
             ("read_3\twith_tab", 900, 77, [(0, 40)])]                                   # header text is copied verbatim
    for rid, rlen, nh, tup in cases:
        text += h.format_read(rid, rlen, nh, [a for a, _ in tup], [b for _, b in tup])
        want[rid.split(" ")[0]] = (rlen, nh, tup)
    h.close()
    header_cols = HEADER.rstrip("\n").split("\t")
    assert len(header_cols) == 10 and header_cols[5] == "QUERY_LEN" and header_cols[6] == "QHASH_COUNT" and header_cols[7] == "QHASH_MATCH"
    results, taxpath = parse_search_results_like_taxor_profile(text)
    r1 = results["read_1"]
    assert [x["accession_id"] for x in r1] == [sp[2]["accession_id"], sp[5]["accession_id"]]      # 100 < 0.8*300 dropped
    assert all(x["query_len"] == 5000 and x["query_hash_count"] == 430 for x in r1)
    assert [x["query_hash_match"] for x in r1] == [300, 290]
    assert r1[0]["ref_len"] == sp[2]["seq_len"] and r1[0]["tax_id"] == sp[2]["taxid"]
    assert taxpath[sp[2]["accession_id"]] == (sp[2]["taxid_string"], sp[2]["taxnames_string"])
    assert results["read_2"] == [dict(accession_id="-", query_len=7123)]
    # an id containing a tab shifts the columns in the reference's own output as well (it writes the id verbatim,
    # taxor_search.cpp:270,287): the formatter must not "repair" it
    assert "read_3\twith_tab\t" in text


def test_store_from_a_source_writes_the_same_file(tmp_path):
    """store_hixf(data_of=...): the fingerprint bytes come through a taxor_ixf_source, one IXF at a time (an index resident on a
    GPU is written without a host copy of all of it) -- byte-identical to the file written from host arrays; and a loaded
    file's own source (pread) delivers the same bytes as its mapping"""
    import ctypes as C
    import filecmp
    from taxor_amd import _lib
    lay, host, _ = small_layout()
    sp = make_species(lay)
    a, b = tmp_path / "a.hixf", tmp_path / "b.hixf"
    store_hixf(a, host, lay["n_user_bins"], sp)
    calls = []

    def data_of(i):
        calls.append(i)
        return host[i]["data"]

    store_hixf(b, [dict(f, data=None) for f in host], lay["n_user_bins"], sp, data_of=data_of)
    assert filecmp.cmp(a, b, shallow=False) and calls == list(range(len(host)))
    h = HixfFile(a)
    v = _lib.lib().taxor_hixf_get_view(h._h).contents
    assert v.source                                           # the loader offers a pread() reader of the file
    src = C.cast(v.source, C.POINTER(_lib.IxfSource)).contents
    for i, f in enumerate(h.ixfs):
        n = f["data"].size
        buf = np.empty(n, dtype=np.uint8)
        assert src.read(src.ctx, i, 0, n, buf.ctypes.data) == 0 and np.array_equal(buf, f["data"])
        if n > 100:
            assert src.read(src.ctx, i, 37, 50, buf.ctypes.data) == 0 and np.array_equal(buf[:50], f["data"][37:87])
    assert src.read(src.ctx, len(h.ixfs), 0, 1, buf.ctypes.data) != 0
    _lib.lib().taxor_hixf_release_data(h._h)                  # gives the mapping's data pages back; the source still reads
    assert src.read(src.ctx, 0, 0, 16, buf.ctypes.data) == 0 and np.array_equal(buf[:16], host[0]["data"][:16])
    h.close()
