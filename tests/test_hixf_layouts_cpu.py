"""Fingerprint layouts of a .hixf this library did not write (taxor_amd/csrc/ixf_layout.h): the serialiser of
seqan3::interleaved_xor_filter is un-vendored (hierarchical_interleaved_xor_filter.hpp:152-158), so the loader must do the
bookkeeping for every layout the variant scan can name -- row-interleaved at a foreign pitch, bin-major, bit-sliced 64-bin words,
rows in position-major order -- and index creation transposes them on the device (tests/test_gpu_relayout.py).  Here, without a
GPU: files written under each layout hold exactly the bytes an independent numpy restatement of the layout produces
(taxor_amd.search.to_source_layout), load back with the right strides / segment lengths / pitches, and a layout that contradicts
the array lengths is refused."""
import numpy as np
import pytest

from taxor_amd import _lib, synth
from taxor_amd._lib import TaxorError
from taxor_amd.hixf_file import HixfFile, default_schema, describe_layout, make_schema, parse_layout, probe_hixf, store_hixf
from taxor_amd.search import source_bytes, to_source_layout
from tests.test_hixf_file_cpu import make_species

R, B, S = _lib.LAYOUT_ROWS, _lib.LAYOUT_BIN_MAJOR, _lib.LAYOUT_BIT_SLICED
PM, UNP, STO = _lib.LAYOUT_POSITION_MAJOR, _lib.LAYOUT_PITCH_BINS, _lib.LAYOUT_PITCH_STORED
LAYOUTS = [R | UNP, R | PM, R | UNP | PM, B, B | UNP, B | PM, B | UNP | PM, B | STO, S, S | PM]


def odd_layout(seed=6):
    """bin counts that are not multiples of 64 (66 and 24) and odd segment lengths: padded and unpadded pitches differ"""
    rng = np.random.default_rng(seed)
    planted = [np.unique(rng.integers(0, 2**63, size=int(rng.integers(200, 400)), dtype=np.uint64)) for _ in range(6)]
    lay = synth.make_layout(planted, root_bins=66, child_bins=24, n_children=2, seed=seed)
    return lay, synth.materialize_host(lay), planted


def test_layout_spec_round_trip():
    for code in [0] + LAYOUTS:
        assert parse_layout(describe_layout(code)) == code
    assert parse_layout("bin-major") == B and parse_layout("bin-major,unpadded,position-major") == B | UNP | PM
    assert parse_layout("") == 0 and describe_layout(0) == "interleaved,padded,segment-major"
    for bad in ("column-major", "bit-sliced,unpadded", "bin-major,,sideways"):
        with pytest.raises(TaxorError):
            parse_layout(bad)


def test_numpy_restatement_follows_the_definition():
    """to_source_layout (vectorised numpy) against a byte-by-byte reading of ixf_layout.h on a tiny IXF -- so the two checkers of
    the device transposition (this and the C header's ixf_src_fingerprint, which the variant scan uses) are pinned to each other"""
    rng = np.random.default_rng(1)
    bins, stride, seg = 70, 128, 5
    rows = 3 * seg
    D = np.zeros((rows, stride), np.uint8)
    D[:, :bins] = rng.integers(0, 256, (rows, bins), dtype=np.uint8)
    f = dict(bins=bins, stride=stride, seg_len=seg, data=D.reshape(-1))
    for code in [0] + LAYOUTS:
        raw, pitch = to_source_layout(f, code)
        assert raw.size == source_bytes(code, dict(f, src_stride=pitch))
        groups = (bins + 63) // 64
        for r in range(rows):
            rs = (r % seg) * 3 + r // seg if code & PM else r
            for b in range(bins):
                kind = code & 0xFF
                if kind == R:
                    got = raw[rs * pitch + b]
                elif kind == B:
                    got = raw[b * rows + rs]
                else:
                    base = (rs * groups + b // 64) * 64
                    got = sum(((int(raw[base + p * 8 + (b % 64) // 8]) >> (b % 8)) & 1) << p for p in range(8))
                assert got == D[r, b], (code, r, b)


@pytest.mark.parametrize("code", LAYOUTS)
def test_files_written_under_a_layout_hold_its_bytes_and_load_back(tmp_path, code):
    lay, host, _ = odd_layout()
    sp = make_species(lay)
    sc = default_schema()
    sc.layout = code
    p = tmp_path / "foreign.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp, schema=sc)
    h = HixfFile(p, schema=sc)
    # kind and row order as given; where the schema leaves the pitch rule open and the records store a pitch, the loader takes that
    assert h.layout & 0x1FF == code & 0x1FF and h.layout & 0x600 == (code & 0x600 or (0 if code & 0xFF == S else STO))
    for a, b in zip(h.ixfs, host):
        raw, pitch = to_source_layout(b, code)
        assert a["bins"] == b["bins"] and a["seg_len"] == b["seg_len"] and a["seed"] == b["seed"]
        assert a["stride"] == (b["bins"] + 63) // 64 * 64 and a["src_stride"] == pitch
        assert np.array_equal(a["data"], raw), describe_layout(code)
        assert np.array_equal(a["next_ixf"], b["next_ixf"]) and np.array_equal(a["fname_idx"], b["fname_idx"])
    assert h.species == sp
    h.close()
    # a schema that stores neither pitch nor segment length: everything comes from the array lengths under the layout's rule
    if code & 0x600 != STO:
        bare = make_schema(2, 1, 0, -1, -1, 1, layout=code)
        q = tmp_path / "bare.hixf"
        store_hixf(q, host, lay["n_user_bins"], sp, schema=bare)
        h = HixfFile(q, schema=bare)
        for a, b in zip(h.ixfs, host):
            assert (a["bins"], a["seg_len"], a["seed"]) == (b["bins"], b["seg_len"], b["seed"])
            assert np.array_equal(a["data"], to_source_layout(b, code)[0])
        h.close()


def test_plain_load_settles_on_what_the_lengths_admit_and_set_layout_switches(tmp_path):
    """a foreign file is loaded before anyone knows its layout (`taxor pin` decides by probing the bytes): the loader takes the
    pitch rule the array lengths admit, set_layout then switches kind / row order and recomputes every IXF; a layout whose pitch
    contradicts a length is refused and leaves the view as it was"""
    lay, host, _ = odd_layout(8)
    sp = make_species(lay)
    bare = make_schema(2, 1, 0, -1, -1, 1, layout=B | UNP)          # bin-major, exactly `bins` columns, nothing else stored
    p = tmp_path / "binmajor.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp, schema=bare)
    sc, report = probe_hixf(p)
    assert (sc.n_before, sc.n_after) == (2, 1) and "pitch = exactly bins" in report
    h = HixfFile(p)                                                   # default schema fails, probed one loads: row-interleaved, unpadded
    assert h.foreign_schema and h.layout == R | UNP
    assert all(a["src_stride"] == a["bins"] and a["seg_len"] == b["seg_len"] for a, b in zip(h.ixfs, host))
    with pytest.raises(TaxorError):
        h.set_layout(B)                                               # padded columns: 66 -> 128 does not divide the arrays
    assert h.layout == R | UNP and h.ixfs[0]["src_stride"] == 66
    with pytest.raises(TaxorError):
        h.set_layout(S)
    with pytest.raises(TaxorError):
        h.set_layout(B | STO)                                         # the records store no pitch
    with pytest.raises(TaxorError):
        h.set_layout(0x003)                                           # not a layout
    h.set_layout(B | UNP | PM)
    assert h.layout == B | UNP | PM
    h.set_layout(B | UNP)
    for a, b in zip(h.ixfs, host):
        assert (a["bins"], a["stride"], a["seg_len"], a["src_stride"]) == (b["bins"], (b["bins"] + 63) // 64 * 64, b["seg_len"], b["bins"])
        assert np.array_equal(a["data"], to_source_layout(b, B | UNP)[0])
    h.close()


def test_search_layout_files_are_untouched_by_the_layout_machinery(tmp_path):
    """this library's own files: layout 0, stored stride == the search stride, bytes as they lie (the upload path of round 4)"""
    lay, host, _ = odd_layout(9)
    p = tmp_path / "own.hixf"
    store_hixf(p, host, lay["n_user_bins"], make_species(lay))
    h = HixfFile(p)
    assert not h.foreign_schema and h.layout & 0x1FF == 0
    for a, b in zip(h.ixfs, host):
        assert a["stride"] == b["stride"] == a["src_stride"] and np.array_equal(a["data"], b["data"])
    h.close()


@pytest.mark.parametrize("frame", [(8, 0, 0), (64, 0, 0), (64, 0, 1), (64, 1, 0), (1, 0, 1), (8, 1, 0)])
def test_probe_finds_other_framings_of_the_fingerprint_vector(tmp_path, frame):
    """the vector's length word need not count bytes: a std::vector<uint64_t> counts words, an sdsl int_vector counts BITS (stored in
    whole 64-bit words, with a width byte next to the length) -- `taxor probe` tries those framings, and a plain load falls back to
    what it finds; here with bare records and bit-sliced plane words, the shape such a class would most likely hold"""
    unit, before, after = frame
    lay, host, _ = odd_layout(12)
    sp = make_species(lay)
    sc = make_schema(2, 1, 0, -1, -1, 1, layout=S, len_unit=unit, skip_before_len=before, skip_after_len=after)
    p = tmp_path / "framed.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp, schema=sc)
    found, report = probe_hixf(p)
    assert (found.n_before, found.n_after, found.len_unit, found.skip_before_len, found.skip_after_len) == (2, 1, unit, before, after), report
    assert ("bits" in report) == (unit == 64) and ("64-bit words" in report) == (unit in (8, 64))
    h = HixfFile(p)
    assert h.foreign_schema
    h.set_layout(S)
    for a, b in zip(h.ixfs, host):
        assert (a["bins"], a["seg_len"], a["seed"]) == (b["bins"], b["seg_len"], b["seed"])
        assert np.array_equal(a["data"], to_source_layout(b, S)[0])
    h.close()
