"""Fingerprint bytes in ANOTHER writer's layout are transposed into the search layout on the device while the index is uploaded
(taxor_amd/csrc/relayout.hip; layouts: taxor_amd/csrc/ixf_layout.h).  The serialiser of seqan3::interleaved_xor_filter is
un-vendored (hierarchical_interleaved_xor_filter.hpp:152-158): whatever a published .hixf turns out to hold must be searchable
the day it arrives.  Checked here: every layout x awkward shapes -> the resident rows equal the original ones byte for byte
(source bytes made by an independent numpy restatement, taxor_amd.search.to_source_layout); arrays of several upload chunks; and
a three-level index stored under each layout, loaded from the file and searched -- tuples identical to the oracle's."""
import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import GpuIndex, Searcher, _lib, synth
from taxor_amd.hixf_file import HixfFile, default_schema, describe_layout, store_hixf
from taxor_amd.search import to_source_layout
from tests.test_hixf_file_cpu import make_species
from tests.test_hixf_layouts_cpu import LAYOUTS

pytestmark = pytest.mark.gpu


def _random_ixf(rng, bins, seg_len):
    stride = (bins + 63) // 64 * 64
    D = np.zeros((3 * seg_len, stride), np.uint8)
    D[:, :bins] = rng.integers(0, 256, (3 * seg_len, bins), dtype=np.uint8)
    return dict(bins=bins, stride=stride, seg_len=seg_len, seed=7, data=D.reshape(-1), next_ixf=np.zeros(bins, np.int64),
                fname_idx=np.arange(bins, dtype=np.int64))


@pytest.mark.parametrize("code", LAYOUTS)
def test_every_layout_lands_in_the_search_layout(code):
    """bins below / at / above multiples of 64 and 128, one bin, segment lengths that leave rows unaligned; padding columns zero"""
    rng = np.random.default_rng(code)
    for bins, seg in [(1, 7), (63, 40), (64, 33), (65, 50), (130, 21), (200, 999), (1000, 77), (2049, 130)]:
        f = _random_ixf(rng, bins, seg)
        raw, pitch = to_source_layout(f, code)
        idx = GpuIndex([dict(f, data=raw, src_stride=pitch)], bins, layout=code)
        got = idx.download_ixf(0)
        assert np.array_equal(got, f["data"]), (describe_layout(code), bins, seg)
        idx.close()


@pytest.mark.parametrize("code", LAYOUTS)
def test_arrays_of_several_upload_chunks(code):
    """8 MiB pieces: a wide IXF of many row chunks / column groups, and a narrow one whose bin-major columns are cut into row strips"""
    rng = np.random.default_rng(100 + code)
    for bins, seg in [(4096, 2999), (130, 50001)]:
        f = _random_ixf(rng, bins, seg)
        raw, pitch = to_source_layout(f, code)
        assert raw.size > (16 << 20)
        idx = GpuIndex([dict(f, data=raw, src_stride=pitch)], bins, layout=code)
        assert np.array_equal(idx.download_ixf(0), f["data"]), (describe_layout(code), bins, seg)
        idx.close()


@pytest.mark.parametrize("code", LAYOUTS)
def test_an_index_stored_under_a_layout_is_searched_like_the_original(tmp_path, code):
    g, go = synth.random_genomes(8, 12000, seed=31)
    planted = [np.unique(orc.seq_to_syncmers(bytes(g[int(go[i]):int(go[i + 1])]))) for i in range(8)]
    lay = synth.make_layout(planted, root_bins=200, child_bins=70, n_children=4, seed=32)
    host = synth.materialize_host(lay)
    sc = default_schema()
    sc.layout = code
    p = tmp_path / "foreign.hixf"
    store_hixf(p, host, lay["n_user_bins"], make_species(lay), schema=sc)
    h = HixfFile(p, schema=sc)
    idx = GpuIndex(h.ixfs, h.n_user_bins, layout=h.layout)
    for i, f in enumerate(host):
        assert np.array_equal(idx.download_ixf(i).reshape(-1, idx.shapes[i][1])[:, :f["bins"]],
                              np.asarray(f["data"]).reshape(-1, f["stride"])[:, :f["bins"]])
    bases, offs, _ = synth.synth_reads(g, go, 400, 3000, error_rate=0.02, frac_random=0.15, seed=33)
    want = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host]).search_batch(bases, offs, threads=4)
    sr = Searcher(idx)
    res = sr.search_batch(bases, offs)
    assert np.array_equal(res.n_hashes, want[0]) and np.array_equal(res.read_off, want[1])
    assert np.array_equal(res.user_bin, want[2]) and np.array_equal(res.count, want[3]) and res.user_bin.size > 300
    sr.close(); idx.close(); h.close()


def test_a_layout_that_contradicts_the_view_is_refused():
    rng = np.random.default_rng(5)
    f = _random_ixf(rng, 100, 20)
    with pytest.raises(_lib.TaxorError):
        GpuIndex([dict(f, data=np.zeros(60 * 64, np.uint8), src_stride=64)], 100, layout=_lib.LAYOUT_BIN_MAJOR | _lib.LAYOUT_PITCH_BINS)   # 64 columns for 100 bins
    with pytest.raises(_lib.TaxorError):
        GpuIndex([f], 100, layout=0x003)


def test_the_pitch_a_layout_code_names_counts_when_src_stride_is_left_zero():
    """a C-ABI caller who passes the code `taxor pin` prints -- 0x200, rows at exactly `bins` bytes -- but leaves
    taxor_ixf_view::src_stride 0 used to get the bytes uploaded as they lie (wrong answers, no error).  The pitch rule of the code
    now counts: unpadded -> bins; padded / none -> the index's own stride; "the record's stored scalar" without a scalar is refused."""
    rng = np.random.default_rng(11)
    for bins, seg in [(100, 37), (65, 50), (130, 21)]:
        f = _random_ixf(rng, bins, seg)
        for code in (_lib.LAYOUT_ROWS | _lib.LAYOUT_PITCH_BINS, _lib.LAYOUT_BIN_MAJOR | _lib.LAYOUT_PITCH_BINS,
                     _lib.LAYOUT_ROWS | _lib.LAYOUT_POSITION_MAJOR | _lib.LAYOUT_PITCH_BINS):
            raw, pitch = to_source_layout(f, code)
            assert pitch == bins
            idx = GpuIndex([dict(f, data=raw, src_stride=0)], bins, layout=code)          # src_stride left 0: the code says "bins"
            assert np.array_equal(idx.download_ixf(0), f["data"]), (describe_layout(code), bins, seg)
            idx.close()
        with pytest.raises(_lib.TaxorError):
            GpuIndex([dict(f, data=f["data"], src_stride=0)], bins, layout=_lib.LAYOUT_ROWS | _lib.LAYOUT_PITCH_STORED)
