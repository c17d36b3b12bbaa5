"""Run-time choice of the IXF arithmetic (taxor_amd/csrc/ixf_arith.h): the host-side XOR-filter builder under
an arithmetic code and the oracle parametrised the same way (its own C restatement) agree -- every key of a bin built under a
code is found under that code and, for codes that really differ, not under the default reading."""
import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import _lib
from taxor_amd.search import arith_code

import ctypes as C


def _build(keys, seed, seg, code):
    col = np.zeros(3 * seg, dtype=np.uint8)
    rc = _lib.lib().taxor_ixf_build_bin_arith(keys.ctypes.data, keys.size, seed, seg, code, col.ctypes.data)
    return rc, col


def test_code_zero_is_the_default_reading_and_roundtrips():
    assert arith_code() == 0
    v = _lib.IxfVariant()
    for kh in range(4):
        for sm in range(4):
            for rot in (1, 16, 21, 32, 63):
                for red in range(3):
                    for fp in range(4):
                        code = arith_code(kh, sm, rot, red, fp)
                        assert (code == 0) == ((kh, sm, rot, red, fp) == (0, 0, 21, 0, 0))
                        _lib.lib().taxor_ixf_arith_decode(code, C.byref(v))
                        assert (v.key_hash, v.seed_mode, v.rot, v.reduce, v.fp_mode) == (kh, sm, rot, red, fp)


@pytest.mark.parametrize("kh,sm,rot,red,fp", [(0, 0, 21, 0, 0), (0, 1, 21, 1, 1), (2, 2, 16, 2, 2), (3, 0, 16, 0, 3), (1, 1, 21, 1, 0),
                                               (0, 3, 21, 0, 0), (2, 0, 7, 2, 1)])
def test_builder_and_oracle_agree_under_a_code(kh, sm, rot, red, fp):
    rng = np.random.default_rng(kh * 1000 + sm * 100 + rot)
    keys = np.unique(rng.integers(0, 1 << 63, size=3000, dtype=np.uint64))
    seg = int(_lib.lib().taxor_ixf_seg_len(keys.size + 64))
    code = arith_code(kh, sm, rot, red, fp)
    col = None
    for seed in (11, 0xDEADBEEFCAFEF00D, 0x123456789ABCDEF):
        rc, col = _build(keys, seed, seg, code)
        if rc == 0:
            break
    assert rc == 0
    bins, stride = 3, 64
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    data[:, 1] = col
    host = [dict(bins=bins, stride=stride, seg_len=seg, seed=seed, data=data.reshape(-1))]
    nx, fn = [np.zeros(bins, np.int64)], [np.arange(bins, dtype=np.int64)]
    h = orc.Hixf(host, nx, fn, arith=code)
    counts = h.ixf_bulk_count(0, keys)
    assert counts[1] == keys.size                       # every key answers in its bin under the code it was built with
    assert counts[0] < keys.size * 0.02 and counts[2] < keys.size * 0.02
    other = rng.integers(0, 1 << 63, size=3000, dtype=np.uint64)
    assert h.ixf_bulk_count(0, other)[1] < 3000 * 0.02   # non-keys: the 2^-8 false-positive floor
    if code != 0:
        assert orc.Hixf(host, nx, fn).ixf_bulk_count(0, keys)[1] < keys.size * 0.05   # the default reading does not find them
