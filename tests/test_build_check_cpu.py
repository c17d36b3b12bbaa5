"""The oracle's checker of BUILT filters (orc_ixf_synth_keys_found, used by the GPU builder's full-size test) on filters built by
the host-side builder (taxor_ixf_build_bin_arith): synthetic keys are regenerated identically by the library, the oracle and
numpy; every key of a bin is found in its own column, a foreign column answers at the 2^-8 rate, a damaged column is noticed."""
import ctypes as C

import numpy as np

from oracle import oracle as orc
from taxor_amd import _lib, synth


def test_three_generators_of_the_synthetic_keys_agree():
    L = _lib.lib()
    want = synth.synth_keys_host(10**12, 1000, 20250523)
    got = np.zeros(1000, dtype=np.uint64)
    O = orc.lib()
    O.orc_synth_keys.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    O.orc_synth_keys(10**12, 1000, 20250523, got.ctypes.data_as(C.c_void_p))
    assert np.array_equal(got, want)
    assert [L.taxor_synth_key(10**12 + i, 20250523) for i in range(50)] == [int(x) for x in want[:50]]
    assert np.unique(synth.synth_keys_host(0, 200000, 7)).size == 200000          # a bijection of the index: distinct


def test_checker_finds_every_key_and_notices_damage():
    salt, n, bins, stride = 99, 30000, 5, 64
    seg = synth.seg_len_for(n)
    rng = np.random.default_rng(4)
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    first = {0: 0, 3: 10**9}                                                       # two built columns, three of random bytes
    seed, cols = synth.build_columns({b: synth.synth_keys_host(f, n, salt) for b, f in first.items()}, seg, 11)
    for b, col in cols.items():
        data[:, b] = col
    h = orc.Hixf([dict(bins=bins, stride=stride, seg_len=seg, seed=seed, data=data.reshape(-1))], [np.zeros(bins, np.int64)], [np.arange(bins)])
    for b, f in first.items():
        counts = np.zeros(bins, dtype=np.uint64)
        found, sampled = h.synth_keys_found(0, b, f, n, salt, sample_step=1, counts=counts)
        assert found == n and sampled == n and counts[b] == n
        assert np.array_equal(counts.astype(np.uint32), h.ixf_bulk_count(0, synth.synth_keys_host(f, n, salt)))     # = bulk_count itself
        foreign = np.delete(counts, b).sum() / (n * (bins - 1))
        assert abs(foreign - 1 / 256) < 0.1 / 256 * 3                               # 120 k trials: a loose band
    found, _ = h.synth_keys_found(0, 1, 0, n, salt)                                # a column of random bytes
    assert abs(found / n - 1 / 256) < 0.002
    data[5:3 * seg:7, 0] ^= 1                                                       # damage column 0
    found, _ = h.synth_keys_found(0, 0, 0, n, salt)
    assert found < n
