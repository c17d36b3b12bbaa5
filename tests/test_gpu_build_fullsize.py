"""GPU construction at full size: a hierarchy of more than 10 GB in which EVERY bin is a real filter --
80 child IXFs of 128 leaf bins with 422 000 keys each (the leaf sizes of bench.py's GTDB-class workload) under a root of 80
merged bins of 54 M keys -- built by taxor_gpu_index_build_hixf_ex from keys generated on the device, then downloaded and checked
by the CPU oracle alone:
  * every one of the 4.3 G leaf keys is found in its own bin AND in the merged bin of the root above it (no false negative),
    with orc_ixf_bulk_count's rule on that column (oracle/taxor_oracle.c, orc_ixf_synth_keys_found);
  * every 64th key goes through orc_ixf_bulk_count itself, over all bins: the foreign-bin hit rate is 2^-8 +- 10 %;
  * a second build of two of the IXFs gives the same bytes (the columns are a function of keys and seed).
The keys are never on the host: device, oracle and numpy regenerate them from their indices (tests/test_build_check_cpu.py)."""
import os
import time

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import GpuIndex, synth

pytestmark = pytest.mark.gpu

SALT = 20250523


def test_ten_gigabyte_hierarchy_every_bin_built_and_checked_by_the_oracle():
    nc = int(os.environ.get("TAXOR_BUILD_TEST_CHILDREN", "80"))
    cb, kpb = 128, 422000
    shapes, ub, counts = synth.full_hierarchy_shapes(nc, cb, kpb)
    idx = GpuIndex(shapes, ub)
    if nc >= 80:
        assert idx.data_bytes > 10e9
    t0 = time.time()
    st, off = idx.build_hixf_synth(counts, salt=SALT, seed0=5)
    t_build = time.time() - t0
    n_leaf = int(counts.sum())
    assert st["keys_inserted"] == 2 * n_leaf                  # every key at its leaf and once more in the root's merged bin
    rate = st["keys_inserted"] / st["seconds_total"]
    print(f"\nbuilt {idx.data_bytes / 1e9:.2f} GB, {idx.n_ixf} IXFs, {st['keys_inserted'] / 1e9:.2f} G insertions in {st['seconds_total']:.2f} s "
          f"= {rate / 1e9:.2f} G/s (peel {st['seconds_peel']:.2f} s, assign + verify {st['seconds_assign']:.2f} s, unions {st['seconds_union']:.2f} s; "
          f"{st['chunks']} chunks, {st['rounds_max']} rounds at most, {st['reseeds']} reseeds, scratch {st['scratch_bytes'] / 1e9:.2f} GB; wall {t_build:.1f} s)")
    assert rate > 0.5e9, "the builder is an order of magnitude faster than this on an idle MI355X"

    # ---- the oracle, on the downloaded bytes --------------------------------------------------------------------------------
    t0 = time.time()
    ixfs = []
    for i, f in enumerate(shapes):
        ixfs.append(dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=idx.ixf_seed(i), data=idx.download_ixf(i)))
    h = orc.Hixf(ixfs, [f["next_ixf"] for f in shapes], [f["fname_idx"] for f in shapes])
    t_down = time.time() - t0
    t0 = time.time()
    rb = shapes[0]["bins"]
    root_counts = np.zeros(rb, dtype=np.uint64)
    missing_own = missing_up = 0
    all_hits_child = sampled_child = sampled_root = 0
    for c in range(1, nc + 1):
        child_counts = np.zeros(cb, dtype=np.uint64)
        for b in range(cb):
            g = rb + (c - 1) * cb + b
            first, n = int(off[g]), int(off[g + 1] - off[g])
            found, s = h.synth_keys_found(c, b, first, n, SALT, sample_step=64, counts=child_counts)
            missing_own += n - found
            sampled_child += s
            found_up, s_up = h.synth_keys_found(0, c - 1, first, n, SALT, sample_step=64, counts=root_counts)
            missing_up += n - found_up
            sampled_root += s_up
        all_hits_child += int(child_counts.sum())
    assert missing_own == 0 and missing_up == 0, (missing_own, missing_up)
    t_check = time.time() - t0
    # bulk_count over all bins: a sampled key hits its own bin (it is a member: no key is missing) and every other bin with
    # probability 2^-8
    foreign_child = (all_hits_child - sampled_child) / (sampled_child * (cb - 1))
    foreign_root = (int(root_counts[:nc].sum()) - sampled_root) / (sampled_root * (nc - 1)) if nc > 1 else 1 / 256
    print(f"oracle: {n_leaf / 1e9:.2f} G keys found in their own bin and in the root's merged bin; foreign-bin hit rate {foreign_child * 256:.4f} / 256 in the "
          f"children ({sampled_child} keys x {cb - 1} bins), {foreign_root * 256:.4f} / 256 in the root ({sampled_root} keys x {nc - 1} bins); "
          f"download {t_down:.1f} s, check {t_check:.1f} s")
    assert abs(foreign_child * 256 - 1) < 0.1 and abs(foreign_root * 256 - 1) < 0.1
    # ---- the same build again: the same bytes ---------------------------------------------------------------------------------
    idx2 = GpuIndex(shapes, ub)
    st2, _ = idx2.build_hixf_synth(counts, salt=SALT, seed0=5)
    assert st2["keys_inserted"] == st["keys_inserted"]
    for i in (0, 1, nc):
        assert idx2.ixf_seed(i) == idx.ixf_seed(i)
        a = idx2.download_ixf(i)
        if i == 0:
            a2 = a.reshape(-1, shapes[0]["stride"])[:, :nc]
            b2 = ixfs[0]["data"].reshape(-1, shapes[0]["stride"])[:, :nc]
            assert np.array_equal(a2, b2), "two builds of the root differ"
        else:
            assert np.array_equal(a, ixfs[i]["data"]), f"two builds of IXF {i} differ"
    idx2.close()
    idx.close()
