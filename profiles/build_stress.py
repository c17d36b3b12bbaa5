#!/usr/bin/env python3
"""Stress of the GPU hierarchical builder: many builds of mid-size hierarchies under different seeds, salts and scratch budgets;
every build verifies every key on the device (the call fails otherwise), every fourth is built twice and compared byte for byte,
and a sample of bins is looked up through the query kernel.  usage: python profiles/build_stress.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TAXOR_TUNING"] = "1"
from taxor_amd import GpuIndex, Searcher, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20250523)
t0 = time.time()
n = ins = twice = reseeds = in_lds = 0
while time.time() - t0 < budget:
    nc = int(rng.integers(1, 12))
    cb = int(rng.choice([3, 17, 64, 100, 128]))
    kpb = int(rng.choice([40, 900, 5000, 30000, 120000, 420000, 1100000]))
    if kpb > 120000:        # bins whose degree words are built in LDS in up to 12 passes per segment, and bins beyond that (global adds again)
        cb = int(rng.choice([3, 17]))
    # all bins of an IXF must peel under ONE seed (the reference's rule): bins of 100 k keys may be full, a hundred bins of forty keys
    # need room (a 40-key bin at load 0.48 still fails under 2 % of the seeds)
    slack = 1.0 if kpb >= 100000 else 3.0 if kpb < 1000 else float(rng.choice([1.1, 1.3, 2.0]))
    os.environ["TAXOR_BUILD_SCRATCH_MB"] = str(int(rng.choice([1, 8, 64, 3072])))
    shapes, ub, counts = synth.full_hierarchy_shapes(nc, cb, kpb, slack=slack)
    counts = counts.copy()
    lo = shapes[0]["bins"]
    counts[lo:] = rng.integers(max(1, kpb // 3), kpb + 1, size=counts.size - lo)        # unequal bins
    counts[lo + int(rng.integers(0, counts.size - lo))] = 0                             # and an empty one
    salt, seed0 = int(rng.integers(1, 2**62)), int(rng.integers(1, 2**62))
    idx = GpuIndex(shapes, ub)
    try:
        st, off = idx.build_hixf_synth(counts, salt=salt, seed0=seed0)
    except Exception:
        print(f"FAILED: children {nc}, bins {cb}, keys per bin <= {kpb}, slack {slack}, scratch {os.environ['TAXOR_BUILD_SCRATCH_MB']} MB, salt {salt}, seed0 {seed0}; "
              f"largest bins {sorted(int(c) for c in counts)[-3:]}", flush=True)
        raise
    assert st["keys_inserted"] == 2 * int(counts.sum()), (st, int(counts.sum()))
    ins += st["keys_inserted"]
    reseeds += st["reseeds"]
    in_lds += st["keys_counted_in_lds"]
    sr = Searcher(idx, ratio=0.5)
    for _ in range(3):
        c, b = int(rng.integers(1, nc + 1)), int(rng.integers(0, cb))
        g = lo + (c - 1) * cb + b
        m = int(off[g + 1] - off[g])
        if m:
            keys = synth.synth_keys_host(int(off[g]), m, salt)
            assert sr.ixf_bulk_count(c, keys)[b] == m and sr.ixf_bulk_count(0, keys)[c - 1] == m, (nc, cb, kpb, c, b)
    sr.close()
    if n % 4 == 0:
        idx2 = GpuIndex(shapes, ub)
        idx2.build_hixf_synth(counts, salt=salt, seed0=seed0)
        for i in (0, 1, nc):
            # (columns of bins that were built: a bin without keys keeps what the freshly allocated shell happened to hold)
            cols = np.arange(nc) if i == 0 else np.flatnonzero(counts[lo + (i - 1) * cb: lo + i * cb] > 0)
            a = idx.download_ixf(i).reshape(-1, shapes[i]["stride"])[:, cols]
            b_ = idx2.download_ixf(i).reshape(-1, shapes[i]["stride"])[:, cols]
            assert idx.ixf_seed(i) == idx2.ixf_seed(i) and np.array_equal(a, b_), f"two builds of IXF {i} differ ({nc}, {cb}, {kpb})"
        idx2.close()
        twice += 1
    idx.close()
    n += 1
print(f"{n} hierarchies built ({ins / 1e9:.2f} G insertions, {in_lds / 1e9:.2f} G of them with degree words built in LDS, every key verified on the device; {reseeds} IXFs redone under a redrawn seed), "
      f"{twice} of them twice with identical bytes, {time.time() - t0:.0f} s")
