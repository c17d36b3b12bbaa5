#!/usr/bin/env python3
"""Throughput of the GPU hierarchical builder (taxor_gpu_index_build_hixf): root of merged bins over child IXFs of leaf
bins with random keys.  usage: python profiles/build_bench.py [children] [bins_per_child] [keys_per_bin]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxor_amd import GpuIndex, synth  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kpb = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
rng = np.random.default_rng(1)
rb = max(64, nc)
nx = np.zeros(rb, np.int64)
fn = np.arange(rb, dtype=np.int64)
nx[:nc] = np.arange(1, nc + 1)
fn[:nc] = -1
shapes = [dict(bins=rb, stride=(rb + 63) // 64 * 64, seg_len=synth.seg_len_for(cb * kpb), seed=1, next_ixf=nx, fname_idx=fn, data=None)]
ub = rb
for c in range(nc):
    shapes.append(dict(bins=cb, stride=(cb + 63) // 64 * 64, seg_len=synth.seg_len_for(kpb), seed=2 + c,
                       next_ixf=np.full(cb, c + 1, np.int64), fname_idx=np.arange(ub, ub + cb, dtype=np.int64), data=None))
    ub += cb
t0 = time.time()
allk = rng.integers(1, 2**63, size=nc * cb * kpb, dtype=np.uint64)     # distinct with overwhelming probability
leaf = {}
p = 0
for c in range(nc):
    for b in range(cb):
        leaf[(c + 1, b)] = allk[p:p + kpb]
        p += kpb
print(f"{allk.size/1e6:.1f} M leaf keys generated in {time.time()-t0:.1f} s", flush=True)
idx = GpuIndex(shapes, ub)
print(f"index shell: {idx.data_bytes/1e9:.2f} GB, {idx.n_ixf} IXFs", flush=True)
t0 = time.time()
rounds = idx.build_hixf(leaf, seed0=5)
dt = time.time() - t0
tot = allk.size * 2      # every key is inserted at its leaf and once more in the root's merged bin
print(f"build_hixf: {dt:.2f} s wall (incl. host marshalling), {rounds} peeling rounds max, {tot/dt/1e6:.1f} M key insertions/s")
# spot check through the query kernel: the keys of one leaf bin are all found there and in the root's merged bin
from taxor_amd import Searcher  # noqa: E402
sr = Searcher(idx, ratio=0.5)
keys = leaf[(3, 5)]
assert sr.ixf_bulk_count(3, keys)[5] == keys.size and sr.ixf_bulk_count(0, keys)[2] == keys.size
print("spot check OK")
