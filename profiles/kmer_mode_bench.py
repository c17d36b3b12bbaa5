#!/usr/bin/env python3
"""Throughput of the search on an index built WITHOUT --use-syncmer (every canonical k-mer, or window minimisers):
viral-class footprint, reads resident in HBM.  usage: python profiles/kmer_mode_bench.py [k] [window] [reads] [read_len]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxor_amd import GpuIndex, Searcher, synth  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
w = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_reads = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
read_len = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
g, go = synth.random_genomes(32, 100000)
bins = 64
dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins),
                       data=np.zeros(3 * 16 * 64, np.uint8))], bins, k=k, s=0, t=0, use_syncmer=False, window_size=w)
hs = Searcher(dummy, ratio=0.5)
hoff, hashes = hs.seq_to_syncmers(g, go)
hs.close()
dummy.close()
planted = [np.unique(hashes[int(hoff[i]):int(hoff[i + 1])]) for i in range(32)]
per_bin = max(len(p) for p in planted)
lay = synth.make_layout(planted, root_bins=256, child_bins=64, n_children=252, root_max_elems=per_bin * 20, child_max_elems=per_bin + 64, build="gpu")
idx = synth.device_index(lay, k=k, s=0, t=0, use_syncmer=False, window_size=w)
bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=0.02, frac_random=0.1, threads=os.cpu_count() or 8)
sr = Searcher(idx, error_rate=0.04, time_kernels=True)
sr.upload(bases, offs)
sr.run(); sr.sync()
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    sr.run(); sr.sync()
dt = (time.perf_counter() - t0) / steps
st = sr.stats()
res = sr.fetch()
hit = sum(1 for r in range(n_reads) if origin[r] >= 0 and lay["planted_user_bin"][origin[r]] in res.user_bin[int(res.read_off[r]):int(res.read_off[r + 1])].tolist())
prof = sr.phase_profile() if os.environ.get("TAXOR_PROFILE_PHASES") else None
if prof is not None:
    p = prof[8:].astype(float); print("-- k_query_level phases:", ", ".join(f"{100*v/p.sum():.1f}%" for v in p))
print(f"algorithmic {st['query_bytes']/1e9:.1f} GB requested {st['query_touched_bytes']/1e9:.1f} GB per step")
print(f"k={k} window={w} model={sr.model}: index {idx.data_bytes/1e9:.2f} GB, {n_reads} x {read_len} bp, {st['n_hashes']/n_reads:.0f} hashes/read, "
      f"{n_reads*read_len/dt/1e6:.0f} Mbp/s ({dt*1e3:.1f} ms/step; hashing {st['syncmer_ms']:.1f} ms, query {st['query_ms']:.1f} ms), "
      f"requested {st['query_touched_bytes']/(st['query_ms']*1e-3)/1e9:.0f} GB/s, planted reads classified {hit}/{int((origin >= 0).sum())}")
