#!/usr/bin/env python3
"""k_syncmers alone on the chip: a searcher over a one-IXF dummy index (the query work is negligible), a resident batch of
synthetic reads, `steps` runs; prints the kernel's own rate from the searcher's HIP events.
usage: python profiles/syncmer_alone.py [reads] [read_len] [steps]      (under rocprofv3 --pmc ... for the counters)"""
import os
import sys

import numpy as np

# TAXOR_AB_ROOT: a directory holding ANOTHER build of the package (an A/B run against an older library)
sys.path.insert(0, os.environ.get("TAXOR_AB_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from taxor_amd import GpuIndex, Searcher, synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
bins = 64
idx = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins),
                     data=np.zeros(3 * 16 * 64, np.uint8))], bins)
g, go = synth.random_genomes(16, 2_000_000, seed=3)
bases, offs, _ = synth.synth_reads(g, go, n_reads, read_len, error_rate=0.02, frac_random=0.1, seed=5, threads=os.cpu_count() or 8)
sr = Searcher(idx, ratio=0.99, time_kernels=True)
sr.upload(bases, offs)
sr.run()
sr.sync()
ms = 0.0
for _ in range(steps):
    sr.run()
    sr.sync()
    ms += sr.stats()["syncmer_ms"]
st = sr.stats()
print(f"{n_reads} reads x {read_len} bp, {steps} runs: k_syncmers {ms / steps:.3f} ms per batch = {n_reads * read_len / (ms / steps) / 1e6:.1f} Gbp/s "
      f"({st['n_hashes'] / n_reads:.1f} distinct hashes per read)")
sr.close()
idx.close()
