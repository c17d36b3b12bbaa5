#!/usr/bin/env python3
"""Experiment: do two query pipelines that run concurrently on one GPU (each with resident batches) classify more per
second than one?  The narrow-row levels are bound by DRAM row activations, the root level by bytes -- if the memory
system overlaps the two, running level 1 of one pipeline beside level 0 of another is worth doing inside one searcher."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

import bench  # noqa: E402
from taxor_amd import Searcher  # noqa: E402

args = bench.parse_args(sys.argv[1:] + ["--batches", "4"])
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
srs = []
for b, o in batches:
    s = Searcher(idx, error_rate=args.error_rate)
    s.upload(b, o)
    s.run(); s.sync()
    srs.append(s)
bases_per = [int(o[-1]) for _, o in batches]


def loop(ids, reps):
    for _ in range(reps):
        for i in ids:
            srs[i].run()
            srs[i].sync()


for name, groups in (("one pipeline", [[0, 1, 2, 3]]), ("two concurrent", [[0, 1], [2, 3]]), ("four concurrent", [[0], [1], [2], [3]])):
    reps = 4 * len(groups) // 1
    th = [threading.Thread(target=loop, args=(g, reps)) for g in groups]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = sum(bases_per[i] for g in groups for i in g) * reps
    print(f"{name}: {total / dt / 1e6:.0f} Mbp/s ({dt * 1e3:.1f} ms for {total / 1e9:.2f} Gbp)")
