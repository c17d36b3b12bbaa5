#!/usr/bin/env python3
"""Host-side marks of a few 1024-read calls (TAXOR_TUNING=1 TAXOR_TRACE_BATCH=1): where a piece's microseconds go before its kernels are
enqueued.  usage: python profiles/r04/small_host_trace.py [reads per call]"""
import os
import sys
import time

os.environ["TAXOR_TUNING"] = "1"
os.environ["TAXOR_TRACE_BATCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

import bench  # noqa: E402
from taxor_amd import Searcher  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
args = bench.parse_args(["--workload", "gtdb", "--batches", "1", "--reads", "16384"])
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
bases, offs = batches[0]
sr = Searcher(idx, error_rate=args.error_rate)
for rep in range(6):
    lo = rep * n
    o = offs[lo:lo + n + 1]
    b = bases[int(o[0]):int(o[-1])]
    t0 = time.perf_counter()
    r = sr.search_batch(b, (o - o[0]).astype(np.uint64), copy=False)
    print(f"call {rep}: {1e6 * (time.perf_counter() - t0):.0f} us, {r.user_bin.size} tuples", file=sys.stderr, flush=True)
sr.close()
idx.close()
