#!/usr/bin/env python3
"""Per-phase cycle shares of k_syncmers and k_query_level (s_memtime marks in instrumented instantiations of the same
kernels, TAXOR_PROFILE_PHASES=1) on bench.py's workload.  usage: python3 profiles/phase_profile.py [bench.py flags]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TAXOR_TUNING"] = "1"          # the library reads its measurement knobs only behind this gate
os.environ["TAXOR_PROFILE_PHASES"] = "1"
import torch  # noqa: F401,E402  (its HIP runtime first)

import bench  # noqa: E402
from taxor_amd import Searcher  # noqa: E402

SYNC = ["cursor", "stage words", "s-mer values", "window argmins", "scan+chain+select", "hash emit", "dedup passes", "copy-out"]
QUERY = ["cursor+flush", "metadata+probes", "dense gathers", "prune check", "sparse gathers", "tally", "final flush", "-"]

args = bench.parse_args(sys.argv[1:] + ["--batches", "2"])
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
sr = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
for rep, (bases, offs) in enumerate(batches):
    sr.upload(bases, offs)
    sr.run()
    sr.sync()
    prof = sr.phase_profile()
    st = sr.stats()
    if rep == 0:
        continue                   # warm-up
    print(f"== {args.workload}: {offs.size - 1} reads x {info['read_len']} bp, family size {info['fam_size']}, read error {args.read_error}; "
          f"instrumented run: syncmers {st['syncmer_ms']:.2f} ms, query {st['query_ms']:.2f} ms, total {st['total_ms']:.2f} ms, "
          f"{st['n_work_items'] / st['n_reads']:.2f} work items/read")
    for name, labels, p in (("k_syncmers", SYNC, prof[:8]), ("k_query_level", QUERY, prof[8:])):
        tot = float(p.sum()) or 1.0
        print(f"-- {name}: " + ", ".join(f"{l} {100.0 * float(v) / tot:.1f}%" for l, v in zip(labels, p) if l != "-"))
sr.close()
idx.close()
