#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace CSV of bench.py into per-kernel and per-level figures.
  * k_query_level: average duration per HIXF level (launches come in groups of `depth` per sub-batch)
  * k_syncmers   : search launches only -- the launches before the first k_query_level hash the planted GENOMES
                   (3-Mbp "reads", global dedup table) and are listed separately
usage: trace_summary.py <dir with *kernel_trace.csv> <depth> [pmc counter_collection dir]"""
import csv
import glob
import os
import sys

d, depth = sys.argv[1], int(sys.argv[2])
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first_q = next((i for i, r in enumerate(rows) if "k_query_level" in r["Kernel_Name"]), len(rows))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
gen = [dur(r) for r in rows[:first_q] if "k_syncmers" in r["Kernel_Name"]]
syn = [dur(r) for r in rows[first_q:] if "k_syncmers" in r["Kernel_Name"]]
q = [dur(r) for r in rows if "k_query_level" in r["Kernel_Name"]]
print(f"== kernel trace summary ({len(rows)} dispatches)")
if gen:
    print(f"k_syncmers, genome hashing (before the first query launch): {len(gen)} launches, avg {sum(gen)/len(gen):.3f} ms")
if syn:
    print(f"k_syncmers, search launches: {len(syn)}, avg {sum(syn)/len(syn):.3f} ms, min {min(syn):.3f}, max {max(syn):.3f} "
          f"(wall duration of launches that overlap the query kernel at two blocks per CU, except each step's first)")
if q:
    print(f"k_query_level: {len(q)} launches, avg {sum(q)/len(q):.4f} ms")
    for lvl in range(depth):
        sel = q[lvl::depth]
        print(f"  level {lvl}: {len(sel)} launches, avg {sum(sel)/len(sel):.4f} ms")
tot = {}
for r in rows[first_q:]:
    n = r["Kernel_Name"].split("(")[0][-48:]
    tot[n] = tot.get(n, 0.0) + dur(r)
print("time by kernel after the first query launch (ms, kernels on different streams overlap):")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {v:10.3f}  {n}")
if len(sys.argv) > 3:
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(sys.argv[3], "**", "*counter_collection.csv"), recursive=True):
            rr = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter]
            if not rr:
                continue
            rr.sort(key=lambda r: int(r["Dispatch_Id"]))
            fq = next((i for i, r in enumerate(rr) if "k_query_level" in r["Kernel_Name"]), len(rr))
            s = [float(r["Counter_Value"]) for r in rr[fq:] if "k_syncmers" in r["Kernel_Name"]]
            if s:
                corr = 2.0 if counter == "FETCH_SIZE" else 1.0
                print(f"{counter} k_syncmers search launches: {len(s)}, {sum(s)/len(s)*1024*corr/1e6:.1f} MB per launch (x{corr:g} gfx950 correction)")
