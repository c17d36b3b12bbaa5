#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory (see run_profiles.sh) into the small text summary committed under
profiles/: per-kernel stats, and FETCH_SIZE / WRITE_SIZE per k_query_level launch with the gfx950 correction
of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B: double it for wide coalesced reads; unit KB)."""
import csv
import glob
import json
import os
import sys

out, wl = sys.argv[1], sys.argv[2]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


traffic = {}
print(f"== rocprofv3 summary: {os.path.basename(out)} ({wl})")
for f in find("stats/**/*kernel_stats.csv"):
    print(f"-- {os.path.relpath(f, out)}")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("   {:<60s} calls={:>6s} total_ns={:>14s} avg_ns={:>12s} pct={:>6s}".format(
            r.get("Name", "")[:60], r.get("Calls", ""), r.get("TotalDurationNs", ""), r.get("AverageNs", ""),
            r.get("Percentage", "")))
for name in ("bench_stats.json", "bench_pmc_fetch.json", "bench_pmc_write.json"):
    p = os.path.join(out, name)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                j = json.loads(line)
                print(f"-- {name}: value={j['value']} {j['unit']} ms_per_step={j['ms_per_step']} roofline={json.dumps(j['roofline'])}")
for kind, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(f"{kind}/**/*counter_collection.csv"):
        tot = {}
        n = {}
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r.get("Kernel_Name", "")
            key = "k_query_level" if "k_query_level" in k else ("k_syncmers" if "k_syncmers" in k else None)
            if key is None:
                continue
            tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"])
            n.setdefault(key, set()).add(r.get("Dispatch_Id"))
        for key in tot:
            launches = len(n[key])
            kb = tot[key]
            corr = 2.0 if counter == "FETCH_SIZE" else 1.0
            print(f"-- {counter} {key}: launches={launches} raw_sum_KB={kb:.0f} per_launch_bytes_raw={kb*1024/launches:.3e} "
                  f"per_launch_bytes_corrected(x{corr:g})={kb*1024*corr/launches:.3e}")
            traffic.setdefault(key, 0.0)
            traffic[key] += kb * 1024 * corr / launches
if traffic.get("k_query_level"):
    j = {"workload": wl, "k_query_level_bytes_per_launch": round(traffic["k_query_level"], 1),
         "k_syncmers_bytes_per_launch": round(traffic.get("k_syncmers", 0.0), 1),
         "source": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + --pmc WRITE_SIZE, separate passes, "
                   + os.path.basename(out)}
    print("-- traffic:", json.dumps(j))        # informative only: bench.py measures its own traffic live (--traffic live)
