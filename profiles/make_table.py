#!/usr/bin/env python3
"""DESIGN.md section 5's workload table from a directory of bench_*.json lines.  usage: python profiles/make_table.py profiles/r03"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"
ROWS = [("gtdb", "**GTDB-class 113 GB, root 1024, families, e = 0.02**"), ("fam_e04", "same, read error 0.04"),
        ("unrel_e02", "same, unrelated genomes (round-1 workload), e = 0.02"), ("root64", "root **64** bins"), ("root256", "root **256**"),
        ("root4096", "root **4096**"), ("len1k", "1-kb reads"), ("len3k", "3-kb reads"), ("len30k", "30-kb reads"),
        ("ont", "ONT-like length mix (1–100 kb)"), ("unrel_len1k", "1-kb reads, unrelated genomes"),
        ("refseq", "RefSeq-class 9.9 GB (root 512, children 64), families"), ("refseq_len1k", "RefSeq-class, 1-kb reads"),
        ("viral", "viral-class 0.37 GB (root 256, children 64), families"), ("viral_len1k", "viral-class, 1-kb reads"),
        ("mode_kmer", "viral-class footprint built WITHOUT syncmers, every 20-mer (`bench.py --mode kmer`)"),
        ("mode_minimiser", "same, window 32 minimisers (`--mode minimiser`)")]
print("| index (resident) | reads | Mbp/s | ms/step | `frac` (sector64) | line128 frac | root level frac | deeper levels G rows/s | unpruned frac (Mbp/s) | `vs_dense` | single call / sustained Mbp/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name, label in ROWS:
    f = os.path.join(d, f"bench_{name}.json")
    try:
        j = json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(f"| {label} | (no line: {e}) |")
        continue
    r = j["roofline"]
    lv = r.get("levels", [])
    c = j["config"]
    wl = c.get("workload", "")
    import re
    m = re.search(r"(\d+) reads x (\d+) bp", wl)
    if name == "ont":
        reads = f"{c['reads_per_gpu']} reads, mean {round(j['value'] * j['ms_per_step'] * 1e3 / c['reads_per_gpu'] / 100) / 10:.1f} kb"
    elif m:
        n, L = int(m.group(1)), int(m.group(2))
        reads = f"{n} × {L // 1000} kb" if L % 1000 == 0 else f"{n} × {L} bp"
    else:
        m2 = re.search(r"(\d+) reads", wl)
        reads = f"{round(int(m2.group(1)) / 1000)} k reads" if m2 else "?"
    u = r.get("unpruned") or {}
    p = j.get("pcie_inclusive") or {}
    s = j.get("sustained") or {}
    print(f"| {label} | {reads} | {j['value']:,.0f} | {j['ms_per_step']:.1f} | {r['frac']:.3f} | {r.get('requested_accounting', {}).get('frac_line128', 0):.3f} | "
          f"{lv[0]['frac'] if lv else 0:.2f} | {lv[1]['row_reads_G_per_s'] if len(lv) > 1 else 0:.0f} | {u.get('frac', 0):.3f} ({u.get('value_Mbp_s', 0):,.0f}) | "
          f"{r['vs_dense']:.2f} | {p.get('value', 0):,.0f} / {s.get('value', 0):,.0f} |")
