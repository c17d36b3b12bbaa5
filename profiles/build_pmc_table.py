#!/usr/bin/env python3
"""Per-kernel table of one rocprofv3 counter pass over a build (profiles/rNN/build_pmc.txt):

  rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
            -d DIR -o b -- python3 bench.py --mode build --build-children 8 --steps 1 --warmup 0 --no-cpu-baseline
  python profiles/build_pmc_table.py DIR/b_counter_collection.csv <keys per level>

keys per level = leaf keys of the hierarchy (children x bins x keys per bin): every level inserts that many keys, the leaf level on
32-bit state words; 54 M-key root bins were on 64-bit words when `r06/build_pmc.txt` was taken and are on 32-bit words since
(degree field of 6 bits, builder.hip `dbits`)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(k_[a-z_]+)(<[^>]*>)?", name)
    if not m:
        return None
    t = m.group(2) or ""
    t = "<u64>" if "long" in t else "<u32>" if "int" in t else ""
    return m.group(1) + t


def main():
    path, keys = sys.argv[1], float(sys.argv[2])
    ctr = defaultdict(lambda: defaultdict(float))
    ns = defaultdict(float)
    launches = defaultdict(int)
    seen = set()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k is None or k.startswith("k_query") or k.startswith("k_synth"):
                continue
            ctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
            d = r["Dispatch_Id"]
            if d not in seen:
                seen.add(d)
                ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                launches[k] += 1
    print(f"{'kernel':20s} {'launches':>8s} {'ms':>9s} {'TCC_ATOMIC':>12s} {'per key':>8s} {'G atomics/s':>12s} {'of them to EA':>14s} {'TCC_REQ per key':>16s}")
    for k in sorted(ctr):
        a, ea, req = ctr[k].get("TCC_ATOMIC_sum", 0.0), ctr[k].get("TCC_EA0_ATOMIC_sum", 0.0), ctr[k].get("TCC_REQ_sum", 0.0)
        s = ns[k] * 1e-9
        print(f"{k:20s} {launches[k]:8d} {s * 1e3:9.2f} {a:12.4e} {a / keys:8.3f} {a / s / 1e9 if s else 0:12.2f} {ea / a if a else 0:14.3f} {req / keys:16.2f}")


if __name__ == "__main__":
    main()
