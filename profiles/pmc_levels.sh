#!/bin/bash
# Per-level PMC averages of k_query_level for a short-read workload (diagnosis of the small-item launches).
# usage: bash profiles/pmc_levels.sh "<counters for pass 1>" "<counters for pass 2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_lv_$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lv_$i -o t -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dropin --reads 1310720 --read-len 1000 > /dev/null 2>&1
  python3 - $R/gpurun_out/pmc_lv_$i <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_query_level" in r["Kernel_Name"]]
by_disp = collections.OrderedDict()
for r in rows:
    by_disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(by_disp.values())
disp = disp[-120:]                      # the timed step: 40 sub-batches x 3 levels
for lvl in range(3):
    sel = disp[lvl::3]
    names = sorted(sel[0])
    print("level", lvl, "launches", len(sel), " ".join(f"{n}={sum(x[n] for x in sel)/len(sel):.4g}" for n in names))
PY
done
