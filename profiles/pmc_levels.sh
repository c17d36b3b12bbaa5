#!/bin/bash
# Per-level PMC averages of k_query_level for the 1-kb-read workload (diagnosis of the small-item launches).
# usage: bash profiles/pmc_levels.sh "<counters for pass 1>" "<counters for pass 2>" ...     (PMC passes carry --kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for C in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_lv_$i
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lv_$i -o t -- python3 $R/bench.py --steps 1 --warmup 0 --batches 1 --traffic none --no-cpu-baseline --no-dropin --no-unpruned --no-ceiling --family-size ${FAMILY:-1} --reads 1310720 --read-len 1000 > /dev/null 2>&1
  python3 - $R/gpurun_out/pmc_lv_$i <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_query_level" in r["Kernel_Name"]]
by_disp = collections.OrderedDict()
for r in rows:
    by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    by_disp[int(r["Dispatch_Id"])]["_small"] = "Li64ELi256" in r["Kernel_Name"] or "64, 256" in r["Kernel_Name"]
disp = [by_disp[k] for k in sorted(by_disp)]
n = len(disp) // 2                      # set-up run + timed step: keep the timed step's launches
disp = disp[-n:]
for lvl in range(3):
    sel = disp[lvl::3]
    if not sel:
        continue
    names = sorted(k for k in sel[0] if not k.startswith("_"))
    print("level", lvl, "launches", len(sel), "single-wave blocks" if sel[0]["_small"] else "256-thread blocks",
          " ".join(f"{n}={sum(x[n] for x in sel)/len(sel):.4g}" for n in names))
    if "SQ_WAIT_ANY" in sel[0] and "SQ_WAVE_CYCLES" in sel[0]:
        print("   waves waiting: %.1f %% of their cycles" % (100 * sum(x["SQ_WAIT_ANY"] for x in sel) / sum(x["SQ_WAVE_CYCLES"] for x in sel)))
    if "SQ_ACTIVE_INST_VALU" in sel[0] and "SQ_WAVE_CYCLES" in sel[0]:
        print("   VALU issue: %.1f %% of wave cycles" % (100 * sum(x["SQ_ACTIVE_INST_VALU"] for x in sel) / sum(x["SQ_WAVE_CYCLES"] for x in sel)))
PY
done
