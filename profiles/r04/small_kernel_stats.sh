#!/bin/bash
# average kernel durations of small calls (one piece each when N <= 256) under the small-path knobs: usage small_kernel_stats.sh N "ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 TAXOR_TUNING=1
for cfg in "$@"; do
  O=$R/gpurun_out/r04_kstats; rm -rf $O; mkdir -p $O
  env $cfg rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/profiles/small_calls.py --sizes $N $SMALL_ARGS > $O/out.txt 2>&1
  echo "== $N reads per call, $cfg: $(grep 'reads per call' $O/out.txt | sed 's/^ *//')"
  O=$O python3 - <<'PY'
import csv, glob, os, re
O=os.environ["O"]
f=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
d={}
rows=list(csv.DictReader(open(f)))
# only the search phase: after the last k_build / fill kernel
last=max([i for i,r in enumerate(rows) if "k_build" in r["Kernel_Name"] or "k_fill" in r["Kernel_Name"]]+[0])
for r in rows[last:]:
    m=re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]); n=m.group(1) if m else r["Kernel_Name"][:24]
    d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v); print(f"   {n:28s} n={len(v):6d}  avg {sum(v)/len(v):8.1f} us  median {v2[len(v2)//2]:8.1f}  min {v2[0]:8.1f}")
PY
done
