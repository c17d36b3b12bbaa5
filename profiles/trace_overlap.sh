#!/bin/bash
# kernel-trace of one bench step: prints start/end (ms) of k_syncmers (S) and k_query_level (Q) launches to show
# how the two streams overlap.  usage: bash profiles/trace_overlap.sh [workload]
WL=${1:-gtdb}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_overlap
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_overlap -o t -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --traffic none --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
f=glob.glob(R+"/gpurun_out/trace_overlap/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_syncmers" in r["Kernel_Name"] or "k_query_level" in r["Kernel_Name"]]
rows=rows[-16:]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    a,b=(int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-t0)/1e6
    print(("S" if "sync" in r["Kernel_Name"] else "Q"), "queue", r.get("Queue_Id"), f"start {a:9.3f} end {b:9.3f} dur {b-a:7.3f} ms")
PY
