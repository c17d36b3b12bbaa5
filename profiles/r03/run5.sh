#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run5
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --durations=6 -x > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
( time python bench.py ) > $O/bench_gtdb.json 2> $O/bench_gtdb.err
tail -4 $O/bench_gtdb.err
python - <<'PY'
import json,os
j=json.loads([l for l in open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_run5/bench_gtdb.json") if l.startswith("{")][0])
r=j["roofline"]
print("value",j["value"],"ms/step",j["ms_per_step"],"frac",r["frac"],"traffic",r.get("traffic"),"t/req",r.get("traffic_over_requested"),"t/line128",r.get("traffic_over_line128"),"traffic_frac",r.get("traffic_frac_of_peak"))
print("acct",r.get("requested_accounting"))
print("unpruned",r.get("unpruned",{}).get("frac"),"pcie",j.get("pcie_inclusive",{}).get("value"),"sustained",j.get("sustained",{}).get("value"),"cpu",j.get("cpu_baseline",{}).get("value"), j.get("cpu_baseline",{}).get("all_cores"))
print("binding",j.get("host_binding"))
PY
