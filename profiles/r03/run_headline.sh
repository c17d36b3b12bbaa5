#!/bin/bash
# the default line and its rocprofv3 --kernel-trace --stats companion on the last kernels of the round (XCD-sliced queue included)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_headline
mkdir -p $O
cd $R
( time python bench.py ) > $O/bench_gtdb.json 2> $O/bench_gtdb.err
cd /tmp && export TMPDIR=/tmp
Q="--traffic none --no-cpu-baseline --no-dropin --no-unpruned --no-ceiling"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o gtdb -- python3 $R/bench.py --steps 4 --warmup 1 --batches 2 $Q > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 $R/profiles/trace_summary.py $O/stats 3 > $O/trace_summary.txt 2>&1
head -8 $O/trace_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/gtdb_kernel_stats.csv \;
find $O/stats -type f -size +200k -delete
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_headline"
j=json.loads([l for l in open(O+"/bench_gtdb.json") if l.startswith("{")][0]); r=j["roofline"]
print(j["value"], j["ms_per_step"], "frac", r["frac"], "avg_launch_ms", r["avg_launch_ms"], "line128", r["requested_accounting"]["frac_line128"], "traffic", r["traffic"], r.get("traffic_frac_of_peak"), "unpruned", r["unpruned"]["frac"], "ceiling", r["gather_ceiling"]["root"]["GBps"], "single", j["pcie_inclusive"]["value"], "sust", j["sustained"]["value"], "cpu", j["cpu_baseline"]["value"])
PY
