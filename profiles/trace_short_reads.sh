R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace1k
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace1k -o t -- python3 $R/bench.py --steps 1 --warmup 0 --traffic none --no-cpu-baseline --no-dropin --reads 1310720 --read-len 1000 > $R/gpurun_out/trace1k.json 2>/dev/null
python3 - <<'PY'
import csv,glob,os,json
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
f=glob.glob(R+"/gpurun_out/trace1k/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_syncmers" in r["Kernel_Name"] or "k_query_level" in r["Kernel_Name"] or "k_s" in r["Kernel_Name"]]
rows=rows[-14:]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    a,b=(int(r["Start_Timestamp"])-t0)/1e6,(int(r["End_Timestamp"])-t0)/1e6
    print(r["Kernel_Name"][:28].ljust(28), f"start {a:9.3f} end {b:9.3f} dur {b-a:7.3f} ms")
d=json.loads(open(R+"/gpurun_out/trace1k.json").read().strip().split("\n")[-1]); print(d["value"], d["roofline"]["requested_GBps"], d["config"]["work_items_per_read"], d["config"]["hashes_per_read"])
PY
