#!/bin/bash
# GPU timeline of ONE taxor_gpu_search_batch call of N x 10 kb reads (default 1024, the reference's chunk size): kernels and copies
# with their gaps and the hardware queue each ran on.   usage: trace_small_call.sh [reads per call] [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1024}
TAG=${2:-small}
O=$R/gpurun_out/r04_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8    # the profiler brings the runtime up before python can set it (api.hip, runtime_env_once)
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python3 $R/profiles/small_calls.py --sizes $N > $O/small_calls.txt 2>&1
grep "reads per call" $O/small_calls.txt
O=$O python3 - <<'PY'
import csv, glob, os, re
O=os.environ["O"]
f=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
for mf in glob.glob(O+"/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(mf)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction",""),"-"))
rows.sort()
def short(n):
    m=re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:28]
fins=[i for i,r in enumerate(rows) if "k_finalize_small" in r[2] or "k_sort_big" in r[2]]
packs=[i for i,r in enumerate(rows) if "k_pack_dna4" in r[2]]
# a window of the single-searcher phase: from a few rows before a pack kernel a quarter of the way into the run, ROWS rows on
ROWS=int(os.environ.get("ROWS","70"))
i0=packs[len(packs)//4]
start=max(0,i0-3)
end=min(len(rows),start+ROWS)
t0=rows[start][0]
prev_end=t0
for r in rows[start:end]:
    print(f"{(r[0]-t0)/1e3:8.1f} us  +{(r[1]-r[0])/1e3:7.1f} us  gap {max(0,(r[0]-prev_end))/1e3:6.1f}  q{r[3]:>3}  {short(r[2])}")
    prev_end=max(prev_end,r[1])
print(f"span {(prev_end-t0)/1e3:.1f} us, busy {sum(r[1]-r[0] for r in rows[start:end])/1e3:.1f} us (overlapping streams count twice)")
PY
