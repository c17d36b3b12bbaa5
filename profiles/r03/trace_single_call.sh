#!/bin/bash
# kernel timeline of one streamed single call (page-locked input): where does the query stream idle?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_single
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python3 $R/profiles/single_call.py --reps 1 > $O/single_call.txt 2>&1
grep -E "resident|single call|plain" $O/single_call.txt
python3 - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_single"
f=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
for mf in glob.glob(O+"/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(mf)):
        if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>20000:
            rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")+" "+r.get("Name","")[:20]))
rows.sort()
def short(n):
    for k in ("k_query_level","k_syncmers_wave","k_syncmers","k_pack_dna4","k_queue_hist","k_queue_scan","k_queue_scatter","k_scan_offsets","k_scatter_hits","k_sort_small","k_sort_big","k_fill_random","k_build","k_gather"):
        if k in n: return k
    return n[:30]
# the last call = everything after the last k_pack_dna4 burst begins: find the last group of launches separated by > 20 ms idle
groups=[[rows[0]]]
for r in rows[1:]:
    if r[0]-max(x[1] for x in groups[-1])>5_000_000: groups.append([r])
    else: groups[-1].append(r)
cand=[g for g in groups if any("k_pack_dna4" in x[2] for x in g)]
g=cand[-1]
t0=g[0][0]
with open(O+"/timeline.txt","w") as out:
    for s,e,n in g:
        out.write(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  {short(n)}\n")
    q=[(s,e) for s,e,n in g if "k_query_level" in n]
    busy=sum(e-s for s,e in q)
    out.write(f"query launches {len(q)}, busy {busy/1e6:.2f} ms, span {(q[-1][1]-q[0][0])/1e6:.2f} ms, first query starts at {(q[0][0]-t0)/1e6:.2f} ms, call span {(g[-1][1]-t0)/1e6:.2f} ms\n")
    gaps=[(q[i+1][0]-q[i][1])/1e3 for i in range(len(q)-1)]
    out.write("gaps between consecutive query launches (us): "+" ".join(f"{x:.0f}" for x in gaps)+"\n")
print(open(O+"/timeline.txt").read()[-1500:])
PY
find $O/trace -type f -size +2000k -delete
