#!/bin/bash
# GPU timeline of ONE taxor_gpu_search_batch call of 1024 x 10 kb reads (the reference's chunk size): kernels and copies with their gaps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_small
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o t -- python3 $R/profiles/small_calls.py --sizes 1024 > $O/small_calls.txt 2>&1
grep "reads per call" $O/small_calls.txt
python3 - <<'PY'
import csv, glob, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_small"
f=glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
for mf in glob.glob(O+"/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(mf)):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),"COPY "+r.get("Direction","")))
rows.sort()
def short(n):
    import re
    m=re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else n[:28]
# calls = groups of activity that contain a k_pack_dna4; take one from the single-searcher phase (a third of the way in)
packs=[i for i,r in enumerate(rows) if "k_pack_dna4" in r[2]]
i0=packs[len(packs)//4]
# walk back to the H2D copies that precede this pack (same call), forward to the next pack's first copy
start=i0
while start>0 and rows[i0][0]-rows[start-1][0] < 600_000 and "k_pack_dna4" not in rows[start-1][2]: start-=1
end=packs[len(packs)//4+1]
t0=rows[start][0]
prev_end=t0
for r in rows[start:end]:
    print(f"{(r[0]-t0)/1e3:8.1f} us  +{(r[1]-r[0])/1e3:7.1f} us  gap {max(0,(r[0]-prev_end))/1e3:6.1f}  {short(r[2])}")
    prev_end=max(prev_end,r[1])
print(f"span {(prev_end-t0)/1e3:.1f} us, busy {sum(r[1]-r[0] for r in rows[start:end])/1e3:.1f} us (overlapping streams count twice)")
PY
