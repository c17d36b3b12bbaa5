#!/bin/bash
# re-measures the workloads whose sub-batch size the library now chooses larger (reads shorter than ~8 kb)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_short
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
B="--traffic none --no-cpu-baseline"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run viral --workload viral
run len1k --reads 1310720 --read-len 1000 --batches 2
run len3k --reads 436906 --read-len 3000 --batches 2
run ont --len-mix ont --batches 2
run unrel_len1k --reads 1310720 --read-len 1000 --batches 2 --family-size 1
run refseq_len1k --workload refseq --reads 1310720 --read-len 1000 --batches 2
run viral_len1k --workload viral --reads 1310720 --read-len 1000 --batches 2
python bench.py --mode kmer > $O/bench_mode_kmer.json 2> $O/bench_mode_kmer.err
python bench.py --mode minimiser > $O/bench_mode_minimiser.json 2> $O/bench_mode_minimiser.err
python profiles/single_call.py --reps 4 --reads 1310720 --read-len 1000 2>&1 | grep -E "resident step|single call" > $O/single_call_1kb.txt
cat $O/single_call_1kb.txt
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_short"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(os.path.basename(f),"NO LINE",e); continue
    r=j["roofline"]; lv=r.get("levels",[])
    print(f"{os.path.basename(f):32s} {j['value']:9.0f} Mbp/s {j['ms_per_step']:7.1f} ms frac {r['frac']:.3f} line128 {r.get('requested_accounting',{}).get('frac_line128',0):.3f} "
          f"root {lv[0]['frac'] if lv else 0:.2f} deeper {lv[1]['row_reads_G_per_s'] if len(lv)>1 else 0:5.1f} Grows/s unpruned {r.get('unpruned',{}).get('frac',0):.3f} "
          f"({r.get('unpruned',{}).get('value_Mbp_s',0):.0f}) vs_dense {r['vs_dense']:.2f} single {j.get('pcie_inclusive',{}).get('value',0):.0f} sustained {(j.get('sustained') or {}).get('value',0):.0f}")
PY
