#!/bin/bash
# candidate-unit tally + info-word reuse: parity first (gpu suite, fuzz), then the lines the change is aimed at
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_tally
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 400 python tests/fuzz_parity.py 300 777000 > $O/fuzz_parity.txt 2>&1
tail -2 $O/fuzz_parity.txt
B="--traffic none --no-cpu-baseline"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run gtdb_quick
run len1k --reads 1310720 --read-len 1000 --batches 2
run len3k --reads 436906 --read-len 3000 --batches 2
run unrel_len1k --reads 1310720 --read-len 1000 --batches 2 --family-size 1
run refseq_len1k --workload refseq --reads 1310720 --read-len 1000 --batches 2
run root4096 --root-bins 4096
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_tally"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(os.path.basename(f),"NO LINE",e); continue
    r=j["roofline"]
    print(f"{os.path.basename(f):32s} {j['value']:9.0f} Mbp/s {j['ms_per_step']:7.1f} ms frac {r['frac']:.3f} line128 {r.get('requested_accounting',{}).get('frac_line128',0):.3f}")
PY
