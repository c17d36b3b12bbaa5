#!/bin/bash
# round-2 run 6: all-pairs dedup for short reads, adaptive query occupancy; overlap on/off by read length
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore6
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_minimiser.py tests/test_gpu_cli.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for ov in 0 1; do
  export TAXOR_NO_OVERLAP=$ov
  run fam10k_no$ov
  run unrel10k_no$ov --family-size 1
  run unrel1k_no$ov --reads 1310720 --read-len 1000 --batches 2 --family-size 1
  run fam1k_no$ov --reads 1310720 --read-len 1000 --batches 2
  run fam3k_no$ov --reads 436906 --read-len 3000 --batches 2
  run viral_no$ov --workload viral
done
unset TAXOR_NO_OVERLAP
python profiles/phase_profile.py --reads 1310720 --read-len 1000 > $O/phase_1k.txt 2>&1
grep -h "^==\|^--" $O/phase_1k.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
