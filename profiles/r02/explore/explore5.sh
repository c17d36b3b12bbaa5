#!/bin/bash
# round-2 run 5: sparse-phase unroll; occupancy / unroll knobs for short reads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore5
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run fam10k
run unrel10k --family-size 1
for cfg in "2 3" "2 4" "4 3"; do
  set -- $cfg
  export TAXOR_QUERY_UNROLL=$1 TAXOR_QUERY_BPC=$2
  run unrel1k_u$1_b$2 --reads 1310720 --read-len 1000 --batches 2 --family-size 1
  run fam1k_u$1_b$2 --reads 1310720 --read-len 1000 --batches 2
  run fam3k_u$1_b$2 --reads 436906 --read-len 3000 --batches 2
done
unset TAXOR_QUERY_UNROLL TAXOR_QUERY_BPC
TAXOR_QUERY_BPC=4 run fam10k_b4
TAXOR_QUERY_UNROLL=4 run fam10k_u4
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
