#!/bin/bash
# round-2 run 8: 4-way unrolled dense loop now that it fits 128 VGPRs (no register widening accumulators)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore8
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_minimiser.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for u in 2 4; do
  export TAXOR_QUERY_UNROLL=$u
  run fam10k_u$u
  run unrel10k_u$u --family-size 1
  run unrel1k_u$u --reads 1310720 --read-len 1000 --batches 2 --family-size 1
  run fam3k_u$u --reads 436906 --read-len 3000 --batches 2
  run root4096_u$u --root-bins 4096 --steps 4 --warmup 1 --batches 2
done
unset TAXOR_QUERY_UNROLL
python profiles/kmer_mode_bench.py > $O/kmer_w20.txt 2>&1
grep -h "^k=" $O/kmer_w20.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
