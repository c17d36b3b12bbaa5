#!/bin/bash
# round-2 run 22: plain vs non-temporal loads at the root for indexes whose root fits the memory-side cache
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore22
mkdir -p $O
cd $R
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for nt in 1 0; do
  export TAXOR_QUERY_NT=$nt
  run viral_nt$nt --workload viral
  run viral1k_nt$nt --workload viral --reads 1310720 --read-len 1000 --batches 2
  run refseq_nt$nt --workload refseq
  run fam10k_nt$nt
done
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], ' '.join('L%d:%.1fms/%.2f' % (x['level'], x['ms_per_step'], x['frac']) for x in r['levels']))
")"; done
