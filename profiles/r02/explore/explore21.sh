#!/bin/bash
# round-2 run 21: sub-batch size with IXF-grouped work queues (more items per child per launch = more cache reuse?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore21
mkdir -p $O
cd $R
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for sr in 32768 65536 131072; do
  export TAXOR_SUB_READS=$sr
  run fam10k_sub$sr
  run refseq_sub$sr --workload refseq
  run unrel10k_sub$sr --family-size 1
done
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], ' '.join('L%d:%.1fms/%.0fG' % (x['level'], x['ms_per_step'], x['row_reads_G_per_s']) for x in r['levels']), j['stage_ms_per_step']['syncmers'])
")"; done
