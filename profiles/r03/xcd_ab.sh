#!/bin/bash
# XCD-sliced work queue below the root (TAXOR_QUERY_XCD=1: block b starts in slice b % 8 of the IXF-grouped queue) against one cursor (=0), same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() {
  python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --no-dropin --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=j['roofline']
print('  TAXOR_QUERY_XCD=${TAXOR_QUERY_XCD}: value',j['value'],'ms',j['ms_per_step'],'levels',[(l['level'],l['ms_per_step'],l['row_reads_G_per_s']) for l in r['levels']])"
}
for w in "" "--workload refseq" "--workload viral" "--workload refseq --reads 1310720 --read-len 1000 --batches 2" "--workload viral --reads 1310720 --read-len 1000 --batches 2" "--reads 1310720 --read-len 1000 --batches 2"; do
  echo "bench.py $w"
  for m in 1 0 1 0; do TAXOR_QUERY_XCD=$m one $w; done
done
