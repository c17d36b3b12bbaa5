#!/bin/bash
# A/B on one box: TAXOR_QUERY_TALLY bit 0 = the tally walks every bin (as before), bit 1 = bin info fetched per item (as before)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() {
  python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --no-dropin --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=j['roofline']
print('  TAXOR_QUERY_TALLY=${TAXOR_QUERY_TALLY:-0}: value',j['value'],'ms',j['ms_per_step'],'levels',[(l['level'],l['ms_per_step']) for l in r['levels']])"
}
for w in "--reads 1310720 --read-len 1000" "" "--workload refseq --reads 1310720 --read-len 1000" "--workload viral --reads 1310720 --read-len 1000"; do
  echo "bench.py $w"
  for rep in 1 2; do for m in 0 3 1 2; do TAXOR_QUERY_TALLY=$m one $w; done; done
done
