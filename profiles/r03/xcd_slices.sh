#!/bin/bash
# how many slices: TAXOR_QUERY_XCD = 8 (one per XCD), 4, 2 (neighbouring XCDs share one), 0 (one cursor)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() {
  python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --no-dropin --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=j['roofline']
print('  TAXOR_QUERY_XCD=${TAXOR_QUERY_XCD}: value',j['value'],'levels',[(l['level'],l['ms_per_step']) for l in r['levels']])"
}
for w in "" "--workload refseq"; do
  echo "bench.py $w"
  for m in 8 4 2 0 8 4 2; do TAXOR_QUERY_XCD=$m one $w; done
done
