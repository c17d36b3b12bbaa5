#!/bin/bash
# work items per cursor atomic (TAXOR_QUERY_CHUNK) on 1-kb reads, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() {
  python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --no-dropin --steps 8 --warmup 2 --reads 1310720 --read-len 1000 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=j['roofline']
print('  TAXOR_QUERY_CHUNK=${TAXOR_QUERY_CHUNK:-default(4)}: value',j['value'],'ms',j['ms_per_step'],'levels',[(l['level'],l['ms_per_step']) for l in r['levels']])"
}
for rep in 1 2; do for c in 4 8 16; do TAXOR_QUERY_CHUNK=$c one; done; done
