#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for m in 4.5 3.5 2.5 1.5; do
  TAXOR_QUERY_MARGIN=$m python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --no-dropin --steps 8 --warmup 2 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=j['roofline']
print('margin $m: value',j['value'],'ms',j['ms_per_step'],'line128/launch GB',round(r['requested_accounting']['line128_bytes_per_launch']/1e9,3),'levels',[(l['level'],l['ms_per_step']) for l in r['levels']])"
done
