#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for sr in 0 262144 524288 1048576; do
  env $( [ $sr != 0 ] && echo TAXOR_SUB_READS=$sr ) python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --no-dropin --steps 6 --warmup 2 --reads 1310720 --read-len 1000 --batches 2 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=j['roofline']
print('sub_reads $sr: value',j['value'],'ms',j['ms_per_step'],'stage',{k:v for k,v in j['stage_ms_per_step'].items() if k!='note'},'levels',[(l['level'],l['ms_per_step']) for l in r['levels']])"
done
