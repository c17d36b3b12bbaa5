#!/bin/bash
# two searchers in flight (bench.py `sustained`): is the streamed sub-batch ramp still worth anything?  TAXOR_STREAM_FIRST_DIV=1 = no ramp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in "--reads 1310720 --read-len 1000 --batches 2" "--batches 4" "--workload refseq --reads 1310720 --read-len 1000 --batches 2" "--len-mix ont --batches 2"; do
  for d in 8 1 8 1; do
    echo -n "[$w] first 1/$d: "
    TAXOR_STREAM_FIRST_DIV=$d python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --steps 4 --warmup 1 $w 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('resident',j['value'],'single',j['pcie_inclusive']['value'],'sustained',j['sustained']['value'])"
  done
done
