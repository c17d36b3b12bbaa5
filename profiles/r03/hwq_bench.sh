#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for q in 4 16 32 64; do
  GPU_MAX_HW_QUEUES=$q python bench.py --traffic none --no-cpu-baseline --no-unpruned --no-ceiling --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('queues $q: value',j['value'],'single',j['pcie_inclusive']['value'],'sustained',j['sustained']['value'])"
done
