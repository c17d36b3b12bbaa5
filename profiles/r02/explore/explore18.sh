#!/bin/bash
# round-2 run 18: ramp of the streamed sub-batch sizes (first = 1/div of a full sub-batch, then x growth)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore18
mkdir -p $O
cd $R
for cfg in "8 1.25" "4 1.25" "4 1.33" "2 1.33" "4 1.5" "2 1.5" "1 1.25"; do
  set -- $cfg
  export TAXOR_STREAM_FIRST_DIV=$1 TAXOR_RAMP_GROWTH=$2
  for fam in 16 1; do
    python bench.py --traffic none --no-cpu-baseline --no-ceiling --no-unpruned --family-size $fam --steps 4 --warmup 1 --batches 4 --sustained-reads 4000000 > $O/b.json 2> $O/b.err
    echo "div $1 growth $2 family $fam: $(python3 -c "
import json
for l in open('$O/b.json'):
    if l.startswith('{'):
        j=json.loads(l); print('resident', j['value'], 'single call', j['pcie_inclusive']['value'], j['pcie_inclusive']['seconds'], 'sustained', j['sustained']['value'])
")"
  done
done
