#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for env in "" "TAXOR_RAMP_GROWTH=1.5" "TAXOR_RAMP_GROWTH=2" "TAXOR_RAMP_GROWTH=2 TAXOR_STREAM_FIRST_DIV=4" "TAXOR_RAMP_GROWTH=3 TAXOR_STREAM_FIRST_DIV=16" "TAXOR_STREAM_FIRST_DIV=16 TAXOR_RAMP_GROWTH=2"; do
  echo "== ${env:-default}"
  env $env python profiles/single_call.py --reps 4 2>&1 | grep -E "resident step|single call"
done
echo "== unrelated genomes (round-1 workload), default"
python profiles/single_call.py --reps 4 --family-size 1 2>&1 | grep -E "resident step|single call"
echo "== 1-kb reads, default"
python profiles/single_call.py --reps 4 --reads 1310720 --read-len 1000 2>&1 | grep -E "resident step|single call"
