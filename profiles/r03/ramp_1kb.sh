#!/bin/bash
# streamed sub-batch ramp (first sub-batch 1/div of a full one, then x growth) for ONE drop-in call on 1-kb reads
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for g in 1.25 1.6 2 3 8; do for d in 8 16 4; do
  echo -n "growth $g first 1/$d: "
  TAXOR_RAMP_GROWTH=$g TAXOR_STREAM_FIRST_DIV=$d python profiles/single_call.py --reps 4 --reads 1310720 --read-len 1000 2>&1 | grep -E "single call, pageable|single call, page-locked" | sed 's/GPU total.*(query/(query/' | tr '\n' ' ' | cut -c1-260
  echo
done; done
