#!/bin/bash
# the new ramp defaults (first sub-batch 1/16, growth 1.6) against the old ones (1/8, 1.25) on the other workloads, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in "" "--family-size 1" "--reads 436906 --read-len 3000" "--reads 1310720 --read-len 1000" "--workload refseq" "--workload refseq --reads 1310720 --read-len 1000"; do
  for cfg in "1.6 16" "1.25 8" "1.6 16" "1.25 8"; do set -- $cfg
    echo -n "[$w] growth $1 first 1/$2: "
    TAXOR_RAMP_GROWTH=$1 TAXOR_STREAM_FIRST_DIV=$2 python profiles/single_call.py --reps 4 $w 2>&1 | grep -E "single call, pageable" | sed 's/GPU total.*(query/(query/' | cut -c1-200
  done
done
