#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python profiles/single_call.py --reps 4 2>&1 | grep -E "resident step|single call"
done
