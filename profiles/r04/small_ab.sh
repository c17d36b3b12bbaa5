#!/bin/bash
# A/B runs of the small-call path under its measurement knobs (each run a process of its own: the knobs are read once)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TAXOR_TUNING=1
run() { echo "== $*"; env "$@" timeout 300 python profiles/small_calls.py --sizes ${SIZES:-256,1024} 2>&1 | grep "reads per call"; }
for cfg in "$@"; do run $cfg; done
