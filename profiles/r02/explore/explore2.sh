#!/bin/bash
# round-2 run 2: all gpu tests (incl. the new full-size and multi-rank tests) with durations, then the new default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -5 $O/bench_default.err
cat $O/bench_default.json
