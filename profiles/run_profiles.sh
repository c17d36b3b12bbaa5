#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's roofline object (run on the GPU box via gpurun).
# usage: bash profiles/run_profiles.sh <round-tag> [workload]
# Pass 1: --kernel-trace --stats  (per-kernel average duration; must agree with bench.py's HIP-event time)
# Pass 2: --pmc FETCH_SIZE        (HBM read bytes per launch; separate pass, kernel-trace only)
# Pass 3: --pmc WRITE_SIZE
set -u
TAG=${1:-r01}
WL=${2:-gtdb}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $WL -- python3 $REPO/bench.py --workload $WL --steps 3 --warmup 1 --traffic none --no-cpu-baseline --no-dropin > $OUT/bench_stats.json 2> $OUT/bench_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o $WL -- python3 $REPO/bench.py --workload $WL --steps 1 --warmup 0 --traffic none --no-cpu-baseline --no-dropin > $OUT/bench_pmc_fetch.json 2> $OUT/bench_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o $WL -- python3 $REPO/bench.py --workload $WL --steps 1 --warmup 0 --traffic none --no-cpu-baseline --no-dropin > $OUT/bench_pmc_write.json 2> $OUT/bench_pmc_write.err
python3 $REPO/profiles/summarize.py $OUT $WL > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
