#!/bin/bash
# round-2 run 10: wave kernel on its own stream (concurrent with the block kernel of the long reads)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore10
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_minimiser.py tests/test_gpu_cli.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run ont --len-mix ont --batches 2
TAXOR_SYNC_WAVE=0 run ont_wave0 --len-mix ont --batches 2
run unrel1k --reads 1310720 --read-len 1000 --batches 2 --family-size 1
run fam1k --reads 1310720 --read-len 1000 --batches 2
run fam10k
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
