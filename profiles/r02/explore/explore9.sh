#!/bin/bash
# round-2 run 9: wave-per-read syncmer kernel for short reads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore9
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_minimiser.py tests/test_gpu_cli.py tests/test_gpu_builder.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for wv in 1 0; do
  export TAXOR_SYNC_WAVE=$wv
  run unrel1k_wave$wv --reads 1310720 --read-len 1000 --batches 2 --family-size 1
  run fam1k_wave$wv --reads 1310720 --read-len 1000 --batches 2
  run fam2k_wave$wv --reads 655360 --read-len 2000 --batches 2
  run ont_wave$wv --len-mix ont --batches 2
  TAXOR_NO_OVERLAP=1 run unrel1k_serial_wave$wv --reads 1310720 --read-len 1000 --batches 2 --family-size 1
done
unset TAXOR_SYNC_WAVE
run fam10k
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
