#!/bin/bash
# round-2 run 16: single-wave query blocks (64 threads, 256 probe slots, 16 per CU) for levels of narrow IXFs under short reads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore16
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_minimiser.py tests/test_gpu_builder.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 300 python tests/fuzz_parity.py 120 1200000 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for sm in 1 0; do
  export TAXOR_QUERY_SMALL=$sm
  run fam1k_small$sm --reads 1310720 --read-len 1000 --batches 2
  run unrel1k_small$sm --reads 1310720 --read-len 1000 --batches 2 --family-size 1
  run fam2k_small$sm --reads 655360 --read-len 2000 --batches 2
  run refseq1k_small$sm --workload refseq --reads 1310720 --read-len 1000 --batches 2
  run viral1k_small$sm --workload viral --reads 1310720 --read-len 1000 --batches 2
done
unset TAXOR_QUERY_SMALL
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], ' '.join('L%d:%.1fms/%.0fG' % (x['level'], x['ms_per_step'], x['row_reads_G_per_s']) for x in r['levels']))
")"; done
