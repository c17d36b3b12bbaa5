#!/bin/bash
# round-2 run 19/20: work items of the levels below the root grouped by IXF (block-aggregated counting sort between levels),
# plain instead of non-temporal loads there
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore20
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py tests/test_gpu_minimiser.py tests/test_gpu_builder.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
timeout 400 python tests/fuzz_parity.py 180 5000000 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for cfg in "1 0" "0 1" "0 0"; do
  set -- $cfg
  export TAXOR_QUERY_GROUP=$1 TAXOR_QUERY_NT_L1=$2
  run fam10k_g$1_nt$2
  run refseq_g$1_nt$2 --workload refseq
  run viral_g$1_nt$2 --workload viral
  run fam1k_g$1_nt$2 --reads 1310720 --read-len 1000 --batches 2
  run unrel10k_g$1_nt$2 --family-size 1
  run unrel1k_g$1_nt$2 --reads 1310720 --read-len 1000 --batches 2 --family-size 1
  run ont_g$1_nt$2 --len-mix ont --batches 2
done
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], ' '.join('L%d:%.1fms/%.0fG' % (x['level'], x['ms_per_step'], x['row_reads_G_per_s']) for x in r['levels']))
")"; done
