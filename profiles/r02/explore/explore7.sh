#!/bin/bash
# round-2 run 7: occupancy / unroll of the levels below the root (narrow rows) on the family workload
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore7
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_minimiser.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
TAXOR_QUERY_BPC_L1=3 run fam10k_l1b3
run fam10k_l1b4
TAXOR_QUERY_BPC_L1=3 TAXOR_QUERY_UNROLL_L1=4 run fam10k_l1b3_u4
run refseq_l1b4 --workload refseq
TAXOR_QUERY_BPC_L1=3 run refseq_l1b3 --workload refseq
TAXOR_QUERY_BPC_L1=3 TAXOR_QUERY_UNROLL_L1=4 run refseq_l1b3_u4 --workload refseq
run viral_l1b4 --workload viral
TAXOR_QUERY_BPC_L1=3 run viral_l1b3 --workload viral
run ont --len-mix ont --batches 2
python profiles/kmer_mode_bench.py > $O/kmer_w20.txt 2>&1
grep -h "^k=" $O/kmer_w20.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
