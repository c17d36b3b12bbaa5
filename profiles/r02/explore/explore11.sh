#!/bin/bash
# round-2 run 11: sub-batch size sweep on the default workload; the CLI end to end (10 GB FASTQ -> TSV)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore11
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_minimiser.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -2 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling --no-unpruned"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
for sr in 16384 32768 65536 131072; do TAXOR_SUB_READS=$sr run sub$sr; done
for sr in 32768 131072 262144; do TAXOR_SUB_READS=$sr run sub${sr}_1k --reads 1310720 --read-len 1000 --batches 2 --family-size 1; done
python profiles/cli_e2e.py 2000000 5000 > $O/cli_e2e.txt 2>&1
tail -20 $O/cli_e2e.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
