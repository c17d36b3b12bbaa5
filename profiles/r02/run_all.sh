#!/bin/bash
# round-2 evidence: gpu tests, the default bench line (with its live PMC passes), rocprofv3 --kernel-trace --stats of the
# same command, per-level / clean-syncmer figures from the trace, and the workload tables of DESIGN.md section 5.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1
tail -14 $O/pytest_gpu.log
( time python bench.py ) > $O/bench_gtdb.json 2> $O/bench_gtdb.err
tail -3 $O/bench_gtdb.err
cd /tmp && export TMPDIR=/tmp
Q="--traffic none --no-cpu-baseline --no-dropin --no-unpruned --no-ceiling"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o gtdb -- python3 $R/bench.py --steps 4 --warmup 1 --batches 2 $Q > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o gtdb -- python3 $R/bench.py --steps 1 --warmup 0 --batches 1 $Q > /dev/null 2>&1
python3 $R/profiles/trace_summary.py $O/stats 3 $O/pmc_fetch > $O/trace_summary.txt 2>&1
cat $O/trace_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/gtdb_kernel_stats.csv \;
find $O/stats $O/pmc_fetch -type f -size +200k -delete
cd $R
B="--traffic none --no-cpu-baseline --no-dropin"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run fam_e02                       # default workload again (table row)
run fam_e04 --read-error 0.04
run unrel_e02 --family-size 1
run unrel_e04 --family-size 1 --read-error 0.04
for rb in 64 256 4096; do run root$rb --root-bins $rb; done
run refseq --workload refseq
run viral --workload viral
run len1k --reads 1310720 --read-len 1000 --batches 2
run len3k --reads 436906 --read-len 3000 --batches 2
run len30k --reads 43690 --read-len 30000 --batches 2
run ont --len-mix ont --batches 2
run unrel_len1k --reads 1310720 --read-len 1000 --batches 2 --family-size 1
run refseq_len1k --workload refseq --reads 1310720 --read-len 1000 --batches 2
run viral_len1k --workload viral --reads 1310720 --read-len 1000 --batches 2
python profiles/kmer_mode_bench.py > $O/kmer_w20.txt 2>&1
python profiles/kmer_mode_bench.py 20 32 > $O/kmer_w32.txt 2>&1
python profiles/phase_profile.py > $O/phase_10k.txt 2>&1
python profiles/phase_profile.py --reads 1310720 --read-len 1000 > $O/phase_1k.txt 2>&1
grep -h "^==\|^--\|^k=\|^algorithmic" $O/phase_10k.txt $O/phase_1k.txt $O/kmer_w20.txt $O/kmer_w32.txt
TAXOR_NO_OVERLAP=1 run serial_10k
TAXOR_NO_OVERLAP=1 run serial_unrel_len1k --reads 1310720 --read-len 1000 --batches 2 --family-size 1
timeout 900 python tests/fuzz_parity.py 240 1500000 > $O/fuzz_parity_c.txt 2>&1
tail -2 $O/fuzz_parity_c.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], 'alg', r['algorithmic_GBps'], 'vs_dense', r['vs_dense'], 'unpruned', r.get('unpruned',{}).get('frac'), r.get('unpruned',{}).get('value_Mbp_s'), 'tuples/read', c['tuples_per_read'], 'items/read', c['work_items_per_read'], 'hits', c['reads_with_hits_last_step'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
