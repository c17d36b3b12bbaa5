#!/bin/bash
# round-2 run 3: all gpu tests, default bench, phase profiles at 10 kb / 1 kb, k-mer mode
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1
tail -22 $O/pytest_gpu.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
python bench.py --read-error 0.04 --traffic none --no-cpu-baseline --no-dropin > $O/bench_err04.json 2> $O/bench_err04.err
python bench.py --family-size 1 --traffic none --no-cpu-baseline --no-dropin > $O/bench_unrelated.json 2> $O/bench_unrelated.err
python profiles/phase_profile.py > $O/phase_10k.txt 2>&1
python profiles/phase_profile.py --reads 1310720 --read-len 1000 > $O/phase_1k.txt 2>&1
python profiles/kmer_mode_bench.py > $O/kmer.txt 2>&1
TAXOR_PROFILE_PHASES=1 python profiles/kmer_mode_bench.py > $O/kmer_prof.txt 2>&1
grep -h "^==\|^--\|^k=" $O/phase_10k.txt $O/phase_1k.txt $O/kmer.txt $O/kmer_prof.txt
for f in $O/bench_*.json; do echo "$f: $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], 'vs_dense', r['vs_dense'], 'unpruned', r.get('unpruned',{}).get('frac'), 'tuples/read', c['tuples_per_read'], 'items/read', c['work_items_per_read'], j['stage_ms_per_step'])
")"; done
