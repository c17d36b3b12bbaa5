#!/bin/bash
# round-2 run 4: after LDS bin info + chunk metadata (query) and chunk metadata + word prefetch (syncmers)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore4
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
B="--traffic none --no-cpu-baseline --no-dropin --no-ceiling"
python bench.py $B > $O/bench_fam10k.json 2> $O/bench_fam10k.err
python bench.py $B --family-size 1 > $O/bench_unrel10k.json 2> $O/bench_unrel10k.err
python bench.py $B --reads 1310720 --read-len 1000 --batches 2 > $O/bench_fam1k.json 2> $O/bench_fam1k.err
python bench.py $B --reads 1310720 --read-len 1000 --batches 2 --family-size 1 > $O/bench_unrel1k.json 2> $O/bench_unrel1k.err
python bench.py $B --reads 436906 --read-len 3000 --batches 2 > $O/bench_fam3k.json 2> $O/bench_fam3k.err
python bench.py $B --workload refseq > $O/bench_refseq.json 2> $O/bench_refseq.err
python bench.py $B --workload viral > $O/bench_viral.json 2> $O/bench_viral.err
python profiles/phase_profile.py > $O/phase_10k.txt 2>&1
python profiles/phase_profile.py --reads 1310720 --read-len 1000 > $O/phase_1k.txt 2>&1
grep -h "^==\|^--" $O/phase_10k.txt $O/phase_1k.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; c=j['config']; print(j['value'], j['ms_per_step'], 'frac', r['frac'], 'vs_dense', r['vs_dense'], 'unpruned', r.get('unpruned',{}).get('frac'), 'tuples/read', c['tuples_per_read'], 'items/read', c['work_items_per_read'], {k:v for k,v in j['stage_ms_per_step'].items() if k!='note'})
")"; done
