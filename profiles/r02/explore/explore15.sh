cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r02_explore15; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_multirank.py -m gpu -q -x 2>&1 | tail -3
python bench.py --traffic none --no-cpu-baseline --no-dropin --no-unpruned > $O/bench_fam.json 2> $O/bench_fam.err
python bench.py --traffic none --no-cpu-baseline --no-dropin --no-unpruned --root-bins 64 > $O/bench_root64.json 2> $O/bench_root64.err
python bench.py --traffic none --no-cpu-baseline --no-dropin --no-unpruned --workload refseq > $O/bench_refseq.json 2> $O/bench_refseq.err
for f in $O/bench_*.json; do python3 -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print('$f', j['value'], r['frac']); print(' levels', r['levels']); print(' ceiling', r.get('gather_ceiling'))
"; done
