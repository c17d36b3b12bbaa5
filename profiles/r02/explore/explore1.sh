#!/bin/bash
# round-2 exploration 1: box facts, gpu tests, baseline, occupancy sweep at 1 kb / 10 kb reads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_explore1
mkdir -p $O
cd $R
( free -g; nproc; rocm-smi --showmeminfo vram 2>/dev/null | head -8 ) > $O/box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python bench.py --steps 10 --warmup 3 > $O/bench_gtdb.json 2> $O/bench_gtdb.err
for bpc in 3 4 5 6; do
  TAXOR_QUERY_BPC=$bpc python bench.py --steps 5 --warmup 2 --reads 1310720 --read-len 1000 --no-cpu-baseline --no-dropin > $O/bench_1k_bpc$bpc.json 2> $O/bench_1k_bpc$bpc.err
  TAXOR_QUERY_BPC=$bpc python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin > $O/bench_10k_bpc$bpc.json 2> $O/bench_10k_bpc$bpc.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1k -o t -- python3 $R/bench.py --traffic none --steps 2 --warmup 1 --reads 1310720 --read-len 1000 --no-cpu-baseline --no-dropin > $O/bench_stats1k.json 2> $O/bench_stats1k.err
find $O/stats1k -name "*kernel_stats.csv" -exec head -12 {} \; > $O/stats1k_summary.txt
find $O/stats1k -type f ! -name "*kernel_stats.csv" -delete
for f in $O/bench_*.json; do echo "$f: $(python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; print(j['value'], j['ms_per_step'], 'req', r['requested_GBps'], 'avg_ms', r['avg_launch_ms'], j['stage_ms_last_step'])
")"; done
