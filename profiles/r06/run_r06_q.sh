cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
python3 -m pytest tests/test_gpu_builder.py tests/test_gpu_build_fullsize.py tests/test_gpu_bench_multirank.py -q -x -k "build or Build or keys" 2>&1 | tail -4
python3 bench.py --mode build > $O/bench_build.json 2> $O/bench_build.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/build_trace -o build -- python3 bench.py --mode build --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_build_traced.json 2> $O/bench_build_traced.err
find $O/build_trace -name "*kernel_stats.csv" -exec cp {} $O/build_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
timeout 200 python3 tests/fuzz_parity.py 120 7700000 | tail -1
python3 -c "
import json,csv
d=json.loads(open('$O/bench_build.json').read().strip().splitlines()[-1]);print(d['value'],d['value_median_step'],d['stage_s_per_step']['peel'],d['stage_s_per_step']['assign_verify'],d['stage_s_per_step']['unions'])
rows=sorted(csv.DictReader(open('$O/build_kernel_stats.csv')),key=lambda r:-float(r['MaxNs']))
for r in rows[:5]: print(round(float(r['MaxNs'])/1e6,2),r['Name'][:60])"
