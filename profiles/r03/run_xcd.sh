#!/bin/bash
# after the XCD-sliced queue: gpu suite, fuzz, the default line (live traffic) and the rows the change moves
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_xcd
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
timeout 300 python tests/fuzz_parity.py 240 8800000 > $O/fuzz_parity_xcd.txt 2>&1
tail -1 $O/fuzz_parity_xcd.txt
python bench.py > $O/bench_gtdb.json 2> $O/bench_gtdb.err
B="--traffic none --no-cpu-baseline"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run refseq --workload refseq
run refseq_len1k --workload refseq --reads 1310720 --read-len 1000 --batches 2
run viral_len1k --workload viral --reads 1310720 --read-len 1000 --batches 2
run viral --workload viral
run len1k --reads 1310720 --read-len 1000 --batches 2
run len3k --reads 436906 --read-len 3000 --batches 2
run ont --len-mix ont --batches 2
python profiles/make_table.py $O 2>/dev/null | grep -v "no line"
