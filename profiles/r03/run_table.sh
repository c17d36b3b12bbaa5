#!/bin/bash
# DESIGN.md section 5's workload table on the last kernels of the round, one box, one call
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_table
mkdir -p $O
cd $R
B="--traffic none --no-cpu-baseline"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
run gtdb
run fam_e04 --read-error 0.04
run unrel_e02 --family-size 1
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
timeout 300 python bench.py --mode kmer --no-cpu-baseline > $O/bench_mode_kmer.json 2> $O/bench_mode_kmer.err
timeout 300 python bench.py --mode minimiser --no-cpu-baseline > $O/bench_mode_minimiser.json 2> $O/bench_mode_minimiser.err
python profiles/make_table.py $O
