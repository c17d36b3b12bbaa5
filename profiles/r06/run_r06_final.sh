#!/bin/bash
# Round-6 closing evidence for the builder's last change (32-bit state words up to 2^26 keys per bin), one gpurun call.  Outputs under gpurun_out/r06p/:
# the GPU suite, the build line (default steps), its kernel trace, one TCC counter pass with the per-kernel table.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
python3 bench.py --mode build > $O/bench_build.json 2> $O/bench_build.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/build_trace -o build -- python3 bench.py --mode build --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_build_traced.json 2> $O/bench_build_traced.err
rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/build_pmc -o b -- python3 bench.py --mode build --build-children 8 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_build_pmc.json 2> $O/bench_build_pmc.err
python3 profiles/build_pmc_table.py $O/build_pmc/b_counter_collection.csv 1296384000 > $O/build_pmc_table.txt
find $O/build_trace -name "*kernel_stats.csv" -exec cp {} $O/build_kernel_stats.csv \;
find $O -name "*kernel_trace.csv" -delete
timeout 1200 python3 -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
cat $O/build_pmc_table.txt
