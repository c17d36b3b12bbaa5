#!/bin/bash
# Round-6 evidence, one gpurun call on the round's last code.  Outputs under gpurun_out/r06/ (copied into profiles/r06/ afterwards).
#   pytest_gpu.log            the GPU suite
#   bench_gtdb.json/.err      the driver's command: headline + legs (incl. exact_fill)
#   gtdb_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command without the legs
#   bench_forced_dist.json    TAXOR_BENCH_FORCE_DIST=1: the N > 1 branches with one rank on RCCL (comm object, strong leg)
#   bench_build.json/.err     bench.py --mode build; build_kernel_stats.csv its kernel trace; build_pmc.txt one TCC counter pass
#   bench_viral.json          the viral-class line (step split syncmers / query)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -q -m gpu --durations=12 > $O/pytest_gpu.log 2>&1
python3 bench.py > $O/bench_gtdb.json 2> $O/bench_gtdb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gtdb_trace -o gtdb -- python3 bench.py --no-layouts --traffic none --no-cpu-baseline --no-dropin > $O/bench_gtdb_traced.json 2> $O/bench_gtdb_traced.err
TAXOR_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-layouts > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
python3 bench.py --mode build --steps 4 --warmup 1 > $O/bench_build.json 2> $O/bench_build.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/build_trace -o build -- python3 bench.py --mode build --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_build_traced.json 2> $O/bench_build_traced.err
rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/build_pmc -o b -- python3 bench.py --mode build --build-children 8 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_build_pmc.json 2> $O/bench_build_pmc.err
python3 bench.py --workload viral --no-layouts --traffic none > $O/bench_viral.json 2> $O/bench_viral.err
tail -3 $O/pytest_gpu.log
ls $O
