#!/bin/bash
# Round-6 evidence, one gpurun call: k_syncmers alone (rate + one SQ counter pass), the builder's kernel trace and bench line,
# the default bench line (headline + legs incl. exact_fill), the forced one-rank distributed line.  Outputs under gpurun_out/r06/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
python3 profiles/syncmer_alone.py 131072 5000 10 > $O/syncmer_alone_5kb.txt 2>&1
python3 profiles/syncmer_alone.py 131072 10000 10 > $O/syncmer_alone_10kb.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $O/pmc_sync -o t -- python3 profiles/syncmer_alone.py 131072 5000 3 > $O/pmc_sync.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/build_trace -o build -- python3 bench.py --mode build --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_build_traced.json 2> $O/bench_build_traced.err
python3 bench.py --mode build --steps 3 --warmup 1 > $O/bench_build.json 2> $O/bench_build.err
python3 bench.py > $O/bench_gtdb.json 2> $O/bench_gtdb.err
TAXOR_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --no-layouts > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
python3 bench.py --workload viral --no-layouts --traffic none > $O/bench_viral.json 2> $O/bench_viral.err
ls -la $O
