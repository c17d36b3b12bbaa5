#!/bin/bash
# round 5: the GPU suite, the default line (driver's command) with its `layouts` legs, its rocprofv3 --kernel-trace --stats companion, the
# N > 1 code on one rank, a differential fuzz that now draws fingerprint layouts, and the upload rate per layout at 64 GB -- one gpurun
# call on the round's code.   usage: bash profiles/r05/run_headline.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_gtdb.json 2> $O/bench_gtdb.err; tail -12 $O/bench_gtdb.err
cd /tmp && export TMPDIR=/tmp
Q="--traffic none --no-cpu-baseline --no-dropin --no-unpruned --no-ceiling --no-e04 --no-layouts"
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o gtdb -- python3 $R/bench.py --steps 4 --warmup 1 --batches 2 $Q > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 $R/profiles/trace_summary.py $O/stats 3 > $O/trace_summary.txt 2>&1
head -8 $O/trace_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/gtdb_kernel_stats.csv \;
find $O/stats -type f -size +200k -delete
cd $R
TAXOR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --traffic none --no-cpu-baseline --no-unpruned --no-ceiling --no-e04 > $O/bench_forced_dist.json 2> $O/bench_forced_dist.err
timeout 700 python tests/fuzz_parity.py ${FUZZ_SECONDS:-600} 70000 > $O/fuzz_parity.txt 2>&1; tail -2 $O/fuzz_parity.txt
RELAYOUT_CODES=0,1 timeout 900 python profiles/relayout_rates.py 64 > $O/relayout_64gb.txt 2>&1; tail -6 $O/relayout_64gb.txt
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r05_final"
j=json.loads([l for l in open(O+"/bench_gtdb.json") if l.startswith("{")][0]); r=j["roofline"]
print("value", j["value"], "ms/step", j["ms_per_step"], "host_fed", j.get("value_host_fed"), "e04", j.get("value_e04"), j.get("value_e04_host_fed"), "frac", r["frac"], "contract", r.get("contract_frac"), "moved", r.get("moved_frac"),
      "avg_launch_ms", r["avg_launch_ms"], "traffic", r["traffic"], "ceiling", r["gather_ceiling"]["root"]["GBps"], "single", j["pcie_inclusive"]["value"], "cpu", j["cpu_baseline"]["value"])
for l in j.get("layouts", []): print("  layouts:", l["layout"], l["value"], l["frac"], l["root_bins"], l["child_bins"], l["n_ixf"], l["work_items_per_read"], l["tuples_per_read"])
f=json.loads([l for l in open(O+"/bench_forced_dist.json") if l.startswith("{")][0])
print("forced dist (1 rank, nccl): value", f["value"], "host_fed", f.get("value_host_fed"), "host_fed_scaling", f.get("host_fed_scaling"))
PY
