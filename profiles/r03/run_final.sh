#!/bin/bash
# round-3 evidence, final kernels (after the margin / candidate-tally / chunk changes): gpu tests, the default bench line (live PMC passes by request size), rocprofv3 --kernel-trace --stats of the
# same command, the workload table of DESIGN.md section 5, indexes built without syncmers, the single drop-in call, the whole
# CLI chain at GTDB scale, and a differential fuzz of the final kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_final2
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
( time python bench.py ) > $O/bench_gtdb.json 2> $O/bench_gtdb.err
tail -3 $O/bench_gtdb.err
cd /tmp && export TMPDIR=/tmp
Q="--traffic none --no-cpu-baseline --no-dropin --no-unpruned --no-ceiling"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o gtdb -- python3 $R/bench.py --steps 4 --warmup 1 --batches 2 $Q > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 $R/profiles/trace_summary.py $O/stats 3 > $O/trace_summary.txt 2>&1
cat $O/trace_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/gtdb_kernel_stats.csv \;
find $O/stats -type f -size +200k -delete
cd $R
B="--traffic none --no-cpu-baseline"
run() { name=$1; shift; python bench.py $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; }
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
python bench.py --mode kmer > $O/bench_mode_kmer.json 2> $O/bench_mode_kmer.err
python bench.py --mode minimiser > $O/bench_mode_minimiser.json 2> $O/bench_mode_minimiser.err
python profiles/single_call.py --reps 4 2>&1 | grep -E "resident step|single call|plain H2D" > $O/single_call_family.txt
python profiles/single_call.py --reps 4 --family-size 1 2>&1 | grep -E "resident step|single call" > $O/single_call_unrelated.txt
python profiles/single_call.py --reps 4 --reads 1310720 --read-len 1000 2>&1 | grep -E "resident step|single call" > $O/single_call_1kb.txt
cat $O/single_call_*.txt
export TAXOR_E2E_TMP=/dev/shm TAXOR_E2E_RUNS=32,32,16,8
timeout 1500 python profiles/cli_e2e_class.py gtdb 4000000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $O/cli_e2e_gtdb.txt
grep -E "RATE|sustained|identical|index:" $O/cli_e2e_gtdb.txt
timeout 700 python tests/fuzz_parity.py 600 424242 > $O/fuzz_parity.txt 2>&1
tail -2 $O/fuzz_parity.txt
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r03_final2"
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][0])
    except Exception as e:
        print(os.path.basename(f),"NO LINE",e); continue
    r=j["roofline"]; lv=r.get("levels",[])
    print(f"{os.path.basename(f):32s} {j['value']:9.0f} Mbp/s {j['ms_per_step']:7.1f} ms frac {r['frac']:.3f} line128 {r.get('requested_accounting',{}).get('frac_line128',0):.3f} "
          f"root {lv[0]['frac'] if lv else 0:.2f} deeper {lv[1]['row_reads_G_per_s'] if len(lv)>1 else 0:5.1f} Grows/s unpruned {r.get('unpruned',{}).get('frac',0):.3f} "
          f"({r.get('unpruned',{}).get('value_Mbp_s',0):.0f}) vs_dense {r['vs_dense']:.2f} single {j.get('pcie_inclusive',{}).get('value',0):.0f} sustained {(j.get('sustained') or {}).get('value',0):.0f}")
PY
