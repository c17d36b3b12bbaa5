#!/bin/bash
# instruction counters of k_inflate (one wave per gzip chunk): what a symbol costs.  usage: bash profiles/r04/inflate_pmc.sh [GB]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_m; mkdir -p $O; cd $R
python profiles/r04/gz_single_member.py --gb ${1:-2} --threads 16 --keep --inflate-args "--gpu 0" 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-160
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"; do
  rm -rf $O/pmc; rocprofv3 --pmc $set --output-format csv -d $O/pmc -o p -- $R/taxor_amd/taxor inflate --query-file /dev/shm/taxor_gz/reads.fastq.gz --threads 16 --gpu 0 --chunk-mb 0.5 --batch-chunks 2048 > $O/pmc_out.txt 2>&1
  python3 - $O/pmc <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(float)
for fn in f:
    for r in csv.DictReader(open(fn)):
        if "k_inflate" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: f"{v:.4g}" for k, v in acc.items()})
PY
done
grep "bytes in" $O/pmc_out.txt | cut -c1-120
rm -rf /dev/shm/taxor_gz
