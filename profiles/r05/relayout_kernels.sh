#!/bin/bash
# round 5: per-kernel time of the device re-layout (relayout.hip) -- `taxor search` on a 8-GB index file per foreign layout under
# rocprofv3 --kernel-trace --stats; the kernels' rate = fingerprint bytes / summed kernel time (read + write = twice that in HBM traffic).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_final/relayout_stats
mkdir -p $O
cd $R
RELAYOUT_KEEP=/dev/shm/taxor_relayout_r05 RELAYOUT_CODES=0x100,0x001,0x002 python profiles/relayout_rates.py 8 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
for spec in interleaved,padded,position-major bin-major,padded,segment-major bit-sliced,segment-major; do
    f=/dev/shm/taxor_relayout_r05/$(echo $spec | tr ',' '_').hixf
    d=$O/$(echo $spec | tr ',' '_')
    TAXOR_TUNING=1 TAXOR_CLI_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o run -- $R/taxor_amd/taxor search --index-file $f --query-file /dev/shm/taxor_relayout_r05/reads.fq --output-file /dev/shm/taxor_relayout_r05/out.tsv --percentage 0.02 --ixf-layout $spec > /dev/null 2> $d.err
    echo "== $spec"
    find $d -name "*kernel_stats.csv" -exec grep -E "Name|k_rows_repitch|k_bin_major|k_bit_sliced" {} \;
done
rm -rf /dev/shm/taxor_relayout_r05
find $O -type f -size +200k -delete
