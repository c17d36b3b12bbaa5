#!/bin/bash
# round 5, VERDICT r04 item 4 ("decide by measurement"): does the device inflate's rate keep following the number of waves per batch beyond the
# round-4 point (2048 half-MiB chunks = 9.3 G symbols/s)?  One 4-GB FASTQ member, chunk size halved while the chunks per batch double -- the
# same bytes per batch, 2 / 4 / 8 / 16 waves per SIMD offered.  The kernel holds 14 KB of LDS per wave: 11 waves per CU can be resident.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
T=/dev/shm/taxor_gz_r05
python profiles/r04/gz_single_member.py --gb 4 --threads 16 --tmp $T --keep 2>&1 | grep -E "plain FASTQ|one gzip member|taxor inflate --threads 16"
export TAXOR_TUNING=1 TAXOR_INFLATE_TRACE=1
for cfg in "0.5 2048" "0.25 4096" "0.125 8192" "0.0625 16384"; do
    set -- $cfg
    echo "== --chunk-mb $1 --batch-chunks $2"
    timeout 300 taxor_amd/taxor inflate --query-file $T/reads.fastq.gz --threads 16 --gpu 0 --chunk-mb $1 --batch-chunks $2 2>&1 | grep -E "k_inflate|bytes in|device:|worker seconds" | head -8
done
rm -rf $T
