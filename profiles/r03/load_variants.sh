#!/bin/bash
# which way does a .hixf reach HBM fastest?  class file in tmpfs, tiny query; "[upload]" = the library's upload alone,
# "index:" = file open to resident (HIP start-up and hipMalloc included), then the teardown marks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_load
mkdir -p $O
cd $R
export TAXOR_E2E_TMP=/dev/shm TAXOR_E2E_KEEP=1 TAXOR_E2E_RUNS=${RUNS:-32,32,8}
timeout 2400 python profiles/cli_e2e_class.py ${1:-refseq} ${2:-131072} > $O/e2e_${1:-refseq}.txt 2>&1
grep -E "RATE|sustained|identical|wall|index:|written|memory|refusing" $O/e2e_${1:-refseq}.txt
D=$(grep "^kept:" $O/e2e_${1:-refseq}.txt | cut -d' ' -f2)
[ -d "$D" ] || { tail -5 $O/e2e_${1:-refseq}.txt; exit 1; }
head -n 400 $D/reads.fastq > $D/small.fastq 2>/dev/null
T=$R/taxor_amd/taxor
run() { name=$1; shift; for i in 1 2; do env TAXOR_CLI_TRACE=1 TAXOR_TRACE_UPLOAD=1 "$@" $T search --index-file $D/*.hixf --query-file $D/small.fastq --output-file $D/o.tsv --threads 8 2>&1 | grep -E "upload\]|index:|host index released|output closed" | sed -e 's/\[trace\] *//' | tr '\n' ' '; echo " <- $name"; done; }
run "pread 8 threads x 8 MB (default)" > $O/variants_${1:-refseq}.txt
run "pread 16 x 8" TAXOR_UPLOAD_THREADS=16 >> $O/variants_${1:-refseq}.txt
run "pread 24 x 4" TAXOR_UPLOAD_THREADS=24 TAXOR_UPLOAD_PIECE_MB=4 >> $O/variants_${1:-refseq}.txt
run "pread 16 x 16" TAXOR_UPLOAD_THREADS=16 TAXOR_UPLOAD_PIECE_MB=16 >> $O/variants_${1:-refseq}.txt
run "pread 8 x 32" TAXOR_UPLOAD_PIECE_MB=32 >> $O/variants_${1:-refseq}.txt
run "map, runtime pageable path, no prefault (round 2)" TAXOR_HIXF_UPLOAD_FROM_MAP=1 TAXOR_UPLOAD_PREFAULT=0 >> $O/variants_${1:-refseq}.txt
run "map + 16 prefault threads" TAXOR_HIXF_UPLOAD_FROM_MAP=1 TAXOR_UPLOAD_PREFAULT=16 >> $O/variants_${1:-refseq}.txt
cat $O/variants_${1:-refseq}.txt
rm -rf $D
