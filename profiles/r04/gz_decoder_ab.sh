#!/bin/bash
# builds of the host deflate decoder side by side on one box: usage gz_decoder_ab.sh GB binary...   (interleaved, three rounds, 16 threads)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
GB=$1; shift
python profiles/r04/gz_single_member.py --gb $GB --threads 16 --keep 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-200
export LD_LIBRARY_PATH=$R/taxor_amd
for round in 1 2 3; do
  for b in "$@"; do
    echo -n "$(basename $b): "; $b inflate --query-file /dev/shm/taxor_gz/reads.fastq.gz --threads 16 | tr "\n" " " | sed 's/1 member.*verified//' | cut -c1-230; echo
  done
done
rm -rf /dev/shm/taxor_gz
