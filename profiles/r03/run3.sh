#!/bin/bash
# round 3: new CLI pipeline -- functional tests, then the drop-in chain at scale (RefSeq-class 1.3 M reads, GTDB-class 10 M reads)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run3
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_cli.py tests/test_gpu_comm.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_cli.log 2>&1
tail -5 $O/pytest_cli.log
export TAXOR_E2E_TMP=/dev/shm
timeout 900 python profiles/cli_e2e_class.py refseq 1310720 > $O/cli_e2e_refseq.txt 2>&1
grep -E "RATE|sustained|identical|wall|index:" $O/cli_e2e_refseq.txt
timeout 2400 python profiles/cli_e2e_class.py gtdb 10000000 > $O/cli_e2e_gtdb.txt 2>&1
grep -E "RATE|sustained|identical|wall|index:|written" $O/cli_e2e_gtdb.txt
