#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (new: communicator, 8 ranks, CLI syntax), then the FETCH_SIZE calibration
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_run1
mkdir -p $O
cd $R
nproc > $O/host.txt; free -g >> $O/host.txt; df -h /dev/shm /tmp >> $O/host.txt; numactl -H >> $O/host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -x > $O/pytest_gpu.log 2>&1
tail -25 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o -E "TCC_EA0?_RDREQ[A-Za-z0-9_]*|TCC_[A-Z0-9_]*128B[A-Za-z0-9_]*|TCC_BUBBLE[A-Za-z0-9_]*|FETCH_SIZE|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*" | sort -u > $O/counters_avail.txt
cat $O/counters_avail.txt | tr '\n' ' '
cd $R
timeout 1200 python3 profiles/calibrate_fetch.py --out $O > $O/calibrate.log 2>&1
tail -20 $O/calibrate.log
