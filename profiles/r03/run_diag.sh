#!/bin/bash
# diagnostics of the final kernels: phase shares (10-kb and 1-kb reads), SQ wait fractions per level at 1 kb, and the
# host-side marks of one drop-in call on 1-kb reads
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_diag
mkdir -p $O
cd $R
python profiles/phase_profile.py 2>&1 | grep -E "^==|^--" > $O/phase_10k.txt
python profiles/phase_profile.py --reads 1310720 --read-len 1000 2>&1 | grep -E "^==|^--" > $O/phase_1k.txt
cat $O/phase_10k.txt $O/phase_1k.txt
{ echo "== family workload, 1310720 x 1 kb reads: SQ counters per k_query_level launch, averaged per level";
  FAMILY=16 bash profiles/pmc_levels.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; } > $O/pmc_levels_1kb.txt 2>&1
cat $O/pmc_levels_1kb.txt
cd $R
TAXOR_TRACE_BATCH=1 python profiles/single_call.py --reps 3 --reads 1310720 --read-len 1000 > $O/single_call_1kb_trace.txt 2>&1
grep -E "resident step|single call" $O/single_call_1kb_trace.txt
tail -60 $O/single_call_1kb_trace.txt | cut -c1-260
