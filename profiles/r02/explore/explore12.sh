cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r02_explore12
python profiles/r02/explore/two_searchers.py > gpurun_out/r02_explore12/two_fam.txt 2>&1
python profiles/r02/explore/two_searchers.py --family-size 1 > gpurun_out/r02_explore12/two_unrel.txt 2>&1
TAXOR_QUERY_BPC=2 TAXOR_QUERY_BPC_L1=2 python profiles/r02/explore/two_searchers.py > gpurun_out/r02_explore12/two_fam_bpc2.txt 2>&1
grep -h "pipeline\|concurrent" gpurun_out/r02_explore12/*.txt
