#!/bin/bash
# round 4: variants of the CLI's text stage, interleaved and repeated on one box (the spread between runs is +-10 %)
# usage: bash profiles/r04/cli_matrix.sh <out-file-prefix>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_h; mkdir -p $O; cd $R
P=$R/build_ab/taxor_prev
V="32@TAXOR_CLI_FORMATTER_DIV=2@TAXOR_CLI_TURN_SPIN=1;32@TAXOR_CLI_FORMATTER_DIV=2@TAXOR_CLI_TURN_SPIN=0;32@TAXOR_CLI_FORMATTER_DIV=4@TAXOR_CLI_TURN_SPIN=1;32@TAXOR_CLI_FORMATTER_DIV=4@TAXOR_CLI_TURN_SPIN=0;32@TAXOR_CLI_FORMATTER_DIV=8@TAXOR_CLI_TURN_SPIN=0"
[ -x $P ] && V="$V;32@BIN=$P"
RUNS="32;$V;$V;$V;$V"
summ() { python3 - "$1" <<'PY'
import re, sys, collections
cur, acc = "default", collections.defaultdict(list)
first = True
for l in open(sys.argv[1]):
    if l.startswith("run with"): cur = l[9:].strip()
    m = re.match(r"RATE .*search phase (\d+) Mbp/s = ([0-9.]+) x", l)
    if m:
        if first: first = False
        else: acc[cur].append((int(m.group(1)), float(m.group(2))))
        cur = "default"
for k, v in acc.items():
    r = sorted(x for x, _ in v)
    print(f"{k:70s} n={len(v)} Mbp/s: " + " ".join(str(x) for x in r) + f"  median {r[len(r)//2]}  ratio median {sorted(y for _, y in v)[len(v)//2]:.3f}")
PY
}
TAXOR_E2E_READ_LEN=1000 TAXOR_E2E_FORMAT=fasta TAXOR_E2E_TMP=/dev/shm TAXOR_E2E_RUNS="$RUNS" python profiles/cli_e2e_class.py refseq 12000000 > $O/$1_1kb.txt 2>&1
echo "== 12 M x 1 kb"; grep "library sustained" $O/$1_1kb.txt; summ $O/$1_1kb.txt
TAXOR_E2E_FORMAT=fasta TAXOR_E2E_TMP=/dev/shm TAXOR_E2E_RUNS="$RUNS" python profiles/cli_e2e_class.py refseq 4194304 > $O/$1_10kb.txt 2>&1
echo "== 4.2 M x 10 kb"; grep "library sustained" $O/$1_10kb.txt; summ $O/$1_10kb.txt
