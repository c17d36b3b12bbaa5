#!/usr/bin/env python3
"""Upload rate of a .hixf per fingerprint layout (taxor_amd/csrc/relayout.hip vs the plain upload of the search layout).
One index of `gb` GB (a 4096-bin root and 128-bin children, random fingerprints) is written to tmpfs under every layout with the
library's writer, then `taxor search` loads each file for a handful of reads with TAXOR_TRACE_UPLOAD=1: its "[upload]" line is
the rate from first read() to resident, PCIe and the device transposition included.  The TSVs of all layouts must be identical.
usage: python profiles/relayout_rates.py [gb=4] > profiles/r05/relayout.txt"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taxor_amd import _lib  # noqa: E402
from taxor_amd.hixf_file import default_schema, describe_layout, store_hixf  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
rng = np.random.default_rng(5)
root_bins, child_bins, n_children = 4096, 128, 64
root_seg = int(gb * 0.5e9 / (3 * root_bins))
child_seg = int(gb * 0.5e9 / n_children / (3 * child_bins))
ixfs = []
nx = np.zeros(root_bins, np.int64)
fn = np.arange(root_bins, dtype=np.int64)
for c in range(n_children):
    nx[c * 8] = c + 1
    fn[c * 8] = -1
ub = root_bins


def rand_bytes(n):
    return np.frombuffer(rng.bytes(n), dtype=np.uint8)


ixfs.append(dict(bins=root_bins, stride=root_bins, seg_len=root_seg, seed=11, data=rand_bytes(3 * root_seg * root_bins), next_ixf=nx, fname_idx=fn))
for c in range(n_children):
    ixfs.append(dict(bins=child_bins, stride=child_bins, seg_len=child_seg + c, seed=100 + c, data=rand_bytes(3 * (child_seg + c) * child_bins),
                     next_ixf=np.zeros(child_bins, np.int64), fname_idx=np.arange(ub, ub + child_bins, dtype=np.int64)))
    ub += child_bins
total = sum(f["data"].size for f in ixfs)
species = [dict(organism_name=f"o{i}", accession_id=f"a{i}", taxid=str(i), taxnames_string="x", taxid_string="1", user_bin=i, seq_len=1) for i in range(0, ub, 97)]
keep = os.environ.get("RELAYOUT_KEEP")          # keep the files (one per layout) and the reads in this directory for a profiler run
if keep:
    os.makedirs(keep, exist_ok=True)
tmp = keep or tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fq = os.path.join(tmp, "reads.fq")
with open(fq, "w") as f:
    for i in range(2000):
        s = "".join("ACGT"[x] for x in rng.integers(0, 4, 3000))
        f.write(f"@r{i}\n{s}\n+\n{'I' * 3000}\n")
print(f"index: {len(ixfs)} IXFs, {total / 1e9:.2f} GB of fingerprints (root {root_bins} bins x {3 * root_seg} rows, {n_children} children of {child_bins} bins)")
R, B, S, PM, UNP = _lib.LAYOUT_ROWS, _lib.LAYOUT_BIN_MAJOR, _lib.LAYOUT_BIT_SLICED, _lib.LAYOUT_POSITION_MAJOR, _lib.LAYOUT_PITCH_BINS
digests = {}
codes = [int(x, 0) for x in os.environ["RELAYOUT_CODES"].split(",")] if os.environ.get("RELAYOUT_CODES") else [0, R | PM, B, S]
for code in codes:
    sc = default_schema()
    sc.layout = code
    spec = describe_layout(code)
    path = os.path.join(tmp, spec.replace(",", "_") + ".hixf" if keep else "x.hixf")
    store_hixf(path, ixfs, ub, species, schema=sc)
    for rep in range(2):
        out = os.path.join(tmp, "out.tsv")
        cmd = [os.path.join(ROOT, "taxor_amd", "taxor"), "search", "--index-file", path, "--query-file", fq, "--output-file", out, "--percentage", "0.02"]
        if code:
            cmd += ["--ixf-layout", spec]
        cp = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, TAXOR_TUNING="1", TAXOR_TRACE_UPLOAD="1"))
        assert cp.returncode == 0, cp.stderr[-2000:]
        line = [l for l in cp.stderr.splitlines() if l.startswith("[upload]")]
        digests[code] = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16]
        print(f"{spec:45s} run {rep}: {line[-1] if line else '(no upload line)'}   tsv {digests[code]}")
    if not keep:
        os.remove(path)
assert len(set(digests.values())) == 1, digests
print("TSV identical under every layout")
