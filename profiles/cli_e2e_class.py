#!/usr/bin/env python3
"""The whole drop-in chain at BASELINE configs[2] scale: a RefSeq-class `.hixf` (9.9 GB, family workload, built on the
GPU, read back and written to disk with the library's writer), a FASTQ of 10-kb reads, the C++ `taxor search` CLI --
index load + upload, parallel parsing, GPU search, TSV -- and a per-read comparison of its TSV with the library's own
formatter over the Python searcher's results for the same reads.
usage: python profiles/cli_e2e_class.py [workload=refseq] [n_reads=400000]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

import bench  # noqa: E402
from taxor_amd import Searcher  # noqa: E402
from taxor_amd.hixf_file import HixfFile, store_hixf  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "refseq"
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
args = bench.parse_args(["--workload", workload, "--reads", str(n_reads), "--batches", "1"])
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
bases, offs = batches[0]
read_len = info["read_len"]
import shutil  # noqa: E402
need = idx.data_bytes * 1.05 + n_reads * (2 * read_len + 64)
base = next((d for d in (os.environ.get("TAXOR_E2E_TMP"), "/tmp", "/dev/shm") if d and os.path.isdir(d) and shutil.disk_usage(d).free > need), None)
if base is None:
    raise SystemExit(f"no scratch directory with {need/1e9:.0f} GB free")
tmp = tempfile.mkdtemp(prefix="taxor_e2e_", dir=base)
print(f"scratch: {tmp}", flush=True)
t0 = time.time()
host = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=idx.ixf_seed(i), next_ixf=f["next_ixf"], fname_idx=f["fname_idx"],
             data=idx.download_ixf(i)) for i, f in enumerate(lay["ixfs"])]
species = [dict(organism_name=f"Organism {u}", accession_id=f"GCF_{u:09d}.1", taxid=str(1000 + u), taxnames_string=f"k__Bacteria;s__Organism {u}",
                taxid_string=f"2;{1000 + u}", user_bin=u, seq_len=info["genome_len"]) for u in range(lay["n_user_bins"])]
idx_path = os.path.join(tmp, f"{workload}.hixf")
store_hixf(idx_path, host, lay["n_user_bins"], species)
del host
print(f"{workload}-class index read back from HBM and written: {os.path.getsize(idx_path)/1e9:.2f} GB, {time.time()-t0:.1f}s", flush=True)
fq = os.path.join(tmp, "reads.fastq")
t0 = time.time()
qual = b"I" * read_len
bb = bases.tobytes()
with open(fq, "wb") as f:
    for i in range(n_reads):
        f.write(b"@read_%d\n" % i)
        f.write(bb[i * read_len:(i + 1) * read_len])
        f.write(b"\n+\n")
        f.write(qual)
        f.write(b"\n")
print(f"fastq {os.path.getsize(fq)/1e9:.2f} GB written in {time.time()-t0:.1f}s", flush=True)
sr = Searcher(idx, error_rate=args.error_rate)
res = sr.search_batch(bases, offs)
sr.close()
idx.close()                      # the CLI loads its own replica
out = os.path.join(tmp, "out.tsv")
for extra in (["--threads", "32"], ["--threads", "32"], ["--threads", "8"]):
    t0 = time.time()
    cp = subprocess.run([os.path.join(ROOT, "taxor_amd", "taxor"), "search", "--index-file", idx_path, "--query-file", fq, "--output-file", out] + extra,
                        capture_output=True, text=True, env=dict(os.environ, TAXOR_CLI_TRACE="1"))
    dt = time.time() - t0
    print(" ".join(extra), "rc", cp.returncode, f"wall {dt:.2f}s -> {n_reads*read_len/dt/1e6:.0f} Mbp/s end to end (index load included)")
    print(cp.stdout.strip().replace("\n", " | "))
    print("\n".join(l for l in cp.stderr.splitlines() if "trace" in l)[:1200], flush=True)
# the CLI's text against the library formatter over the Python searcher's tuples, read by read
hx = HixfFile(idx_path)
want = []
for i in range(n_reads):
    lo, hi = int(res.read_off[i]), int(res.read_off[i + 1])
    want.append(hx.format_read(f"read_{i}", read_len, int(res.n_hashes[i]), res.user_bin[lo:hi], res.count[lo:hi]))
hx.close()
got = open(out).read()
hdr, body = got.split("\n", 1)
same = body == "".join(want)
print(f"TSV: {got.count(chr(10))} lines, {res.user_bin.size} tuples before the 0.8*max filter; identical to formatter(searcher results): {same}")
subprocess.run(["rm", "-rf", tmp])
sys.exit(0 if same else 1)
