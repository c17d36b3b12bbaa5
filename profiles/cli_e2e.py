#!/usr/bin/env python3
"""End-to-end `taxor search` (C++ host) on a viral-class .hixf written to disk: FASTQ file in, TSV out.
Prints the CLI's own phase timers and the whole-command Mbp/s.  usage: python profiles/cli_e2e.py [n_reads] [read_len]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taxor_amd import GpuIndex, Searcher, synth  # noqa: E402
from taxor_amd.hixf_file import store_hixf  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
tmp = tempfile.mkdtemp(prefix="taxor_e2e_", dir="/tmp")
g, go = synth.random_genomes(64, 100000)
bins = 64
dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                       fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins)
hs = Searcher(dummy, ratio=0.5)
hoff, hashes = hs.seq_to_syncmers(g, go)
hs.close()
dummy.close()
planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(64)]
total = 373e6
lay = synth.make_layout(planted, root_bins=256, child_bins=64, n_children=252,
                        root_max_elems=int((total * 0.4 / 256 - 32) / 1.23),
                        child_max_elems=max(int((total * 0.6 / (253 * 64) - 32) / 1.23), max(len(p) for p in planted) + 64))
host = synth.materialize_host(lay)
species = [dict(organism_name=f"Organism {u}", accession_id=f"GCF_{u:09d}.1", taxid=str(1000 + u),
                taxnames_string=f"k__Viruses;s__Organism {u}", taxid_string=f"10239;{1000 + u}", user_bin=u, seq_len=100000)
           for u in range(lay["n_user_bins"])]
idx_path = os.path.join(tmp, "viral.hixf")
store_hixf(idx_path, host, lay["n_user_bins"], species)
bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=0.02, frac_random=0.1, threads=os.cpu_count() or 8)
fq = os.path.join(tmp, "reads.fastq")
t0 = time.time()
qual = b"I" * read_len
with open(fq, "wb") as f:
    bb = bases.tobytes()
    for i in range(n_reads):
        f.write(b"@read_%d\n" % i)
        f.write(bb[i * read_len:(i + 1) * read_len])
        f.write(b"\n+\n")
        f.write(qual)
        f.write(b"\n")
print(f"index {os.path.getsize(idx_path)/1e6:.0f} MB, fastq {os.path.getsize(fq)/1e9:.2f} GB written in {time.time()-t0:.1f}s", flush=True)
out = os.path.join(tmp, "out.tsv")
configs = ([], [], ["--threads", "16"], ["--gpu-list", "0,0"], ["--threads", "1"])
for extra in configs:
    t0 = time.time()
    cp = subprocess.run([os.path.join(ROOT, "taxor_amd", "taxor"), "search", "--index-file", idx_path, "--query-file", fq,
                         "--output-file", out] + (["--threads", "32"] if "--threads" not in extra else []) + extra, capture_output=True, text=True, env=dict(os.environ, TAXOR_TUNING="1", TAXOR_CLI_TRACE="1"))
    dt = time.time() - t0
    print(" ".join(extra) or "(default batch)", "rc", cp.returncode, f"wall {dt:.2f}s -> {n_reads*read_len/dt/1e6:.0f} Mbp/s end to end")
    print(cp.stdout.strip().replace("\n", " | "))
    print(cp.stderr.strip()[:1500])
# gzip input: one stream inflates on one thread, several files inflate concurrently
n_gz = 8
per = n_reads // 5 // n_gz            # a fifth of the reads, in eight files
parts = []
for j in range(n_gz):
    pth = os.path.join(tmp, f"part{j}.fastq")
    with open(pth, "wb") as f:
        for i in range(j * per, (j + 1) * per):
            f.write(b"@read_%d\n" % i)
            f.write(bb[i * read_len:(i + 1) * read_len])
            f.write(b"\n+\n")
            f.write(qual)
            f.write(b"\n")
    parts.append(pth)
procs = [subprocess.Popen(["gzip", "-1", "-f", pth]) for pth in parts]
for pr in procs:
    pr.wait()
gz = [pth + ".gz" for pth in parts]
subprocess.run("cat " + " ".join(gz) + " > " + os.path.join(tmp, "all.fastq.gz"), shell=True, check=True)
gz_bases = n_gz * per * read_len
for label, qf in (("one gzip file (8 members)", os.path.join(tmp, "all.fastq.gz")), ("eight gzip files", ",".join(gz))):
    t0 = time.time()
    cp = subprocess.run([os.path.join(ROOT, "taxor_amd", "taxor"), "search", "--index-file", idx_path, "--query-file", qf,
                         "--output-file", out, "--threads", "32"], capture_output=True, text=True)
    dt = time.time() - t0
    print(label, "rc", cp.returncode, f"wall {dt:.2f}s -> {gz_bases/dt/1e6:.0f} Mbp/s end to end")
    print(cp.stdout.strip().replace("\n", " | "))
lines = sum(1 for _ in open(out))
print("tsv lines", lines)
subprocess.run(["rm", "-rf", tmp])
