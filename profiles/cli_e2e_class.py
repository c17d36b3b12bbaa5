#!/usr/bin/env python3
"""The whole drop-in chain at BASELINE configs[2] / [3] scale: a RefSeq- or GTDB-class `.hixf` (family workload, built on the
GPU, read back and written to disk with the library's writer), a FASTQ of 10-kb reads (10 M of them = 200 GB for the
round-3 measurement), the C++ `taxor search` CLI -- index load + upload, parallel parsing, GPU search, TSV -- and a
byte comparison of its TSV with the library's own formatter over the Python searcher's results for the same reads.
usage: python profiles/cli_e2e_class.py [workload=refseq] [n_reads=400000]      (TAXOR_E2E_RUNS=32,32,8: --threads of the runs)"""
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
_rl = int(os.environ.get("TAXOR_E2E_READ_LEN", "0"))       # 0: the class's own read length (10 kb; viral 5 kb)
per = 131072 if not _rl else max(131072, 131072 * 10000 // _rl)      # reads per generated batch: ~1.3 Gbp like bench.py's batches
args = bench.parse_args(["--workload", workload, "--reads", str(min(n_reads, per)), "--batches", "1"] + (["--read-len", str(_rl)] if _rl else []))
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
bases, offs = batches[0]
read_len = info["read_len"]
import shutil  # noqa: E402
need = idx.data_bytes * 1.05 + n_reads * (2 * read_len + 64)


def _mem_limit():
    """what this process tree may use: the cgroup limit if there is one, else MemAvailable"""
    lim = None
    for pth in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(pth).read().strip()
            if v.isdigit() and int(v) < (1 << 60):
                lim = int(v)
        except OSError:
            pass
    try:
        avail = next(int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable"))
    except (OSError, StopIteration):
        avail = None
    return min(x for x in (lim, avail) if x is not None) if (lim or avail) else None


# index file + FASTQ + two TSVs live in tmpfs (= RAM) beside a host copy of the index while it is written: bound the
# footprint against what the box (or its cgroup) really has, instead of finding out by exhausting it
footprint = max(1.5 * idx.data_bytes,                                   # file + the largest IXF on the host while the index is written
                1.05 * idx.data_bytes + n_reads * (2 * read_len + 64) + 2 * n_reads * 1200 + 20e9)   # file + FASTQ + two TSVs + the CLI's buffers
lim = _mem_limit()
print(f"memory: footprint ~{footprint/1e9:.0f} GB, limit/available {lim/1e9 if lim else float('nan'):.0f} GB", flush=True)
if lim is not None and footprint > 0.8 * lim:
    raise SystemExit(f"refusing: ~{footprint/1e9:.0f} GB of host memory needed, {lim/1e9:.0f} GB available")
base = next((d for d in (os.environ.get("TAXOR_E2E_TMP"), "/tmp", "/dev/shm") if d and os.path.isdir(d) and shutil.disk_usage(d).free > need), None)
if base is None:
    raise SystemExit(f"no scratch directory with {need/1e9:.0f} GB free")
tmp = tempfile.mkdtemp(prefix="taxor_e2e_", dir=base)
print(f"scratch: {tmp}", flush=True)
t0 = time.time()
# the file is written IXF by IXF straight out of HBM (one IXF on the host at a time: the root's 45 GB, not all 113 GB)
host = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=idx.ixf_seed(i), next_ixf=f["next_ixf"], fname_idx=f["fname_idx"],
             data=None) for i, f in enumerate(lay["ixfs"])]
species = [dict(organism_name=f"Organism {u}", accession_id=f"GCF_{u:09d}.1", taxid=str(1000 + u), taxnames_string=f"k__Bacteria;s__Organism {u}",
                taxid_string=f"2;{1000 + u}", user_bin=u, seq_len=info["genome_len"]) for u in range(lay["n_user_bins"])]
idx_path = os.path.join(tmp, f"{workload}.hixf")
# TAXOR_E2E_LAYOUT=bin-major,padded,segment-major: the file is written in ANOTHER writer's fingerprint layout (ixf_layout.h) and the CLI
# reads it with --ixf-layout: the whole chain through the device re-layout at class scale (round 5)
e2e_layout = os.environ.get("TAXOR_E2E_LAYOUT", "")
schema = None
if e2e_layout:
    from taxor_amd.hixf_file import default_schema, parse_layout  # noqa: E402
    schema = default_schema()
    schema.layout = parse_layout(e2e_layout)
store_hixf(idx_path, host, lay["n_user_bins"], species, data_of=idx.download_ixf, schema=schema)
del host
print(f"{workload}-class index read back from HBM and written: {os.path.getsize(idx_path)/1e9:.2f} GB, {time.time()-t0:.1f}s", flush=True)
fq = os.path.join(tmp, "reads.fastq")
t0 = time.time()
# the reads, batch by batch (same generator and seeds as bench.py's distinct batches): searched through the library for the
# expected text, and appended to the FASTQ.  Fixed-width ids make every record the same length, so a batch is one 2-D array.
from taxor_amd import synth  # noqa: E402
sr = Searcher(idx, error_rate=args.error_rate)
hx = HixfFile(idx_path, schema=schema)          # (only its formatter is used here)
g, go = info.get("genomes"), info.get("genome_off")
expected_sizes, n_tuples, n_lines, kept = [], 0, 0, []
want_path = os.path.join(tmp, "want.tsv")
with open(fq, "wb") as f, open(want_path, "w") as wf:
    done = 0
    b = 0
    while done < n_reads:
        n = min(per, n_reads - done)
        if b == 0:
            bb, oo = bases, offs
            if oo.size - 1 > n:
                bb, oo = bb[: int(oo[n])], oo[: n + 1]
            n = oo.size - 1
        else:
            bb, oo, _ = synth.synth_reads(g, go, n, read_len, error_rate=args.read_error, frac_random=0.1, seed=synth.DEFAULT_SEED + 1000 * b,
                                          threads=info["ncpu"])
        if len(kept) < 8:
            kept.append((np.ascontiguousarray(bb), np.ascontiguousarray(oo)))
        r = sr.search_batch(bb, oo)
        ids = [f"read_{done + i:09d}" for i in range(n)]
        text = hx.format_reads(ids, np.full(n, read_len), r.n_hashes, r.read_off, r.user_bin, r.count)
        wf.write(text)
        expected_sizes.append(len(text))
        n_tuples += int(r.user_bin.size)
        n_lines += text.count("\n")
        fasta = os.environ.get("TAXOR_E2E_FORMAT", "fastq") == "fasta"
        rec = np.empty((n, 1 + 14 + 1 + read_len + 1 if fasta else 1 + 14 + 1 + read_len + 3 + read_len + 1), dtype=np.uint8)
        rec[:, 0] = ord(">") if fasta else ord("@")
        rec[:, 1:6] = np.frombuffer(b"read_", np.uint8)
        num = np.arange(done, done + n, dtype=np.int64)
        rec[:, 6:15] = (num[:, None] // 10 ** np.arange(8, -1, -1)) % 10 + 48
        rec[:, 15] = 10
        rec[:, 16:16 + read_len] = bb.reshape(n, read_len)
        if not fasta:
            rec[:, 16 + read_len:19 + read_len] = np.frombuffer(b"\n+\n", np.uint8)
            if os.environ.get("TAXOR_E2E_QUAL") == "skewed":
                # quality strings as a sequencer writes them (a skewed alphabet, no structure): windows of one 64-MB random block at
                # random offsets -- repeats lie far outside deflate's 32 KiB, so the text compresses like real FASTQ (about 2:1)
                if "qblock" not in globals():
                    _q = np.frombuffer(b"%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJK", np.uint8)
                    _p = np.exp(-0.5 * ((np.arange(_q.size) - 24) / 7.0) ** 2)
                    globals()["qblock"] = np.random.default_rng(9).choice(_q, size=(64 << 20) + read_len, p=_p / _p.sum())
                qo = np.random.default_rng(done).integers(0, 64 << 20, size=n)
                rec[:, 19 + read_len:19 + 2 * read_len] = qblock[qo[:, None] + np.arange(read_len)[None, :]]
            else:
                rec[:, 19 + read_len:19 + 2 * read_len] = ord("I")
        rec[:, -1] = 10
        rec.tofile(f)
        done += n
        b += 1
sr.close()
hx.close()
# the library's own host-fed rate on these reads (bench.py's `sustained`: page-locked staging buffers, two searchers in
# flight, results fetched every call) -- what the CLI's search phase is held against
keep = kept[:8]
sargs = bench.parse_args(["--workload", workload, "--sustained-reads", str(max(n_reads, 4 * per))] + (["--read-len", str(_rl)] if _rl else []))
_, sustained = bench.dropin_measurements(sargs, idx, keep, read_len)
print(f"library sustained (host-fed, two searchers, {sustained['reads']} reads): {sustained['value']:.0f} Mbp/s", flush=True)
del keep, kept
idx.close()                      # the CLI loads its own replica
print(f"fastq {os.path.getsize(fq)/1e9:.2f} GB ({n_reads} reads) written and searched through the library in {time.time()-t0:.1f}s", flush=True)
out = os.path.join(tmp, "out.tsv")
if os.environ.get("TAXOR_E2E_GZ"):
    # the same reads as ONE gzip member (pieces deflated in parallel, each primed with the 32 KiB before it: back-references across
    # every piece boundary, like gzip's own output; profiles/r04/gz_single_member.py) -- the CLI then runs on reads.fastq.gz
    from multiprocessing import Pool
    sys.path.insert(0, os.path.join(ROOT, "profiles", "r04"))
    import gz_single_member as gzsm       # (a plain import: the pool's workers must be able to find the module by name)
    t0 = time.time()
    size = os.path.getsize(fq)
    jobs = [(fq, off, min(gzsm.PIECE, size - off), off + gzsm.PIECE >= size, 6) for off in range(0, size, gzsm.PIECE)]
    crc, total = 0, 0
    with Pool(min(len(os.sched_getaffinity(0)), 96)) as pool, open(fq + ".gz", "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for o, c, n in pool.imap(gzsm.deflate_piece, jobs, chunksize=1):
            f.write(o)
            crc = gzsm.crc_combine(crc, c, n) if total else c
            total += n
        f.write(crc.to_bytes(4, "little") + (total & 0xFFFFFFFF).to_bytes(4, "little"))
    print(f"one gzip member: {os.path.getsize(fq + '.gz')/1e9:.2f} GB of {size/1e9:.2f} GB, {time.time()-t0:.0f} s", flush=True)
    if os.environ.get("TAXOR_E2E_GZ") == "only":
        os.remove(fq)
    fq = fq + ".gz"
_r = os.environ.get("TAXOR_E2E_RUNS", "32,32,8")
runs = _r.split(";") if ";" in _r else _r.split(",")       # "16;16:--gpu-list:0,0" when a run's extra arguments contain commas
for thr in runs:
    thr, *run_env = thr.split("@")                                      # "32@TAXOR_CLI_FILL_MS=5": tuning variables of this run only
    extra = ["--threads", thr.split(":")[0]] + thr.split(":")[1:]       # "16:--gpu-list:0,0" = extra arguments after the thread count
    if e2e_layout:
        extra += ["--ixf-layout", e2e_layout]
    if run_env:
        print("run with", " ".join(run_env))
    binary = next((kv[4:] for kv in run_env if kv.startswith("BIN=")), os.path.join(ROOT, "taxor_amd", "taxor"))     # "32@BIN=/path/taxor_old": A/B of two builds
    run_env = [kv for kv in run_env if not kv.startswith("BIN=")]
    t0 = time.time()
    cp = subprocess.run([binary, "search", "--index-file", idx_path, "--query-file", fq, "--output-file", out] + extra,
                        capture_output=True, text=True,
                        env=dict(os.environ, TAXOR_TUNING="1", TAXOR_CLI_TRACE="1", **dict(kv.split("=", 1) for kv in os.environ.get("TAXOR_E2E_ENV", "").split() + run_env)))
    dt = time.time() - t0
    print(" ".join(extra), "rc", cp.returncode, f"wall {dt:.2f}s -> {n_reads*read_len/dt/1e6:.0f} Mbp/s end to end (index load included)")
    print(cp.stdout.strip().replace("\n", " | "))
    tr = [l for l in cp.stderr.splitlines() if "trace" in l]
    print("\n".join(tr)[:2400], flush=True)
    import re  # noqa: E402
    thr = " ".join(extra[1:])
    stamp = {}
    for l in tr:
        m = re.match(r"\[trace\]\s+([0-9.]+) s  (.*)", l)
        if m:
            stamp[m.group(2)] = float(m.group(1))
    m = re.search(r"search phase ([0-9.]+) s wall after the index was resident = ([0-9.]+) Mbp/s", cp.stderr)
    if m and "writer done" in stamp and "output closed" in stamp:
        print(f"RATE --threads {thr}: search phase {float(m.group(2)):.0f} Mbp/s = {float(m.group(2)) / sustained['value']:.3f} x library sustained; "
              f"teardown (writer done -> output closed) {stamp['output closed'] - stamp['writer done']:.3f} s; process exit {dt - stamp['output closed']:.3f} s later", flush=True)
# the CLI's text against the library formatter over the Python searcher's tuples, batch by batch
same = True
with open(out, "rb") as fo, open(want_path, "rb") as fw:
    hdr = fo.readline()
    while True:
        x, y = fo.read(64 << 20), fw.read(64 << 20)
        if x != y:
            same = False
            break
        if not x:
            break
print(f"TSV: {n_lines} lines, {n_tuples} tuples before the 0.8*max filter; identical to formatter(searcher results): {same}")
if os.environ.get("TAXOR_E2E_KEEP"):
    print(f"kept: {tmp}", flush=True)
else:
    subprocess.run(["rm", "-rf", tmp])
sys.exit(0 if same else 1)
