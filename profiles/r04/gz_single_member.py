#!/usr/bin/env python3
"""One SINGLE-MEMBER .fastq.gz of N GB (decompressed) through the parallel inflate (taxor_amd/csrc/pgz.h):
  1. synthetic FASTQ (10-kb reads cut from planted genomes, skewed quality alphabet) written to tmpfs;
  2. compressed into ONE gzip member the way pigz does it -- pieces deflated in parallel, each primed with the 32 KiB before it as
     dictionary (so the stream has back-references across every piece boundary like gzip's own output), sync-flushed and
     concatenated, one CRC-32 / length trailer -- because `gzip` itself needs minutes per 10 GB;
  3. `taxor inflate` at several thread counts (rate, chunks decoded twice), byte-compared with the plain file once;
(that `taxor search` writes the same TSV for the .gz as for the plain file is tests/test_gpu_cli.py::test_cli_single_member_gzip_equals_plain.)
usage: gz_single_member.py [--gb 10] [--threads 1,8,32,64] [--tmp /dev/shm/taxor_gz]"""
import argparse
import os
import subprocess
import sys
import time
import zlib
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TAXOR = os.path.join(ROOT, "taxor_amd", "taxor")
PIECE = 16 << 20


def deflate_piece(job):
    path, off, n, last, level = job
    with open(path, "rb") as f:
        f.seek(max(0, off - 32768))
        pre = f.read(min(off, 32768))
        data = f.read(n)
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, zlib.Z_DEFAULT_STRATEGY, pre) if pre else zlib.compressobj(level, zlib.DEFLATED, -15, 9)
    out = co.compress(data) + (co.flush(zlib.Z_FINISH) if last else co.flush(zlib.Z_SYNC_FLUSH))
    return out, zlib.crc32(data), len(data)


def crc_combine(c1, c2, len2):      # zlib's crc32_combine, via the gf2 matrices
    def gf2_times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def gf2_square(mat):
        return [gf2_times(mat, mat[n]) for n in range(32)]
    if len2 <= 0:
        return c1
    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = gf2_square(odd)
    odd = gf2_square(even)
    while True:
        even = gf2_square(odd)
        if len2 & 1:
            c1 = gf2_times(even, c1)
        len2 >>= 1
        if not len2:
            break
        odd = gf2_square(even)
        if len2 & 1:
            c1 = gf2_times(odd, c1)
        len2 >>= 1
        if not len2:
            break
    return c1 ^ c2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=10.0)
    ap.add_argument("--threads", default="1,8,16,32,64")
    ap.add_argument("--tmp", default="/dev/shm/taxor_gz")
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--inflate-args", default="", help='extra arguments of `taxor inflate`, e.g. "--gpu 0 --batch-chunks 1024"')
    a = ap.parse_args()
    os.makedirs(a.tmp, exist_ok=True)
    plain, gz = os.path.join(a.tmp, "reads.fastq"), os.path.join(a.tmp, "reads.fastq.gz")
    from taxor_amd import synth
    ncpu = len(os.sched_getaffinity(0))
    g, go = synth.random_genomes(64, 2000000, seed=11)
    n_reads = int(a.gb * 1e9 / 20100)
    t0 = time.time()
    rng = np.random.default_rng(3)
    qual = np.frombuffer(b"%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJK", np.uint8)
    pq = np.exp(-0.5 * ((np.arange(qual.size) - 24) / 7.0) ** 2)
    pq /= pq.sum()
    with open(plain, "wb") as f:
        done = 0
        while done < n_reads:
            m = min(100000, n_reads - done)
            bases, offs, _ = synth.synth_reads(g, go, m, 10000, error_rate=0.04, frac_random=0.1, seed=100 + done, threads=ncpu)
            q = rng.choice(qual, size=int(offs[-1]), p=pq)
            for i in range(m):
                lo, hi = int(offs[i]), int(offs[i + 1])
                f.write(b"@read_%d runid=0123456789abcdef ch=%d start_time=2024-01-01T00:%02d:%02dZ\n" % (done + i, (done + i) % 2048, i % 60, (i * 7) % 60))
                f.write(bases[lo:hi].tobytes())
                f.write(b"\n+\n")
                f.write(q[lo:hi].tobytes())
                f.write(b"\n")
            done += m
    size = os.path.getsize(plain)
    print(f"plain FASTQ: {n_reads} reads, {size/1e9:.2f} GB, {time.time()-t0:.0f} s", flush=True)
    t0 = time.time()
    jobs = [(plain, off, min(PIECE, size - off), off + PIECE >= size, a.level) for off in range(0, size, PIECE)]
    crc, total = 0, 0
    with Pool(min(ncpu, 96)) as pool, open(gz, "wb") as f:
        f.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        for out, c, n in pool.imap(deflate_piece, jobs, chunksize=1):
            f.write(out)
            crc = crc_combine(crc, c, n) if total else c
            total += n
        f.write(crc.to_bytes(4, "little") + (total & 0xFFFFFFFF).to_bytes(4, "little"))
    print(f"one gzip member: {os.path.getsize(gz)/1e9:.2f} GB (ratio {size/os.path.getsize(gz):.2f}), {time.time()-t0:.0f} s; pieces primed with the 32 KiB before them", flush=True)
    out = os.path.join(a.tmp, "inflated.fastq")
    first = True
    for th in [int(x) for x in a.threads.split(",")]:
        cmd = [TAXOR, "inflate", "--query-file", gz, "--threads", str(th)] + a.inflate_args.split() + (["--output-file", out] if first else [])
        cp = subprocess.run(cmd, capture_output=True, text=True)
        print(f"taxor inflate --threads {th}{' (written to tmpfs)' if first else ''}: " + cp.stdout.strip().replace("\n", " | "), cp.stderr.strip()[-300:], flush=True)
        if first:
            same = subprocess.run(["cmp", out, plain]).returncode == 0
            print("byte-identical to the plain file:", same, flush=True)
            os.remove(out)
            assert same
            first = False
    t0 = time.time()
    rc = subprocess.run(["bash", "-c", f"gzip -dc {gz} | head -c 2000000000 > /dev/null"]).returncode
    print(f"gzip -dc (zlib, one thread) on the first 2 GB of output: {2.0/(time.time()-t0):.2f} GB/s", flush=True)
    if not a.keep:
        for p in (plain, gz):
            os.remove(p)


if __name__ == "__main__":
    main()
