#!/usr/bin/env python3
"""A bgzip-like FASTQ (members of 64 KB of text) through `taxor reads`: the rate of the multi-member reader (inflate + parse).
usage: gz_multi_member.py [--gb 8] [--threads 16] binary..."""
import argparse
import os
import subprocess
import sys
import time
import zlib
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def members(job):
    path, off, n = job
    with open(path, "rb") as f:
        f.seek(off)
        data = f.read(n)
    out = []
    for a in range(0, len(data), 65280):
        piece = data[a:a + 65280]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        out.append(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body + zlib.crc32(piece).to_bytes(4, "little") + len(piece).to_bytes(4, "little"))
    return b"".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--threads", default="16")
    ap.add_argument("binaries", nargs="+")
    a = ap.parse_args()
    tmp = "/dev/shm/taxor_gzm"
    os.makedirs(tmp, exist_ok=True)
    import gz_single_member as gzsm
    sys.argv = ["x", "--gb", str(a.gb), "--threads", "1", "--keep", "--tmp", tmp]
    plain = os.path.join(tmp, "reads.fastq")
    if not os.path.exists(plain):
        # the generator of the single-member profile writes reads.fastq (and a .gz that is not needed here)
        gzsm.main()
    size = os.path.getsize(plain)
    gz = os.path.join(tmp, "reads.bgzf.fastq.gz")
    t0 = time.time()
    step = 65280 * 256
    with Pool(min(len(os.sched_getaffinity(0)), 64)) as pool, open(gz, "wb") as f:
        for blob in pool.imap(members, [(plain, off, min(step, size - off)) for off in range(0, size, step)], chunksize=1):
            f.write(blob)
    print(f"bgzip-like file: {os.path.getsize(gz)/1e9:.2f} GB of {size/1e9:.2f} GB in members of 64 KB, {time.time()-t0:.0f} s", flush=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "taxor_amd"))
    extra_envs = [e for e in os.environ.get("GZM_ENVS", "").split(";") if e]          # e.g. "TAXOR_CLI_GZ_PARSERS=8;TAXOR_CLI_GZ_PARSERS=12": one more run each
    for rnd in range(2):
        for b in a.binaries:
            for th in a.threads.split(","):
                t0 = time.time()
                cp = subprocess.run([b, "reads", "--query-file", gz, "--threads", th], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
                print(f"{os.path.basename(b)} --threads {th}: {cp.stderr.strip().splitlines()[-1]} -> {size/1e9/(time.time()-t0):.2f} GB/s of FASTQ (wall, printing included)", flush=True)
                for ee in extra_envs:
                    cp = subprocess.run([b, "reads", "--query-file", gz, "--threads", th], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                        env=dict(env, TAXOR_TUNING="1", **dict([ee.split("=", 1)])))
                    print(f"{os.path.basename(b)} --threads {th} {ee}: {cp.stderr.strip().splitlines()[-1]}", flush=True)
    subprocess.run(["rm", "-rf", tmp])


if __name__ == "__main__":
    main()
