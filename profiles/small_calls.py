#!/usr/bin/env python3
"""The drop-in call at the REFERENCE's chunk size: taxor_search.cpp:315 hands its workers 1024 records at a time.  Rate of
taxor_gpu_search_batch on host buffers as a function of the reads per call (one searcher, calls back to back; and two
searchers on two host threads), GTDB-class family workload.
usage: python profiles/small_calls.py [--workload gtdb] [--read-len 10000]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

import bench  # noqa: E402
from taxor_amd import Searcher  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="gtdb")
ap.add_argument("--read-len", type=int, default=0)
ap.add_argument("--sizes", default="256,1024,4096,16384,65536,131072")
ap.add_argument("--registered", action="store_true", help="page-lock the caller's buffer once (taxor_gpu_host_register): the copies are then asynchronous")
a = ap.parse_args()
extra = ["--read-len", str(a.read_len)] if a.read_len else []
args = bench.parse_args(["--workload", a.workload, "--batches", "1"] + extra)
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
bases, offs = batches[0]
n_all = offs.size - 1


def chunks(n):
    out = []
    for lo in range(0, n_all - n + 1, n):
        o = offs[lo:lo + n + 1]
        out.append((bases[int(o[0]):int(o[-1])], (o - o[0]).astype(np.uint64)))
    return out


def run(sr, cs, reps):
    t = 0
    for _ in range(reps):
        for b, o in cs:
            r = sr.search_batch(b, o, copy=False)
            t += int(r.user_bin.size)
    return t


if a.registered:
    import ctypes as C
    from taxor_amd import _lib
    assert _lib.lib().taxor_gpu_host_register(bases.ctypes.data_as(C.c_void_p), bases.nbytes) == 0
srs = [Searcher(idx, error_rate=args.error_rate) for _ in range(2)]
for n in [int(x) for x in a.sizes.split(",")]:
    if n > n_all:
        continue
    cs = chunks(n)[:max(1, min(len(chunks(n)), (1 << 17) // n, 64))]
    nb = sum(int(o[-1]) for _, o in cs)
    for sr in srs:
        run(sr, cs[:2], 1)
    reps = max(1, int(2e9 // max(nb, 1)) if n <= 4096 else 2)
    t0 = time.perf_counter()
    run(srs[0], cs, reps)
    dt1 = time.perf_counter() - t0
    th = [threading.Thread(target=run, args=(srs[i], cs, reps)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt2 = time.perf_counter() - t0
    calls = len(cs) * reps
    print(f"{n:7d} reads per call: one searcher {nb*reps/dt1/1e6:8.0f} Mbp/s ({dt1/calls*1e3:7.3f} ms per call); "
          f"two searchers {2*nb*reps/dt2/1e6:8.0f} Mbp/s ({dt2/calls*1e3:7.3f} ms per call and thread)", flush=True)
for sr in srs:
    sr.close()
idx.close()
