#!/usr/bin/env python3
"""Where does a single drop-in call (taxor_gpu_search_batch on host buffers) lose against the resident-batch step?
GTDB-class family workload, one batch of 131072 x 10 kb reads: resident run, then the streamed call from pageable and from
page-locked memory, each with its HIP-event stage sums (TAXOR_TRACE_BATCH=1 prints the host-side marks).
usage: python profiles/single_call.py [--workload gtdb] [--reps 3]      env knobs are read by the library as usual"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402

import bench  # noqa: E402
from taxor_amd import Searcher, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="gtdb")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--family-size", type=int, default=16)
ap.add_argument("--reads", type=int, default=0)
ap.add_argument("--read-len", type=int, default=0)
a = ap.parse_args()
extra = []
if a.reads:
    extra += ["--reads", str(a.reads)]
if a.read_len:
    extra += ["--read-len", str(a.read_len)]
args = bench.parse_args(["--workload", a.workload, "--batches", "2", "--family-size", str(a.family_size)] + extra)
wl, idx, lay, batches, info = bench.build_workload(args, 0, 0, 1)
bases, offs = batches[0]
nb = float(offs[-1])


def stats_line(sr):
    st = sr.stats()
    return (f"GPU total {st['total_ms']:.1f} ms (query {st['query_ms']:.1f}, syncmers {st['syncmer_ms']:.1f}, finalize {st['finalize_ms']:.1f}; "
            f"{st['query_launches']} query launches)")


sr = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
sr.upload(bases, offs)
sr.run(); sr.sync()
best = 1e9
for _ in range(a.reps):
    t0 = time.perf_counter()
    sr.run(); sr.sync()
    best = min(best, time.perf_counter() - t0)
print(f"resident step            : {best*1e3:7.1f} ms = {nb/best/1e6:8.0f} Mbp/s   {stats_line(sr)}", flush=True)
resident = best
sr.close()

for label, pinned in (("pageable", False), ("page-locked", True)):
    for b, o in batches:
        if pinned:
            _lib.lib().taxor_gpu_host_register(b.ctypes.data_as(C.c_void_p), b.nbytes)
    sr = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
    sr.search_batch(*batches[1])                   # scratch sized, code paths warm
    best, line = 1e9, ""
    for r in range(a.reps):
        bb, oo = batches[r % 2]
        t0 = time.perf_counter()
        res = sr.search_batch(bb, oo, copy=False)
        dt = time.perf_counter() - t0
        if dt < best:
            best, line = dt, stats_line(sr)
    print(f"single call, {label:11s}: {best*1e3:7.1f} ms = {nb/best/1e6:8.0f} Mbp/s = {resident/best:.3f} x resident   {line}", flush=True)
    sr.close()
    for b, o in batches:
        if pinned:
            _lib.lib().taxor_gpu_host_unregister(b.ctypes.data_as(C.c_void_p))
# plain copy rates of this box, for scale: one hipMemcpy of the batch from pageable / page-locked memory
t = torch.empty(int(nb), dtype=torch.uint8, device="cuda")
src = torch.from_numpy(bases[: int(nb)])
for label, s in (("pageable", src), ("page-locked", src.pin_memory())):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.copy_(s, non_blocking=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"plain H2D copy of the batch, {label:11s}: {dt*1e3:6.1f} ms = {nb/dt/1e9:5.1f} GB/s", flush=True)
idx.close()
