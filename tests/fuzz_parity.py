#!/usr/bin/env python3
"""Long randomized differential run (GPU vs CPU oracle): python tests/fuzz_parity.py [seconds] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from tests.fuzz_common import run_trial  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
t0 = time.time()
n = bad = tuples = gpu_built = void = 0
while time.time() - t0 < budget:
    ok, cfg, nt = run_trial(seed)
    n += 1
    tuples += nt
    gpu_built += bool(cfg.get("gpu_built"))
    void += "void" in cfg
    if not ok:
        bad += 1
        print("MISMATCH", cfg, flush=True)
    seed += 1
print(f"{n} trials ({gpu_built} on indexes constructed by the GPU builder, {void} void), {tuples} tuples compared, {bad} mismatches, {time.time()-t0:.0f} s")
sys.exit(1 if bad else 0)
