#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE on THIS path's access shapes (MI355X_MICROARCH.md, HBM: "calibrate on a
known byte count in your own access pattern before trusting an absolute").

    python3 profiles/calibrate_fetch.py [--out profiles/r03] [--gb 6]

Parent: one `rocprofv3 --pmc <counters> --kernel-trace` pass per counter set over a child that launches kernels with a
KNOWN request count and nothing else, in the two shapes of k_query_level:
  rows   : whole fingerprint rows at random row indices (dense phase) -- 64-B, 128-B, 256-B and 1-KiB rows
  sparse : one 16-B load per lane, every lane on a row of its own (pruned phase)
each with plain and with non-temporal loads, on tables of several GB (far beyond L2 and the 256 MB memory-side cache).
Writes <out>/fetch_calibration.json (+ .txt): per shape, raw FETCH_SIZE bytes per launch / requested bytes per launch.
`bench.py` reads profiles/fetch_calibration.json (a committed copy of the latest run) to turn the raw counter of a
k_query_level launch into bytes for the launch's own mix of shapes.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHAPES = [(64, "rows64"), (128, "rows128"), (256, "rows256"), (1024, "rows1024")]


def child(path, gb):
    """launch the known-size kernels; write the launch list (in launch order) to `path`"""
    import numpy as np
    from taxor_amd import GpuIndex
    ixfs = []
    nroot = 64
    nx = np.zeros(nroot, np.int64)
    fn = np.arange(nroot, dtype=np.int64)
    for j in range(len(SHAPES)):
        nx[j], fn[j] = j + 1, -1
    fn[len(SHAPES):] = np.arange(nroot - len(SHAPES))
    ixfs.append(dict(bins=nroot, stride=64, seg_len=64, seed=1, next_ixf=nx, fname_idx=fn, data=None))
    ub = nroot - len(SHAPES)
    for bins, _ in SHAPES:
        stride = max(64, bins)
        seg = int(gb * (1 << 30)) // (3 * stride)
        ixfs.append(dict(bins=bins, stride=stride, seg_len=seg, seed=2, next_ixf=np.zeros(bins, np.int64),
                         fname_idx=ub + np.arange(bins, dtype=np.int64), data=None))
        ub += bins
    idx = GpuIndex(ixfs, ub)
    for i in range(1, len(ixfs)):
        idx.fill_random(i, 100 + i)
    launches = []
    for i, (bins, label) in enumerate(SHAPES, start=1):
        for pattern in (0, 1):
            if pattern == 1 and bins not in (128, 1024):
                continue
            for nt in (0, 1):
                reps = 2
                gbps, nbytes, reqs = idx.gather_pattern(i, pattern, nt, want_bytes=(4 << 30) if pattern == 0 else (1 << 30), reps=reps)
                name = "k_gather_ceiling" if pattern == 0 else "k_gather_sparse"
                for r in range(reps + 1):
                    launches.append(dict(kernel=name, shape=(label if pattern == 0 else f"sparse16_in_{label}"), nt=nt,
                                         row_bytes=max(64, bins), requested_bytes=nbytes, requests=reqs, warmup=(r == 0),
                                         GBps=round(gbps, 1)))
    idx.close()
    with open(path, "w") as f:
        json.dump(launches, f)


def run_pass(counters, gb, tmp):
    exe = shutil.which("rocprofv3")
    d = os.path.join(tmp, "_".join(counters))
    lj = os.path.join(tmp, "launches_" + "_".join(counters) + ".json")
    cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--",
           sys.executable, os.path.abspath(__file__), "--child", lj, "--gb", str(gb)]
    cp = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
    if cp.returncode != 0 or not os.path.exists(lj):
        return None, f"rc {cp.returncode}: {cp.stderr[-400:]}"
    launches = json.load(open(lj))
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            if "k_gather_ceiling" in kn or "k_gather_sparse" in kn:
                rows.append((int(r["Dispatch_Id"]), "k_gather_sparse" if "k_gather_sparse" in kn else "k_gather_ceiling",
                             r["Counter_Name"], float(r["Counter_Value"])))
    by_disp = {}
    for disp, kn, cn, v in rows:
        by_disp.setdefault(disp, {"kernel": kn})[cn] = by_disp.get(disp, {}).get(cn, 0.0) + v
    disps = [by_disp[k] for k in sorted(by_disp)]
    if len(disps) != len(launches):
        return None, f"{len(disps)} profiled dispatches vs {len(launches)} launches"
    for L, D in zip(launches, disps):
        if L["kernel"] != D["kernel"]:
            return None, "dispatch order does not match the launch list"
        L["counters"] = {k: v for k, v in D.items() if k != "kernel"}
    return launches, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default="")
    ap.add_argument("--gb", type=float, default=6.0, help="size of each calibration table in GiB")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03"))
    ap.add_argument("--only", default="", help="comma-separated pass names (counters joined by '+') to run")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.gb)
        return
    if shutil.which("rocprofv3") is None:
        raise SystemExit("rocprofv3 not found")
    os.makedirs(a.out, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="taxor_cal_", dir="/tmp")
    report = {"passes": {}}
    try:
        # FETCH_SIZE is the counter bench.py uses; the raw request counters behind it, where this rocprofv3 has them, say
        # HOW the bytes are tallied (MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B whatever the request size)
        for counters in (["FETCH_SIZE"], ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCC_REQ_sum", "TCC_MISS_sum"],
                         ["TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
                         ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
                         ["TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_DRAM_32B_sum"], ["WRITE_SIZE"]):
            if a.only and "+".join(counters) not in a.only.split(","):
                continue
            launches, err = run_pass(counters, a.gb, tmp)
            report["passes"]["+".join(counters)] = launches if launches is not None else {"error": err}
            print("+".join(counters), "ok" if launches is not None else f"FAILED: {err}", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fs = report["passes"].get("FETCH_SIZE")
    table = {}
    lines = ["shape                nt  row_B  requested/launch  raw FETCH_SIZE/launch  raw/requested  raw/(requests*64B)  GB/s(requested)"]
    if isinstance(fs, list):
        groups = {}
        for L in fs:
            if L["warmup"]:
                continue
            groups.setdefault((L["shape"], L["nt"]), []).append(L)
        for (shape, nt), ls in groups.items():
            req = ls[0]["requested_bytes"]
            raw = sum(l["counters"].get("FETCH_SIZE", 0.0) for l in ls) / len(ls) * 1024.0
            sect = ls[0]["requests"] * 64.0
            e = dict(shape=shape, nt=nt, row_bytes=ls[0]["row_bytes"], requested_bytes_per_launch=req, raw_fetch_bytes_per_launch=raw,
                     raw_over_requested=raw / req, raw_over_sectors64=raw / sect, requested_GBps=ls[0]["GBps"])
            table[f"{shape}/{'nt' if nt else 'plain'}"] = e
            lines.append(f"{shape:20s} {nt:2d} {e['row_bytes']:6d} {req:17.4e} {raw:22.4e} {e['raw_over_requested']:14.4f} "
                         f"{e['raw_over_sectors64']:19.4f} {e['requested_GBps']:16.1f}")
    report["table"] = table
    report["note"] = ("raw = rocprofv3 FETCH_SIZE (KB) x 1024, uncorrected.  rows*: requested = rows x row bytes.  sparse16: requested = loads "
                      "x 16 B; raw/(requests*64B) compares with one 64-B sector per load.  bench.py divides a k_query_level launch's raw "
                      "counter by the mix-weighted raw/requested of its shapes.")
    with open(os.path.join(a.out, "fetch_calibration.json"), "w") as f:
        json.dump(report, f, indent=1)
    with open(os.path.join(a.out, "fetch_calibration.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
