#!/usr/bin/env python3
"""bench.py -- `taxor search` hot path on MI355X: Mbp/s classified against a GTDB-class k22/s12 HIXF.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path (syncmers -> dedup -> threshold -> level-synchronous HIXF query -> DFS
ordered per-read tuples) over one batch of synthetic long reads that is already resident in HBM (2-bit packed).
`--batches` DISTINCT batches are resident and the steps rotate through them, so no step re-reads the rows its
predecessor touched.  With N>1 every rank holds a replica of the index, classifies its own shard of reads and the
per-read results are gathered on rank 0 over RCCL inside the timed region (`--scaling weak`: per-GPU work fixed;
`--scaling strong`: the N=1 batch is cut into N shards).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     : k_query_level (dominant kernel).  achieved = bytes the kernel REQUESTED (device-counted) / its
                 HIP-event time, frac = achieved / 8 TB/s (<= 1 by construction).  The reference's formulation
                 counts every hash against every bin (n_h*3*bins per visited IXF); threshold-aware
                 pruning requests fewer bytes with identical results, so the algorithmic rate is reported
                 separately (algorithmic_GBps, vs_dense) and `unpruned` holds the same steps with pruning switched
                 off, where requested == algorithmic and the contract formula is physical.
                 traffic = memory-side bytes per launch from two live rocprofv3 --pmc passes over one step of the same
                 workload (reads: the L2's read requests BY SIZE x their size -- every miss of this kernel is one 128-B
                 request, profiles/r03/fetch_calibration.txt; writes: WRITE_SIZE), run as child processes before this
                 process touches the GPU; null when rocprofv3 is unavailable (never a constant from a file).
                 requested_accounting bills the same launches three ways: useful16 / sector64 (= achieved, frac) / line128
                 (the 128-B lines HBM has to move; traffic is judged against this one).
  cpu_baseline : the CPU oracle (oracle/, a port of the reference path) timed on this box's host cores on a
                 bounded sample of the same reads and index; also used to re-check parity on that sample.
  sustained    : >= 10 M reads fed from HOST buffers through taxor_gpu_search_batch_begin/_end (PCIe inside),
                 rotating through the same distinct batches.  Top level: value_host_fed (= sustained.value, or the sum over
                 ranks at N > 1).  `value` itself stays the contract's figure -- K timed steps over HBM-resident batches (the
                 task statement: a PCIe-inclusive rate "is never `value`") -- and the two sit side by side in every line.
  value_e04    : the same index with reads at 4 % error (BASELINE.md section 3's workload; the default is 0.02, see
                 --read-error), resident and host-fed, detail under "read_error_0.04".
  roofline.contract_frac / moved_frac : SURVEY 8(d)'s formula on the kernel that does all of that work (= unpruned.frac) and
                 the bytes the memory side moved (= traffic_frac_of_peak), beside `frac` (requested bytes of the pruned kernel).
  layouts      : the headline's workload choices varied one at a time in the same invocation (N = 1): strand-mixed reads
                 (frac_reverse 0.5, SURVEY 8(d) as written), chopper-shaped children (as wide as the root) and a 4096-bin root --
                 value / frac / work per read for each, 3 timed steps, no PMC pass (moved_frac null); config.frac_reverse and
                 config.layout_note say what the headline itself runs.
  TAXOR_BENCH_FORCE_DIST=1 : every N > 1 branch with one rank (RCCL process group, probe gather, collectives) -- tests.
  N > 1        : pcie_inclusive_per_rank -- rank 0's host-fed rate alone, then every rank's at once, and
                 host_fed_scaling = their sum / rank 0's solo rate (the resident `value` scales by construction);
                 host_binding = the NUMA node each rank bound itself to before its first HIP call.
  --mode kmer|minimiser : the same line for an index built without --use-syncmer (viral-class footprint).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

# the HIP runtime maps streams onto this many hardware queues per device (default 4) and streams that share one run in
# submission order; a searcher's copy / hashing / query streams must not (taxor_amd/csrc/api.hip, runtime_env_once).  Set
# before torch brings the runtime up; a value the user exported wins.
# A rank of a distributed run also carries torch's and RCCL's streams: with 8 queues the searchers' streams then share queues with
# each other (measured: forced one-rank RCCL run 4.8 % behind the plain run with 2 resident batches, 1.6 % with 8; with 24 queues
# 0.7 % / 0.6 % -- profiles/r06/hw_queues_dist.txt; 32 and more lose a third, api.hip).
_DIST = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("TAXOR_BENCH_FORCE_DIST") == "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24" if _DIST else "8")
# several ranks on one node: this pool's host driver shares device memory between processes through dmabuf only, and
# RCCL's intra-node transport fails with `hipIpcGetMemHandle: invalid argument` under the legacy mode (exported on the
# boxes already; set here too so that a hand-built environment does not lose it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: root_bins, child_bins, n_children, total index bytes, reads/step, read_len
    # GTDB-220 k22/s12 is 113 GB (README.md:51); RefSeq-ABFV 9.9 GB (:52); Genbank-viral 373 MB (:50)
    # planted genomes are sized so that one step's reads cover them only ~2.5x (131072 x 10 kb over 384 Mbp):
    # little row reuse between reads, like a diverse metagenomic sample
    "gtdb": dict(root_bins=1024, child_bins=128, n_children=1020, total_bytes=113e9, root_frac=0.40,
                 reads=131072, read_len=10000, genomes=128, genome_len=3000000),
    "refseq": dict(root_bins=512, child_bins=64, n_children=508, total_bytes=9.9e9, root_frac=0.40,
                   reads=131072, read_len=10000, genomes=64, genome_len=2000000),
    "viral": dict(root_bins=256, child_bins=64, n_children=252, total_bytes=373e6, root_frac=0.40,
                  reads=131072, read_len=5000, genomes=64, genome_len=100000),
    "tiny": dict(root_bins=64, child_bins=32, n_children=8, total_bytes=8e6, root_frac=0.40,
                 reads=2048, read_len=3000, genomes=16, genome_len=50000),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("TAXOR_BENCH_WORKLOAD", "gtdb"), choices=sorted(WORKLOADS))
    ap.add_argument("--root-bins", type=int, default=0, help="root t_max (taxor_build.cpp:177-184: 64..4096); 0 = workload default")
    ap.add_argument("--child-bins", type=int, default=0, help="bins of every IXF below the root; 0 = workload default.  The reference applies ONE t_max "
                                                               "at every level (taxor_build.cpp:168-187,473): chopper-shaped = --child-bins equal to --root-bins")
    ap.add_argument("--frac-reverse", type=float, default=0.0,
                    help="fraction of the planted reads drawn from the reverse strand.  Default 0: with the reference's t = ceil((k-s+1)/2) in INTEGER "
                         "division (taxor_build.cpp:509-510; 5 at k22/s12) open-syncmer selection is not strand-symmetric, a reverse-strand read shares "
                         "no syncmers with a forward-indexed genome and stops at the root -- forward-only reads are the harder case.  SURVEY 8(d) says "
                         "'strand uniformly': the `layouts` array of the line carries that leg (0.5)")
    ap.add_argument("--no-layouts", action="store_true", help="skip the `layouts` legs (strand-mixed reads; chopper-shaped children; root 4096)")
    ap.add_argument("--reads", type=int, default=0, help="reads per step and GPU (0 = workload default)")
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--genomes", type=int, default=0, help="planted genomes (0 = workload default)")
    ap.add_argument("--genome-len", type=int, default=0)
    ap.add_argument("--family-size", type=int, default=16,
                    help="planted genomes come in families of this many related strains (99.7-93.6 %% identical) laid out in "
                         "adjacent bins of several child IXFs; 1 = unrelated genomes (the round-1 workload)")
    ap.add_argument("--read-error", type=float, default=0.02,
                    help="per-base error of the synthetic reads.  BASELINE.md section 3 names 0.04, but with uniformly placed errors only "
                         "(1-0.036)^22 = 45 %% of a read's 22-mers survive that, below the 0.508 the reference's model demands at "
                         "--error-rate 0.04, so nothing would classify; 0.02 (68 %% survive) exercises the whole traversal.  DESIGN.md "
                         "section 5 reports both.")
    ap.add_argument("--error-rate", type=float, default=0.04, help="taxor search --error-rate")
    ap.add_argument("--batches", type=int, default=8, help="distinct resident batches the steps rotate through")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--mode", default="syncmer", choices=("syncmer", "kmer", "minimiser", "build"),
                    help="hashing of the index: syncmer = k22/s12 open syncmers (the BASELINE configs); kmer = an index built WITHOUT "
                         "--use-syncmer, the reference's default build mode (every canonical 20-mer, k-mer threshold model); minimiser = "
                         "the same with window 32 (FracMinHash model).  kmer / minimiser run on the viral-class footprint with 5-kb reads.  "
                         "build = GPU construction of a hierarchy (SURVEY 8(f) #3): a step is one whole build, the line carries key insertions/s")
    ap.add_argument("--build-children", type=int, default=64, help="--mode build: child IXFs under the root")
    ap.add_argument("--build-child-bins", type=int, default=128)
    ap.add_argument("--build-keys-per-bin", type=int, default=422000, help="--mode build: keys per leaf bin (default: the leaf size of the gtdb workload)")
    ap.add_argument("--build-cpu-keys", type=int, default=200000,
                    help="--mode build: keys the reference's own builder (xorfilter.hpp AddAll) is timed on; beyond ~213 000 it never returns "
                         "(its deferred-block path is cut short by a debugging `break`, xorfilter.hpp:237-238, and its seed is fixed)")
    ap.add_argument("--len-mix", default="", help="'ont': skewed read lengths 1-100 kb (same total bases) instead of a fixed length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurements")
    ap.add_argument("--no-unpruned", action="store_true", help="skip the pruning-off pass")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the gather-ceiling microkernel")
    ap.add_argument("--no-e04", action="store_true", help="skip the extra leg at read error 0.04 (BASELINE.md section 3's workload)")
    ap.add_argument("--sustained-reads", type=int, default=10_000_000, help="host-fed reads of the `sustained` figure (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--traffic", default="live", choices=("live", "none"),
                    help="live: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) over one step before the run")
    ap.add_argument("--dump-results", default="", help="rank 0 writes the last step's (gathered) CSR results to this .npz (tests)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# live HBM traffic: rocprofv3 --pmc passes over one step of the same workload, as child processes, BEFORE this process
# initialises the GPU (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; FETCH_SIZE is in KB and
# reports half the bytes of wide coalesced reads on gfx950 -> doubled; counters get --kernel-trace only)
# ---------------------------------------------------------------------------------------------------------------------
def live_traffic(args):
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found", {}
    # this process is itself being profiled (rocprofv3 -- python3 bench.py ...): nested counter passes would inherit the
    # tool's environment and contend for the counters -- skip, and say so
    if any("rocprofiler" in os.environ.get(v, "") or "rocprofv3" in os.environ.get(v, "")
           for v in ("ROCP_TOOL_LIBRARIES", "LD_PRELOAD", "ROCPROFILER_REGISTER_FORCE_LOAD")) or os.environ.get("ROCP_TOOL_LIBRARIES"):
        return None, "skipped: this process runs under a rocprofiler tool already (use --traffic none when profiling bench.py)", {}
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", "1", "--warmup", "0",
             "--workload", args.workload, "--family-size", str(args.family_size), "--read-error", str(args.read_error),
             "--error-rate", str(args.error_rate), "--batches", "1", "--mode", args.mode]
    for flag, v in (("--reads", args.reads), ("--read-len", args.read_len), ("--genomes", args.genomes),
                    ("--genome-len", args.genome_len), ("--root-bins", args.root_bins)):
        if v:
            child += [flag, str(v)]
    if args.len_mix:
        child += ["--len-mix", args.len_mix]
    per_launch = {}
    launches = None
    tmp = tempfile.mkdtemp(prefix="taxor_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")

    def one_pass(counters):
        """-> ({counter: summed value over the k_query_level dispatches}, n dispatches) or (None, reason)"""
        d = os.path.join(tmp, "_".join(counters))
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--"] + child
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = proc.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            try:                       # the profiler AND the profiled child (its own session = its own process group)
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.communicate()
            return None, f"rocprofv3 --pmc {' '.join(counters)} pass timed out"
        if proc.returncode != 0:
            return None, f"rocprofv3 --pmc {' '.join(counters)} pass failed (rc {proc.returncode}): {err.decode(errors='replace')[-300:]}"
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            return None, f"rocprofv3 --pmc {' '.join(counters)}: no counter_collection.csv"
        tot, ids = {}, set()
        for f in files:
            for r in csv.DictReader(open(f)):
                if "k_query_level" in r.get("Kernel_Name", ""):
                    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    ids.add(r.get("Dispatch_Id"))
        if not ids or any(c not in tot for c in counters):
            return None, f"rocprofv3 --pmc {' '.join(counters)}: no k_query_level dispatch (or a counter missing) in the trace"
        return tot, len(ids)

    try:
        # READ bytes: the L2's memory-side read requests BY SIZE (32 / 64 / 128 B), so no correction factor is assumed.
        # profiles/r03/fetch_calibration*.json (known request counts in this kernel's access shapes): every miss is ONE
        # 128-B request -- a 1-KiB row is eight, a 64-B row is one, a lone 16-B load of the pruned phase is one -- and
        # FETCH_SIZE tallies each request as 64 B, so FETCH_SIZE x 2 is the same number (the fallback if this rocprofv3
        # lacks the by-size counters).
        t0 = time.time()
        by_size = ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
        tot, n = one_pass(by_size)
        detail = {}
        if tot is not None:
            n32, n64, n128 = tot["TCC_EA0_RDREQ_32B_sum"], tot["TCC_EA0_RDREQ_64B_sum"], tot["TCC_EA0_RDREQ_128B_sum"]
            other = max(0.0, tot["TCC_EA0_RDREQ_sum"] - n32 - n64 - n128)         # requests of no listed size: tallied at 64 B like FETCH_SIZE does
            per_launch["read"] = (32.0 * n32 + 64.0 * (n64 + other) + 128.0 * n128) / n
            detail = {"read_requests_per_launch": round(tot["TCC_EA0_RDREQ_sum"] / n, 1), "requests_128B": round(n128 / n, 1),
                      "requests_64B": round(n64 / n, 1), "requests_32B": round(n32 / n, 1)}
            read_src = "TCC_EA0_RDREQ by request size (32/64/128 B) x size"
        else:
            log("traffic: by-size request counters unavailable (" + str(n) + "); falling back to FETCH_SIZE x 2")
            tot, n = one_pass(["FETCH_SIZE"])
            if tot is None:
                return None, n, {}
            per_launch["read"] = tot["FETCH_SIZE"] * 1024.0 * 2.0 / n
            read_src = "FETCH_SIZE x 2 (every request is a 128-B line tallied at 64 B: profiles/r03/fetch_calibration.txt)"
        launches = n
        log(f"rocprofv3 read pass: {n} k_query_level launches, {per_launch['read']/1e9:.3f} GB read per launch, {time.time()-t0:.0f}s")
        t0 = time.time()
        tot, n = one_pass(["WRITE_SIZE"])
        if tot is None:
            return None, n, {}
        per_launch["write"] = tot["WRITE_SIZE"] * 1024.0 / n
        log(f"rocprofv3 write pass: {per_launch['write']/1e6:.3f} MB written per launch, {time.time()-t0:.0f}s")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    detail.update({"read_bytes_per_launch": round(per_launch["read"], 1), "write_bytes_per_launch": round(per_launch["write"], 1)})
    return per_launch["read"] + per_launch["write"], \
        f"live rocprofv3 --pmc passes over one step ({launches} launches) of this workload in this invocation: reads = {read_src}; " \
        f"writes = WRITE_SIZE", detail


def build_workload(args, local_rank, rank, world):
    """planted genomes, index resident in HBM, and this rank's distinct read batches"""
    from taxor_amd import GpuIndex, Searcher, synth
    if args.mode != "syncmer":
        return build_workload_no_syncmer(args, local_rank, rank, world)
    wl = dict(WORKLOADS[args.workload])
    if args.root_bins:
        wl["root_bins"] = args.root_bins
    if getattr(args, "child_bins", 0):
        wl["child_bins"] = args.child_bins
    n_reads = args.reads or wl["reads"]
    read_len = args.read_len or wl["read_len"]
    n_genomes = args.genomes or wl["genomes"]
    genome_len = args.genome_len or wl["genome_len"]
    fam_size = max(1, min(args.family_size, n_genomes))
    k, s, t = 22, 12, 5
    ncpu = len(os.sched_getaffinity(0)) or 8       # this rank's cores (bound to its GPU's NUMA node in main())

    # ---- planted genomes and their syncmer hashes (hashed on the GPU; same seed on every rank) -------------
    t0 = time.time()
    if fam_size > 1:
        n_genomes = (n_genomes // fam_size) * fam_size
        g, go, family = synth.family_genomes(n_genomes // fam_size, fam_size, genome_len, seed=synth.DEFAULT_SEED)
    else:
        g, go = synth.random_genomes(n_genomes, genome_len, seed=synth.DEFAULT_SEED)
        family = None
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins, k, s, t,
                     device=local_rank)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(n_genomes)]
    hs.close()
    dummy.close()
    log(f"{n_genomes} genomes x {genome_len} bp ({'families of %d' % fam_size if fam_size > 1 else 'unrelated'}) hashed on GPU: "
        f"{int(hoff[-1])} syncmers, {time.time()-t0:.1f}s")

    # ---- footprint-faithful layout -----------------------------------------------------------------------------
    t0 = time.time()
    spread = 4
    n_children = min(wl["n_children"], wl["root_bins"] - 8)
    root_rows = wl["total_bytes"] * wl["root_frac"] / max(64, wl["root_bins"])
    child_rows = wl["total_bytes"] * (1 - wl["root_frac"]) / ((n_children + 1) * max(64, wl["child_bins"]))
    root_max = max(int((root_rows - 32) / 1.23), 8)
    child_max = max(int((child_rows - 32) / 1.23), 8)
    biggest = max(len(p) for p in planted)
    if fam_size > 1:
        per_child = -(-fam_size // spread) + 1
        need_root = biggest * per_child
    else:
        need_root = max(sum(len(p) for p in planted[2:]) // max(1, min(n_children, len(planted) - 2)) * 2, biggest)
    root_max = max(root_max, need_root)
    child_max = max(child_max, biggest + 1024)
    # every child must hold the largest planted genome in one bin; where wide children (chopper-shaped: --child-bins = --root-bins)
    # would blow the footprint, there are fewer of them instead -- like a layout step that merges more user bins per child
    min_child_rows = 1.23 * child_max + 32
    fit = int(wl["total_bytes"] * (1 - wl["root_frac"]) / (min_child_rows * max(64, wl["child_bins"]))) - 1
    if fit < n_children:
        n_children = max(2 * spread, fit)
        log(f"children of {wl['child_bins']} bins: {n_children} of them fit the {wl['total_bytes']/1e9:.0f} GB footprint")
    if fam_size > 1:
        lay = synth.make_family_layout(planted, family, root_bins=wl["root_bins"], child_bins=wl["child_bins"],
                                       n_children=n_children, spread=spread, root_max_elems=root_max,
                                       child_max_elems=child_max, seed=synth.DEFAULT_SEED, build="gpu")
    else:
        lay = synth.make_layout(planted, root_bins=wl["root_bins"], child_bins=wl["child_bins"],
                                n_children=n_children, root_max_elems=root_max, child_max_elems=child_max,
                                seed=synth.DEFAULT_SEED, build="gpu")
    idx = synth.device_index(lay, k, s, t, device=local_rank)   # planted columns constructed on the GPU
    log(f"index in HBM: {idx.data_bytes/1e9:.2f} GB, {idx.n_ixf} IXFs, depth {idx.depth}, root {wl['root_bins']} bins, "
        f"children {wl['child_bins']} bins, {time.time()-t0:.1f}s")

    # ---- this rank's distinct batches (seed + rank + batch) ----------------------------------------------------
    t0 = time.time()
    from taxor_amd import distributed as td
    if args.scaling == "strong" and world > 1:
        lo, hi = td.shard_range(n_reads, rank, world)
    else:
        lo, hi = 0, n_reads
    batches, origins = [], []
    for b in range(max(1, args.batches)):
        seed = synth.DEFAULT_SEED + 1000 * b + (rank if args.scaling == "weak" else 0)
        if args.len_mix == "ont":
            # ONT-like skew with the same total bases: 1, 3, 10, 30, 100 kb carrying 10/20/40/20/10 % of the bases, shuffled
            total = n_reads * read_len
            parts = []
            for L, frac in ((1000, 0.1), (3000, 0.2), (10000, 0.4), (30000, 0.2), (100000, 0.1)):
                cnt = max(1, int(total * frac / L))
                bb, oo, _ = synth.synth_reads(g, go, cnt, L, error_rate=args.read_error, frac_random=0.1, seed=seed + L, threads=ncpu)
                parts += [(bb, int(oo[i]), int(oo[i + 1])) for i in range(cnt)]
            perm = np.random.default_rng(seed).permutation(len(parts))
            bases = np.concatenate([parts[i][0][parts[i][1]:parts[i][2]] for i in perm])
            offs = np.concatenate([[0], np.cumsum([parts[i][2] - parts[i][1] for i in perm])]).astype(np.uint64)
            origin = None
        else:
            bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=args.read_error, frac_random=0.1,
                                                    seed=seed, threads=ncpu, frac_reverse=getattr(args, "frac_reverse", 0.0))
        if args.scaling == "strong" and world > 1:  # this rank's contiguous shard of the common batch
            lo_, hi_ = (lo, hi) if not args.len_mix else td.shard_range(offs.size - 1, rank, world)
            bases, offs = bases[int(offs[lo_]):int(offs[hi_])], offs[lo_:hi_ + 1] - offs[lo_]
            origin = origin[lo_:hi_] if origin is not None else None
        batches.append((np.ascontiguousarray(bases), np.ascontiguousarray(offs)))
        origins.append(origin)
    log(f"{len(batches)} distinct batches of {batches[0][1].size - 1} reads x {read_len} bp generated ({ncpu} threads), {time.time()-t0:.1f}s")
    return wl, idx, lay, batches, dict(n_reads=n_reads, read_len=read_len, n_genomes=n_genomes, genome_len=genome_len,
                                       fam_size=fam_size, ncpu=ncpu, origins=origins, genomes=g, genome_off=go)


def build_workload_no_syncmer(args, local_rank, rank, world):
    """An index built without --use-syncmer (taxor build's default; taxor_search.cpp:210-212,239-260): every canonical k-mer
    (window == k) or window minimisers, on the viral-class footprint (root 256 bins, children 64), 5-kb reads.  Every k-mer
    of a read is a hash (4981 per 5-kb read against 435 syncmers), nothing is deduplicated, and the threshold is the
    reference's k-mer / FracMinHash model, evaluated on the host."""
    from taxor_amd import GpuIndex, Searcher, synth
    from taxor_amd import distributed as td
    k = 20
    w = 20 if args.mode == "kmer" else 32
    n_reads = args.reads or 32768
    read_len = args.read_len or 5000
    n_genomes = args.genomes or 32
    genome_len = args.genome_len or 100000
    ncpu = len(os.sched_getaffinity(0)) or 8
    t0 = time.time()
    g, go = synth.random_genomes(n_genomes, genome_len, seed=synth.DEFAULT_SEED)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64), fname_idx=np.arange(bins),
                           data=np.zeros(3 * 16 * 64, np.uint8))], bins, k=k, s=0, t=0, use_syncmer=False, window_size=w, device=local_rank)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    hs.close()
    dummy.close()
    planted = [np.unique(hashes[int(hoff[i]):int(hoff[i + 1])]) for i in range(n_genomes)]
    per_bin = max(len(p) for p in planted)
    wl = dict(root_bins=args.root_bins or 256, child_bins=64, n_children=252)
    lay = synth.make_layout(planted, root_bins=wl["root_bins"], child_bins=64, n_children=min(252, wl["root_bins"] - 4), root_max_elems=per_bin * 20,
                            child_max_elems=per_bin + 64, seed=synth.DEFAULT_SEED, build="gpu")
    idx = synth.device_index(lay, k=k, s=0, t=0, use_syncmer=False, window_size=w, device=local_rank)
    log(f"index built without syncmers (k={k}, window={w}) in HBM: {idx.data_bytes/1e9:.2f} GB, {idx.n_ixf} IXFs, {time.time()-t0:.1f}s")
    lo, hi = td.shard_range(n_reads, rank, world) if (args.scaling == "strong" and world > 1) else (0, n_reads)
    batches, origins = [], []
    for b in range(max(1, args.batches)):
        seed = synth.DEFAULT_SEED + 1000 * b + (rank if args.scaling == "weak" else 0)
        bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=args.read_error, frac_random=0.1, seed=seed, threads=ncpu)
        if args.scaling == "strong" and world > 1:
            bases, offs = bases[int(offs[lo]):int(offs[hi])], offs[lo:hi + 1] - offs[lo]
            origin = origin[lo:hi]
        batches.append((np.ascontiguousarray(bases), np.ascontiguousarray(offs)))
        origins.append(origin)
    return wl, idx, lay, batches, dict(n_reads=n_reads, read_len=read_len, n_genomes=n_genomes, genome_len=genome_len, fam_size=1, ncpu=ncpu,
                                       origins=origins, genomes=g, genome_off=go, k=k, s=0, t=0, window=w)


def build_cpu_reference(n_keys, salt):
    """cpu_baseline of --mode build: the REFERENCE's own xorfilter::XorFilter<uint64_t, uint8_t>::AddAll (src/main/xorfilter.hpp:142-334,
    compiled where it lies into oracle/_ref) on n_keys synthetic keys of one bin, one core, best of 3"""
    import ctypes as C
    from taxor_amd import synth
    so = os.path.join(ROOT, "oracle", "_ref", "libtaxor_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    L.ref_xor_build.restype = C.c_void_p
    L.ref_xor_build.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ref_xor_free.argtypes = [C.c_void_p]
    keys = synth.synth_keys_host(0, n_keys, salt)
    seed, bl, al = C.c_uint64(), C.c_uint64(), C.c_uint64()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        h = L.ref_xor_build(keys.ctypes.data_as(C.c_void_p), n_keys, C.byref(seed), C.byref(bl), C.byref(al))
        dt = time.perf_counter() - t0
        if not h:
            return None
        L.ref_xor_free(h)
        best = dt if best is None else min(best, dt)
    return {"value": round(n_keys / best, 1), "unit": "key insertions/s", "cores": 1, "kind": "reference",
            "sample": f"xorfilter::XorFilter<uint64_t,uint8_t>::AddAll (src/main/xorfilter.hpp:142-334, oracle/_ref) on {n_keys} synthetic keys of one "
                      "bin, best of 3; one filter at a time on one core is how the reference builds (construct_ixf.cpp:80-130)"}


# algorithmic bytes one key insertion moves in builder.hip (32-bit state words; DESIGN.md section 4 derives the sum):
#   key 8 B x 4 (count, round, assign, verify) + state word: 3 adds + 2 subs (4 B read + 4 B written each), 1 load, the seed scan over
#   1.23 slots + list entry 8 B written and read for 1.12 listed rows + log 4 + 4 + round marks: 2 B written per listed row, ~1 read
#   + fingerprints: 2 read + 1 written + 3 verified + 1.23 cleared
BUILD_BYTES_PER_INSERTION = 32 + 24 + 16 + 4 + 1.23 * 4 + 1.12 * 16 + 8 + 1.12 * 2 + 2 + 6 + 1.23
BUILD_RMW_PER_INSERTION = 5          # 3 adds (k_count; none for bins counted in LDS), 2 subs (k_round: the peeling row's own slot is not updated)
BUILD_RANDOM_ACCESSES_PER_INSERTION = 12   # loads / stores that are not atomics: key x 3, state word, round marks ~2, fingerprints 6
RMW_CEILING_G_PER_S = 27.1           # random 4-B atomics on a <= 256 MB set, any scope, returning or not (profiles/r06/atomics_bench.txt)


def build_mode(args):
    """bench.py --mode build: GPU construction of a two-level hierarchy in which every bin is a real filter.
    A step = one whole build (leaf IXFs, key unions, root) from keys resident in HBM; value = key insertions/s."""
    import torch  # noqa: F401  (first: its HIP runtime is the one the process uses)
    from taxor_amd import GpuIndex, Searcher, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or args.gpus > 1:
        raise SystemExit("--mode build is a single-GPU line: index construction does not shard across devices here (an index is built once and replicated)")
    nc, cb, kpb = args.build_children, args.build_child_bins, args.build_keys_per_bin
    shapes, ub, counts = synth.full_hierarchy_shapes(nc, cb, kpb)
    salt = synth.DEFAULT_SEED
    idx = GpuIndex(shapes, ub)
    steps, warmup = max(1, args.steps), max(0, args.warmup)
    sts = []
    for i in range(warmup + steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, off = idx.build_hixf_synth(counts, salt=salt, seed0=5 + i)
        torch.cuda.synchronize()
        st["wall_s"] = time.perf_counter() - t0
        if i >= warmup:
            sts.append(st)
        log(f"build {i}: {st['keys_inserted'] / st['seconds_total'] / 1e9:.3f} G insertions/s ({st['seconds_total']:.3f} s; peel {st['seconds_peel']:.3f}, "
            f"assign+verify {st['seconds_assign']:.3f}, unions {st['seconds_union']:.3f}, scratch alloc {st['seconds_alloc']:.3f}; count kernels {st['seconds_count']:.3f}, seed + rounds "
            f"{st['seconds_rounds']:.3f}; {st['chunks']} chunks, {st['rounds_max']} rounds, {st['reseeds']} reseeds)")
    ins = sum(s_["keys_inserted"] for s_ in sts)
    secs = sum(s_["seconds_total"] for s_ in sts)
    wall = sum(s_["wall_s"] for s_ in sts)
    value = ins / secs
    # spot check through the query kernel against the oracle's answer for the same keys (the full check is tests/test_gpu_build_fullsize.py)
    sr = Searcher(idx, ratio=0.5)
    rb = shapes[0]["bins"]
    for c, b in ((1, 0), (nc, cb - 1)):
        g = rb + (c - 1) * cb + b
        keys = synth.synth_keys_host(int(off[g]), int(off[g + 1] - off[g]), salt)[::8]      # (every 8th key of the bin: a spot check)
        own, up = sr.ixf_bulk_count(c, keys), sr.ixf_bulk_count(0, keys)
        if own[b] != keys.size or up[c - 1] != keys.size:
            raise SystemExit("PARITY FAILURE: a built bin does not hold its keys")
    sr.close()
    # the same from keys in HOST memory, as a binding has them (construct_ixf.cpp reads the bins' hashes from temp files): PCIe inside,
    # never `value`.  A smaller hierarchy (<= 16 children: 7 GB of keys), two builds.
    import ctypes as C
    host_fed = None
    if not args.no_dropin:
        from taxor_amd import _lib
        nc_h = min(nc, 16)
        shapes_h, ub_h, counts_h = synth.full_hierarchy_shapes(nc_h, cb, kpb)
        off_h = np.zeros(counts_h.size + 1, dtype=np.uint64)
        np.cumsum(counts_h, out=off_h[1:])
        n_h = int(off_h[-1])
        L = _lib.lib()
        d_tmp = C.c_void_p()
        _lib.check(L.taxor_gpu_malloc(0, n_h * 8, C.byref(d_tmp)))
        _lib.check(L.taxor_gpu_synth_keys(0, d_tmp, 0, n_h, salt + 1))
        hkeys = np.empty(n_h, dtype=np.uint64)
        _lib.check(L.taxor_gpu_memcpy_to_host(hkeys.ctypes.data_as(C.c_void_p), d_tmp, n_h * 8))
        L.taxor_gpu_free(d_tmp)
        idx_h = GpuIndex(shapes_h, ub_h)
        hs = [idx_h.build_hixf_host_keys(hkeys, off_h, seed0=77 + i) for i in range(2)][-1]
        idx_h.close()
        del hkeys
        host_fed = {"value": round(hs["keys_inserted"] / hs["seconds_total"], 1), "unit": "key insertions/s", "children": nc_h, "insertions": int(hs["keys_inserted"]),
                    "seconds_total": round(hs["seconds_total"], 4), "seconds_upload": round(hs["seconds_upload"], 4),
                    "upload_GBps": round(n_h * 8 / max(1e-9, hs["seconds_upload"]) / 1e9, 1),
                    "note": "taxor_gpu_index_build_hixf_ex with the keys in pageable host memory: allocation + upload (4 threads through page-locked staging) + build; "
                            "the second of two builds"}
        log(f"host-fed build: {host_fed['value'] / 1e9:.3f} G insertions/s ({host_fed['seconds_total']:.3f} s of which upload {host_fed['seconds_upload']:.3f} s = {host_fed['upload_GBps']} GB/s)")
    kern_s = sum(s_["seconds_peel"] + s_["seconds_assign"] for s_ in sts)
    t_count = max(1e-9, sum(s_["seconds_count"] for s_ in sts))
    t_rounds = max(1e-9, sum(s_["seconds_rounds"] for s_ in sts))
    ins_lds = sum(s_["keys_counted_in_lds"] for s_ in sts)            # keys whose degree words were built in LDS: no global adds
    rmw_total = 3 * (ins - ins_lds) + 2 * ins
    achieved = BUILD_BYTES_PER_INSERTION * ins / kern_s / 1e9
    out = {"metric": "key insertions/s (GPU IXF/HIXF construction, SURVEY 8(f) #3)", "value": round(value, 1), "unit": "key insertions/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(secs / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u64 keys -> u8 fingerprints", "data": "synthetic",
           "config": {"workload": f"hierarchy of {idx.n_ixf} IXFs, {idx.data_bytes / 1e9:.2f} GB: {nc} children x {cb} leaf bins x {kpb} keys (GTDB-class leaf size) under a root "
                                  f"of {nc} merged bins of {cb * kpb} keys; every bin built; {ins // steps} insertions per step",
                      "timed_region": "keys resident in HBM (generated on the device): count -> seed scan -> peeling rounds -> clear -> assignment in reverse -> "
                                      "verification of every key, for every chunk of bins; duplicate-free union of each child's keys for the root (a hash set in HBM); scratch allocation inside; "
                                      "the index shell's allocation and the release of scratch afterwards outside",
                      "index_bytes": idx.data_bytes, "n_ixf": idx.n_ixf, "children": nc, "child_bins": cb, "keys_per_bin": kpb,
                      "insertions_per_step": ins // steps, "chunks_per_step": sts[-1]["chunks"], "rounds_max": max(s_["rounds_max"] for s_ in sts),
                      "reseeds": sum(s_["reseeds"] for s_ in sts), "scratch_bytes": max(s_["scratch_bytes"] for s_ in sts)},
           "value_median_step": round(float(np.median([s_["keys_inserted"] / s_["seconds_total"] for s_ in sts])), 1),
           "step_seconds": [round(s_["seconds_total"], 4) for s_ in sts],
           "step_note": "`value` = all insertions / all step seconds (the contract); value_median_step = the median step's rate.  A build allocates tens of GB (keys' unions) "
                        "and now and then the driver takes seconds over such a hipMalloc and stalls the queue meanwhile -- such a step is in `value`, not in the median",
           "stage_s_per_step": {"peel": round(sum(s_["seconds_peel"] for s_ in sts) / steps, 4), "assign_verify": round(sum(s_["seconds_assign"] for s_ in sts) / steps, 4),
                                "unions": round(sum(s_["seconds_union"] for s_ in sts) / steps, 4), "total": round(secs / steps, 4),
                                "release_after": round(sum(s_["seconds_release"] for s_ in sts) / steps, 4), "wall": round(wall / steps, 4),
                                "note": "total = call -> last IXF built and verified; release_after = hipFree of keys, unions and scratch, outside `value`; wall also "
                                        "holds the generation of the synthetic keys"},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "kernel": "k_count + k_round + k_assign + k_verify (builder.hip)",
                        "algorithmic_bytes_per_insertion": round(BUILD_BYTES_PER_INSERTION, 1),
                        "note": "achieved = algorithmic bytes per insertion x insertions / (peel + assign + verify time).  The builder is NOT bound by HBM bytes: "
                                "every one of its accesses is a random 1-8-byte access, and what the chip limits is their NUMBER -- see `rmw`",
                        "rmw": {"per_insertion": round(rmw_total / ins, 3), "per_insertion_by_design": BUILD_RMW_PER_INSERTION,
                                "insertions_counted_in_lds": int(ins_lds // steps),
                                "achieved_G_per_s": round(rmw_total / kern_s / 1e9, 2),
                                "ceiling_G_per_s": RMW_CEILING_G_PER_S, "frac": round(rmw_total / kern_s / 1e9 / RMW_CEILING_G_PER_S, 4),
                                "random_loads_stores_per_insertion": BUILD_RANDOM_ACCESSES_PER_INSERTION,
                                "kernels": [
                                    {"kernel": "k_count (+ k_count_lds)", "rmw_per_insertion": 3, "seconds_per_step": round(t_count / steps, 4),
                                     "achieved_G_per_s": round(3 * (ins - ins_lds) / t_count / 1e9, 2),
                                     "frac": round(3 * (ins - ins_lds) / t_count / 1e9 / RMW_CEILING_G_PER_S, 4),
                                     "keys_G_per_s": round(ins / t_count / 1e9, 2),
                                     "note": "3 global atomic adds per key for the bins that go through k_count (more than ~1 M keys -- too many LDS passes --, or < 16 k keys); the others' "
                                             "degree words are built in LDS by k_count_lds, in the same event bracket: achieved / frac count the global adds only"},
                                    {"kernel": "k_seed + k_round", "rmw_per_insertion": 2, "seconds_per_step": round(t_rounds / steps, 4),
                                     "achieved_G_per_s": round(2 * ins / t_rounds / 1e9, 2), "frac": round(2 * ins / t_rounds / 1e9 / RMW_CEILING_G_PER_S, 4),
                                     "note": "plus ~4 random loads / stores per insertion (state word, key, round marks) and the list traffic"}],
                                "kernels_note": "HIP events on the builder's stream inside the library (taxor_build_stats::seconds_count, seconds_rounds), summed over the chunks",
                                "note": "random atomic read-modify-writes per second against the chip's measured rate for them (profiles/r06/atomics_bench.txt: 27 G/s "
                                        "up to 256 MB, 18-20 G/s beyond, whatever the scope, width or use of the return value); the builder also does ~12 random loads "
                                        "and stores per insertion (54 G/s ceiling), so 5 RMW / 20-27 G/s + 12 / 54 G/s = 0.41-0.47 ns per insertion is the floor of this "
                                        "design: 2.1-2.4 G insertions/s in the peeling + assignment kernels"}}}
    if host_fed is not None:
        out["pcie_inclusive"] = host_fed
        out["value_host_fed"] = host_fed["value"]
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = build_cpu_reference(args.build_cpu_keys, salt)
    idx.close()
    print(json.dumps(out), flush=True)


def main():
    args = parse_args()
    if args.mode == "build":
        return build_mode(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # TAXOR_BENCH_FORCE_DIST=1: take every `world > 1` branch with WORLD_SIZE = 1 -- process group on RCCL with a device id,
    # the probe gather on device tensors, all_gather / all_reduce / barrier, the per-step result gather, the solo-then-
    # concurrent host-fed block, destroy_process_group -- so that ONE GPU executes the code the 8-GPU launch runs
    # (tests/test_gpu_bench_multirank.py).  The line it prints must agree with the plain N = 1 run.
    force_dist = os.environ.get("TAXOR_BENCH_FORCE_DIST") == "1"
    dist_on = world > 1 or force_dist

    # Bind this rank's host threads (and, by first touch, its staging buffers) to the NUMA node of its GPU before
    # anything touches the GPU: eight ranks feeding eight GPUs from host memory otherwise stage across the socket
    # fabric and share cores at random.  sysfs only, no numactl, no re-exec.
    from taxor_amd import numa
    full_affinity = os.sched_getaffinity(0)
    numa_info = {"bound": False, "reason": "TAXOR_BENCH_NUMA=0"}
    if os.environ.get("TAXOR_BENCH_NUMA", "1") != "0" and not args.pmc_child:
        numa_info = numa.bind_to_gpu(0 if os.environ.get("TAXOR_BENCH_SAME_GPU") == "1" else local_rank)

    # live PMC passes run as child processes before anything here touches the GPU
    traffic, traffic_src, traffic_detail = None, "not collected (--traffic none, a child run, or N > 1)", {}
    if args.traffic == "live" and not args.pmc_child and world == 1:
        try:
            traffic, traffic_src, traffic_detail = live_traffic(args)
        except Exception as e:      # the profiler is an aid: its failure must not take the benchmark down
            traffic, traffic_src, traffic_detail = None, f"rocprofv3 pass raised {type(e).__name__}: {e}", {}
        if traffic is None:
            log("traffic:", traffic_src)

    import torch                      # first: its bundled HIP runtime is the one the process uses
    import torch.distributed as dist

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # TAXOR_BENCH_BACKEND=gloo + TAXOR_BENCH_SAME_GPU=1 exercise the N>1 flow on a single-GPU box (tests only)
    backend = os.environ.get("TAXOR_BENCH_BACKEND", "nccl")
    if os.environ.get("TAXOR_BENCH_SAME_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:      # forced single-rank group launched without torch.distributed.run
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # Fail loudly and early: a rendezvous or first-transfer failure must end the job with one readable line and a
        # non-zero exit code, not a hang in the timed region or a silent change of transport (no fallback, no re-exec).
        try:
            import datetime
            tmo = datetime.timedelta(seconds=float(os.environ.get("TAXOR_BENCH_RDZV_TIMEOUT", "600")))
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
            else:
                dist.init_process_group(backend, timeout=tmo)
            from taxor_amd import distributed as td_
            dev_ = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
            probe = td_.gather_csr(torch.tensor([0, 1], dtype=torch.int64, device=dev_), torch.tensor([rank], dtype=torch.int64, device=dev_),
                                   torch.tensor([rank], dtype=torch.int32, device=dev_), torch.tensor([1], dtype=torch.int32, device=dev_), dst=0)
            if rank == 0 and [int(x) for x in probe[1].cpu()] != list(range(world)):
                raise RuntimeError(f"first point-to-point gather returned {probe[1].tolist()}")
            if backend == "nccl":
                torch.cuda.synchronize()
        except Exception as e:
            print(f"[bench] FATAL rank {rank}/{world}: {backend} ({'RCCL' if backend == 'nccl' else backend}) process group or first "
                  f"point-to-point gather failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            os._exit(13)

    from taxor_amd import Searcher

    wl, idx, lay, batches, info = build_workload(args, local_rank, rank, world)
    n_reads, read_len, ncpu = info["n_reads"], info["read_len"], info["ncpu"]
    shard_reads = [b[1].size - 1 for b in batches]
    shard_bases = [int(b[1][-1]) for b in batches]

    # one searcher per resident batch: each keeps its packed reads (and its scratch) in HBM
    searchers = []
    for bases, offs in batches:
        sr = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
        sr.upload(bases, offs)
        sr.run()          # set-up, untimed: sizes this searcher's result buffers for its batch (an overflow reruns the
        sr.sync()         # batch once with grown buffers; that must not land in another searcher's timed step)
        searchers.append(sr)

    from taxor_amd import distributed as td
    gathered = {}
    comm_acc = {"ms": 0.0, "n": 0, "bytes": 0, "sent": 0, "events": [], "export_host_ms": 0.0, "gather_host_ms": 0.0}

    def gather_results(sr):
        """per-read results of every rank -> rank 0 over RCCL (point-to-point, one xGMI link per peer)"""
        if not dist_on:
            return
        t_x = time.perf_counter()
        nr, nt = sr.result_sizes()
        dev = torch.device("cuda", local_rank)
        ro = torch.empty(nr + 1, dtype=torch.int64, device=dev)
        ub = torch.empty(nt, dtype=torch.int64, device=dev)
        ct = torch.empty(nt, dtype=torch.int32, device=dev)
        nh = torch.empty(nr, dtype=torch.int32, device=dev)
        sr.export_device(ro.data_ptr(), ub.data_ptr() if nt else None, ct.data_ptr() if nt else None,
                         nh.data_ptr() if nr else None)
        if backend != "nccl":
            ro, ub, ct, nh = ro.cpu(), ub.cpu(), ct.cpu(), nh.cpu()
        # HIP events around the exchange on the stream torch's collectives are ordered against (device tensors); wall clock for gloo
        if backend == "nccl":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        t_g = time.perf_counter()
        comm_acc["export_host_ms"] += (t_g - t_x) * 1e3
        gathered["last"] = td.gather_csr(ro, ub, ct, nh, dst=0)
        comm_acc["gather_host_ms"] += (time.perf_counter() - t_g) * 1e3
        if backend == "nccl":
            e1.record()
            comm_acc["events"].append((e0, e1))        # read after the timed loop's final synchronize: no host sync per step for the clock
        else:
            comm_acc["ms"] += (time.perf_counter() - t_g) * 1e3
        comm_acc["n"] += 1
        comm_acc["bytes"] += td.last_gather["bytes_received"]
        comm_acc["sent"] += sum(x.numel() * x.element_size() for x in (ro, ub, ct, nh))

    def step(i, pool):
        sr = pool[i % len(pool)]
        sr.run()
        sr.sync()
        gather_results(sr)
        return sr

    def timed(pool, steps, warmup):
        for i in range(warmup):
            step(i, pool)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = dict(q_ms=0.0, q_bytes=0.0, q_touched=0.0, launches=0, bases=0, sync_ms=0.0, query_ms=0.0, fin_ms=0.0, total_ms=0.0,
                   hashes=0, tuples=0, work=0, reads=0, alg=0, lvl_ms=[0.0] * 8, lvl_bytes=[0] * 8, lvl_rows=[0] * 8, lvl_sparse=[0] * 8)
        comm_acc.update(ms=0.0, n=0, bytes=0, sent=0, events=[], export_host_ms=0.0, gather_host_ms=0.0)
        acc["local_s"] = 0.0
        for i in range(steps):
            t_s = time.perf_counter()
            sr = step(warmup + i, pool)
            acc["local_s"] += time.perf_counter() - t_s
            st = sr.stats()
            acc["q_ms"] += st["query_ms"]
            acc["q_bytes"] += st["query_bytes"]
            acc["q_touched"] += st["query_touched_bytes"]
            acc["launches"] += st["query_launches"]
            acc["bases"] += st["n_bases"]
            acc["sync_ms"] += st["syncmer_ms"]
            acc["fin_ms"] += st["finalize_ms"]
            acc["total_ms"] += st["total_ms"]
            acc["hashes"] += st["n_hashes"]
            acc["tuples"] += st["n_tuples"]
            acc["work"] += st["n_work_items"]
            acc["reads"] += st["n_reads"]
            acc["alg"] += st["algorithmic_bytes"]
            for l in range(8):
                acc["lvl_ms"][l] += st["level_ms"][l]
                acc["lvl_bytes"][l] += st["level_requested_bytes"][l]
                acc["lvl_rows"][l] += st["level_row_reads"][l]
                acc["lvl_sparse"][l] += st["level_sparse_loads"][l]
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        comm_acc["ms"] += sum(a_.elapsed_time(b_) for a_, b_ in comm_acc["events"])
        if dist_on:
            dev_t = torch.device("cuda", local_rank) if backend == "nccl" else "cpu"
            tt = torch.tensor([elapsed, float(acc["bases"])], dtype=torch.float64, device=dev_t)
            dist.all_reduce(tt[:1], op=dist.ReduceOp.MAX)
            dist.all_reduce(tt[1:], op=dist.ReduceOp.SUM)
            elapsed, acc["all_bases"] = float(tt[0].item()), float(tt[1].item())
            mine_t = torch.tensor([acc["local_s"] / max(1, steps) * 1e3, comm_acc["ms"] / max(1, comm_acc["n"]), float(comm_acc["sent"]) / max(1, comm_acc["n"])],
                                  dtype=torch.float64, device=dev_t)
            all_t = [torch.zeros_like(mine_t) for _ in range(dist.get_world_size())]
            dist.all_gather(all_t, mine_t)
            acc["per_rank_ms"] = [float(v[0]) for v in all_t]
            acc["per_rank_gather_ms"] = [float(v[1]) for v in all_t]
            acc["per_rank_sent_bytes"] = [float(v[2]) for v in all_t]
            acc["gather_bytes_received"] = comm_acc["bytes"] / max(1, comm_acc["n"])
            acc["export_host_ms"] = comm_acc["export_host_ms"] / max(1, comm_acc["n"])
            acc["gather_host_ms"] = comm_acc["gather_host_ms"] / max(1, comm_acc["n"])
            acc["gather_sizes"] = list(td.last_gather["sizes"])
        else:
            acc["all_bases"] = float(acc["bases"])
        return elapsed, acc

    elapsed, acc = timed(searchers, args.steps, args.warmup)
    if dist_on and rank == 0:
        g_off, g_ub, g_cnt, g_nh = gathered["last"]
        assert g_off.numel() == g_nh.numel() + 1 and int(g_off[-1]) == g_ub.numel() == g_cnt.numel()
        want = n_reads if args.scaling == "strong" else n_reads * world
        assert g_nh.numel() == want, (g_nh.numel(), want)
    value = acc["all_bases"] / elapsed / 1e6
    if args.dump_results and rank == 0:
        if dist_on:
            d_off, d_ub, d_cnt, d_nh = (x.cpu().numpy() for x in gathered["last"])
        else:
            r_ = searchers[(args.warmup + args.steps - 1) % len(searchers)].fetch()
            d_off, d_ub, d_cnt, d_nh = r_.read_off, r_.user_bin, r_.count, r_.n_hashes
        np.savez(args.dump_results, read_off=d_off.astype(np.int64), user_bin=d_ub.astype(np.int64), count=d_cnt.astype(np.int64),
                 n_hashes=d_nh.astype(np.int64))

    # N > 1 (or the forced one-rank group): the same invocation also runs three STRONG-scaling steps -- one common batch, rank r
    # classifies its contiguous shard, the shards are gathered -- and rank 0 classifies the whole batch alone: the gathered CSR must
    # be the single-rank CSR, bit for bit.  That holds only if every rank's replica of the index is the same (the builder is
    # deterministic) and every rank's tuples arrived in rank order.
    strong_leg = None
    if dist_on and args.mode == "syncmer" and not args.len_mix and not args.pmc_child:
        import hashlib
        from taxor_amd import synth as synth_s

        def digest(ro, ub, ct, nh):
            h_ = hashlib.sha256()
            for x in (ro, ub, ct, nh):
                h_.update(np.ascontiguousarray(np.asarray(x).astype(np.int64)).tobytes())
            return h_.hexdigest()

        wsz = dist.get_world_size()
        cb_, co_, _ = synth_s.synth_reads(info["genomes"], info["genome_off"], n_reads, read_len, error_rate=args.read_error, frac_random=0.1,
                                          seed=synth_s.DEFAULT_SEED + 424242, threads=ncpu)
        lo_, hi_ = td.shard_range(n_reads, rank, wsz)
        s_sh = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
        s_sh.upload(np.ascontiguousarray(cb_[int(co_[lo_]):int(co_[hi_])]), np.ascontiguousarray(co_[lo_:hi_ + 1] - co_[lo_]))
        s_sh.run()
        s_sh.sync()
        e_s, a_s = timed([s_sh], 3, 1)
        s_sh.close()
        if rank == 0:
            g_ = [x.cpu().numpy() for x in gathered["last"]]
            s_full = Searcher(idx, error_rate=args.error_rate)
            r_full = s_full.search_batch(np.ascontiguousarray(cb_), np.ascontiguousarray(co_))
            d_g, d_1 = digest(*g_), digest(r_full.read_off, r_full.user_bin, r_full.count, r_full.n_hashes)
            s_full.close()
            strong_leg = {"scaling": "strong", "steps": 3, "reads": int(n_reads), "value": round(a_s["all_bases"] / e_s / 1e6, 2), "unit": "Mbp/s",
                          "ms_per_step": round(e_s / 3 * 1e3, 3), "tuples": int(g_[1].size), "digest_gathered": d_g, "digest_single_rank": d_1, "equal": d_g == d_1,
                          "note": "one common batch, rank r classifies reads [r*n/N, (r+1)*n/N), gathered on rank 0; digest = sha256 over read_off, user_bin, "
                                  "count, n_hashes; digest_single_rank = rank 0 classifying the whole batch alone in the same process"}
            if d_g != d_1:
                print(json.dumps({"PARITY FAILURE": "the gathered strong-scaling CSR differs from the single-rank result", "strong_leg": strong_leg}), file=sys.stderr, flush=True)
                os._exit(14)
        del cb_, co_
        dist.barrier()

    # N > 1: the drop-in call with HOST buffers on every rank at the same time -- the ranks share the host's memory
    # bandwidth and PCIe root complexes, which is where a sharded run is won or lost (the resident-batch `value` is not)
    # The resident-batch `value` scales with N by construction; what decides >= 6x at N = 8 is the host side.  So rank 0
    # first runs the host-fed measurement ALONE (the other ranks wait at a barrier: the N = 1 condition inside this very
    # job), then every rank runs it at once; host_fed_scaling = sum of the concurrent per-rank rates / rank 0's solo rate.
    per_rank = None
    if dist_on and not args.no_dropin and not args.pmc_child:
        small = argparse.Namespace(**vars(args))
        small.sustained_reads = max(1, args.sustained_reads // 4)

        def host_fed():
            try:
                return dropin_measurements(small, idx, batches, read_len)
            except Exception as e:   # an aid: its failure on one rank must not desynchronise the collectives below
                log(f"host-fed measurement failed: {type(e).__name__}: {e}")
                return {"value": 0.0}, None

        # every rank once, untimed and short: page-locking the staging buffers, the searchers' first allocations and the first
        # launches are one-time costs that would otherwise sit in rank 0's solo figure only and inflate host_fed_scaling
        warm = argparse.Namespace(**vars(small))
        warm.sustained_reads = max(1, small.sustained_reads // 8)
        try:
            dropin_measurements(warm, idx, batches, read_len, single=False)
        except Exception as e:
            log(f"host-fed warm-up failed: {type(e).__name__}: {e}")
        dist.barrier()
        solo = host_fed() if rank == 0 else None
        dist.barrier()
        single, sustained = host_fed()
        mine = torch.tensor([single["value"], sustained["value"] if sustained else 0.0, float(numa_info.get("numa_node", -1)),
                             float(numa_info.get("cpus", 0))], dtype=torch.float64,
                            device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        if rank == 0:
            solo_sus = solo[1]["value"] if solo and solo[1] else 0.0
            ssum = float(sum(v[1] for v in allv))
            per_rank = {"single_call_Mbp_s": [round(float(v[0]), 1) for v in allv], "sustained_Mbp_s": [round(float(v[1]), 1) for v in allv],
                        "sustained_sum_Mbp_s": round(ssum, 1),
                        "solo_rank0": {"single_call_Mbp_s": round(float(solo[0]["value"]), 1) if solo else None, "sustained_Mbp_s": round(solo_sus, 1)},
                        "host_fed_scaling": round(ssum / solo_sus, 3) if solo_sus > 0 else None,
                        "numa_node": [int(v[2]) for v in allv], "bound_cpus": [int(v[3]) for v in allv],
                        "note": "taxor_gpu_search_batch on host buffers (PCIe inside): rank 0 alone while the others wait, then all "
                                "ranks at once; host_fed_scaling = sum of the concurrent per-rank sustained rates / rank 0's solo rate"}

    if args.pmc_child:                 # profiled child: the launches above are all the parent wanted
        for sr in searchers:
            sr.close()
        idx.close()
        return

    out = None
    if rank == 0:
        q_s = acc["q_ms"] * 1e-3
        requested = acc["q_touched"] / q_s / 1e9 if q_s > 0 else 0.0
        algorithmic = acc["q_bytes"] / q_s / 1e9 if q_s > 0 else 0.0
        launches = max(1, acc["launches"])
        roof = {"bound": "hbm", "achieved": round(requested, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(requested / HBM_PEAK_GBS, 4),
                "traffic": None if traffic is None else round(traffic, 1), "traffic_source": traffic_src,
                "kernel": "k_query_level",
                "note": "achieved = bytes k_query_level requested (device-counted) / its HIP-event time on the searcher's "
                        "stream.  algorithmic_* = the reference's n_h*3*bins per visited IXF (SURVEY 8(d)); pruning asks for "
                        "fewer bytes with identical tuples, so vs_dense > 1 is saved work, not bandwidth.  `unpruned`: same "
                        "steps with pruning off, where the contract formula (algorithmic bytes / time) is physical.  Below the "
                        "root, items are grouped by IXF and re-read children partly out of L2 / the memory-side cache (levels[].frac "
                        "can then exceed what HBM alone would deliver); traffic (PMC) counts L2 misses.",
                "launches": acc["launches"], "avg_launch_ms": round(acc["q_ms"] / launches, 4),
                "requested_bytes_per_launch": round(acc["q_touched"] / launches, 1),
                "algorithmic_bytes_per_launch": round(acc["q_bytes"] / launches, 1),
                "algorithmic_GBps": round(algorithmic, 1),
                "vs_dense": round(acc["q_bytes"] / max(1.0, acc["q_touched"]), 4)}
        # The same launches billed three ways.  A dense-phase row read asks for the 16-B units that hold bins; a load of the
        # pruned (sparse) phase is 16 useful bytes.  What the memory system moves for either is whole 128-B lines (measured:
        # profiles/r03/fetch_calibration.txt -- one 128-B request per line, also for a 64-B row and for a lone 16-B load).
        #   useful16 : dense bytes + 16 B per sparse load          (what the arithmetic consumes)
        #   sector64 : dense bytes + 64 B per sparse load          (= requested_bytes_per_launch / achieved / frac, as in round 2)
        #   line128  : 128-B lines touched by either               (what HBM has to deliver; `traffic` is judged against this)
        sparse = sum(acc["lvl_sparse"])
        dense_bytes = acc["q_touched"] - 64.0 * sparse
        lvl_row_bytes = [((wl["root_bins"] if l == 0 else wl["child_bins"]) + 15) // 16 * 16 for l in range(8)]
        line128 = 0.0
        for l in range(8):
            dense_rows = acc["lvl_rows"][l] - acc["lvl_sparse"][l]
            line128 += 128.0 * (dense_rows * -(-lvl_row_bytes[l] // 128) + acc["lvl_sparse"][l])
        roof["requested_accounting"] = {
            "useful16_bytes_per_launch": round((dense_bytes + 16.0 * sparse) / launches, 1),
            "sector64_bytes_per_launch": round(acc["q_touched"] / launches, 1),
            "line128_bytes_per_launch": round(line128 / launches, 1),
            "frac_useful16": round((dense_bytes + 16.0 * sparse) / q_s / 1e9 / HBM_PEAK_GBS, 4) if q_s > 0 else 0.0,
            "frac_sector64": roof["frac"],
            "frac_line128": round(line128 / q_s / 1e9 / HBM_PEAK_GBS, 4) if q_s > 0 else 0.0,
            "sparse_loads_per_launch": round(sparse / launches, 1),
            "note": "line128 is what the memory system must move (one 128-B request per line touched); a 16-B load of the pruned phase "
                    "and a 64-B row each cost a whole line"}
        if traffic is not None:
            roof["traffic_over_requested"] = round(traffic / max(1.0, acc["q_touched"] / launches), 4)
            roof["traffic_over_line128"] = round(traffic / max(1.0, line128 / launches), 4)
            roof["traffic_GBps"] = round(traffic / (acc["q_ms"] / launches * 1e-3) / 1e9, 1)
            roof["traffic_frac_of_peak"] = round(roof["traffic_GBps"] / HBM_PEAK_GBS, 4)
            roof["traffic_detail"] = traffic_detail

        # per HIXF level: wide rows are bound by bytes, rows of <= 128 B by the number of DRAM rows opened per second
        roof["levels"] = [{"level": l, "ms_per_step": round(acc["lvl_ms"][l] / args.steps, 3),
                           "requested_GBps": round(acc["lvl_bytes"][l] / (acc["lvl_ms"][l] * 1e-3) / 1e9, 1),
                           "frac": round(acc["lvl_bytes"][l] / (acc["lvl_ms"][l] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "row_reads_G_per_s": round(acc["lvl_rows"][l] / (acc["lvl_ms"][l] * 1e-3) / 1e9, 2),
                           "bytes_per_row_read": round(acc["lvl_bytes"][l] / max(1, acc["lvl_rows"][l]), 1)}
                          for l in range(8) if acc["lvl_ms"][l] > 0]

        # ---- the contract formula on a kernel that does all the algorithmic work: pruning off -------------------
        if not args.no_unpruned and world == 1:
            pool = []
            for bases, offs in batches[:2]:
                s2 = Searcher(idx, error_rate=args.error_rate, time_kernels=True, prune=False)
                s2.upload(bases, offs)
                pool.append(s2)
            e2, a2 = timed(pool, max(2, min(args.steps, 4)), 1)
            q2 = a2["q_ms"] * 1e-3
            assert a2["q_touched"] == a2["q_bytes"]
            roof["unpruned"] = {"achieved": round(a2["q_bytes"] / q2 / 1e9, 1), "frac": round(a2["q_bytes"] / q2 / 1e9 / HBM_PEAK_GBS, 4),
                                "avg_launch_ms": round(a2["q_ms"] / max(1, a2["launches"]), 4),
                                "algorithmic_bytes_per_launch": round(a2["q_bytes"] / max(1, a2["launches"]), 1),
                                "value_Mbp_s": round(a2["all_bases"] / e2 / 1e6, 2),
                                "note": "taxor_gpu_search_params.flags = TAXOR_SEARCH_NO_PRUNE: every hash against every bin of every visited IXF, like the reference"}
            ra, rb = pool[0].fetch(), searchers[0].fetch()      # both hold batch 0: pruning must not change a single tuple
            same = (np.array_equal(ra.read_off, rb.read_off) and np.array_equal(ra.user_bin, rb.user_bin)
                    and np.array_equal(ra.count, rb.count) and np.array_equal(ra.n_hashes, rb.n_hashes))
            if not same:
                raise SystemExit("PARITY FAILURE: pruning changed the results")
            for s2 in pool:
                s2.close()

        # the three readings of the roofline side by side at top level, so that they cannot be confused:
        #   frac          requested bytes (sector64 accounting) of the pruned kernel / its time / peak  -- what the kernel asks for
        #   contract_frac SURVEY 8(d)'s formula, n_h*3*bins per visited IXF / time / peak, on the kernel that does ALL of
        #                 that work (pruning off): the contract's number
        #   moved_frac    bytes the memory side actually moved (PMC request counters) / time / peak
        roof["contract_frac"] = roof.get("unpruned", {}).get("frac")
        roof["moved_frac"] = roof.get("traffic_frac_of_peak")

        # ---- BASELINE.md section 3's read error (0.04) beside the default 0.02 (why 0.02: --read-error's help text) ---------
        e04 = None
        if world == 1 and not args.no_e04 and args.mode == "syncmer" and not args.len_mix and abs(args.read_error - 0.04) > 1e-9:
            from taxor_amd import synth as synth_
            t0_ = time.time()
            pool, b04 = [], []
            for b in range(2):
                bb, oo, _ = synth_.synth_reads(info["genomes"], info["genome_off"], n_reads, read_len, error_rate=0.04, frac_random=0.1,
                                               seed=synth_.DEFAULT_SEED + 77000 + b, threads=ncpu)
                bb, oo = np.ascontiguousarray(bb), np.ascontiguousarray(oo)
                b04.append((bb, oo))
                s2 = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
                s2.upload(bb, oo)
                s2.run()
                s2.sync()
                pool.append(s2)
            e4, a4 = timed(pool, max(2, min(args.steps, 6)), 2)
            q4 = a4["q_ms"] * 1e-3
            e04 = {"read_error": 0.04, "value": round(a4["all_bases"] / e4 / 1e6, 2), "unit": "Mbp/s",
                   "frac": round(a4["q_touched"] / q4 / 1e9 / HBM_PEAK_GBS, 4) if q4 > 0 else None,
                   "tuples_per_read": round(a4["tuples"] / max(1, a4["reads"]), 3), "work_items_per_read": round(a4["work"] / max(1, a4["reads"]), 3),
                   "note": "the same index and searcher settings, reads with 4 % errors (BASELINE.md section 3): few of them keep enough "
                           "22-mers to pass the 0.508 the reference's model demands, so most stop at the root -- the easier case"}
            for s2 in pool:
                s2.close()
            e04["_batches"] = b04          # host-fed below, once the resident searchers (and their streams) are gone
            log(f"read error 0.04 leg: {e04['value']:.0f} Mbp/s resident, {time.time()-t0_:.1f}s")

        # measured gather ceiling: random whole-row reads of the same IXFs by a kernel that does nothing else
        if not args.no_ceiling:
            ceiling = {}
            gbps, row_bytes = idx.gather_ceiling(0, want_bytes=16 << 30, reps=3)
            ceiling["root"] = {"ixf": 0, "row_bytes": row_bytes, "GBps": round(gbps, 1), "row_reads_G_per_s": round(gbps / row_bytes, 2)}
            if idx.n_ixf > 1:          # all equally shaped children at once: one of them alone would sit in the caches
                gbps, row_bytes, used = idx.gather_ceiling(1, want_bytes=16 << 30, reps=3, span=idx.n_ixf - 1)
                ceiling["children"] = {"first_ixf": 1, "ixfs": used, "row_bytes": row_bytes, "GBps": round(gbps, 1),
                                       "row_reads_G_per_s": round(gbps / row_bytes, 2)}
            roof["gather_ceiling"] = ceiling

        last = searchers[(args.warmup + args.steps - 1) % len(searchers)]
        res = last.fetch()
        st = last.stats()
        classified = int(((res.read_off[1:] - res.read_off[:-1]) > 0).sum())
        nr = max(1, acc["reads"])
        fam = f"{info['n_genomes'] // info['fam_size']} families x {info['fam_size']} strains" if info["fam_size"] > 1 else f"{info['n_genomes']} unrelated genomes"
        out = {
            "metric": "Mbp/s classified (taxor search) vs GTDB k22/s12", "value": round(value, 2), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"{args.workload}-class HIXF k22/s12 {idx.data_bytes/1e9:.0f} GB in HBM, root {wl['root_bins']} bins, "
                                    + (f"{shard_reads[0]} reads x {int(read_len)} bp/GPU/step, {fam}" if not args.len_mix else
                                       f"{shard_reads[0]} reads of 1-100 kb (ONT-like mix: 1/3/10/30/100 kb carrying 10/20/40/20/10 % of the bases, "
                                       f"mean {shard_bases[0] // max(1, shard_reads[0])} bp)/GPU/step, {fam}")) if args.mode == "syncmer" else
                                   (f"viral-class HIXF built WITHOUT --use-syncmer (k={info['k']}, window={info['window']}: "
                                    f"{'every canonical k-mer, k-mer threshold model' if args.mode == 'kmer' else 'window minimisers, FracMinHash threshold model'}) "
                                    f"{idx.data_bytes/1e9:.2f} GB in HBM, root {wl['root_bins']} bins, {shard_reads[0]} reads x {int(read_len)} bp/GPU/step"),
                       "mode": args.mode,
                       "timed_region": "resident 2-bit batches: syncmers -> query -> CSR (+ gather on rank 0 at N > 1); H2D, dna4 pack (a1) and the D2H of the "
                                       "CSR outside -- see value_host_fed for the PCIe-inclusive rate",
                       "index_bytes": idx.data_bytes, "n_ixf": idx.n_ixf, "root_bins": wl["root_bins"],
                       "child_bins": wl["child_bins"], "depth": idx.depth, "reads_per_gpu": shard_reads[0], "read_len": read_len,
                       "distinct_batches": len(batches), "read_error": args.read_error, "search_error_rate": args.error_rate,
                       "frac_reverse": args.frac_reverse, "frac_random": 0.1,
                       "layout_note": (f"reads: forward strand {'only' if args.frac_reverse == 0 else f'{1 - args.frac_reverse:.2f}'} (reverse-strand reads share no "
                                       "syncmers with the index at the reference's t = 5 and stop at the root: forward-only is the harder case); "
                                       f"index: children {wl['child_bins']} bins under a {wl['root_bins']}-bin root (the reference's chopper layout applies one t_max "
                                       "at every level, taxor_build.cpp:168-187,473).  `layouts` carries the same invocation's legs with strand-mixed reads, "
                                       "chopper-shaped children and a 4096-bin root"),
                       "planted_genomes": info["n_genomes"], "family_size": info["fam_size"],
                       "sharding": "reads by rank, index replicated",
                       "hashes_per_read": round(acc["hashes"] / nr, 1),
                       "tuples_per_read": round(acc["tuples"] / nr, 3),
                       "reads_with_hits_last_step": classified, "work_items_per_read": round(acc["work"] / nr, 3)},
            "roofline": roof,
            "whole_step": {"algorithmic_GBps": round(acc["alg"] / elapsed / 1e9, 1),
                           "note": "SURVEY 8(d) A(read) summed over the timed steps / wall time (includes syncmers and CSR assembly)"},
            "stage_ms_per_step": {"syncmers": round(acc["sync_ms"] / args.steps, 3), "query": round(acc["q_ms"] / args.steps, 3),
                                  "finalize": round(acc["fin_ms"] / args.steps, 3), "total": round(acc["total_ms"] / args.steps, 3),
                                  "note": "HIP-event sums per stream; syncmers of sub-batch i+1 overlap the query of sub-batch i"},
        }
        if world == 1 and not dist_on and not args.no_dropin:
            # the resident searchers (and their streams) are not needed any more: a drop-in user's process holds the one or two
            # searchers it feeds, and the figures below are measured like that
            for sr in searchers:
                sr.close()
            searchers = []
            out["pcie_inclusive"], out["sustained"] = dropin_measurements(args, idx, batches, read_len)
            if out["sustained"]:
                out["value_host_fed"] = out["sustained"]["value"]
            if e04 is not None and args.sustained_reads > 0:
                a04 = argparse.Namespace(**vars(args))
                a04.sustained_reads = max(1, min(args.sustained_reads, 2_000_000))
                _, sus04 = dropin_measurements(a04, idx, e04["_batches"], read_len, single=False)
                e04["value_host_fed"] = sus04["value"]
                e04["host_fed_reads"] = sus04["reads"]
        if dist_on:
            sizes = acc.get("gather_sizes", [])
            prm = acc.get("per_rank_ms", [])
            out["comm"] = {"backend": "nccl (RCCL)" if backend == "nccl" else backend,
                           "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                           "world": dist.get_world_size(), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                           "ranks_in_last_gather": [p_ for p_, (n_, _t) in enumerate(sizes) if n_ > 0],
                           "reads_tuples_per_rank_last_gather": [[int(n_), int(t_)] for n_, t_ in sizes],
                           "gather_bytes_per_step": round(acc.get("gather_bytes_received", 0.0), 1),
                           "gather_ms_per_step": round(acc.get("per_rank_gather_ms", [0.0])[0], 4),
                           "host_ms_per_step": {"export_to_torch": round(acc.get("export_host_ms", 0.0), 4), "gather_csr": round(acc.get("gather_host_ms", 0.0), 4),
                                                "note": "rank 0's wall clock inside a step for the hand-off: result sizes + four torch.empty + the D2D export of the CSR; "
                                                        "gather_csr (all_gather of the sizes + the point-to-point exchange)"},
                           "sent_bytes_per_rank_per_step": [round(v, 1) for v in acc.get("per_rank_sent_bytes", [])],
                           "ms_per_step_per_rank": {"min": round(min(prm), 3) if prm else None, "max": round(max(prm), 3) if prm else None,
                                                    "rank0": round(prm[0], 3) if prm else None, "all": [round(v, 3) for v in prm]},
                           "strong_leg": strong_leg,
                           "note": "written from what the exchange moved, not from WORLD_SIZE: ranks_in_last_gather = peers whose reads arrived in the last gathered "
                                   "CSR; gather_bytes_per_step = payload bytes rank 0 received from its peers (0 with one rank); gather_ms_per_step = HIP events "
                                   "around gather_csr on rank 0 (export of the four device arrays excluded); ms_per_step_per_rank = each rank's own clock around "
                                   "its steps (run + sync + gather), without the barriers that bracket `value`"}
        if per_rank is not None:
            out["pcie_inclusive_per_rank"] = per_rank
            out["sustained_sum_Mbp_s"] = per_rank["sustained_sum_Mbp_s"]
            out["host_fed_scaling"] = per_rank["host_fed_scaling"]
            out["value_host_fed"] = per_rank["sustained_sum_Mbp_s"]
        if "value_host_fed" in out:
            out["value_note"] = ("`value` is the contract's figure: K timed steps over batches already resident in HBM (2-bit packed).  "
                                 "`value_host_fed` is the drop-in figure -- >= --sustained-reads reads fed from host buffers through "
                                 "taxor_gpu_search_batch (PCIe, on-device packing and the D2H copy of the results inside), whole job")
        if e04 is not None:
            e04.pop("_batches", None)
            out["value_e04"] = e04["value"]
            if "value_host_fed" in e04:
                out["value_e04_host_fed"] = e04["value_host_fed"]
            out["read_error_0.04"] = e04
        out["host_binding"] = numa_info
        if world == 1 and not args.no_cpu_baseline:
            os.sched_setaffinity(0, full_affinity)      # the CPU baseline is timed on the box's host cores, all sockets
            ncpu = len(full_affinity)
            hash_kw = dict(k=info["k"], s=info["s"], t=info["t"], window=info["window"]) if args.mode != "syncmer" else {}
            out["cpu_baseline"] = cpu_baseline(args, idx, lay, res, batches[(args.warmup + args.steps - 1) % len(batches)], read_len, ncpu, hash_kw)
        if world == 1 and not dist_on and not args.no_layouts and args.mode == "syncmer" and not args.len_mix:
            # the legs build more 113-GB-class indexes: should one of them take the process down (a host out-of-memory kill cannot be caught),
            # the headline measured so far is on disk already
            side = os.environ.get("TAXOR_BENCH_SIDE_FILE", os.path.join(ROOT, "gpurun_out", "bench_headline_before_legs.json"))
            try:
                os.makedirs(os.path.dirname(side), exist_ok=True)
                with open(side, "w") as fh:
                    fh.write(json.dumps(out) + "\n")
            except OSError:
                pass
            info["lay"], info["leg_batches"] = lay, batches[:2]
            out["layouts"] = layout_legs(args, idx, info, searchers, timed, local_rank, out)     # (never raises: a leg that fails is reported as such)
            idx = None
        print(json.dumps(out), flush=True)
    for sr in searchers:
        sr.close()
    if idx is not None:
        idx.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def layout_legs(args, idx, info, searchers, timed, local_rank, out):
    """The headline's workload choices, varied one at a time in the SAME invocation (2 resident batches, 3 timed steps each, HIP-event
    kernel time, no PMC pass): (a) strand-mixed reads on the headline index -- SURVEY 8(d) as written; (b) chopper-shaped index:
    every IXF below the root as wide as the root (the reference applies one t_max at every level, taxor_build.cpp:168-187,473);
    (c) a 4096-bin root (the largest t_max the reference offers, :177-184).  (b) and (c) build their own 113-GB-class index, so
    the headline index is released first (closes `idx` and `searchers`)."""
    from taxor_amd import Searcher, synth
    n_reads, read_len, ncpu = info["n_reads"], info["read_len"], info["ncpu"]
    legs = []

    def measure(index, batches, label, what, extra):
        pool = []
        try:
            for bb, oo in batches:
                s2 = Searcher(index, error_rate=args.error_rate, time_kernels=True)
                pool.append(s2)
                s2.upload(bb, oo)
                s2.run()
                s2.sync()
            e, a = timed(pool, 3, 1)
        except BaseException:
            for s2 in pool:          # searchers go before their index, also when a leg fails
                s2.close()
            raise
        q = a["q_ms"] * 1e-3
        sparse = sum(a["lvl_sparse"])
        leg = {"layout": label, "what": what, "value": round(a["all_bases"] / e / 1e6, 2), "unit": "Mbp/s", "steps": 3,
               "frac": round(a["q_touched"] / q / 1e9 / HBM_PEAK_GBS, 4) if q > 0 else None,
               "algorithmic_frac": round(a["q_bytes"] / q / 1e9 / HBM_PEAK_GBS, 4) if q > 0 else None,
               "moved_frac": None, "sparse_loads_per_launch": round(sparse / max(1, a["launches"]), 1),
               "hashes_per_read": round(a["hashes"] / max(1, a["reads"]), 1), "tuples_per_read": round(a["tuples"] / max(1, a["reads"]), 3),
               "work_items_per_read": round(a["work"] / max(1, a["reads"]), 3)}
        leg.update(extra)
        for s2 in pool:
            s2.close()
        log(f"layout leg {label}: {leg['value']:.0f} Mbp/s, frac {leg['frac']}")
        return leg

    shape = {"root_bins": out["config"]["root_bins"], "child_bins": out["config"]["child_bins"], "index_bytes": out["config"]["index_bytes"],
             "n_ixf": out["config"]["n_ixf"], "depth": out["config"]["depth"]}
    # the random-filled headline index with two of its own batches, measured like every leg: what exact_fill (below) is held against
    twin = None
    try:
        twin = measure(idx, info["leg_batches"], "headline_2_batches", "the headline index, its first two batches, 3 steps", dict(shape, frac_reverse=args.frac_reverse))
    except Exception as e:
        log(f"twin measurement failed: {type(e).__name__}: {e}")
    # (a) the headline index, reads from both strands
    both = []
    for b in range(2):
        bb, oo, _ = synth.synth_reads(info["genomes"], info["genome_off"], n_reads, read_len, error_rate=args.read_error, frac_random=0.1,
                                      seed=synth.DEFAULT_SEED + 55000 + b, threads=ncpu, frac_reverse=0.5)
        both.append((np.ascontiguousarray(bb), np.ascontiguousarray(oo)))
    try:
        legs.append(measure(idx, both, "strand_mixed", "headline index, planted reads from either strand with probability 0.5 (SURVEY 8(d) 'strand uniformly'); "
                            "reverse-strand reads share no syncmers with the index and stop at the root", dict(shape, frac_reverse=0.5)))
    except Exception as e:        # the legs are additions to the line: a failure in one must not cost the headline
        legs.append({"layout": "strand_mixed", "error": f"{type(e).__name__}: {e}"})
        log(f"layout leg strand_mixed failed: {type(e).__name__}: {e}")
    del both
    for sr in searchers:
        sr.close()
    del searchers[:]
    idx.close()
    # (b), (c): indexes of the same footprint in other shapes, each with its own forward-strand reads like the headline
    for label, root_bins, child_bins, what in (
            (f"chopper_{out['config']['root_bins']}", out["config"]["root_bins"], out["config"]["root_bins"],
             "children as wide as the root: one t_max at every level, as the reference's layout step applies it"),
            ("root_4096", 4096, out["config"]["child_bins"], "t_max 4096 at the root (the largest the reference offers), the headline's children")):
        a2 = argparse.Namespace(**vars(args))
        a2.root_bins, a2.child_bins, a2.batches = root_bins, child_bins, 2
        idx2 = None
        try:
            wl2, idx2, lay2, batches2, info2 = build_workload(a2, local_rank, 0, 1)
            legs.append(measure(idx2, batches2, label, what, {"root_bins": wl2["root_bins"], "child_bins": wl2["child_bins"], "index_bytes": idx2.data_bytes,
                                                              "n_ixf": idx2.n_ixf, "depth": idx2.depth, "frac_reverse": args.frac_reverse}))
            del lay2, batches2, info2
        except Exception as e:
            legs.append({"layout": label, "error": f"{type(e).__name__}: {e}"})
            log(f"layout leg {label} failed: {type(e).__name__}: {e}")
        if idx2 is not None:
            idx2.close()
    # (d) every bin a real filter: the headline's own layout and reads, against the random-filled headline index measured the same way
    try:
        if twin is None:
            raise RuntimeError("the random-filled twin was not measured")
        legs.append(exact_fill_leg(args, measure, local_rank, info, twin))
    except Exception as e:
        legs.append({"layout": "exact_fill", "error": f"{type(e).__name__}: {e}"})
        log(f"layout leg exact_fill failed: {type(e).__name__}: {e}")
    return legs


def exact_fill_leg(args, measure, local_rank, info, twin):
    """`layouts` leg exact_fill: the HEADLINE'S OWN layout -- same bins, same planted genomes, same size -- built once more
    with EVERY bin a real XOR filter (taxor_gpu_index_build_hixf_gen): the planted bins from their genomes' hashes, every decoy leaf bin from
    generated keys (no key memory: a 113-GB index holds 7e10 keys, 560 GB if they had to be resident), merged bins from the union of their
    child; searched with the headline's own first two batches and held against the random-filled headline index measured the same way
    (`twin`: 2 batches, 3 steps, before the headline index was released)."""
    from taxor_amd import synth
    lay, batches = info["lay"], info["leg_batches"]
    t0 = time.time()
    idx_e, st = synth.exact_fill_index(lay, device=local_rank, fill_frac=0.95)
    log(f"exact-fill index: {idx_e.data_bytes/1e9:.2f} GB, {idx_e.n_ixf} IXFs, {st['keys_inserted']/1e9:.2f} G insertions in {st['seconds_total']:.2f} s "
        f"({st['keys_inserted']/st['seconds_total']/1e9:.2f} G/s; {st['chunks']} chunks, {st['reseeds']} IXFs redone), {time.time()-t0:.1f}s")
    shape = {"root_bins": lay["ixfs"][0]["bins"], "child_bins": lay["ixfs"][1]["bins"] if len(lay["ixfs"]) > 1 else None, "index_bytes": idx_e.data_bytes,
             "n_ixf": idx_e.n_ixf, "depth": idx_e.depth, "frac_reverse": args.frac_reverse}
    try:
        leg = measure(idx_e, batches, "exact_fill", "the headline's own layout at full size with EVERY bin a real filter: planted bins from their genomes, decoy leaf bins from "
                      "generated keys (up to 95 % of capacity, less where the root's bin bounds the union of a child), merged bins = union of their child, all constructed "
                      "on the GPU; the headline's own reads", shape)
    finally:
        idx_e.close()
    leg["build"] = {"insertions": int(st["keys_inserted"]), "seconds": round(st["seconds_total"], 3), "insertions_per_s": round(st["keys_inserted"] / st["seconds_total"], 1),
                    "chunks": int(st["chunks"]), "rounds_max": int(st["rounds_max"]), "reseeds": int(st["reseeds"]), "scratch_bytes": int(st["scratch_bytes"]),
                    "note": "decoy keys are generated by the kernels (synth_key of a running index), not read: not the builder's bench line (bench.py --mode build)"}
    leg["random_fill_twin"] = {kk: twin[kk] for kk in ("value", "frac", "algorithmic_frac", "tuples_per_read", "work_items_per_read", "hashes_per_read")}
    leg["value_over_random_fill_twin"] = round(leg["value"] / twin["value"], 4)
    leg["within_3_percent"] = bool(abs(leg["value"] / twin["value"] - 1.0) < 0.03)
    return leg


def dropin_measurements(args, idx, batches, read_len, single=True):
    """The drop-in boundary with HOST buffers: (a) one taxor_gpu_search_batch call on pageable memory; (b) the sustained
    rate over >= --sustained-reads reads, rotating through the distinct batches from page-locked staging buffers with two
    searchers in flight (what the CLI's GPU workers do), results fetched to the host every call."""
    from taxor_amd import Searcher, _lib
    import ctypes as C
    sr = Searcher(idx, error_rate=args.error_rate)
    bases, offs = batches[0]
    sr.search_batch(bases, offs)
    if single:
        samples = []
        for _ in range(3):     # the blocking copies out of pageable memory vary from call to call (26-41 ms for the same 1.3 GB): median of three
            t0 = time.perf_counter()
            sr.search_batch(bases, offs, copy=False)   # the C call, results in the library's host arrays (what a C++ host gets)
            samples.append(time.perf_counter() - t0)
        dt = sorted(samples)[1]
        single = {"seconds": round(dt, 4), "value": round(float(offs[-1]) / dt / 1e6, 2), "unit": "Mbp/s",
                  "samples_s": [round(x, 4) for x in samples],
                  "note": "one taxor_gpu_search_batch on host buffers (median of three calls): ASCII bases from pageable memory, streamed "
                          "H2D + on-device pack overlapped with compute, results fetched to host; per GPU"}
    else:
        single = None
    sustained = None
    if args.sustained_reads > 0:
        L = _lib.lib()
        for b, _ in batches:
            L.taxor_gpu_host_register(b.ctypes.data_as(C.c_void_p), b.nbytes)
        per = sum(o.size - 1 for _, o in batches)
        rounds = max(1, -(-args.sustained_reads // per))
        order = [i % len(batches) for i in range(rounds * len(batches))]
        workers = [sr, Searcher(idx, error_rate=args.error_rate)]
        tuples = [0, 0]

        def run(w):
            for j in range(w, len(order), len(workers)):
                b, o = batches[order[j]]
                r = workers[w].search_batch(b, o, copy=False)
                tuples[w] += int(r.user_bin.size)

        for w in range(len(workers)):          # warm both searchers' scratch
            workers[w].search_batch(*batches[w % len(batches)])
        t0 = time.perf_counter()
        th = [threading.Thread(target=run, args=(w,)) for w in range(len(workers))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        nb = sum(int(batches[i][1][-1]) for i in order)
        nr = sum(batches[i][1].size - 1 for i in order)
        sustained = {"reads": nr, "calls": len(order), "distinct_batches": len(batches), "seconds": round(dt, 3),
                     "value": round(nb / dt / 1e6, 2), "unit": "Mbp/s", "tuples": int(sum(tuples)),
                     "note": "host-fed: page-locked ASCII staging buffers -> taxor_gpu_search_batch (H2D, pack, search, D2H of "
                             "the CSR) with two searchers in flight on one GPU; PCIe inside the timed region"}
        for b, _ in batches:
            L.taxor_gpu_host_unregister(b.ctypes.data_as(C.c_void_p))
        workers[1].close()
    sr.close()
    return single, sustained


def _interleave_host_memory(on):
    """MPOL_INTERLEAVE over all NUMA nodes for pages this thread touches from now on (off: back to the default policy).  The
    oracle's copy of the index is first touched here, on one thread: without this the 45 GB root table lands on one socket
    and, at all hardware threads, the other socket reads every fingerprint row remotely (the round-2 `all_cores` figure was
    slower than 32 threads for that reason).  Returns the node count used, 0 if the policy could not be set."""
    import ctypes
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        if not on:
            libc.syscall(238, 0, None, 0)                    # set_mempolicy(MPOL_DEFAULT)
            return 0
        nodes = set()
        for part in open("/sys/devices/system/node/online").read().strip().split(","):
            a, _, b = part.partition("-")
            nodes.update(range(int(a), int(b or a) + 1))
        if len(nodes) < 2:
            return 0
        words = max(nodes) // 64 + 1
        mask = (ctypes.c_ulong * words)()
        for n in nodes:
            mask[n // 64] |= 1 << (n % 64)
        rc = libc.syscall(238, 3, mask, ctypes.c_ulong(words * 64 + 1))      # set_mempolicy(MPOL_INTERLEAVE, mask, maxnode)
        return len(nodes) if rc == 0 else 0
    except Exception:
        return 0


def cpu_quota():
    """CPUs the container's cgroup allows (cpu.max: quota / period), or None.  The pool's GPU box shows 256 hardware threads and
    allows 16 (profiles/r04/cpu_quota_probe.txt): beyond 16 runnable threads every thread of the process is stopped for the rest of
    each 100-ms period."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        return None


def cpu_baseline(args, idx, lay, res, batch, read_len, ncpu, hash_kw=None):
    """The CPU oracle (a port of the reference path) on a bounded sample of the same reads + index, same box."""
    from oracle import oracle as orc
    hash_kw = hash_kw or {}
    bases, offs = batch
    quota = cpu_quota()
    cap32 = min(ncpu, 32)                        # the reference caps --threads at 32 (taxor_search.cpp:51-55)
    threads = max(1, min(cap32, int(quota + 0.5))) if quota else cap32      # ... and the box may allow fewer CPUs than that
    # host copy of the IXFs the sample can visit: the root plus every IXF holding a planted path; the rest get
    # untouched virtual memory (never read: the traversal enters a child only when its merged bin passes the
    # threshold, which random fingerprints cannot)
    needed = {0}
    for i, f in enumerate(lay["ixfs"]):
        if (f["columns"] or f.get("key_sets")) and i > 0:
            needed.add(i)
    interleaved = _interleave_host_memory(True)
    try:
        host = []
        for i, f in enumerate(lay["ixfs"]):
            nbytes = 3 * f["seg_len"] * f["stride"]
            data = np.empty(nbytes, dtype=np.uint8)
            if i in needed:
                data[::4096] = 0                    # first touch under the interleave policy, then the bytes out of HBM
                idx.download_ixf(i, out=data)
            host.append(dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], data=data))
        h = orc.Hixf(host, [f["next_ixf"] for f in lay["ixfs"]], [f["fname_idx"] for f in lay["ixfs"]])
    except MemoryError as e:
        return {"value": None, "unit": "Mbp/s", "cores": threads, "kind": "port", "sample": f"skipped: {e}"}
    finally:
        _interleave_host_memory(False)

    # the worker is the port; the SCHEDULER is the reference's own hixf::do_parallel (do_parallel.hpp, compiled into oracle/_ref
    # where /root/reference was mounted at build time; the file travels): 1024-record chunks, `threads` std::async tasks per
    # chunk, floor(n/threads) records each with the remainder on the last, one barrier per chunk -- taxor_search.cpp:315-326.
    # Without that library: the port's own OpenMP slices over the whole sample.
    sched = "reference" if orc.ref_lib() is not None else "openmp"

    def run(n, th, scheduler=None):
        t0 = time.perf_counter()
        o = h.search_batch(bases[: int(offs[n])], offs[: n + 1], err=args.error_rate, threads=th, scheduler=scheduler or sched, **hash_kw)
        return time.perf_counter() - t0, o

    n = min(256 * threads // 8 + 64, len(offs) - 1)
    dt, o = run(n, threads)
    n2 = int(min(len(offs) - 1, max(n, n * args.cpu_seconds / max(dt, 1e-3))))
    if n2 > n:
        n = n2
        dt, o = run(n, threads)
    nh, off, ub, cnt, _ = o
    lo = int(res.read_off[n])
    same = (np.array_equal(res.n_hashes[:n], nh) and np.array_equal(res.read_off[: n + 1], off)
            and np.array_equal(res.user_bin[:lo], ub) and np.array_equal(res.count[:lo], cnt))
    if not same:
        raise SystemExit("PARITY FAILURE: GPU results differ from the CPU oracle on the baseline sample")
    extra = {}
    if sched == "reference":     # beside it: the port's own scheduler (one OpenMP barrier per sample instead of one per 1024 records)
        nq = max(1, n // 3)
        dtq, oq = run(nq, threads, "openmp")
        if not np.array_equal(oq[1], off[: nq + 1]):
            raise SystemExit("PARITY FAILURE: the oracle's two schedulers disagree")
        extra["openmp_slices"] = {"value": round(int(offs[nq]) / dtq / 1e6, 3), "reads": nq,
                                  "note": "same worker, ceil(n/threads) slices of the whole sample under OpenMP"}
    if cap32 > threads:     # the reference's own cap, over the quota: 32 threads on `quota` CPUs
        nb = max(1, n // 3)
        dtb, _ = run(nb, cap32)
        extra["threads_32"] = {"value": round(int(offs[nb]) / dtb / 1e6, 3), "threads": cap32, "reads": nb,
                               "note": f"the reference's cap of 32 threads on the {quota:g} CPUs the cgroup allows"}
    if ncpu > threads and not quota:     # also at all hardware threads (the reference itself caps --threads at 32);
        na = max(1, n // 4)  # a quarter of the sample: beyond 32 threads the port gets slower (remote-socket row reads)
        dta, _ = run(na, ncpu)
        extra["all_cores"] = {"value": round(int(offs[na]) / dta / 1e6, 3), "cores": ncpu, "reads": na}
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {**extra, "value": round(int(offs[n]) / dt / 1e6, 3), "unit": "Mbp/s", "cores": threads, "kind": "port",
            "cpu_model": model, "hardware_threads": ncpu, "cpu_quota": quota, "index_copy_interleaved_over_numa_nodes": interleaved,
            "scheduler": ("the reference's hixf::do_parallel (do_parallel.hpp compiled from /root/reference into oracle/_ref): 1024-record "
                          "chunks, std::async tasks, one barrier per chunk" if sched == "reference" else "port: OpenMP slices of the whole sample"),
            "sample": f"first {n} reads of the last timed batch ({int(offs[n])/1e6:.1f} Mbp), same index, {dt:.1f} s wall, "
                      f"{threads} threads in the reference's do_parallel shape; GPU results bit-identical on the sample"}


if __name__ == "__main__":
    main()
