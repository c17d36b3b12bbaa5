#!/usr/bin/env python3
"""bench.py -- `taxor search` hot path on MI355X: Mbp/s classified against a GTDB-class k22/s12 HIXF.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path (syncmers -> dedup -> threshold -> level-synchronous HIXF query -> DFS
ordered per-read tuples) over one batch of synthetic long reads that is already resident in HBM (2-bit
packed); with N>1 every rank holds a replica of the index, classifies its own shard of reads (weak scaling)
and the per-read results are gathered on rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : k_query_level (dominant kernel) algorithmic gather bytes / its HIP-event time vs 8 TB/s HBM
  cpu_baseline : the CPU oracle (oracle/, a port of the reference path) timed on this box's host cores on a
                 bounded sample of the same reads and index; also used to re-check parity on that sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

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
                 reads=2048, read_len=3000, genomes=8, genome_len=50000),
}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("TAXOR_BENCH_WORKLOAD", "gtdb"), choices=sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=0, help="reads per step and GPU (0 = workload default)")
    ap.add_argument("--read-len", type=int, default=0)
    ap.add_argument("--genomes", type=int, default=0, help="planted genomes (0 = workload default)")
    ap.add_argument("--genome-len", type=int, default=0)
    ap.add_argument("--read-error", type=float, default=0.02)
    ap.add_argument("--error-rate", type=float, default=0.04, help="taxor search --error-rate")
    ap.add_argument("--len-mix", default="", help="'ont': skewed read lengths 1-100 kb (same total bases) instead of a fixed length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement: "
                    "profiling passes then contain only the timed steps' launches")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per k_query_level launch from a separate rocprofv3 --pmc pass")
    args = ap.parse_args()

    import torch                      # first: its bundled HIP runtime is the one the process uses
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from taxor_amd import GpuIndex, Searcher, synth

    wl = dict(WORKLOADS[args.workload])
    n_reads = args.reads or wl["reads"]
    read_len = args.read_len or wl["read_len"]
    args.genomes = args.genomes or wl["genomes"]
    args.genome_len = args.genome_len or wl["genome_len"]
    k, s, t = 22, 12, 5
    ncpu = os.cpu_count() or 8

    # ---- planted genomes and their syncmer hashes (hashed on the GPU; same seed on every rank) -------------
    t0 = time.time()
    g, go = synth.random_genomes(args.genomes, args.genome_len, seed=synth.DEFAULT_SEED)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins, k, s, t,
                     device=local_rank)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(args.genomes)]
    hs.close()
    dummy.close()
    log(f"{args.genomes} genomes x {args.genome_len} bp hashed on GPU: {int(hoff[-1])} syncmers, {time.time()-t0:.1f}s")

    # ---- footprint-faithful layout -----------------------------------------------------------------------------
    t0 = time.time()
    root_rows = wl["total_bytes"] * wl["root_frac"] / wl["root_bins"]
    child_rows = wl["total_bytes"] * (1 - wl["root_frac"]) / ((wl["n_children"] + 1) * max(64, wl["child_bins"]))
    root_max = max(int((root_rows - 32) / 1.23), 8)
    child_max = max(int((child_rows - 32) / 1.23), 8)
    need_root = max(sum(len(p) for p in planted[2:]) // max(1, min(wl["n_children"], len(planted) - 2)) * 2,
                    max(len(p) for p in planted))
    root_max = max(root_max, need_root)
    child_max = max(child_max, max(len(p) for p in planted) + 1024)
    lay = synth.make_layout(planted, root_bins=wl["root_bins"], child_bins=wl["child_bins"],
                            n_children=wl["n_children"], root_max_elems=root_max, child_max_elems=child_max,
                            seed=synth.DEFAULT_SEED, build="gpu")
    idx = synth.device_index(lay, k, s, t, device=local_rank)   # planted columns constructed on the GPU
    log(f"index in HBM: {idx.data_bytes/1e9:.2f} GB, {idx.n_ixf} IXFs, depth {idx.depth}, root {wl['root_bins']} bins, "
        f"children {wl['child_bins']} bins, {time.time()-t0:.1f}s")

    # ---- reads of this rank's shard (seed + rank), resident in HBM before the timed region -----------------------
    t0 = time.time()
    if args.len_mix == "ont":
        # ONT-like skew with the same total bases: 1, 3, 10, 30, 100 kb carrying 10/20/40/20/10 % of the bases, shuffled
        total = n_reads * read_len
        parts = []
        for L, frac in ((1000, 0.1), (3000, 0.2), (10000, 0.4), (30000, 0.2), (100000, 0.1)):
            cnt = max(1, int(total * frac / L))
            b, o, _ = synth.synth_reads(g, go, cnt, L, error_rate=args.read_error, frac_random=0.1,
                                        seed=synth.DEFAULT_SEED + rank + L, threads=ncpu)
            parts += [(b, int(o[i]), int(o[i + 1])) for i in range(cnt)]
        perm = np.random.default_rng(synth.DEFAULT_SEED + rank).permutation(len(parts))
        bases = np.concatenate([parts[i][0][parts[i][1]:parts[i][2]] for i in perm])
        offs = np.concatenate([[0], np.cumsum([parts[i][2] - parts[i][1] for i in perm])]).astype(np.uint64)
        origin = None
        n_reads = len(parts)
        read_len = int(offs[-1]) / n_reads
    else:
        bases, offs, origin = synth.synth_reads(g, go, n_reads, read_len, error_rate=args.read_error, frac_random=0.1,
                                                seed=synth.DEFAULT_SEED + rank, threads=ncpu)
    log(f"{n_reads} reads x {read_len} bp generated ({ncpu} threads), {time.time()-t0:.1f}s")
    sr = Searcher(idx, error_rate=args.error_rate, time_kernels=True)
    # the drop-in call with host buffers (bases cross PCIe inside the call, streamed per sub-batch); reported as
    # pcie_inclusive, never as `value`
    t_dropin = None
    if not args.no_dropin:
        sr.search_batch(bases, offs)
        t0 = time.time()
        sr.search_batch(bases, offs)
        t_dropin = time.time() - t0
    sr.upload(bases, offs)

    from taxor_amd import distributed as td
    gathered = {}

    def gather_results():
        """per-read results of every rank -> rank 0 over RCCL (point-to-point, one xGMI link per peer)"""
        if world == 1:
            return
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
        gathered["last"] = td.gather_csr(ro, ub, ct, nh, dst=0)

    def step():
        sr.run()
        sr.sync()
        gather_results()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q_ms = q_bytes = q_touched = 0.0
    q_launches = 0
    for _ in range(args.steps):
        step()
        st = sr.stats()
        q_ms += st["query_ms"]
        q_bytes += st["query_bytes"]
        q_touched += st["query_touched_bytes"]
        q_launches += st["query_launches"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = sr.stats()
    if world > 1 and rank == 0:
        g_off, g_ub, g_cnt, g_nh = gathered["last"]
        assert g_nh.numel() == n_reads * world and g_off.numel() == n_reads * world + 1
        assert int(g_off[-1]) == g_ub.numel() == g_cnt.numel()
    total_bases = float(n_reads) * read_len * world * args.steps
    value = total_bases / elapsed / 1e6

    out = None
    if rank == 0:
        # measured gather ceiling: random whole-row reads of the same IXFs by a kernel that does nothing
        # else; outside the timed region
        ceiling = {}
        for name, ixf in (("root", 0), ("child", 1 if idx.n_ixf > 1 else 0)):
            gbps, row_bytes = idx.gather_ceiling(ixf, want_bytes=16 << 30, reps=3)
            ceiling[name] = {"ixf": ixf, "row_bytes": row_bytes, "GBps": round(gbps, 1)}
        achieved = q_bytes / (q_ms * 1e-3) / 1e9 if q_ms > 0 else 0.0
        res = sr.fetch()
        classified = int(((res.read_off[1:] - res.read_off[:-1]) > 0).sum())
        out = {
            "metric": "Mbp/s classified (taxor search) vs GTDB k22/s12", "value": round(value, 2), "unit": "Mbp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}-class synthetic HIXF k22/s12/t5, {idx.data_bytes/1e9:.1f} GB resident, "
                                   f"root {wl['root_bins']} bins / children {wl['child_bins']} bins / depth {idx.depth}, "
                                   f"{n_reads} reads x {read_len} bp per GPU per step",
                       "index_bytes": idx.data_bytes, "n_ixf": idx.n_ixf, "root_bins": wl["root_bins"],
                       "child_bins": wl["child_bins"], "reads_per_gpu": n_reads, "read_len": read_len,
                       "read_error": args.read_error, "search_error_rate": args.error_rate,
                       "planted_genomes": args.genomes, "sharding": "reads by rank, index replicated",
                       "hashes_per_read": round(st["n_hashes"] / max(1, n_reads), 1),
                       "tuples_per_read": round(st["n_tuples"] / max(1, n_reads), 3),
                       "reads_with_hits": classified, "work_items_per_read": round(st["n_work_items"] / max(1, n_reads), 3)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic_from_profiles(args), "kernel": "k_query_level",
                         "note": "achieved = algorithmic bytes (sum n_h*3*bins over visited IXFs, SURVEY 8(d)) / HIP-event "
                                 "time; threshold-aware pruning requests fewer bytes than that (requested_*), so "
                                 "achieved can exceed the HBM peak while requested_GBps cannot",
                         "launches": q_launches, "avg_launch_ms": round(q_ms / max(1, q_launches), 4),
                         "algorithmic_bytes_per_launch": round(q_bytes / max(1, q_launches), 1),
                         "requested_bytes_per_launch": round(q_touched / max(1, q_launches), 1),
                         "requested_GBps": round(q_touched / (q_ms * 1e-3) / 1e9, 1) if q_ms > 0 else 0.0,
                         "gather_ceiling": ceiling,
                         "whole_step_achieved": round(st["algorithmic_bytes"] * args.steps / elapsed / 1e9 / 1.0, 1)},
            "stage_ms_last_step": {"syncmers": round(st["syncmer_ms"], 3), "query": round(st["query_ms"], 3),
                                   "finalize": round(st["finalize_ms"], 3), "total": round(st["total_ms"], 3)},
            "pcie_inclusive": None if t_dropin is None else {"seconds": round(t_dropin, 4), "value": round(float(n_reads) * read_len / t_dropin / 1e6, 2),
                               "unit": "Mbp/s",
                               "note": "taxor_gpu_search_batch on host buffers: ASCII bases from pageable memory, streamed "
                                       "H2D + on-device pack overlapped with compute, results fetched to host; per GPU"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, idx, lay, res, bases, offs, read_len, ncpu)
        print(json.dumps(out), flush=True)
    sr.close()
    idx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def traffic_from_profiles(args):
    """HBM bytes per k_query_level launch from the separate rocprofv3 --pmc passes (profiles/run_profiles.sh):
    FETCH_SIZE (doubled, gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE.  None if never collected for
    this workload at its default size."""
    if args.traffic_bytes is not None:
        return args.traffic_bytes
    if args.reads or args.read_len or args.len_mix:
        return None
    p = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f).get("k_query_level_bytes_per_launch")
    return None


def cpu_baseline(args, idx, lay, res, bases, offs, read_len, ncpu):
    """The CPU oracle (a port of the reference path) on a bounded sample of the same reads + index, same box."""
    from oracle import oracle as orc
    threads = min(ncpu, 32)                      # the reference caps --threads at 32 (taxor_search.cpp:51-55)
    # host copy of the IXFs the sample can visit: the root plus every IXF holding a planted path; the rest get
    # untouched virtual memory (never read: the traversal enters a child only when its merged bin passes the
    # threshold, which random fingerprints cannot)
    needed = {0}
    for i, f in enumerate(lay["ixfs"]):
        if (f["columns"] or f.get("key_sets")) and i > 0:
            needed.add(i)
    try:
        host = []
        for i, f in enumerate(lay["ixfs"]):
            nbytes = 3 * f["seg_len"] * f["stride"]
            data = idx.download_ixf(i) if i in needed else np.empty(nbytes, dtype=np.uint8)
            host.append(dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], data=data))
        h = orc.Hixf(host, [f["next_ixf"] for f in lay["ixfs"]], [f["fname_idx"] for f in lay["ixfs"]])
    except MemoryError as e:
        return {"value": None, "unit": "Mbp/s", "cores": threads, "kind": "port", "sample": f"skipped: {e}"}

    def run(n):
        t0 = time.perf_counter()
        o = h.search_batch(bases[: int(offs[n])], offs[: n + 1], err=args.error_rate, threads=threads)
        return time.perf_counter() - t0, o

    n = min(256 * threads // 8 + 64, len(offs) - 1)
    dt, o = run(n)
    n2 = int(min(len(offs) - 1, max(n, n * args.cpu_seconds / max(dt, 1e-3))))
    if n2 > n:
        n = n2
        dt, o = run(n)
    nh, off, ub, cnt, _ = o
    lo = int(res.read_off[n])
    same = (np.array_equal(res.n_hashes[:n], nh) and np.array_equal(res.read_off[: n + 1], off)
            and np.array_equal(res.user_bin[:lo], ub) and np.array_equal(res.count[:lo], cnt))
    if not same:
        raise SystemExit("PARITY FAILURE: GPU results differ from the CPU oracle on the baseline sample")
    extra = {}
    if ncpu > threads:      # also at all hardware threads (the reference itself caps --threads at 32)
        t0 = time.perf_counter()
        h.search_batch(bases[: int(offs[n])], offs[: n + 1], err=args.error_rate, threads=ncpu)
        extra = {"all_cores": {"value": round(int(offs[n]) / (time.perf_counter() - t0) / 1e6, 3), "cores": ncpu}}
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {**extra, "value": round(int(offs[n]) / dt / 1e6, 3), "unit": "Mbp/s", "cores": threads, "kind": "port",
            "cpu_model": model, "hardware_threads": ncpu,
            "sample": f"first {n} of the step's reads ({int(offs[n])/1e6:.1f} Mbp), same index, {dt:.1f} s wall, "
                      f"{threads} threads in the reference's do_parallel shape; GPU results bit-identical on the sample"}


if __name__ == "__main__":
    main()
