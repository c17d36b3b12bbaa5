"""One randomized differential trial: random genomes, a random HIXF around them (taxor_amd.synth.random_layout), random
reads and search parameters; the HIP path (through the C ABI) must equal the CPU oracle tuple for tuple.
Used by tests/test_gpu_fuzz.py (fixed seeds) and tests/fuzz_parity.py (as long as you like)."""
import os

import numpy as np

from oracle import oracle as orc
from taxor_amd import GpuIndex, Searcher, synth


def _rand_read(rng, g, go, kind):
    if kind == 0:      # junk
        return bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 3000))))
    if kind == 1:      # low complexity
        u = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(1, 9))))
        return (u * 600)[:int(rng.integers(10, 2500))]
    gi = int(rng.integers(0, go.size - 1))   # piece of a genome, optionally with IUPAC codes sprinkled in
    a = int(rng.integers(int(go[gi]), int(go[gi + 1]) - 50))
    b = min(int(go[gi + 1]), a + int(rng.integers(30, 6000)))
    r = bytearray(bytes(g[a:b]))
    if kind == 3:
        for p in rng.integers(0, len(r), size=max(1, len(r) // 100)):
            r[int(p)] = int(rng.choice(np.frombuffer(b"NRYKMSWBDHVacgtn", np.uint8)))
    return bytes(r)


def run_trial(seed, verbose=False):
    rng = np.random.default_rng(seed)
    # the trials' batches are small: below 4096 reads the library skips the grouping of the work queues by IXF (read per call),
    # so half of the trials force it on -- drawn from a generator of its own, the trial's other draws stay what they were
    group_always = bool(np.random.default_rng(seed ^ 0x5EED).random() < 0.5)
    syncmer = rng.random() < 0.7
    if syncmer:
        k, s = [(22, 12), (16, 8), (20, 10), (28, 14), (30, 12), (22, 16), (24, 9)][int(rng.integers(0, 7))]
        w = k - s + 1
        t = int(rng.integers(1, w + 1)) if rng.random() < 0.3 else (w // 2 if w > 1 else 1)   # taxor_build.cpp:510 default
        t = max(t, 1)
        win = 0
    else:
        k = int(rng.choice([16, 20, 22, 31, 32]))
        win = k if rng.random() < 0.5 else k + int(rng.integers(1, 30))
        s = t = 0
    scaling = int(rng.choice([1, 1, 1, 2, 5]))
    n_gen = int(rng.integers(2, 8))
    g, go = synth.random_genomes(n_gen, int(rng.integers(2000, 9000)), seed=int(rng.integers(1, 2**31)))
    planted = []
    for i in range(n_gen):
        seq = bytes(g[int(go[i]):int(go[i + 1])])
        hs = orc.seq_to_syncmers(seq, k, s, t) if syncmer else orc.minimiser_hash(seq, k, win)
        if scaling > 1:
            hs = np.array([h for h in hs.tolist() if float(orc.wyhash(h)) <= float(2**64 - 1) / float(scaling)], dtype=np.uint64)
        planted.append(hs)
    # one trial in four: an index that follows ANOTHER reading of the un-vendored IXF arithmetic (built, searched and checked
    # under the same arithmetic code; rotation steps that keep the three rows independent)
    arith = 0
    if rng.random() < 0.25:
        from taxor_amd.search import arith_code
        # (key hash 1 = none is left out: minimiser values are raw 2-bit k-mer codes, and un-mixed they do not spread over a
        # filter's rows -- no seed peels; seed mode 3 = unused likewise, a re-seed could never help)
        arith = arith_code(int(rng.choice([0, 2, 3])), int(rng.integers(0, 3)), int(rng.choice([13, 16, 21, 27])), int(rng.integers(0, 3)),
                           int(rng.integers(0, 4)))
    heavy = rng.random() < 0.08      # wide rows (several block passes, > 32 alive units possible) and reads whose probes
    #                                  do not fit the LDS staging area
    try:
        if heavy:
            lay = synth.random_layout(planted, rng, max_depth=int(rng.integers(1, 3)), bins_choices=(64, 1000, 2049, 4096), max_ixfs=4, arith=arith)
        else:
            lay = synth.random_layout(planted, rng, max_depth=int(rng.integers(1, 5)), arith=arith)
    except RuntimeError:
        if arith == 0:
            raise
        # some drawn readings cannot build a filter at all (a weak mixer over raw k-mer codes, row windows that overlap):
        # no seed peels, so nobody could have written such an index either -- nothing to search, the trial is void
        return True, dict(seed=seed, arith=arith, void="this arithmetic code does not peel"), 0
    host = synth.materialize_host(lay)
    # one trial in three: the index reaches the device in ANOTHER writer's fingerprint layout (taxor_amd/csrc/ixf_layout.h: foreign
    # pitch, bin-major, bit-sliced words, rows position-major) and is transposed there while it is uploaded (relayout.hip) -- drawn
    # from a generator of its own, the trial's other draws stay what they were
    lrng = np.random.default_rng(seed ^ 0x1A707)
    layout = 0
    device_side = host
    if lrng.random() < 0.34:
        from taxor_amd.search import to_source_layout
        kind = int(lrng.integers(0, 3))
        layout = kind | (0x100 if lrng.random() < 0.5 else 0) | (0 if kind == 2 else int(lrng.choice([0, 0x200, 0x400])))
        device_side = []
        for f in host:
            raw, pitch = to_source_layout(f, layout)
            device_side.append(dict(f, data=raw, src_stride=pitch, stride=(f["bins"] + 63) // 64 * 64))
    # one trial in three of those in the search layout: the index is CONSTRUCTED ON THE GPU from the leaf key sets alone
    # (taxor_gpu_index_build_hixf: all IXFs of a level in shared chunks, merged bins = unions computed on the device, an IXF that
    # does not peel redone under a redrawn seed), read back, and the oracle works on those bytes -- after checking that every leaf
    # key is found in its own bin and in the merged bin of every ancestor.  A generator of its own again.
    brng = np.random.default_rng(seed ^ 0xB111D)
    gpu_built = False
    if layout == 0 and brng.random() < 0.34:
        from taxor_amd._lib import TaxorError
        shapes = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], next_ixf=f["next_ixf"], fname_idx=f["fname_idx"], data=None)
                  for f in lay["ixfs"]]
        idx = GpuIndex(shapes, lay["n_user_bins"], k=k, s=s, t=t, use_syncmer=syncmer, window_size=(win or k), scaling=scaling, arith=arith)
        for i, f in enumerate(lay["ixfs"]):
            idx.fill_random(i, f["fill_seed"])
        leaf = {(i, b): keys for i, f in enumerate(lay["ixfs"]) for b, keys in f["leaf_keys"].items()}
        try:
            idx.build_hixf(leaf, seed0=int(brng.integers(1, 2**62)))
        except TaxorError as e:
            idx.close()
            if "no seed peeled" not in str(e):
                raise
            # small full bins: a hundred of them under ONE seed may not peel in 32 seeds (the reference's rule)
            return True, dict(seed=seed, void="the GPU builder found no seed for an IXF of small full bins"), 0
        host = [dict(s_, seed=idx.ixf_seed(i), data=idx.download_ixf(i)) for i, s_ in enumerate(shapes)]
        h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host], arith=arith)
        parent = {int(f["next_ixf"][b]): (i, b) for i, f in enumerate(host) for b in range(f["bins"]) if f["fname_idx"][b] == -1}
        for (i, b), keys in leaf.items():
            node, at = i, b
            while True:
                if h.ixf_bulk_count(node, keys)[at] != len(keys):
                    idx.close()
                    return False, dict(seed=seed, gpu_built=True, missing=f"keys of leaf ({i},{b}) in bin {at} of IXF {node}"), 0
                if node not in parent:
                    break
                node, at = parent[node]
        gpu_built = True
    else:
        idx = GpuIndex(device_side, lay["n_user_bins"], k=k, s=s, t=t, use_syncmer=syncmer, window_size=(win or k), scaling=scaling, arith=arith, layout=layout)
        h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host], arith=arith)
    n_syn = int(rng.integers(20, 200))
    bases, offs, origin = synth.synth_reads(g, go, n_syn, int(rng.integers(200, 4000)), error_rate=float(rng.choice([0.0, 0.01, 0.03, 0.08])),
                                            frac_random=0.1, seed=int(rng.integers(1, 2**31)))
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(n_syn)]
    reads += [_rand_read(rng, g, go, int(rng.integers(0, 4))) for _ in range(int(rng.integers(5, 60)))]
    reads += [b"", b"A" * int(rng.integers(1, 80))]
    if heavy:
        for _ in range(6):
            gi = int(rng.integers(0, n_gen))
            reads.append(bytes(g[int(go[gi]):int(go[gi + 1])]) * int(rng.integers(2, 6)))      # long, repeats -> duplicates
        reads.append(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=30000)))
    order = rng.permutation(len(reads))
    reads = [reads[int(i)] for i in order]
    B = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, np.uint8)
    O = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    Bn = np.frombuffer(orc.dna4_normalise(B.tobytes()), dtype=np.uint8)
    pct = float(rng.choice([-1.0, -1.0, 0.05, 0.3, 0.6, 1.0]))
    err = float(rng.choice([0.0, 0.01, 0.04, 0.1, 0.2])) if syncmer else float(rng.choice([0.01, 0.04, 0.1, 0.3]))
    sub = int(rng.choice([0, 1, 7, 64]))
    cfg = dict(seed=seed, syncmer=syncmer, k=k, s=s, t=t, window=win, scaling=scaling, n_ixf=len(host), depth=lay["depth"],
               bins=[f["bins"] for f in host][:6], heavy=bool(heavy), reads=len(reads), pct=pct, err=err, sub=sub, arith=arith, layout=layout, gpu_built=gpu_built)
    if verbose:
        print(cfg, flush=True)
    sr = Searcher(idx, error_rate=err, percentage=pct, sub_batch_reads=sub, group_always=group_always,
                  small_path=bool(rng.random() < 0.7), split_always=bool(rng.random() < 0.3))
    n_seg = int(rng.choice([0, 0, 1, 2, 5]))
    if n_seg == 0 or len(reads) < n_seg:
        res = sr.search_batch(B, O)
    else:      # the same reads handed over in several host buffers (taxor_gpu_search_segments_begin), cut at random reads
        cuts = [0] + sorted(int(x) for x in rng.integers(0, len(reads) + 1, size=n_seg - 1)) + [len(reads)]
        segs = []
        for a, b in zip(cuts, cuts[1:]):
            pad = int(rng.integers(0, 3))               # a segment's offsets need not start at 0
            segs.append((np.concatenate([np.full(pad, ord("A"), np.uint8), B[int(O[a]):int(O[b])]]), O[a:b + 1] - O[a] + np.uint64(pad)))
        res = sr.search_segments(segs)
    nh, off, ub, cnt, _ = h.search_batch(Bn, O, k=k, s=s, t=t, err=err, percentage=pct, threads=4, scaling=scaling, window=win)
    ok = (np.array_equal(res.n_hashes, nh) and np.array_equal(res.read_off, off) and np.array_equal(res.user_bin, ub) and
          np.array_equal(res.count, cnt))
    sr.close()
    idx.close()
    return ok, cfg, int(off[-1])
