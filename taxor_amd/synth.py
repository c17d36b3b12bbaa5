"""Synthetic, footprint-faithful HIXF layouts with planted genomes, and seeded synthetic long reads
(SURVEY.md section 8(d)).  No published .hixf is available offline, so tests and the bench query indexes
produced here: same k/s/t, uint8 fingerprints, a stated number of bins per IXF and hierarchy depth; planted
genomes are inserted by real XOR-filter construction along full root->leaf paths (including a split user
bin and a depth-3 merged chain); every other bin is seeded pseudo-random bytes, which behaves exactly like a
non-matching bin (false-positive rate 2^-8 per hash and bin).

All construction goes through libtaxor_gpu.so's host helpers; nothing here touches oracle/."""
import ctypes as C

import numpy as np

from . import _lib

DEFAULT_SEED = 20250523


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def random_genomes(n, length, seed=DEFAULT_SEED):
    """n random ACGT genomes -> (bases uint8[n*length], offsets uint64[n+1])"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = acgt[rng.integers(0, 4, size=n * length, dtype=np.uint8)]
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length))
    return bases, offs


def synth_reads(genomes, genome_off, n_reads, read_len, error_rate=0.04, frac_random=0.1, seed=DEFAULT_SEED,
                threads=8, frac_reverse=0.0):
    """ONT-like reads (40/30/30 sub/ins/del) from the genomes + a fraction of uniform random reads.
    Returns (bases uint8[n_reads*read_len], offsets uint64[n_reads+1], origin int32[n_reads])."""
    g = np.ascontiguousarray(genomes, dtype=np.uint8)
    go = np.ascontiguousarray(genome_off, dtype=np.uint64)
    bases = np.empty(n_reads * read_len, dtype=np.uint8)
    offs = np.empty(n_reads + 1, dtype=np.uint64)
    origin = np.empty(n_reads, dtype=np.int32)
    rc = _lib.lib().taxor_synth_reads(_p(g), _p(go), go.size - 1, n_reads, read_len, error_rate, frac_random,
                                      frac_reverse, seed, threads, _p(bases), bases.size, _p(offs), _p(origin))
    _lib.check(rc)
    return bases, offs, origin


def synth_keys_host(first, n, salt):
    """keys first .. first+n-1 of the synthetic key set (ixf_arith.h synth_key: the splitmix64 finaliser of i + salt), as the
    device generates them for GpuIndex.build_hixf_synth"""
    with np.errstate(over="ignore"):
        z = np.arange(first, first + n, dtype=np.uint64) + np.uint64(salt & (2**64 - 1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def full_hierarchy_shapes(n_children, child_bins, keys_per_bin, slack=1.0):
    """Shapes of a two-level hierarchy in which every bin is meant to be built (build bench, full-size build test): a root of
    max(64, n_children) bins, the first n_children merged, over children of child_bins leaf bins with keys_per_bin keys each.
    slack > 1: the IXFs are sized for slack x the keys their bins get (bins of a few ten thousand keys filled to exactly
    1/1.23 of their rows fail to peel too often for a hundred of them to succeed under one seed, see exact_fill_index).
    Returns (shapes for GpuIndex, n_user_bins, keys per technical bin in index bin order)."""
    rb = max(64, n_children)
    nx = np.zeros(rb, np.int64)
    fn = np.arange(rb, dtype=np.int64)
    nx[:n_children] = np.arange(1, n_children + 1)
    fn[:n_children] = -1
    shapes = [dict(bins=rb, stride=_stride(rb), seg_len=seg_len_for(int(child_bins * keys_per_bin * slack)), seed=1, next_ixf=nx, fname_idx=fn, data=None)]
    ub = rb
    for c in range(n_children):
        shapes.append(dict(bins=child_bins, stride=_stride(child_bins), seg_len=seg_len_for(int(keys_per_bin * slack)), seed=2 + c,
                           next_ixf=np.full(child_bins, c + 1, np.int64), fname_idx=np.arange(ub, ub + child_bins, dtype=np.int64), data=None))
        ub += child_bins
    counts = np.zeros(rb + n_children * child_bins, dtype=np.uint64)
    counts[rb:] = keys_per_bin
    return shapes, ub, counts


def capacity_of(seg_len):
    """keys a bin of an IXF with this segment length is sized for: the largest n with seg_len_for(n) <= seg_len"""
    n = int((3 * seg_len - 32) / 1.23)
    while seg_len_for(n + 1) <= seg_len:
        n += 1
    while n > 0 and seg_len_for(n) > seg_len:
        n -= 1
    return n


def exact_fill_index(layout, salt=DEFAULT_SEED, k=22, s=12, t=5, device=0, seed0=7, fill_frac=0.9):
    """An index of the layout's shape in which EVERY bin is a real filter: planted leaf bins hold their genomes' hashes, every
    other leaf bin is filled with synthetic keys (synth_key(running index, salt), generated on the device) to fill_frac of its
    IXF's capacity, merged bins receive the union of their child -- all of it constructed by taxor_gpu_index_build_hixf_gen (the decoys'
    keys are generated inside the kernels: no key memory, so the index may be of class scale).
    fill_frac < 1: an IXF's capacity is that of its LARGEST bin, the others are smaller in any real index; and a bin of a few
    ten thousand keys filled to exactly 1/1.23 of its rows fails to peel under a given seed every so often (the margin to the
    peeling threshold, 0.7 %, is about its own finite-size fluctuation), which a hundred such bins under ONE seed -- the
    reference's rule, construct_ixf.cpp:100-108 -- practically never survive.  The layout must come from make_layout /
    make_family_layout(build="gpu").  Returns (index, build figures)."""
    from .search import GpuIndex
    L = _lib.lib()
    fs = layout["ixfs"]
    idx = GpuIndex([dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"], next_ixf=f["next_ixf"],
                         fname_idx=f["fname_idx"], data=None) for f in fs], layout["n_user_bins"], k, s, t, device)
    # how many synthetic keys a decoy leaf bin of IXF i gets: its IXF's capacity -- unless the IXF hangs below a merged bin, whose
    # capacity P bounds the UNION of everything in the IXF: decoys * fill + planted + merged bins * capacity <= P
    parent_cap = [None] * len(fs)
    for i, f in enumerate(fs):
        for b in range(f["bins"]):
            if f["fname_idx"][b] < 0:
                parent_cap[int(f["next_ixf"][b])] = capacity_of(f["seg_len"])
    counts, planted = [], []            # per technical bin in index order; planted: (global bin, keys)
    for i, f in enumerate(fs):
        cap = capacity_of(f["seg_len"])
        leaf = [b for b in range(f["bins"]) if f["fname_idx"][b] >= 0]
        n_merged = f["bins"] - len(leaf)
        own = sum(len(f["key_sets"][b]) for b in leaf if b in f["key_sets"])
        decoys = sum(1 for b in leaf if b not in f["key_sets"])
        fill = max(1, int(cap * fill_frac))
        if parent_cap[i] is not None and decoys:
            fill = max(1, min(fill, (parent_cap[i] - own - n_merged * cap) // decoys))
        for b in range(f["bins"]):
            if f["fname_idx"][b] < 0:                      # merged: its keys come from its child
                counts.append(0)
            elif b in f["key_sets"]:
                planted.append((len(counts), f["key_sets"][b]))
                counts.append(len(f["key_sets"][b]))
            else:
                counts.append(fill)
    # planted bins bring their keys (host arrays, a few hundred MB at most); decoy bins have theirs GENERATED on the device from a running
    # index -- no key memory at all, so the index may be far larger than its keys would be (taxor_gpu_index_build_hixf_gen)
    n_bins = len(counts)
    is_planted = np.zeros(n_bins, dtype=bool)
    for g, _ in planted:
        is_planted[g] = True
    cnt = np.array(counts, dtype=np.uint64)
    real = np.where(is_planted, cnt, np.uint64(0)).astype(np.uint64)
    gen_count = np.where(is_planted, np.uint64(0), cnt).astype(np.uint64)
    gen_first = np.zeros(n_bins, dtype=np.uint64)
    np.cumsum(gen_count[:-1], out=gen_first[1:])
    off = np.zeros(n_bins + 1, dtype=np.uint64)
    np.cumsum(real, out=off[1:])
    keys = np.concatenate([np.ascontiguousarray(k_, dtype=np.uint64) for _, k_ in planted]) if planted else np.zeros(0, np.uint64)
    assert keys.size == int(off[-1])
    st = _lib.BuildStats()
    _lib.check(L.taxor_gpu_index_build_hixf_gen(idx._h, _p(keys) if keys.size else None, 0, _p(off), _p(gen_first), _p(gen_count), int(salt) & (2**64 - 1),
                                                int(seed0), C.byref(st)))
    for i, f in enumerate(fs):
        f["seed"] = idx.ixf_seed(i)
    return idx, {kk: getattr(st, kk) for kk, _ in _lib.BuildStats._fields_ if kk != "reserved"}


def seg_len_for(max_bin_elements):
    return int(_lib.lib().taxor_ixf_seg_len(int(max_bin_elements)))


def build_columns(bin_keys, seg_len, seed0, arith=0):
    """XOR-filter columns for several bins of one IXF under a common seed; redraws the seed like the
    reference's reseed loop (src/hixf/build/construct_ixf.cpp:100-108) until every bin peels.
    arith: code of the IXF arithmetic the columns follow (search.arith_code; 0 = the library's reading)."""
    L = _lib.lib()
    seed = int(seed0) & (2**64 - 1)
    rows = 3 * seg_len
    for _ in range(64):
        cols = {}
        ok = True
        for b, keys in bin_keys.items():
            k = np.ascontiguousarray(keys, dtype=np.uint64)
            col = np.zeros(rows, dtype=np.uint8)
            if L.taxor_ixf_build_bin_arith(_p(k), k.size, seed, seg_len, int(arith), _p(col)) != 0:
                ok = False
                break
            cols[b] = col
        if ok:
            return seed, cols
        seed = (seed * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
    raise RuntimeError("XOR filter construction failed for 64 seeds (duplicate keys?)")


def _stride(bins):
    return ((bins + 63) // 64) * 64


def make_layout(planted, root_bins=64, child_bins=64, n_children=4, root_max_elems=None,
                child_max_elems=None, seed=DEFAULT_SEED, with_split=True, with_deep=True, build="host"):
    """Build a 2-3 level HIXF layout with the planted hash sets on full root->leaf paths.

    planted: list of uint64 arrays (distinct syncmer hashes of genome i).  Returns a dict:
      ixfs: [{bins, stride, seg_len, seed, next_ixf, fname_idx, columns{bin: uint8[rows]}, fill_seed}]
      n_user_bins, planted_user_bin[i], depth
    Root: planted[0] split over bins 0..2 (one user bin), planted[1] a single leaf at bin 3, bins
    4..4+n_children-1 merged (children), the rest decoy leaves.  The other planted sets go round-robin into
    the children; in child 0 one of them is split over two bins and, with_deep, another sits in a grandchild
    reached through a merged bin of child 0 (depth-3 chain).
    build="host": XOR-filter columns are constructed here (CPU); build="gpu": only the key sets are returned
    (`key_sets`) and device_index() constructs the columns with taxor_gpu_index_build_ixf."""
    planted = [np.unique(np.ascontiguousarray(p, dtype=np.uint64)) for p in planted]
    P = len(planted)
    assert P >= 2 and root_bins >= 4 + n_children and child_bins >= 8
    rng = np.random.default_rng(seed)
    next_ub = [0]

    def new_ub():
        next_ub[0] += 1
        return next_ub[0] - 1

    planted_ub = [None] * P
    ixfs = []

    def new_ixf(bins, max_elems):
        ixfs.append(dict(bins=bins, stride=_stride(bins), keys={}, next_ixf=None,
                         fname_idx=np.full(bins, -2, dtype=np.int64), child_of={}, max_elems=max_elems))
        return len(ixfs) - 1

    root = new_ixf(root_bins, root_max_elems)
    # --- root leaves ---
    b = 0
    if with_split:
        ub = new_ub()
        planted_ub[0] = ub
        for j in range(3):
            ixfs[root]["keys"][b] = planted[0][j::3]
            ixfs[root]["fname_idx"][b] = ub
            b += 1
    else:
        ub = new_ub()
        planted_ub[0] = ub
        ixfs[root]["keys"][b] = planted[0]
        ixfs[root]["fname_idx"][b] = ub
        b = 3
    ub = new_ub()
    planted_ub[1] = ub
    ixfs[root]["keys"][3] = planted[1]
    ixfs[root]["fname_idx"][3] = ub
    # --- children ---
    children = []
    for c in range(n_children):
        ci = new_ixf(child_bins, child_max_elems)
        children.append(ci)
        ixfs[root]["fname_idx"][4 + c] = -1
        ixfs[root]["child_of"][4 + c] = ci
    rest = list(range(2, P))
    deep_member = rest.pop() if (with_deep and len(rest) >= 2) else None
    slot = [1] * n_children      # next free bin per child (bin 0 stays a decoy)
    for n_, pi in enumerate(rest):
        c = n_ % n_children
        ci = children[c]
        ub = new_ub()
        planted_ub[pi] = ub
        if n_ == 0 and with_split:   # split over two technical bins of child 0
            for j in range(2):
                ixfs[ci]["keys"][slot[c]] = planted[pi][j::2]
                ixfs[ci]["fname_idx"][slot[c]] = ub
                slot[c] += 1
        else:
            ixfs[ci]["keys"][slot[c]] = planted[pi]
            ixfs[ci]["fname_idx"][slot[c]] = ub
            slot[c] += 1
        slot[c] += 1                 # leave a decoy bin between planted ones
        assert slot[c] < child_bins - 2, "child_bins too small for the planted genomes"
    if deep_member is not None:
        gi = new_ixf(child_bins, child_max_elems)
        c0 = children[0]
        mb = child_bins - 2
        ixfs[c0]["fname_idx"][mb] = -1
        ixfs[c0]["child_of"][mb] = gi
        ub = new_ub()
        planted_ub[deep_member] = ub
        ixfs[gi]["keys"][2] = planted[deep_member]
        ixfs[gi]["fname_idx"][2] = ub
    out = _finalize_layout(ixfs, new_ub, rng, build)
    depth = 3 if deep_member is not None else 2
    return dict(ixfs=out, n_user_bins=next_ub[0], planted_user_bin=planted_ub, depth=depth)


def _finalize_layout(ixfs, new_ub, rng, build):
    """decoy leaves get their own user bins; merged bins hold the union of their child's keys; XOR-filter columns are
    constructed (build="host") or left to the device builder as key sets (build="gpu")"""
    for f in ixfs:
        for bb in range(f["bins"]):
            if f["fname_idx"][bb] == -2:
                f["fname_idx"][bb] = new_ub()

    def all_keys(i):
        parts = [k for k in ixfs[i]["keys"].values()]
        for bb, ch in ixfs[i]["child_of"].items():
            parts.append(all_keys(ch))
        return np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)

    for i in range(len(ixfs) - 1, -1, -1):
        for bb, ch in ixfs[i]["child_of"].items():
            ixfs[i]["keys"][bb] = all_keys(ch)
    out = []
    for i, f in enumerate(ixfs):
        mx = max([len(k) for k in f["keys"].values()] + [1])
        cap = f["max_elems"] if f["max_elems"] else mx
        assert cap >= mx, f"IXF {i}: max_elems {cap} < largest bin {mx}"
        seg = seg_len_for(cap)
        nonempty = {bb: k for bb, k in f["keys"].items() if len(k)}   # empty bins stay random fill
        seed0 = int(rng.integers(1, 2**63))
        sd, cols = build_columns(nonempty, seg, seed0) if build == "host" else (seed0, {})
        nx = np.full(f["bins"], i, dtype=np.int64)       # next_ixf_id[i][b] == i  <=>  not merged
        for bb, ch in f["child_of"].items():
            nx[bb] = ch
        out.append(dict(bins=f["bins"], stride=f["stride"], seg_len=seg, seed=sd, next_ixf=nx,
                        fname_idx=f["fname_idx"], columns=cols, fill_seed=int(rng.integers(1, 2**63)),
                        key_sets=nonempty if build != "host" else {}))
    return out


def family_genomes(n_families, family_size, length, seed=DEFAULT_SEED, ladder=(0.001, 0.002, 0.004, 0.008, 0.016, 0.032)):
    """Families of related genomes (strains of one species): every family has a random ancestor, sibling j carries
    substitutions at rate ladder[j % len(ladder)] against it, so two siblings are about 1 - (d_i + d_j) identical --
    99.7 % down to 93.6 % with the default ladder (strains of a species; GTDB clusters species at 95 % ANI).  A read of one sibling then reaches the threshold in several of its
    siblings' bins and in every merged bin above them, like reads against a real, taxonomically clustered index.
    Returns (bases uint8[n*length], offsets uint64[n+1], family int32[n])."""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = n_families * family_size
    codes = np.empty(n * length, dtype=np.uint8)
    fam = np.empty(n, dtype=np.int32)
    for f in range(n_families):
        anc = rng.integers(0, 4, size=length, dtype=np.uint8)
        for j in range(family_size):
            g = f * family_size + j
            fam[g] = f
            d = ladder[j % len(ladder)]
            sib = anc.copy()
            pos = np.flatnonzero(rng.random(length) < d)
            sib[pos] = (sib[pos] + rng.integers(1, 4, size=pos.size, dtype=np.uint8)) & 3   # always a different base
            codes[g * length:(g + 1) * length] = sib
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(length))
    return acgt[codes], offs, fam


def make_family_layout(planted, family, root_bins=64, child_bins=64, n_children=8, spread=4, root_max_elems=None,
                       child_max_elems=None, seed=DEFAULT_SEED, build="host"):
    """HIXF layout for families of related genomes (family_genomes): the members of a family sit in ADJACENT bins of
    `spread` different child IXFs (a taxonomically clustered layout keeps relatives together, and a large clade fills
    several merged bins), every third member is split over two technical bins, member 0 of family 0 is a split root
    leaf over three technical bins, member 0 of family 1 a plain root leaf, and the last member of family 0 sits in a
    grandchild behind a merged bin of its child (depth-3 chain).  A read of one member so passes the threshold in
    several merged bins of the root and in several leaf runs below each.  Same return format as make_layout."""
    planted = [np.unique(np.ascontiguousarray(p, dtype=np.uint64)) for p in planted]
    P = len(planted)
    family = np.asarray(family)
    n_fam = int(family.max()) + 1
    assert P >= 3 and n_children >= spread >= 1 and root_bins >= 4 + n_children
    rng = np.random.default_rng(seed)
    next_ub = [0]

    def new_ub():
        next_ub[0] += 1
        return next_ub[0] - 1

    planted_ub = [None] * P
    ixfs = []

    def new_ixf(bins, max_elems):
        ixfs.append(dict(bins=bins, stride=_stride(bins), keys={}, next_ixf=None,
                         fname_idx=np.full(bins, -2, dtype=np.int64), child_of={}, max_elems=max_elems))
        return len(ixfs) - 1

    root = new_ixf(root_bins, root_max_elems)
    members = [list(np.flatnonzero(family == f)) for f in range(n_fam)]
    # root leaves: a split run of three and a plain leaf
    ub = new_ub()
    m = members[0].pop(0)
    planted_ub[m] = ub
    for j in range(3):
        ixfs[root]["keys"][j] = planted[m][j::3]
        ixfs[root]["fname_idx"][j] = ub
    if n_fam > 1:
        ub = new_ub()
        m = members[1].pop(0)
        planted_ub[m] = ub
        ixfs[root]["keys"][3] = planted[m]
        ixfs[root]["fname_idx"][3] = ub
    children = []
    for c in range(n_children):
        ci = new_ixf(child_bins, child_max_elems)
        children.append(ci)
        ixfs[root]["fname_idx"][4 + c] = -1
        ixfs[root]["child_of"][4 + c] = ci
    deep_member = members[0].pop() if len(members[0]) >= 2 else None
    slot = [1] * n_children                                   # bin 0 of every child stays a decoy
    n_split = 0
    for f in range(n_fam):
        for j, m in enumerate(members[f]):
            c = (f * spread + j % spread) % n_children        # family f occupies children f*spread .. f*spread+spread-1
            ci = children[c]
            ub = new_ub()
            planted_ub[m] = ub
            run = 2 if j % 3 == 2 else 1
            assert slot[c] + run < child_bins - 2, "child_bins too small for the planted families"
            for r in range(run):
                ixfs[ci]["keys"][slot[c]] = planted[m][r::run]
                ixfs[ci]["fname_idx"][slot[c]] = ub
                slot[c] += 1
            n_split += run == 2
    depth = 2
    if deep_member is not None:
        gi = new_ixf(child_bins, child_max_elems)
        c0 = children[0]
        mb = child_bins - 2
        ixfs[c0]["fname_idx"][mb] = -1
        ixfs[c0]["child_of"][mb] = gi
        ub = new_ub()
        planted_ub[deep_member] = ub
        ixfs[gi]["keys"][2] = planted[deep_member]
        ixfs[gi]["fname_idx"][2] = ub
        depth = 3
    out = _finalize_layout(ixfs, new_ub, rng, build)
    return dict(ixfs=out, n_user_bins=next_ub[0], planted_user_bin=planted_ub, depth=depth, split_runs=n_split + 1)


def random_layout(planted, rng, max_depth=4, bins_choices=(5, 17, 40, 64, 100, 130, 300), max_ixfs=12, arith=0):
    """A random HIXF around the planted hash sets, for differential tests: random tree shape, bin counts that are not
    multiples of 16 or 64, planted and decoy split runs of random length at random positions (so that runs straddle
    16-bin units and 64-bin rows), merged bins anywhere, empty bins.  Same return format as make_layout."""
    planted = [np.unique(np.ascontiguousarray(p, dtype=np.uint64)) for p in planted]
    todo = list(range(len(planted)))
    rng.shuffle(todo)
    next_ub = [0]
    planted_ub = [None] * len(planted)
    ixfs = []

    def new_ub():
        next_ub[0] += 1
        return next_ub[0] - 1

    def node(depth):
        me = len(ixfs)
        bins = int(rng.choice(bins_choices))
        f = dict(bins=bins, stride=_stride(bins), keys={}, fname_idx=np.full(bins, -2, dtype=np.int64), child_of={})
        ixfs.append(f)
        b = 0
        while b < bins:
            role = rng.random()
            left = bins - b
            if role < 0.18 and todo:                                   # planted genome, possibly split over a run
                pi = todo.pop()
                run = int(min(left, rng.integers(1, 7)))
                ub = new_ub()
                planted_ub[pi] = ub
                for j in range(run):
                    f["keys"][b + j] = planted[pi][j::run]
                    f["fname_idx"][b + j] = ub
                b += run
            elif role < 0.30 and depth < max_depth and len(ixfs) < max_ixfs:   # merged bin -> child
                f["fname_idx"][b] = -1
                f["child_of"][b] = node(depth + 1)
                b += 1
            elif role < 0.45:                                          # decoy split run (random fill, never prunable early)
                run = int(min(left, rng.integers(2, 24)))
                ub = new_ub()
                for j in range(run):
                    f["fname_idx"][b + j] = ub
                b += run
            else:                                                      # decoy leaf
                f["fname_idx"][b] = new_ub()
                b += 1
        return me

    node(1)
    while todo:                                                        # genomes that found no place: root leaves, overwriting decoys
        pi = todo.pop()
        f = ixfs[0]
        cand = [b for b in range(f["bins"]) if b not in f["keys"] and b not in f["child_of"] and
                (b == 0 or f["fname_idx"][b - 1] != f["fname_idx"][b]) and (b + 1 == f["bins"] or f["fname_idx"][b + 1] != f["fname_idx"][b])]
        if not cand:
            break
        b = int(rng.choice(cand))
        f["keys"][b] = planted[pi]
        planted_ub[pi] = int(f["fname_idx"][b])

    def all_keys(i):
        parts = [k for k in ixfs[i]["keys"].values()]
        for ch in ixfs[i]["child_of"].values():
            parts.append(all_keys(ch))
        return np.unique(np.concatenate(parts)) if parts else np.zeros(0, np.uint64)

    for f in ixfs:                                                     # what a build front end hands to the builder
        f["leaf_keys"] = {bb: k for bb, k in f["keys"].items() if len(k)}
    for i in range(len(ixfs) - 1, -1, -1):
        for bb, ch in ixfs[i]["child_of"].items():
            ixfs[i]["keys"][bb] = all_keys(ch)
    out = []
    depth = [1]

    def walk(i, d):
        depth[0] = max(depth[0], d)
        for ch in ixfs[i]["child_of"].values():
            walk(ch, d + 1)

    walk(0, 1)
    for i, f in enumerate(ixfs):
        mx = max([len(k) for k in f["keys"].values()] + [1])
        seg = seg_len_for(int(mx * float(rng.uniform(1.0, 1.5))) + 1)
        nonempty = {bb: k for bb, k in f["keys"].items() if len(k)}
        sd, cols = build_columns(nonempty, seg, int(rng.integers(1, 2**63)), arith)
        nx = np.full(f["bins"], i, dtype=np.int64)
        for bb, ch in f["child_of"].items():
            nx[bb] = ch
        out.append(dict(bins=f["bins"], stride=f["stride"], seg_len=seg, seed=sd, next_ixf=nx, fname_idx=f["fname_idx"],
                        columns=cols, fill_seed=int(rng.integers(1, 2**63)), key_sets={}, leaf_keys=f["leaf_keys"]))
    return dict(ixfs=out, n_user_bins=next_ub[0], planted_user_bin=planted_ub, depth=depth[0])


def materialize_host(layout):
    """Random-fill every IXF on the host and write the planted columns -> list usable by GpuIndex and by
    a checker (small layouts only: allocates rows*stride bytes per IXF)."""
    res = []
    for f in layout["ixfs"]:
        rows = 3 * f["seg_len"]
        rng = np.random.default_rng(f["fill_seed"])
        data = rng.integers(0, 256, size=(rows, f["stride"]), dtype=np.uint8)
        for b, col in f["columns"].items():
            data[:, b] = col
        res.append(dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"],
                        next_ixf=f["next_ixf"], fname_idx=f["fname_idx"], data=data.reshape(-1)))
    return res


def device_index(layout, k=22, s=12, t=5, device=0, use_syncmer=True, window_size=None, scaling=1):
    """Create the index directly in HBM: rows are filled by a device kernel, planted columns uploaded."""
    from .search import GpuIndex
    ixfs = [dict(bins=f["bins"], stride=f["stride"], seg_len=f["seg_len"], seed=f["seed"],
                 next_ixf=f["next_ixf"], fname_idx=f["fname_idx"], data=None) for f in layout["ixfs"]]
    idx = GpuIndex(ixfs, layout["n_user_bins"], k, s, t, device, use_syncmer=use_syncmer, window_size=window_size, scaling=scaling)
    for i, f in enumerate(layout["ixfs"]):
        idx.fill_random(i, f["fill_seed"])
        for b, col in f["columns"].items():
            idx.upload_bin(i, b, col)
        if f.get("key_sets"):                      # columns constructed on the GPU; it may re-seed the IXF
            f["seed"], _ = idx.build_ixf(i, f["key_sets"], seed0=f["seed"])
    return idx
