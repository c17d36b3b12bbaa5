"""The bench's family workload (taxor_amd.synth.family_genomes / make_family_layout) on the CPU: layout invariants, and
-- through the oracle -- that it is the workload it claims to be: a read of one strain reaches the threshold in several
merged bins and several sibling bins ("a workload where pruning is not a best case").  Also pins the read
generator's independence across reads (round 1 seeded read i with seed + i * splitmix64's own increment, which made all
reads replay one error stream)."""
import numpy as np

from oracle import oracle as orc
from taxor_amd import synth


def _family_case():
    g, go, fam = synth.family_genomes(2, 16, 40000, seed=9)
    planted = [orc.seq_to_syncmers(bytes(g[int(go[i]):int(go[i + 1])])) for i in range(32)]
    lay = synth.make_family_layout(planted, fam, root_bins=72, child_bins=40, n_children=8, spread=4, seed=3)
    return g, go, fam, planted, lay


def test_family_genomes_identity_ladder():
    g, go, fam = synth.family_genomes(3, 6, 20000, seed=1, ladder=(0.001, 0.004, 0.016))
    assert fam.tolist() == [0] * 6 + [1] * 6 + [2] * 6
    G = [g[int(go[i]):int(go[i + 1])] for i in range(18)]
    d01 = float((G[0] != G[1]).mean())            # siblings 0 and 1: 0.1 % + 0.4 % apart
    d12 = float((G[1] != G[2]).mean())            # 0.4 % + 1.6 %
    assert 0.003 < d01 < 0.007 and 0.015 < d12 < 0.025
    assert float((G[0] != G[6]).mean()) > 0.7     # different families are unrelated


def test_family_layout_invariants():
    g, go, fam, planted, lay = _family_case()
    ix = lay["ixfs"]
    root = ix[0]
    assert lay["depth"] == 3 and lay["split_runs"] >= 8
    assert all(u is not None for u in lay["planted_user_bin"])
    # the root: a split run of three technical bins, a plain leaf, eight merged bins
    assert root["fname_idx"][0] == root["fname_idx"][1] == root["fname_idx"][2] != root["fname_idx"][3]
    merged = [b for b in range(root["bins"]) if root["fname_idx"][b] == -1]
    assert merged == list(range(4, 12)) and sorted(int(root["next_ixf"][b]) for b in merged) == list(range(1, 9))
    # siblings sit in ADJACENT bins of `spread` different children
    ub_of = {int(u): i for i, u in enumerate(lay["planted_user_bin"])}
    for f in (0, 1):
        kids = set()
        for c in range(1, 9):
            members = [ub_of[int(u)] for u in ix[c]["fname_idx"] if int(u) in ub_of]
            if any(fam[m] == f for m in members):
                kids.add(c)
                bins_f = [b for b in range(ix[c]["bins"]) if int(ix[c]["fname_idx"][b]) in ub_of and fam[ub_of[int(ix[c]["fname_idx"][b])]] == f]
                assert bins_f == list(range(min(bins_f), max(bins_f) + 1))        # contiguous
        assert len(kids) == 4


def test_family_workload_has_many_tuples_per_read():
    g, go, fam, planted, lay = _family_case()
    host = synth.materialize_host(lay)
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    bases, offs, origin = synth.synth_reads(g, go, 200, 8000, error_rate=0.02, frac_random=0.1, seed=3)
    nh, off, ub, cnt, visited = h.search_batch(bases, offs, threads=4)
    per = np.diff(off.astype(np.int64))
    planted_reads = origin >= 0
    assert per[planted_reads].mean() > 3.0                 # several sibling bins per read (unrelated genomes: ~1)
    assert (per[~planted_reads] == 0).all()
    # the read's own strain is among its tuples
    own = np.array([lay["planted_user_bin"][o] if o >= 0 else -1 for o in origin])
    hit = [own[i] in ub[int(off[i]):int(off[i + 1])].tolist() for i in range(200) if origin[i] >= 0]
    assert np.mean(hit) > 0.9
    # a merged bin of the root answers for every key of every member below it (it holds the union of its child's keys)
    ub_of = {int(u): i for i, u in enumerate(lay["planted_user_bin"])}
    for b in range(4, 12):
        child = int(host[0]["next_ixf"][b])
        for u in set(int(x) for x in host[child]["fname_idx"] if int(x) in ub_of):
            m = ub_of[u]
            assert h.ixf_bulk_count(0, planted[m])[b] == len(np.unique(planted[m]))
    # more than the root is visited: bytes per read well above n_h * 3 * root bins
    root_only = int(nh.astype(np.int64).sum()) * 3 * host[0]["bins"]
    assert visited > 1.3 * root_only


def test_synth_reads_have_independent_error_streams():
    """k-mer survival must not depend on the position in the read, and two reads must not share error positions"""
    g, go = synth.random_genomes(1, 120000, seed=5)
    G = bytes(g)
    K = 22
    S = set(G[i:i + K] for i in range(len(G) - K + 1))
    bases, offs, origin = synth.synth_reads(g, go, 40, 6000, error_rate=0.04, frac_random=0.0, seed=3)
    per_kb = np.zeros(6)
    for i in range(40):
        r = bytes(bases[int(offs[i]):int(offs[i + 1])])
        hit = np.array([r[p:p + K] in S for p in range(6000 - K + 1)])
        per_kb += [hit[b * 1000:(b + 1) * 1000].mean() for b in range(6)]
    per_kb /= 40
    assert per_kb.max() - per_kb.min() < 0.06, per_kb       # round 1: 0.36 ... 0.63 along the read
    assert 0.40 < per_kb.mean() < 0.52                      # (1 - 0.036)^22 = 0.446
