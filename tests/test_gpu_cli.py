"""End to end through the drop-in surface: `taxor search --index-file x.hixf --query-file reads.fq ...`
(the C++ host in taxor_amd/csrc) must write byte-identical TSV to what the reference's per-read driver
(taxor_search.cpp:196-313,340-360) produces from the oracle's tuples."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from taxor_amd import GpuIndex, Searcher, synth
from taxor_amd.hixf_file import store_hixf
from tests.test_hixf_file_cpu import HEADER, expected_lines, make_species

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAXOR = os.path.join(ROOT, "taxor_amd", "taxor")


def _setup(tmp_path, seed):
    g, go = synth.random_genomes(7, 15000, seed=seed)
    bins = 64
    dummy = GpuIndex([dict(bins=bins, stride=64, seg_len=16, seed=1, next_ixf=np.zeros(bins, np.int64),
                           fname_idx=np.arange(bins), data=np.zeros(3 * 16 * 64, np.uint8))], bins)
    hs = Searcher(dummy, ratio=0.5)
    hoff, hashes = hs.seq_to_syncmers(g, go)
    hs.close()
    dummy.close()
    planted = [hashes[int(hoff[i]):int(hoff[i + 1])] for i in range(7)]
    lay = synth.make_layout(planted, root_bins=68, child_bins=40, n_children=3, seed=seed)
    host = synth.materialize_host(lay)
    sp = make_species(lay)
    path = tmp_path / f"idx{seed}.hixf"
    store_hixf(path, host, lay["n_user_bins"], sp)
    return g, go, host, sp, path


def _expected(host, sp, ids, reads, err=0.04, percentage=-1.0, arith=0):
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host], arith=arith)
    norm = [orc.dna4_normalise(r) for r in reads]
    B = np.frombuffer(b"".join(norm), dtype=np.uint8)
    O = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    nh, off, ub, cnt, _ = h.search_batch(B, O, err=err, percentage=percentage, threads=4)
    out = ""
    for i, rid in enumerate(ids):
        tup = [(int(a), int(b)) for a, b in zip(ub[int(off[i]):int(off[i + 1])], cnt[int(off[i]):int(off[i + 1])])]
        out += expected_lines(sp, rid, len(reads[i]), int(nh[i]), tup)
    return out


def test_cli_fastq_gz_fasta_multi(tmp_path):
    g, go, host, sp, idx_path = _setup(tmp_path, 31)
    bases, offs, origin = synth.synth_reads(g, go, 180, 1500, error_rate=0.02, frac_random=0.2, seed=5)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(180)]
    reads += [b"ACGTNNNNRYKM" * 30, b"ACGTACGT", bytes(g[:22])]
    ids = [f"read_{i} len={len(r)} some description" for i, r in enumerate(reads)]
    # FASTQ, gzip
    fq = tmp_path / "reads.fastq.gz"
    with gzip.open(fq, "wb") as f:
        for rid, r in zip(ids, reads):
            f.write(b"@" + rid.encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
    # FASTA, multi-line, plain
    fa = tmp_path / "reads2.fa"
    ids2 = [f"fa_{i}" for i in range(40)]
    reads2 = reads[100:140]
    with open(fa, "wb") as f:
        for rid, r in zip(ids2, reads2):
            f.write(b">" + rid.encode() + b"\n")
            for j in range(0, len(r), 70):
                f.write(r[j:j + 70] + b"\n")
    out = tmp_path / "out.tsv"
    cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", f"{fq},{fa}", "--output-file",
                         str(out), "--threads", "4", "--batch-reads", "50"], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr
    assert "checking input ... done!" in cp.stdout and "use syncmer model" in cp.stdout
    want = HEADER + _expected(host, sp, ids, reads) + _expected(host, sp, ids2, reads2)
    assert open(out).read() == want

    # the parser's other spelling (--opt=value) and the reference's hidden no-op flags (taxor_search.cpp:68-79)
    out_eq = tmp_path / "out_eq.tsv"
    cp = subprocess.run([TAXOR, "search", f"--index-file={idx_path}", f"--query-file={fq},{fa}", f"--output-file={out_eq}",
                         "--threads=4", "--batch-reads=50", "--debug", "--output-verbose-statistics"], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr
    assert open(out_eq).read() == want

    # the report is written in pieces by the threads that format them, each in its turn: pieces of 16 KiB on eight threads, into a
    # file and into a pipe
    for kind in ("file", "pipe"):
        out_p = tmp_path / f"out_pieces_{kind}.tsv"
        env = dict(os.environ, TAXOR_TUNING="1", TAXOR_CLI_PIECE_KB="16", TAXOR_CLI_TRACE="1")
        cmd = [TAXOR, "search", "--index-file", str(idx_path), "--query-file", f"{fq},{fa}", "--threads", "16", "--batch-reads", "50"]
        if kind == "pipe":
            fifo = tmp_path / "report.fifo"
            os.mkfifo(fifo)
            with open(out_p, "wb") as sink:
                p2 = subprocess.Popen(["cat", str(fifo)], stdout=sink)
                cp = subprocess.run(cmd + ["--output-file", str(fifo)], capture_output=True, text=True, timeout=300, env=env)
                assert cp.returncode == 0 and p2.wait(timeout=60) == 0, cp.stderr
            assert open(out_p).read() == want
            assert "inside write()" in cp.stderr
        else:
            cp = subprocess.run(cmd + ["--output-file", str(out_p)], capture_output=True, text=True, timeout=300, env=env)
            assert cp.returncode == 0, cp.stderr
            assert open(out_p).read() == want, kind
            import re
            assert int(re.search(r"in (\d+) pieces", cp.stderr).group(1)) > 6, cp.stderr

    # two (three) workers sharding the chunks -- here on the same device -- must give the identical file in input order
    for devs in ("0,0", "0,0,0"):
        out2 = tmp_path / "out_multi.tsv"
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", f"{fq},{fa}", "--output-file",
                             str(out2), "--gpu-list", devs, "--batch-reads", "17"], capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0, cp.stderr
        assert open(out2).read() == want

    # the communicator paths: rounds of batches + one gather per round (host transport: a device may repeat; RCCL: one rank)
    for extra in (["--gpu-list", "0,0", "--gather", "host"], ["--gpu-list", "0,0,0"], ["--gpu", "0", "--gather", "rccl"],
                  ["--gpu-list", "0,0", "--gather", "none"]):
        out3 = tmp_path / "out_comm.tsv"
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", f"{fq},{fa}", "--output-file",
                             str(out3), "--batch-reads", "23", "--threads", "3"] + extra, capture_output=True, text=True, timeout=300,
                            env=dict(os.environ, TAXOR_TUNING="1", TAXOR_CLI_TRACE="1"))
        assert cp.returncode == 0, cp.stderr
        assert open(out3).read() == want, extra
        if "none" not in extra:
            assert "gathers (" in cp.stderr
    cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(fa), "--output-file", str(out), "--gpu-list", "0,0",
                         "--gather", "rccl"], capture_output=True, text=True, timeout=300)
    assert cp.returncode != 0 and "listed twice" in cp.stderr and "--gather host" in cp.stderr

    # two real devices (skipped on the usual one-GPU box): RCCL between them against both other transports, byte for byte
    import torch
    if torch.cuda.device_count() >= 2:
        for extra in (["--gpus", "2", "--gather", "rccl"], ["--gpus", "2"], ["--gpus", "2", "--gather", "host"], ["--gpus", "2", "--gather", "none"]):
            out4 = tmp_path / "out_two.tsv"
            cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", f"{fq},{fa}", "--output-file",
                                 str(out4), "--batch-reads", "23", "--threads", "3"] + extra, capture_output=True, text=True, timeout=300)
            assert cp.returncode == 0, cp.stderr
            assert open(out4).read() == want, extra

    # --error-rate / --percentage change the threshold exactly like the reference's models
    for extra, kw in ((["--error-rate", "0.1"], dict(err=0.1)), (["--percentage", "0.3"], dict(percentage=0.3))):
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(fa), "--output-file", str(out)]
                            + extra, capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0, cp.stderr
        assert open(out).read() == HEADER + _expected(host, sp, ids2, reads2, **kw)


def test_cli_errors(tmp_path):
    g, go, host, sp, idx_path = _setup(tmp_path, 32)
    out = tmp_path / "o.tsv"
    fa = tmp_path / "r.fa"
    open(fa, "w").write(">a\nACGT\n")
    cases = [(["--index-file", str(tmp_path / "nope.hixf"), "--query-file", str(fa)], "does not exist"),
             (["--index-file", str(idx_path), "--query-file", str(tmp_path / "nope.fa")], "does not exist"),
             (["--index-file", str(idx_path), "--query-file", str(fa), "--threads", "64"], "range"),
             (["--index-file", str(idx_path), "--query-file", str(fa), "--error-rate", "0.5"], "threshold model"),
             (["--query-file", str(fa)], "required")]
    for args, needle in cases:
        cp = subprocess.run([TAXOR, "search", "--output-file", str(out)] + args, capture_output=True, text=True, timeout=120)
        assert cp.returncode != 0
        assert "[TAXOR SEARCH ERROR]" in cp.stderr and needle in cp.stderr, (args, cp.stderr)
    bad = tmp_path / "bad.fa"
    open(bad, "w").write(">a\nACGTACGTACGTACGTACGTACGTACGT#ACGT\n")
    cp = subprocess.run([TAXOR, "search", "--output-file", str(out), "--index-file", str(idx_path), "--query-file", str(bad)],
                        capture_output=True, text=True, timeout=120)
    assert cp.returncode != 0 and "dna15" in cp.stderr


@pytest.mark.parametrize("k,w,msg", [(20, 20, "use kmer-model"), (20, 30, "use frac minhash")])
def test_cli_index_built_without_syncmers(tmp_path, k, w, msg):
    """a .hixf with use_syncmer = 0: minimiser hashing + the k-mer / FracMinHash threshold model end to end"""
    g, go = synth.random_genomes(7, 6000, seed=k + w)
    planted = [orc.minimiser_hash(bytes(g[int(go[i]):int(go[i + 1])]), k, w) for i in range(7)]
    lay = synth.make_layout(planted, root_bins=68, child_bins=40, n_children=3, seed=3)
    host = synth.materialize_host(lay)
    sp = make_species(lay)
    idx_path = tmp_path / "kmer.hixf"
    store_hixf(idx_path, host, lay["n_user_bins"], sp, k=k, s=0, t=0, window_size=w, use_syncmer=False)
    bases, offs, origin = synth.synth_reads(g, go, 120, 1200, error_rate=0.01, frac_random=0.2, seed=6)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(120)] + [b"ACGTNNNNRYKM" * 30, b"ACGTACGT", bytes(g[:k])]
    ids = [f"r{i}" for i in range(len(reads))]
    fa = tmp_path / "r.fa"
    with open(fa, "wb") as f:
        for rid, r in zip(ids, reads):
            f.write(b">" + rid.encode() + b"\n" + r + b"\n")
    out = tmp_path / "o.tsv"
    cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(fa), "--output-file", str(out),
                         "--batch-reads", "40", "--threads", "2"], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr
    assert msg in cp.stdout
    h = orc.Hixf(host, [f["next_ixf"] for f in host], [f["fname_idx"] for f in host])
    B = np.frombuffer(b"".join(orc.dna4_normalise(r) for r in reads), dtype=np.uint8)
    O = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    nh, off, ub, cnt, _ = h.search_batch(B, O, k=k, threads=4, window=w)
    want = HEADER
    for i, rid in enumerate(ids):
        tup = [(int(a), int(b)) for a, b in zip(ub[int(off[i]):int(off[i + 1])], cnt[int(off[i]):int(off[i + 1])])]
        want += expected_lines(sp, rid, len(reads[i]), int(nh[i]), tup)
    assert open(out).read() == want
    assert want.count("\n") > 50 and "Organism" in want      # the control reads do classify


def test_cli_verify_positive_control(tmp_path):
    """`taxor verify`: windows of an indexed genome must match themselves; a foreign genome (or an index whose
    arithmetic is read wrongly) must be reported as FAIL"""
    g, go, host, sp, idx_path = _setup(tmp_path, 33)
    inside, outside = tmp_path / "in.fa", tmp_path / "out.fa"
    with open(inside, "wb") as f:
        f.write(b">genome_2 first half\n" + bytes(g[int(go[2]):int(go[2]) + 7000]) + b"\n>genome_2 second half\n" + bytes(g[int(go[2]) + 7000:int(go[3])]) + b"\n")
    rng = np.random.default_rng(1)
    with open(outside, "wb") as f:
        f.write(b">foreign\n" + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=15000)) + b"\n")
    cp = subprocess.run([TAXOR, "verify", "--index-file", str(idx_path), "--genome-file", str(inside), "--reads", "200", "--read-len", "1500"],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0 and "PASS" in cp.stdout, cp.stdout + cp.stderr
    assert "median 1.0000" in cp.stdout
    cp = subprocess.run([TAXOR, "verify", "--index-file", str(idx_path), "--genome-file", str(outside), "--reads", "200", "--read-len", "1500"],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 2 and "FAIL" in cp.stdout, cp.stdout + cp.stderr
    # the same fingerprints declared with another k: every hash differs -> FAIL
    wrong = tmp_path / "wrong_k.hixf"
    store_hixf(wrong, host, len(sp) if False else max(s["user_bin"] for s in sp) + 1, sp, k=24, s=12, t=6)
    cp = subprocess.run([TAXOR, "verify", "--index-file", str(wrong), "--genome-file", str(inside), "--reads", "200", "--read-len", "1500"],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 2 and "FAIL" in cp.stdout, cp.stdout + cp.stderr


def test_cli_many_query_files_read_concurrently(tmp_path):
    """several gzip / plain query files: read concurrently, index loaded once, output in the order of --query-file"""
    g, go, host, sp, idx_path = _setup(tmp_path, 34)
    bases, offs, origin = synth.synth_reads(g, go, 400, 900, error_rate=0.02, frac_random=0.2, seed=8)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(400)]
    cuts = [0, 150, 150, 151, 260, 300, 390, 400]            # one empty file, one with a single read
    files, want = [], HEADER
    for j in range(len(cuts) - 1):
        ids = [f"f{j}_r{i}" for i in range(cuts[j], cuts[j + 1])]
        rs = reads[cuts[j]:cuts[j + 1]]
        gz = j % 3 != 2
        path = tmp_path / (f"q{j}.fastq.gz" if gz else f"q{j}.fastq")
        with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
            for rid, r in zip(ids, rs):
                f.write(b"@" + rid.encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
        files.append(str(path))
        want += _expected(host, sp, ids, rs) if rs else ""
    out = tmp_path / "many.tsv"
    for extra in (["--threads", "8", "--batch-reads", "40"], ["--threads", "1"], ["--threads", "3", "--gpu-list", "0,0", "--batch-reads", "25"]):
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", ",".join(files), "--output-file", str(out)] + extra,
                            capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0, cp.stderr
        assert cp.stdout.count("use syncmer model") == 1          # the index was loaded once for all files
        assert open(out).read() == want, extra


def test_cli_plain_then_large_gzip_batch(tmp_path):
    """A batch buffer page-locked while a plain file was read (ranged parser) is recycled for a gzip file whose batch
    outgrows it (sequential reader): the registration must be dropped before the buffer is refilled."""
    g, go, host, sp, idx_path = _setup(tmp_path, 35)
    bases, offs, origin = synth.synth_reads(g, go, 3300, 1500, error_rate=0.02, frac_random=0.2, seed=9)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(3300)]
    ids = [f"q{i}" for i in range(3300)]
    plain, gz = tmp_path / "a.fastq", tmp_path / "b.fastq.gz"
    with open(plain, "wb") as f:                                   # 300 reads = 0.45 MB: fits the 2 MB minimum reserve
        for rid, r in zip(ids[:300], reads[:300]):
            f.write(b"@" + rid.encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
    with gzip.open(gz, "wb", compresslevel=1) as f:                # one 4.5 MB batch: the recycled string must grow
        for rid, r in zip(ids[300:], reads[300:]):
            f.write(b"@" + rid.encode() + b"\n" + r + b"\n+\n" + b"I" * len(r) + b"\n")
    out = tmp_path / "o.tsv"
    want = HEADER + _expected(host, sp, ids, reads)
    for extra in (["--threads", "1"], ["--threads", "4"]):
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", f"{plain},{gz},{plain}",
                             "--output-file", str(out), "--batch-reads", "100000"] + extra, capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0, cp.stderr
        assert open(out).read() == want + _expected(host, sp, ids[:300], reads[:300])


def test_cli_sequences_with_spaces_and_digits(tmp_path):
    """seqan3 drops white space and digits inside sequences (numbered / column-formatted files); the CLI cleans a batch
    only when the device reports a character outside dna15, and a really foreign character still fails"""
    g, go, host, sp, idx_path = _setup(tmp_path, 36)
    bases, offs, origin = synth.synth_reads(g, go, 60, 1300, error_rate=0.02, frac_random=0.2, seed=10)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(60)]
    ids = [f"gb_{i}" for i in range(60)]
    fa = tmp_path / "numbered.fa"
    with open(fa, "wb") as f:
        for rid, r in zip(ids, reads):
            f.write(b">" + rid.encode() + b"\n")
            for a in range(0, len(r), 60):                          # GenBank-like: position, blocks of ten
                row = r[a:a + 60]
                f.write(b"%9d " % (a + 1) + b" ".join(row[j:j + 10] for j in range(0, len(row), 10)) + b"\t\n")
    out = tmp_path / "o.tsv"
    for extra in (["--threads", "1"], ["--threads", "4", "--batch-reads", "7"]):
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(fa), "--output-file", str(out)] + extra,
                            capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0, cp.stderr
        assert open(out).read() == HEADER + _expected(host, sp, ids, reads)
    bad = tmp_path / "bad.fa"
    open(bad, "wb").write(b">x\nACGT ACGT 12 ACGTACGTACGTACGTACGT#ACGT\n")
    cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(bad), "--output-file", str(out)],
                        capture_output=True, text=True, timeout=120)
    assert cp.returncode != 0 and "dna15" in cp.stderr


def _foreign_xor_column(keys, seed, seg_len):
    """XOR-filter column of one bin under ANOTHER reading of the un-vendored IXF arithmetic than ixf_arith.h's:
    murmur64(key ^ seed) instead of (key + seed), rows by modulo instead of multiply-shift, fingerprint = low byte."""
    M = (1 << 64) - 1

    def murmur64(h):
        h ^= h >> 33
        h = (h * 0xff51afd7ed558ccd) & M
        h ^= h >> 33
        h = (h * 0xc4ceb9fe1a85ec53) & M
        return h ^ (h >> 33)

    def probe(key):
        h = murmur64((int(key) ^ seed) & M)
        rows = []
        for i in range(3):
            rot = ((h << (21 * i)) | (h >> (64 - 21 * i))) & M if i else h
            rows.append((rot & 0xFFFFFFFF) % seg_len + i * seg_len)
        return rows, h & 0xFF

    pr = {int(k): probe(k) for k in keys}
    deg, acc = {}, {}
    for k, (rows, _) in pr.items():
        for r in rows:
            deg[r] = deg.get(r, 0) + 1
            acc[r] = acc.get(r, 0) ^ k
    stack, queue = [], [r for r, d in deg.items() if d == 1]
    while queue:
        r = queue.pop()
        if deg.get(r, 0) != 1:
            continue
        k = acc[r]
        stack.append((k, r))
        for q in pr[k][0]:
            deg[q] -= 1
            acc[q] ^= k
            if deg[q] == 1:
                queue.append(q)
    if len(stack) != len(pr):
        return None
    col = np.zeros(3 * seg_len, dtype=np.uint8)
    for k, r in reversed(stack):
        rows, fp = pr[k]
        col[r] = fp ^ col[rows[0]] ^ col[rows[1]] ^ col[rows[2]] ^ col[r]
    return col


def test_cli_verify_variant_scan_names_a_foreign_arithmetic(tmp_path):
    """`taxor verify --variants`: an index whose fingerprints follow another reading of the IXF arithmetic than this
    library's must FAIL the positive control and the scan must name that reading; the library's own index must be
    recognised as the library's reading."""
    g, go = synth.random_genomes(1, 12000, seed=77)
    genome = bytes(g)
    keys = np.unique(orc.seq_to_syncmers(genome))
    bins, stride = 40, 64
    seg = synth.seg_len_for(len(keys) + 50)
    rng = np.random.default_rng(5)
    col = None
    for seed in (0x1234567890ABCDEF, 0x0FEDCBA987654321, 0x55AA55AA55AA55AA):
        col = _foreign_xor_column(keys, seed, seg)
        if col is not None:
            break
    assert col is not None
    data = rng.integers(0, 256, size=(3 * seg, stride), dtype=np.uint8)
    data[:, 7] = col
    host = [dict(bins=bins, stride=stride, seg_len=seg, seed=seed, data=data.reshape(-1), next_ixf=np.zeros(bins, np.int64),
                 fname_idx=np.arange(bins, dtype=np.int64))]
    sp = [dict(organism_name=f"O{i}", accession_id=f"A{i}", taxid=str(i), taxnames_string="n", taxid_string="t", user_bin=i, seq_len=1)
          for i in range(bins)]
    foreign = tmp_path / "foreign.hixf"
    store_hixf(foreign, host, bins, sp)
    fa = tmp_path / "g.fa"
    fa.write_bytes(b">g\n" + genome + b"\n")
    cp = subprocess.run([TAXOR, "verify", "--index-file", str(foreign), "--genome-file", str(fa), "--reads", "60", "--read-len", "1500"],
                        capture_output=True, text=True, timeout=600)
    assert cp.returncode == 2 and "FAIL" in cp.stdout, cp.stdout + cp.stderr
    assert "ANOTHER reading" in cp.stdout, cp.stdout
    first = [l for l in cp.stdout.splitlines() if l.startswith("  1.0000")][0]
    assert "murmur64 as h(key ^ seed)" in first and "(u32)rot % seg" in first and "fingerprint (u8)h," in first and "data[row*pitch + bin]" in first, first
    assert f"seed {seed}" in first and "rotl(h, 21*i)" in first
    # ... and the reading it names is a run-time choice, not a rebuild: `taxor search --ixf-arithmetic <spec>`
    # classifies the foreign index, bit for bit like the oracle parametrised the same way; without it the index answers at the
    # false-positive floor
    spec = [l for l in cp.stdout.splitlines() if l.startswith("search it with:")][0].split("--ixf-arithmetic ")[1].split()[0]
    assert spec == "kh=0,sm=1,rot=21,red=1,fp=1", spec
    from taxor_amd.search import arith_code
    code = arith_code(key_hash=0, seed_mode=1, rot=21, reduce=1, fp_mode=1)
    assert code != 0
    bases, offs, _ = synth.synth_reads(g, np.array([0, len(genome)], np.uint64), 150, 1500, error_rate=0.01, frac_random=0.2, seed=9)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(150)]
    ids = [f"r{i}" for i in range(150)]
    rfa = tmp_path / "reads.fa"
    with open(rfa, "wb") as f:
        for rid, r in zip(ids, reads):
            f.write(b">" + rid.encode() + b"\n" + r + b"\n")
    out = tmp_path / "foreign.tsv"
    cp = subprocess.run([TAXOR, "search", "--index-file", str(foreign), "--query-file", str(rfa), "--output-file", str(out), "--ixf-arithmetic", spec],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0, cp.stderr
    assert "[TAXOR SEARCH WARNING]" in cp.stderr and "--ixf-arithmetic kh=0,sm=1" in cp.stderr
    want = HEADER + _expected(host, sp, ids, reads, arith=code)
    got = open(out).read()
    assert got == want
    hit_lines = [l for l in got.splitlines()[1:] if l.split("\t")[1] != "-"]
    assert len(hit_lines) > 80 and all(l.split("\t")[1] == "A7" for l in hit_lines)      # the genome sits in bin 7
    cp = subprocess.run([TAXOR, "search", "--index-file", str(foreign), "--query-file", str(rfa), "--output-file", str(out)],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0 and open(out).read() == HEADER + _expected(host, sp, ids, reads)       # this build's reading: no hits
    assert not [l for l in open(out).read().splitlines()[1:] if l.split("\t")[1] != "-"]
    cp = subprocess.run([TAXOR, "verify", "--index-file", str(foreign), "--genome-file", str(fa), "--reads", "60", "--read-len", "1500",
                         "--ixf-arithmetic", spec], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0 and "PASS" in cp.stdout, cp.stdout + cp.stderr
    # the library's own files: recognised as such
    g2, go2, host2, sp2, own = _setup(tmp_path, 37)
    fa2 = tmp_path / "g2.fa"
    fa2.write_bytes(b">g\n" + bytes(g2[int(go2[3]):int(go2[4])]) + b"\n")
    cp = subprocess.run([TAXOR, "verify", "--variants", "--index-file", str(own), "--genome-file", str(fa2), "--reads", "60", "--read-len", "1500"],
                        capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0 and "PASS" in cp.stdout and "the file follows this library's reading" in cp.stdout, cp.stdout + cp.stderr


def test_cli_expect_tsv_compares_per_read(tmp_path):
    """`--expect ref.tsv`: per-read comparison with a TSV written for the same input -- reads in any order (the reference
    writes them in completion order when --threads > 1), lines of a read in order; a changed count must be reported"""
    g, go, host, sp, idx_path = _setup(tmp_path, 38)
    bases, offs, origin = synth.synth_reads(g, go, 120, 1400, error_rate=0.02, frac_random=0.2, seed=12)
    reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(120)]
    ids = [f"e{i}" for i in range(120)]
    fa = tmp_path / "r.fa"
    with open(fa, "wb") as f:
        for rid, r in zip(ids, reads):
            f.write(b">" + rid.encode() + b"\n" + r + b"\n")
    per_read = [_expected(host, sp, [rid], [r]) for rid, r in zip(ids, reads)]
    order = np.random.default_rng(3).permutation(120)
    ref = tmp_path / "ref.tsv"
    ref.write_text(HEADER + "".join(per_read[i] for i in order))             # the reference's order: arbitrary
    out = tmp_path / "o.tsv"
    cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(fa), "--output-file", str(out), "--expect", str(ref)],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0 and "120 reads identical, 0 differ" in cp.stdout and "PASS" in cp.stdout, cp.stdout + cp.stderr
    hit = next(i for i in range(120) if "\t-\t" not in per_read[i])
    cols = per_read[hit].split("\n")[0].split("\t")
    cols[7] = str(int(cols[7]) + 1)                                             # QHASH_MATCH off by one
    bad = list(per_read)
    bad[hit] = "\t".join(cols) + "\n" + "\n".join(per_read[hit].split("\n")[1:])
    ref.write_text(HEADER + "".join(bad[i] for i in order if i != 5))            # and one read missing
    cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(fa), "--output-file", str(out), "--expect", str(ref)],
                        capture_output=True, text=True, timeout=300)
    assert cp.returncode == 3 and "118 reads identical, 1 differ, 1 only in this run" in cp.stdout and "FAIL" in cp.stdout, cp.stdout
    assert f"read {ids[hit]} differs" in cp.stdout


def test_cli_several_index_files(tmp_path):
    """--index-file a.hixf,b.hixf: every query file is searched against every index in turn and the results are appended
    to one output (search_hixf, taxor_search.cpp:344-358: for query: for index); indexes built with another k-mer
    selection scheme are refused (:118-141)"""
    g1, go1, host1, sp1, idx1 = _setup(tmp_path, 41)
    g2, go2, host2, sp2, idx2 = _setup(tmp_path, 42)
    qa, qb = tmp_path / "qa.fa", tmp_path / "qb.fa"
    sets = []
    for path, (g, go), seed in ((qa, (g1, go1), 13), (qb, (g2, go2), 14)):
        bases, offs, origin = synth.synth_reads(g, go, 70, 1300, error_rate=0.02, frac_random=0.2, seed=seed)
        reads = [bytes(bases[int(offs[i]):int(offs[i + 1])]) for i in range(70)]
        ids = [f"{path.stem}_{i}" for i in range(70)]
        with open(path, "wb") as f:
            for rid, r in zip(ids, reads):
                f.write(b">" + rid.encode() + b"\n" + r + b"\n")
        sets.append((ids, reads))
    out = tmp_path / "o.tsv"
    cp = subprocess.run([TAXOR, "search", "--index-file", f"{idx1},{idx2}", "--query-file", f"{qa},{qb}", "--output-file", str(out),
                         "--threads", "2", "--batch-reads", "32"], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stderr
    want = HEADER
    for ids, reads in sets:                              # query-major, index-minor
        want += _expected(host1, sp1, ids, reads) + _expected(host2, sp2, ids, reads)
    assert open(out).read() == want
    assert want.count("Organism") > 60                  # reads of qa hit index 1, reads of qb hit index 2
    # another (k, s, t): refused before anything is searched
    other = tmp_path / "k24.hixf"
    store_hixf(other, host2, max(s["user_bin"] for s in sp2) + 1, sp2, k=24, s=12, t=6)
    cp = subprocess.run([TAXOR, "search", "--index-file", f"{idx1},{other}", "--query-file", str(qa), "--output-file", str(out)],
                        capture_output=True, text=True, timeout=120)
    assert cp.returncode != 0 and "different kmer selection schemes" in cp.stderr



def test_cli_single_member_gzip_equals_plain(tmp_path):
    """reads.fastq.gz as `gzip` writes it -- ONE member, here 9 MB -- is inflated speculatively from the middle on the reader's
    threads (taxor_amd/csrc/pgz.h): the TSV must be the plain file's, byte for byte, and --sequential (one zlib stream) agrees"""
    g, go, host, sp, idx_path = _setup(tmp_path, 53)
    rng = np.random.default_rng(4)
    bases, offs, origin = synth.synth_reads(g, go, 2400, 4000, error_rate=0.02, frac_random=0.3, seed=6)
    q = np.frombuffer(b"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHI", np.uint8)
    fq = tmp_path / "reads.fastq"
    with open(fq, "wb") as f:
        for i in range(2400):
            r = bytes(bases[int(offs[i]):int(offs[i + 1])])
            f.write(b"@read_%d runid=abc ch=%d\n" % (i, i % 97) + r + b"\n+\n" + bytes(rng.choice(q, size=len(r))) + b"\n")
    gz = tmp_path / "reads.fastq.gz"
    gz.write_bytes(gzip.compress(fq.read_bytes(), 6))
    assert gz.stat().st_size > (8 << 20)
    outs = []
    for path, extra in ((fq, []), (gz, []), (gz, ["--sequential"]), (gz, ["--threads", "2"])):
        out = tmp_path / f"o{len(outs)}.tsv"
        cp = subprocess.run([TAXOR, "search", "--index-file", str(idx_path), "--query-file", str(path), "--output-file", str(out), "--threads", "8"] + extra,
                            capture_output=True, text=True, timeout=600)
        assert cp.returncode == 0, cp.stderr
        outs.append(open(out).read())
    assert outs[0].count("\n") > 2400 and all(o == outs[0] for o in outs[1:])
