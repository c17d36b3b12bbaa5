"""The CLI's FASTA/FASTQ readers (taxor_amd/csrc/fastx.h): the sequential zlib reader and the memory-mapped parallel
parser must produce the same records -- id = full header line, sequence = concatenated lines -- as a plain Python
restatement of what seqan3::sequence_file_input hands the reference (src/main/taxor_search.cpp:181-184,315-321).
`taxor reads` prints id, length and FNV-1a of every record; no GPU involved."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "taxor_amd", "taxor")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="taxor CLI not built")


def fnv1a(b: bytes) -> int:
    h = 1469598103934665603
    for c in b:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def run_reads(path, *extra):
    cp = subprocess.run([EXE, "reads", "--query-file", str(path), *extra], capture_output=True, text=True)
    assert cp.returncode == 0, cp.stderr
    rows = [l.split("\t") for l in cp.stdout.split("\n") if l]
    n_batches = int(cp.stderr.split(" batches")[0].split(", ")[-1])
    return [(r[0], int(r[1]), int(r[2], 16)) for r in rows], n_batches


def expected(records):
    return [(i, len(s), fnv1a(s)) for i, s in records]


def make_records(rng, n, lo=0, hi=400):
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L))
        recs.append((f"read_{i} some description/{i % 7}", s))
    return recs


def write_fastq(path, recs, rng, eol=b"\n", final_eol=True, opener=open):
    qchars = np.frombuffer(b"@+!I5>#", np.uint8)     # quality lines that start with '@', '+' and '>' are legal
    with opener(path, "wb") as f:
        for j, (i, s) in enumerate(recs):
            q = bytes(rng.choice(qchars, size=len(s)))
            last = j == len(recs) - 1
            f.write(b"@" + i.encode() + eol + s + eol + b"+" + (i.encode() if j % 3 == 0 else b"") + eol + q +
                    (eol if (final_eol or not last) else b""))


def write_fasta(path, recs, rng, eol=b"\n", width=60, blank_every=0, opener=open):
    with opener(path, "wb") as f:
        for j, (i, s) in enumerate(recs):
            f.write(b">" + i.encode() + eol)
            w = width if width else max(1, len(s))
            for a in range(0, len(s), w):
                f.write(s[a:a + w] + eol)
            if blank_every and j % blank_every == 0:
                f.write(eol)


MODES = [("--sequential",), ("--threads", "1"), ("--threads", "4"), ("--threads", "4", "--batch-reads", "7"),
         ("--threads", "3", "--batch-reads", "1"), ("--sequential", "--batch-reads", "5")]


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
def test_fastq_all_paths_agree(tmp_path, eol):
    rng = np.random.default_rng(1)
    recs = make_records(rng, 300)
    p = tmp_path / "r.fastq"
    write_fastq(p, recs, rng, eol=eol)
    want = expected(recs)
    for m in MODES:
        got, nb = run_reads(p, *m)
        assert got == want, m
        if "--batch-reads" in m and "--sequential" not in m:
            assert nb > 10          # ranges really were cut inside the file, at record starts


def test_fastq_without_final_newline_and_gz(tmp_path):
    rng = np.random.default_rng(2)
    recs = make_records(rng, 50, lo=1)
    p = tmp_path / "r.fastq"
    write_fastq(p, recs, rng, final_eol=False)
    for m in MODES:
        assert run_reads(p, *m)[0] == expected(recs)
    pz = tmp_path / "r.fastq.gz"
    write_fastq(pz, recs, np.random.default_rng(2), opener=gzip.open)
    assert run_reads(pz, "--threads", "4")[0] == expected(recs)      # gzip: falls back to the sequential reader


@pytest.mark.parametrize("eol,width,blank", [(b"\n", 60, 0), (b"\r\n", 13, 0), (b"\n", 0, 0), (b"\n", 50, 4)])
def test_fasta_all_paths_agree(tmp_path, eol, width, blank):
    rng = np.random.default_rng(3)
    recs = make_records(rng, 200, lo=0, hi=500)
    p = tmp_path / "r.fa"
    write_fasta(p, recs, rng, eol=eol, width=width, blank_every=blank)
    want = expected(recs)
    for m in MODES:
        assert run_reads(p, *m)[0] == want, m


def test_long_records_span_many_ranges(tmp_path):
    """records much longer than a byte range: a range then holds exactly one record"""
    rng = np.random.default_rng(4)
    recs = [(f"long_{i}", bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(L))))
            for i, L in enumerate([300000, 5, 170000, 1, 0, 250000])]
    p = tmp_path / "l.fastq"
    write_fastq(p, recs, rng)
    for m in MODES:
        assert run_reads(p, *m)[0] == expected(recs), m
    p2 = tmp_path / "l.fa"
    write_fasta(p2, recs, rng)
    for m in MODES:
        assert run_reads(p2, *m)[0] == expected(recs), m


@pytest.mark.parametrize("text,msg", [
    (b"@r1\nACGT\n+\nIIII\n@r2\nACGT\n", "FASTQ record: r2"),
    (b"@r1\nACGT\n+\nIIII\n@r2\nACGT\n+\n", "truncated FASTQ record: r2"),
    (b"@r1\nACGT\nIIII\n@r2\nAC\n+\nII\n", "truncated FASTQ record: r1"),   # no '+' line: the sequence runs on
    (b"hello\nworld\n", "neither FASTA nor FASTQ"),
])
def test_damaged_input_fails_loudly(tmp_path, text, msg):
    p = tmp_path / "bad.fq"
    p.write_bytes(text)
    for m in (("--sequential",), ("--threads", "2")):
        cp = subprocess.run([EXE, "reads", "--query-file", str(p), *m], capture_output=True, text=True)
        assert cp.returncode != 0 and msg in cp.stderr, (m, cp.stderr)


def test_multi_line_fastq_like_seqan3(tmp_path):
    """FASTQ whose sequence and quality strings wrap over several lines (seqan3's format_fastq reads the sequence up to
    the '+' line and then as many quality characters as the sequence has): not cut into byte ranges -- the four-line
    resync rule does not hold -- but read by the sequential reader, whatever --threads says."""
    rng = np.random.default_rng(11)
    recs = make_records(rng, 120, lo=1, hi=900)
    qchars = np.frombuffer(b"@+!I5>#", np.uint8)
    p = tmp_path / "wrapped.fastq"
    with open(p, "wb") as f:
        for j, (i, s) in enumerate(recs):
            q = bytes(rng.choice(qchars, size=len(s)))
            w = int(rng.integers(20, 80))
            f.write(b"@" + i.encode() + b"\n")
            for a in range(0, len(s), w):
                f.write(s[a:a + w] + b"\n")
            f.write(b"+\n")
            for a in range(0, len(q), w):
                f.write(q[a:a + w] + b"\n")
    want = expected(recs)
    for m in MODES:
        assert run_reads(p, *m)[0] == want, m


def test_multi_member_gzip_is_inflated_in_parallel_and_in_order(tmp_path):
    """bgzip-like / concatenated gzip files: members are found speculatively and inflated by several threads; the
    records must come out exactly as from a sequential read.  Includes members of very different sizes, an empty
    member, a header whose FNAME field contains a fake member magic, and trailing zero padding."""
    import io
    import zlib
    rng = np.random.default_rng(7)
    recs = make_records(rng, 900, lo=0, hi=700)
    text = io.BytesIO()
    for i, s in recs:
        text.write(b"@" + i.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    raw = text.getvalue()
    cuts = sorted(set([0, 10, 11, 5000, 5000, 40000, 90000, 90001, len(raw) // 2, len(raw) - 7, len(raw)] +
                      [int(x) for x in rng.integers(0, len(raw), 25)]))
    blob = b""
    for a, b in zip(cuts[:-1], cuts[1:]):
        blob += gzip.compress(raw[a:b], compresslevel=int(rng.integers(1, 9)))
    blob += gzip.compress(b"")                                            # an empty member is legal
    # a member whose header carries a file name that looks like another member header
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    tail_text = b"@last_read extra\nACGTACGTAC\n+\nIIIIIIIIII\n"
    body = co.compress(tail_text) + co.flush()
    hdr = b"\x1f\x8b\x08\x08" + b"\x00" * 4 + b"\x00\x03" + b"\x1f\x8b\x08\x01fakefakefake" + b"\x00"
    blob += hdr + body + (zlib.crc32(tail_text) & 0xFFFFFFFF).to_bytes(4, "little") + (len(tail_text) & 0xFFFFFFFF).to_bytes(4, "little")
    blob += b"\x00" * 37                                                  # zero padding after the last member
    p = tmp_path / "multi.fastq.gz"
    p.write_bytes(blob)
    assert gzip.decompress(blob[:-37]) == raw + tail_text      # python agrees that this is a valid file
    want = expected(recs + [("last_read extra", b"ACGTACGTAC")])
    for m in (("--threads", "1"), ("--threads", "4"), ("--threads", "8", "--batch-reads", "50"), ("--sequential",)):
        assert run_reads(p, *m)[0] == want, m
    # corrupt member in the middle: loud failure, not silence
    bad = bytearray(blob)
    bad[len(blob) // 2] ^= 0x55
    q = tmp_path / "bad.fastq.gz"
    q.write_bytes(bytes(bad))
    cp = subprocess.run([EXE, "reads", "--query-file", str(q), "--threads", "4"], capture_output=True, text=True)
    assert cp.returncode != 0 or cp.stdout != "".join(f"{i}\t{n}\t{h:016x}\n" for i, n, h in want)


def test_bgzip_like_members_through_the_cutter_and_the_parser_pool(tmp_path):
    """Members of 64 KB and less (bgzip), boundaries anywhere -- inside lines, inside reads longer than a member -- quality lines that
    start with '@' and '+': the members are the chunks the cutter collects into parser jobs (search_main.cpp), small jobs here so that
    many cuts fall into parts that begin in the middle of a line."""
    rng = np.random.default_rng(21)
    recs = make_records(rng, 260, lo=1, hi=3000) + make_records(rng, 6, lo=90000, hi=150000) + make_records(rng, 200, lo=0, hi=9000)
    recs = [(f"{i} n={j}", s) for j, (i, s) in enumerate(recs)]
    plain = tmp_path / "reads.fastq"
    write_fastq(plain, recs, rng)
    raw = plain.read_bytes()
    blob, pos = b"", 0
    while pos < len(raw):
        n = int(rng.choice([65280, 65280, 4096, 1, 30000]))
        blob += gzip.compress(raw[pos:pos + n], compresslevel=int(rng.integers(1, 7)))
        pos += n
    p = tmp_path / "reads.bgzf.fastq.gz"
    p.write_bytes(blob)
    want = expected(recs)
    for m in (("--threads", "8", "--batch-bases", "300000"), ("--threads", "3", "--batch-bases", "70000"), ("--threads", "4"), ("--sequential",)):
        got, n_batches = run_reads(p, *m)
        assert got == want, m
        if "--batch-bases" in m:
            assert n_batches > 5, (m, n_batches)


def test_single_member_gzip_with_false_member_headers(tmp_path):
    """a one-member .gz whose (stored) data contains byte patterns that look like member headers: the speculative
    member search must not be fooled into a wrong split"""
    rng = np.random.default_rng(13)
    recs = [(f"r{i} \x1f\x8b\x08\x01 tag", bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=300))) for i in range(40)]
    raw = b"".join(b"@" + i.encode("latin1") + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in recs)
    p = tmp_path / "stored.fastq.gz"
    p.write_bytes(gzip.compress(raw, compresslevel=0))
    cp = subprocess.run([EXE, "reads", "--query-file", str(p), "--threads", "4"], capture_output=True)
    assert cp.returncode == 0, cp.stderr
    rows = [l.split(b"\t") for l in cp.stdout.split(b"\n") if l]
    assert [(r[0], int(r[1]), int(r[2], 16)) for r in rows] == [(i.encode("latin1"), len(s), fnv1a(s)) for i, s in recs]


def test_bzip2_input(tmp_path):
    """seqan3's sequence_file_input reads .bz2 (the reference's CMake fetches bzip2): same records as the plain file"""
    import bz2
    rng = np.random.default_rng(17)
    recs = make_records(rng, 150, lo=0, hi=900)
    plain = tmp_path / "r.fastq"
    write_fastq(plain, recs, rng)
    pz = tmp_path / "r.fastq.bz2"
    pz.write_bytes(bz2.compress(plain.read_bytes()))
    for m in (("--threads", "1"), ("--threads", "4", "--batch-reads", "20"), ("--sequential",)):
        assert run_reads(pz, *m)[0] == expected(recs), m
    fa = tmp_path / "r.fa"
    write_fasta(fa, recs, rng, width=70)
    fz = tmp_path / "r.fa.bz2"
    fz.write_bytes(bz2.compress(fa.read_bytes()) + bz2.compress(b">second_stream\nACGTACGT\n"))      # two concatenated streams
    got = run_reads(fz, "--threads", "2")[0]
    assert got[:len(recs)] == expected(recs)
    # ... and the second stream is read too (pbzip2 output, `cat a.bz2 b.bz2`): nothing is dropped silently
    assert got == expected(recs) + expected([("second_stream", b"ACGTACGT")])
    many = tmp_path / "many.fastq.bz2"
    cut = [0, 40, 41, 97, len(recs)]
    parts = []
    for a, b in zip(cut, cut[1:]):
        q = tmp_path / "part.fastq"
        write_fastq(q, recs[a:b], rng)
        parts.append(bz2.compress(q.read_bytes()))
    many.write_bytes(b"".join(parts))
    assert run_reads(many, "--threads", "3")[0] == expected(recs)
    broken = tmp_path / "broken.fastq.bz2"
    broken.write_bytes(parts[0] + parts[1][: len(parts[1]) // 2])
    cp = subprocess.run([EXE, "reads", "--query-file", str(broken)], capture_output=True)
    assert cp.returncode != 0 and b"bzip2" in cp.stderr


def test_fastq_that_starts_four_line_and_wraps_later_is_not_cut_wrongly(tmp_path):
    """the range cutter is chosen from the file's first records.  A FASTQ whose later records wrap their lines must
    not be mis-parsed in parallel mode: the range parser notices (a record's sequence spans several lines, or quality and
    sequence lengths differ) and stops with a message that names --sequential, under which the file parses correctly."""
    rng = np.random.default_rng(23)
    recs = make_records(rng, 400, lo=150, hi=400)
    p = tmp_path / "mixed.fastq"
    with open(p, "wb") as f:
        for j, (i, s) in enumerate(recs):
            q = bytes(rng.choice(np.frombuffer(b"I5>#", np.uint8), size=len(s)))
            if j < 120:
                f.write(b"@" + i.encode() + b"\n" + s + b"\n+\n" + q + b"\n")
            else:                                   # wrapped at 80 columns, sequence and quality alike
                f.write(b"@" + i.encode() + b"\n" + b"\n".join(s[k:k + 80] for k in range(0, len(s), 80)) + b"\n+\n" +
                        b"\n".join(q[k:k + 80] for k in range(0, len(q), 80)) + b"\n")
    cp = subprocess.run([EXE, "reads", "--query-file", str(p), "--threads", "4", "--batch-reads", "40"], capture_output=True, text=True)
    assert cp.returncode != 0 and "--sequential" in cp.stderr, cp.stderr
    assert run_reads(p, "--sequential")[0] == expected(recs)


def test_malformed_fastq_fails_loudly(tmp_path):
    rng = np.random.default_rng(29)
    recs = make_records(rng, 30, lo=50, hi=90)
    good = tmp_path / "good.fastq"
    write_fastq(good, recs, rng)
    lines = good.read_bytes().split(b"\n")
    short = list(lines)
    short[4 * 7 + 3] = short[4 * 7 + 3][:-5]                  # record 7: quality five characters short
    bad1 = tmp_path / "short_quality.fastq"
    bad1.write_bytes(b"\n".join(short))
    longq = list(lines)
    longq[4 * 9 + 3] = longq[4 * 9 + 3] + b"III"              # record 9: quality three characters too long
    bad2 = tmp_path / "long_quality.fastq"
    bad2.write_bytes(b"\n".join(longq))
    for bad in (bad1, bad2):
        for mode in ((), ("--sequential",)):
            cp = subprocess.run([EXE, "reads", "--query-file", str(bad), *mode], capture_output=True, text=True)
            assert cp.returncode != 0 and ("malformed FASTQ" in cp.stderr or "truncated FASTQ" in cp.stderr), (bad, mode, cp.stderr)
