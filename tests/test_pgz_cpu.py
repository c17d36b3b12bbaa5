"""The parallel inflate of ONE gzip member (taxor_amd/csrc/pgz.h: speculative decoding from the middle of a deflate stream on a
decoder of its own, window markers, chunk starts tied to their predecessors' ends, CRC-32 / length of the member verified at the
end) against Python's gzip module: every level, stored-only and fixed-Huffman streams, several members, header fields, text and
binary content, chunk sizes from 32 KiB up; truncated and corrupted streams must end in an error or in the exact original bytes,
never in wrong ones; under AddressSanitizer + UBSan and under ThreadSanitizer; and through the CLI's reader (`taxor reads`)."""
import gzip
import os
import shutil
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "taxor_amd", "csrc")
SAN = os.path.join(ROOT, "tests", "sanitize")
EXE = os.path.join(ROOT, "taxor_amd", "taxor")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")


def _build(tmp_path_factory, name, flags):
    exe = tmp_path_factory.mktemp("pgz") / name
    cp = subprocess.run(["g++", "-std=c++17", "-O2", "-g", "-fno-omit-frame-pointer", *flags, f"-I{CSRC}", os.path.join(SAN, "pgz_inflate.cpp"),
                         "-o", str(exe), "-lz", "-pthread"], capture_output=True, text=True)
    if cp.returncode != 0 and " error: " not in cp.stderr and any(t in cp.stderr for t in ("cannot find -lasan", "cannot find -lubsan", "cannot find -ltsan", "libasan", "libtsan")):
        pytest.skip("sanitizer runtime not available: " + cp.stderr[:200])          # (a compile error is a failure, not a skip)
    assert cp.returncode == 0, cp.stderr
    return exe


@pytest.fixture(scope="module")
def exe_plain(tmp_path_factory):
    return _build(tmp_path_factory, "pgz_plain", [])


@pytest.fixture(scope="module")
def exe_asan(tmp_path_factory):
    return _build(tmp_path_factory, "pgz_asan", ["-fsanitize=address,undefined"])


@pytest.fixture(scope="module")
def exe_tsan(tmp_path_factory):
    return _build(tmp_path_factory, "pgz_tsan", ["-fsanitize=thread"])


def fastq(rng, n_reads, lo=200, hi=6000, genome=None):
    out = []
    if genome is None:
        genome = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=200000))
    q = np.frombuffer(b"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHI", np.uint8)
    for i in range(n_reads):
        L = int(rng.integers(lo, hi))
        a = int(rng.integers(0, len(genome) - L))
        out.append(b"@read_%d ch=%d start_time=2024-01-01T00:00:00Z\n" % (i, i % 512) + genome[a:a + L] + b"\n+\n" + bytes(rng.choice(q, size=L)) + b"\n")
    return b"".join(out)


def run(exe, path, threads=4, chunk=65536, out=None, budget_mb=None):
    cp = subprocess.run([str(exe), str(path), str(threads), str(chunk)] + ([str(out)] if out else ["-"] if budget_mb else []) + ([str(budget_mb)] if budget_mb else []),
                        capture_output=True, text=True, timeout=900)
    assert cp.returncode == 0, cp.stderr[-3000:]
    assert "ERROR: " not in cp.stderr and "runtime error" not in cp.stderr and "WARNING: ThreadSanitizer" not in cp.stderr, cp.stderr[-3000:]
    return cp.stdout.strip()


def want(raw):
    return f"bytes {len(raw)} crc {zlib.crc32(raw):08x}"


def test_levels_chunk_sizes_and_stream_kinds(tmp_path, exe_plain):
    rng = np.random.default_rng(5)
    raw = fastq(rng, 4000)                                   # ~25 MB
    for level in (1, 4, 6, 9):
        p = tmp_path / f"l{level}.fastq.gz"
        p.write_bytes(gzip.compress(raw, level))
        for chunk in (32768, 65536, 1 << 20, 4 << 20):
            got = run(exe_plain, p, threads=4, chunk=chunk)
            assert got.startswith(want(raw)), (level, chunk, got)
        if level == 6:
            got = run(exe_plain, p, threads=4, chunk=65536, out=tmp_path / "out.bin")
            assert (tmp_path / "out.bin").read_bytes() == raw
            n_chunks, redone = int(got.split("chunks ")[1].split()[0]), int(got.split("redecoded ")[1].split()[0])
            assert n_chunks > 50 and redone <= n_chunks // 10, got          # the speculative starts are almost always the right ones
    # stored blocks only (level 0): no dynamic header to find, every chunk is decoded from its predecessor's end
    p = tmp_path / "stored.gz"
    p.write_bytes(gzip.compress(raw[: 3 << 20], 0))
    assert run(exe_plain, p, chunk=65536).startswith(want(raw[: 3 << 20]))
    # tiny inputs: fixed-Huffman blocks, an empty member, one byte
    for tiny in (b"", b"A", b"ACGT\n" * 3, raw[:700]):
        p = tmp_path / "tiny.gz"
        p.write_bytes(gzip.compress(tiny, 6))
        assert run(exe_plain, p, chunk=32768).startswith(want(tiny)), tiny[:20]
    # a name and a comment in the header (what the gzip tool writes), several members, zero padding behind the last
    hdr = b"\x1f\x8b\x08\x18" + b"\0\0\0\0" + b"\x00\x03" + b"reads.fastq\0" + b"a comment\0"
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(raw[: 5 << 20]) + co.flush()
    member = hdr + body + zlib.crc32(raw[: 5 << 20]).to_bytes(4, "little") + ((5 << 20) & 0xFFFFFFFF).to_bytes(4, "little")
    p = tmp_path / "multi.gz"
    p.write_bytes(member + gzip.compress(raw[5 << 20: 9 << 20], 9) + gzip.compress(b"", 6) + gzip.compress(raw[9 << 20:], 1) + b"\0" * 512)
    got = run(exe_plain, p, chunk=1 << 20)
    assert got.startswith(want(raw)) and "members 4" in got, got
    # binary content (no text to validate candidate block starts with): still exact
    blob = bytes(rng.integers(0, 256, size=3 << 20, dtype=np.uint8)) + bytes(1 << 20) + raw[: 2 << 20]
    p = tmp_path / "blob.gz"
    p.write_bytes(gzip.compress(blob, 6))
    assert run(exe_plain, p, chunk=65536).startswith(want(blob))


def test_truncated_and_corrupted_streams_never_give_wrong_bytes(tmp_path, exe_plain):
    rng = np.random.default_rng(6)
    raw = fastq(rng, 1500)
    comp = gzip.compress(raw, 6)
    ok = want(raw)
    for cut in (20, len(comp) // 3, len(comp) // 2, len(comp) - 9, len(comp) - 1):
        p = tmp_path / "cut.gz"
        p.write_bytes(comp[:cut])
        got = run(exe_plain, p, chunk=65536)
        assert got.startswith("error:"), (cut, got)
    outcomes = {"error": 0, "same": 0}
    for k in range(40):
        pos = int(rng.integers(12, len(comp) - 8))
        bad = bytearray(comp)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        p = tmp_path / "bad.gz"
        p.write_bytes(bytes(bad))
        got = run(exe_plain, p, chunk=65536)
        if got.startswith("error:"):
            outcomes["error"] += 1
        else:
            assert got.startswith(ok), (pos, got)        # a flipped bit that changes nothing that is checked cannot exist: CRC-32 covers all
            outcomes["same"] += 1
    assert outcomes["error"] >= 39
    for flip_trailer in (len(comp) - 8, len(comp) - 3):                 # CRC field, length field
        bad = bytearray(comp)
        bad[flip_trailer] ^= 0x10
        p = tmp_path / "badtrailer.gz"
        p.write_bytes(bytes(bad))
        assert "mismatch" in run(exe_plain, p, chunk=65536)


def _bits_to_bytes(bits):
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i >> 3] |= b << (i & 7)
    return bytes(out)


def test_reader_behaves_like_the_zlib_stream_reader_at_the_edges(tmp_path, exe_plain):
    """bytes behind the last member that are no gzip header: gzip(1) ignores them with a warning and keeps what it decoded -- so does
    this reader, after the members' CRCs have been checked; a match that reaches before the START of the stream is invalid deflate
    data ("distance too far back" in zlib), not a zero window found out by the CRC at the end of the member"""
    rng = np.random.default_rng(9)
    raw = fastq(rng, 1200)
    comp = gzip.compress(raw, 6)
    for tail in (b"garbage behind the member", b"\x1f", bytes(rng.integers(1, 256, size=3000, dtype=np.uint8))):
        p = tmp_path / "tail.gz"
        p.write_bytes(comp + b"\0" * 7 + tail)
        cp = subprocess.run([str(exe_plain), str(p), "4", "65536"], capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0 and cp.stdout.startswith(want(raw)) and f"trailing {len(tail)} " in cp.stdout, cp.stdout + cp.stderr
        assert "trailing garbage behind the last member ignored" in cp.stderr
    # fixed-Huffman block: BFINAL=1, BTYPE=01, length code 257 (len 3), distance code 0 (dist 1) as the very first symbol, end of block
    bits = [1, 1, 0] + [0, 0, 0, 0, 0, 0, 1] + [0, 0, 0, 0, 0] + [0, 0, 0, 0, 0, 0, 0]
    body = _bits_to_bytes(bits)
    with pytest.raises(zlib.error):
        zlib.decompress(body, -15)
    p = tmp_path / "far.gz"
    p.write_bytes(b"\x1f\x8b\x08\x00" + b"\0" * 6 + body + zlib.crc32(b"\0\0\0").to_bytes(4, "little") + (3).to_bytes(4, "little"))
    got = run(exe_plain, p, chunk=32768)
    assert got.startswith("error:") and "invalid deflate data" in got, got


@pytest.mark.parametrize("which", ["plain", "asan"])
def test_memory_in_flight_follows_the_compression_ratio(tmp_path, which, request):
    """a very repetitive (valid) member expands ~180:1 (deflate allows 1032:1): what the reader HOLDS -- symbol buffers (two bytes per
    output byte until resolved), resolved output not yet handed out, both pools; its own meter, pgz_detail::MemMeter -- stays within
    the budget plus one chunk (the chunk the consumer waits for is always decoded), instead of 2 x threads + 2 chunks whatever they
    expand to.  The process's RSS is the allocator's business (sixteen threads' arenas keep what they freed) and only printed."""
    exe_plain = request.getfixturevalue("exe_" + which)
    line = b"@r\n" + b"ACGT" * 64 + b"\n+\n" + b"I" * 256 + b"\n"
    raw = line * (400_000_000 // len(line))
    comp = gzip.compress(raw, 6)
    assert len(raw) / len(comp) > 100
    p = tmp_path / "rep.gz"
    p.write_bytes(comp)
    del comp
    chunk = 32768                                     # ~70 chunks of ~6 MB of output (12 MB of symbols) each

    def fields(out):
        w = out.split()
        return {k: int(w[w.index(k) + 1]) for k in ("maxrss_kb", "inflight_peak_kb", "largest_chunk_kb")}

    budget_mb = 64
    bounded = run(exe_plain, p, threads=16, chunk=chunk, budget_mb=budget_mb)
    assert bounded.startswith(want(raw)), bounded[:200]
    b = fields(bounded)
    free = run(exe_plain, p, threads=16, chunk=chunk, budget_mb=65536)
    assert free.startswith(want(raw))
    f = fields(free)
    print(f"budget {budget_mb} MB: held at most {b['inflight_peak_kb'] >> 10} MB (largest chunk {b['largest_chunk_kb'] >> 10} MB), RSS {b['maxrss_kb'] >> 10} MB; "
          f"no budget: held {f['inflight_peak_kb'] >> 10} MB, RSS {f['maxrss_kb'] >> 10} MB")
    assert b["inflight_peak_kb"] <= budget_mb * 1024 + b["largest_chunk_kb"], (b, f)
    assert b["inflight_peak_kb"] < 0.5 * f["inflight_peak_kb"], (b, f)


@pytest.mark.parametrize("which", ["asan", "tsan"])
def test_under_sanitizers(tmp_path, which, request):
    exe = request.getfixturevalue("exe_" + which)
    rng = np.random.default_rng(7)
    raw = fastq(rng, 700 if which == "tsan" else 1500)
    comp = gzip.compress(raw, 6)
    p = tmp_path / "a.fastq.gz"
    p.write_bytes(comp)
    assert run(exe, p, threads=6, chunk=65536).startswith(want(raw))
    p2 = tmp_path / "two.gz"
    p2.write_bytes(comp + gzip.compress(raw[:100000], 0))
    assert run(exe, p2, threads=3, chunk=131072).startswith(want(raw + raw[:100000]))
    if which == "asan":
        for cut in (len(comp) // 2, len(comp) - 5):
            pc = tmp_path / "cut.gz"
            pc.write_bytes(comp[:cut])
            assert run(exe, pc, threads=4, chunk=65536).startswith("error:")
        for k in range(12):
            bad = bytearray(comp)
            bad[int(rng.integers(12, len(comp) - 8))] ^= 0xFF
            pb = tmp_path / "bad.gz"
            pb.write_bytes(bytes(bad))
            got = run(exe, pb, threads=4, chunk=65536)
            assert got.startswith("error:") or got.startswith(want(raw))


@pytest.mark.skipif(not os.path.exists(EXE), reason="taxor CLI not built")
def test_cli_reader_takes_single_member_gzip_through_the_parallel_inflate(tmp_path):
    """`taxor reads` on reads.fastq.gz (one member, > 8 MB) = on the plain file, record for record; --sequential keeps zlib"""
    rng = np.random.default_rng(8)
    raw = fastq(rng, 3000, genome=bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=3000000)))
    plain = tmp_path / "r.fastq"
    plain.write_bytes(raw)
    gz = tmp_path / "r.fastq.gz"
    gz.write_bytes(gzip.compress(raw, 6))
    assert gz.stat().st_size > (8 << 20)

    def reads(path, *extra):
        cp = subprocess.run([EXE, "reads", "--query-file", str(path), "--threads", "4", *extra], capture_output=True, text=True, timeout=600)
        assert cp.returncode == 0, cp.stderr
        return cp.stdout

    ref = reads(plain, "--sequential")
    assert ref.count("\n") == 3000
    assert reads(gz) == ref and reads(gz, "--sequential") == ref
    cp = subprocess.run([EXE, "inflate", "--query-file", str(gz), "--threads", "4", "--output-file", str(tmp_path / "o.fastq")], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0 and "verified" in cp.stdout, cp.stdout + cp.stderr
    assert (tmp_path / "o.fastq").read_bytes() == raw


def test_crc32_by_carryless_multiplication_equals_zlib(tmp_path):
    """pgz.h's crc32_bytes (PCLMULQDQ folding where the host has it, zlib otherwise) against zlib's crc32: 4000 random
    (length, offset, start value) cases, a buffer in random pieces, plain and under ASan + UBSan."""
    for name, flags in (("plain", []), ("asan", ["-fsanitize=address,undefined"])):
        exe = tmp_path / f"crc_{name}"
        cp = subprocess.run(["g++", "-std=c++17", "-O2", "-g", *flags, f"-I{CSRC}", os.path.join(SAN, "crc32_fold.cpp"), "-o", str(exe), "-lz", "-pthread"],
                            capture_output=True, text=True)
        if cp.returncode != 0 and " error: " not in cp.stderr and ("libasan" in cp.stderr or "cannot find -lasan" in cp.stderr):
            continue
        assert cp.returncode == 0, cp.stderr
        run_ = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
        assert run_.returncode == 0 and "bad 0" in run_.stdout and "runtime error" not in run_.stderr, run_.stdout + run_.stderr
