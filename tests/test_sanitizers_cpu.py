"""Host code under AddressSanitizer + UBSan (and the threaded gzip reader under ThreadSanitizer): the .hixf loader /
probe / formatter against mutated files, the multi-member gzip reader against a file of 60 members.  The GPU side has
no sanitizer on this platform; its check is bit-equality with the CPU oracle."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.test_hixf_file_cpu import make_species, small_layout
from taxor_amd.hixf_file import store_hixf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "taxor_amd", "csrc")
SAN = os.path.join(ROOT, "tests", "sanitize")

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")


def _build(tmp_path, name, sources, flags):
    exe = tmp_path / name
    cp = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags, f"-I{ROOT}/include", f"-I{CSRC}",
                         *sources, "-o", str(exe), "-lz", "-pthread"], capture_output=True, text=True)
    if cp.returncode != 0 and " error: " not in cp.stderr and any(t in cp.stderr for t in ("cannot find -lasan", "cannot find -lubsan", "cannot find -ltsan", "libasan", "libtsan")):
        pytest.skip("sanitizer runtime not available: " + cp.stderr[:200])          # (a compile error is a failure, not a skip)
    assert cp.returncode == 0, cp.stderr
    return exe


def test_hixf_loader_probe_formatter_under_asan_ubsan(tmp_path):
    exe = _build(tmp_path, "loader_fuzz", [os.path.join(SAN, "hixf_loader_fuzz.cpp"), os.path.join(CSRC, "hixf_io.cpp"),
                                           os.path.join(CSRC, "host_util.cpp")], ["-fsanitize=address,undefined"])
    lay, host, _ = small_layout(12)
    sp = make_species(lay)[:20]
    p = tmp_path / "base.hixf"
    store_hixf(p, host, lay["n_user_bins"], sp)
    cp = subprocess.run([str(exe), str(p), "0", "0", "400", str(tmp_path / "mut.hixf")], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0 and "ok=" in cp.stdout, cp.stdout + cp.stderr[-2000:]
    assert "ERROR" not in cp.stderr and "runtime error" not in cp.stderr, cp.stderr[-2000:]


@pytest.mark.parametrize("flag", ["-fsanitize=address,undefined", "-fsanitize=thread"])
def test_multi_member_gzip_reader_under_sanitizers(tmp_path, flag):
    exe = _build(tmp_path, "gz_read", [os.path.join(SAN, "gz_members_read.cpp")], [flag])
    rng = np.random.default_rng(1)
    raw = b"".join(b"@r%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=300)) + b"\n+\n" + b"I" * 300 + b"\n"
                   for i in range(6000))
    cuts = sorted(set([0, len(raw)] + [int(x) for x in rng.integers(0, len(raw), 60)]))
    p = tmp_path / "m.fastq.gz"
    p.write_bytes(b"".join(gzip.compress(raw[a:b], 1) for a, b in zip(cuts[:-1], cuts[1:])))
    cp = subprocess.run([str(exe), str(p)], capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0 and "open=1" in cp.stdout and "6000 records 1800000 bases" in cp.stdout, cp.stdout + cp.stderr[-2000:]
    assert "ERROR" not in cp.stderr and "WARNING: ThreadSanitizer" not in cp.stderr, cp.stderr[-2000:]


def test_fastx_readers_under_asan_ubsan(tmp_path):
    """the sequential and the ranged readers over plain / bzip2 (two streams) input, and over damaged FASTQ: clean reports, no
    sanitizer finding"""
    import bz2
    exe = _build(tmp_path, "fastx_read", [os.path.join(SAN, "fastx_read.cpp")], ["-fsanitize=address,undefined", "-ldl"])
    rng = np.random.default_rng(3)
    recs = [(b"r%d x" % i, bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=int(rng.integers(0, 700))))) for i in range(400)]
    raw = b"".join(b"@" + i + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n" for i, s in recs)
    total = sum(len(s) for _, s in recs)
    plain = tmp_path / "a.fastq"
    plain.write_bytes(raw)
    half = raw.index(b"@r200 x")
    bz = tmp_path / "a.fastq.bz2"
    bz.write_bytes(bz2.compress(raw[:half]) + bz2.compress(raw[half:]))
    for args in ([str(plain)], [str(plain), "ranged"], [str(bz)]):
        cp = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300)
        assert cp.returncode == 0 and f"400 records {total} bases" in cp.stdout, cp.stdout + cp.stderr[-1500:]
        assert "ERROR" not in cp.stderr and "runtime error" not in cp.stderr, cp.stderr[-1500:]
    for k, cut in enumerate((len(raw) // 3, len(raw) // 2 + 7, len(raw) - 3)):
        bad = tmp_path / f"bad{k}.fastq"
        damaged = bytearray(raw[:cut] + raw[cut + 5:])          # five bytes missing somewhere
        bad.write_bytes(bytes(damaged))
        for args in ([str(bad)], [str(bad), "ranged"]):
            cp = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300)
            assert cp.returncode == 0 and ("error:" in cp.stdout or "records" in cp.stdout), cp.stdout + cp.stderr[-1500:]
            assert "ERROR" not in cp.stderr and "runtime error" not in cp.stderr, cp.stderr[-1500:]
