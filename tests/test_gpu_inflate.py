"""The chunks of a gzip member decoded on the GPU (taxor_amd/csrc/inflate.hip: one wave per chunk, wave-uniform decoder, 16-bit
symbols with window markers in HBM; windows chained and symbols resolved by two more kernels) through `taxor inflate --gpu 0`,
which verifies every member's CRC-32 and length itself -- against Python's gzip module, byte for byte: every level, chunk sizes
from 32 KiB, stored-only and fixed-Huffman streams, several members, binary content, a batch smaller than the file (so that windows
are carried from batch to batch), an arena too small for the data (every chunk falls back to the host decoder); truncated and
corrupted streams end in an error or in the original bytes; `taxor search` on the .gz with the device decoding equals the plain file."""
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from tests.test_pgz_cpu import fastq

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "taxor_amd", "taxor")


def inflate(path, out, chunk_mb, batch=0, threads=4, ok=True):
    cmd = [EXE, "inflate", "--query-file", str(path), "--threads", str(threads), "--chunk-mb", str(chunk_mb), "--gpu", "0", "--output-file", str(out)]
    if batch:
        cmd += ["--batch-chunks", str(batch)]
    cp = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if ok:
        assert cp.returncode == 0, cp.stdout[-1500:] + cp.stderr[-1500:]
        assert "device decoding not available" not in cp.stderr, cp.stderr
    return cp


def on_device(cp):
    line = [l for l in cp.stdout.splitlines() if l.startswith("device:")][0]
    return int(line.split()[1]), int(line.split(",")[1].split()[0])


def test_levels_chunk_sizes_and_stream_kinds(tmp_path):
    rng = np.random.default_rng(5)
    raw = fastq(rng, 4000)                                   # ~25 MB
    out = tmp_path / "out.bin"
    for level in (1, 4, 6, 9):
        p = tmp_path / f"l{level}.fastq.gz"
        p.write_bytes(gzip.compress(raw, level))
        for chunk_mb, batch in ((0.03125, 0), (0.0625, 7), (1, 0), (4, 0)):
            cp = inflate(p, out, chunk_mb, batch)
            assert out.read_bytes() == raw, (level, chunk_mb)
            dev, host = on_device(cp)
            if chunk_mb < 1:
                assert dev > 5 * host and dev > 50, cp.stdout          # the device decodes nearly every chunk
    # stored blocks only: no dynamic header to find; the device is handed nothing but chunk 0 and the host decodes the rest
    p = tmp_path / "stored.gz"
    p.write_bytes(gzip.compress(raw[: 3 << 20], 0))
    inflate(p, out, 0.0625)
    assert out.read_bytes() == raw[: 3 << 20]
    # tiny inputs: fixed-Huffman blocks, an empty member, one byte
    for tiny in (b"", b"A", b"ACGT\n" * 3, raw[:700]):
        p = tmp_path / "tiny.gz"
        p.write_bytes(gzip.compress(tiny, 6))
        inflate(p, out, 0.03125)
        assert out.read_bytes() == tiny
    # several members (one with header fields, an empty one), zero padding behind the last
    hdr = b"\x1f\x8b\x08\x18" + b"\0\0\0\0" + b"\x00\x03" + b"reads.fastq\0" + b"a comment\0"
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(raw[: 5 << 20]) + co.flush()
    member = hdr + body + zlib.crc32(raw[: 5 << 20]).to_bytes(4, "little") + ((5 << 20) & 0xFFFFFFFF).to_bytes(4, "little")
    p = tmp_path / "multi.gz"
    p.write_bytes(member + gzip.compress(raw[5 << 20: 9 << 20], 9) + gzip.compress(b"", 6) + gzip.compress(raw[9 << 20:], 1) + b"\0" * 512)
    cp = inflate(p, out, 0.25, 5)
    assert out.read_bytes() == raw and "4 member(s)" in cp.stdout, cp.stdout
    # binary content, long runs (matches at distance 1: overlapping copies), text behind it
    blob = bytes(rng.integers(0, 256, size=3 << 20, dtype=np.uint8)) + bytes(1 << 20) + b"ab" * (1 << 19) + raw[: 2 << 20]
    p = tmp_path / "blob.gz"
    p.write_bytes(gzip.compress(blob, 6))
    inflate(p, out, 0.0625, 9)
    assert out.read_bytes() == blob
    # highly compressible text: far more output than the arena's share of a chunk -> those chunks come from the host decoder
    rep = (b"@r\nACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIII\n" * 400000)
    p = tmp_path / "rep.gz"
    p.write_bytes(gzip.compress(rep, 6))
    cp = inflate(p, out, 0.03125)
    assert out.read_bytes() == rep
    assert on_device(cp)[1] > 0, cp.stdout


def test_truncated_and_corrupted_streams_never_give_wrong_bytes(tmp_path):
    rng = np.random.default_rng(6)
    raw = fastq(rng, 1500)
    comp = gzip.compress(raw, 6)
    out = tmp_path / "out.bin"
    for cut in (20, len(comp) // 3, len(comp) // 2, len(comp) - 9, len(comp) - 1):
        p = tmp_path / "cut.gz"
        p.write_bytes(comp[:cut])
        cp = inflate(p, out, 0.0625, ok=False)
        assert cp.returncode != 0, (cut, cp.stdout)
    outcomes = {"error": 0, "same": 0}
    for t in range(24):
        b = bytearray(comp)
        pos = int(rng.integers(10, len(b) - 8))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        p = tmp_path / "bad.gz"
        p.write_bytes(bytes(b))
        cp = inflate(p, out, 0.0625, ok=False)
        if cp.returncode != 0:
            outcomes["error"] += 1
        else:
            assert out.read_bytes() == raw, pos
            outcomes["same"] += 1
    assert outcomes["error"] >= 20, outcomes


def test_c_abi_one_chunk_equals_zlib():
    """Straight through the C ABI: a raw deflate stream as ONE chunk from bit 0 -- symbols (no window markers: nothing lies before the
    chunk) and resolved bytes equal what zlib inflates; a start in the middle of nowhere is reported as invalid, not decoded."""
    import ctypes as C

    from taxor_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    raw = fastq(rng, 300)
    for level in (1, 6, 9):
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(raw) + co.flush()
        h = C.c_void_p()
        assert L.taxor_gpu_inflater_create(0, len(comp) + 64, 4, 4 * (32768 + 256) + 3 * len(raw), C.byref(h)) == 0, L.taxor_gpu_last_error()
        try:
            buf = (C.c_uint8 * len(comp)).from_buffer_copy(comp)
            req = (_lib.InflateChunk * 2)()
            req[0].start_bit, req[0].stop_bit, req[0].weight = 0, len(comp) * 8, len(comp) * 8
            req[1].start_bit, req[1].stop_bit, req[1].weight = 12345, len(comp) * 8, 8          # not a block start
            res = (_lib.InflateResult * 2)()
            assert L.taxor_gpu_inflate_decode(h, buf, len(comp), req, 2, res) == 0, L.taxor_gpu_last_error()
            assert res[0].status == 0 and res[0].final_block == 1 and res[0].n_out == len(raw), (res[0].status, res[0].n_out, len(raw))
            assert res[0].end_bit <= len(comp) * 8 and res[0].end_bit > len(comp) * 8 - 8
            assert res[1].status != 0
            sym = np.empty(len(raw), np.uint16)
            assert L.taxor_gpu_inflate_symbols(h, 0, sym.ctypes.data_as(C.c_void_p)) == 0
            assert sym.max() < 256 and bytes(sym.astype(np.uint8)) == raw
            out = np.empty(len(raw), np.uint8)
            outs = (C.c_void_p * 1)(out.ctypes.data)
            win_in = (C.c_uint8 * 32768)()
            win_out = (C.c_uint8 * 32768)()
            assert L.taxor_gpu_inflate_resolve(h, win_in, 0, 1, outs, win_out) == 0, L.taxor_gpu_last_error()
            assert bytes(out) == raw and bytes(win_out) == raw[-32768:]
        finally:
            L.taxor_gpu_inflater_destroy(h)
