"""Argument syntax of `taxor search` that is decided before any GPU work: the reference's seqan3 parser
(src/main/taxor_search.cpp:32-80) takes `--opt value` and `--opt=value`, carries two hidden no-op flags and a
version string, and rejects unknown options / out-of-range values with `[TAXOR SEARCH ERROR] ...`, exit -1."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAXOR = os.path.join(ROOT, "taxor_amd", "taxor")


def run(*args):
    return subprocess.run([TAXOR, "search", *args], capture_output=True, text=True, timeout=60)


def test_version_and_help():
    cp = run("--version")
    assert cp.returncode == 0 and "0.2.0" in cp.stdout
    for h in ("--help", "-h", "--advanced-help"):
        cp = run(h)
        assert cp.returncode == 0 and "--index-file" in cp.stderr


def test_equals_syntax_reaches_the_same_validators(tmp_path):
    fa = tmp_path / "r.fa"
    fa.write_text(">a\nACGT\n")
    missing = tmp_path / "nope.hixf"
    for args, needle in (
            ([f"--index-file={missing}", f"--query-file={fa}"], "does not exist"),
            ([f"--index-file={missing}", "--threads=64"], "Value not in range [1,32]"),
            ([f"--index-file={missing}", "--threads=4x"], "could not be parsed"),
            ([f"--index-file={missing}", "--error-rate=1.5"], "Value not in range [0,1]"),
            ([f"--index-file={missing}", "--percentage=-0.1"], "Value not in range [0,1]"),
            ([f"--index-file={missing}", "--threads"], "Missing value"),
            ([f"--query-file={fa}"], "required"),
            ([f"--index-file={missing}", "--ixf-layout=sideways"], "Validation failed for option --ixf-layout"),
            ([f"--index-file={missing}", "--ixf-layout", "bit-sliced,unpadded"], "not a layout"),
            ([f"--index-file={missing}", "--ixf-layout=bin-major,unpadded,position-major"], "does not exist"),
            ([f"--index-file={missing}", "--frobnicate"], "Unknown option --frobnicate"),
            ([f"--index-file={missing}", "--frobnicate=3"], "Unknown option --frobnicate")):
        cp = run("--output-file", str(tmp_path / "o.tsv"), *args)
        assert cp.returncode != 0, args
        assert "[TAXOR SEARCH ERROR]" in cp.stderr and needle in cp.stderr, (args, cp.stderr)


def test_hidden_flags_are_accepted(tmp_path):
    """--output-verbose-statistics / --debug (taxor_search.cpp:68-79) parse; the run then fails on the missing index,
    not on the flags"""
    missing = tmp_path / "nope.hixf"
    cp = run("--debug", "--output-verbose-statistics", f"--index-file={missing}", "--output-file", str(tmp_path / "o.tsv"))
    assert cp.returncode != 0 and "does not exist" in cp.stderr and "Unknown option" not in cp.stderr
